// Pointwise (1x1 / stride 1 / pad 0) convolution for the TRAINING step: C3's cba1|cba2 / cba3, the bottlenecks' conv_bn_act_1, the
// neck's 1x1 layers and the Detect convs (utils/layer_tools.py:90-114, 152-169, 454-470), forward and data gradient, with 128, 256
// or 512 input channels in all — one segment, or a virtual concat of two equal halves (C3's cba3, the neck joins) either of which
// may be read through the nearest-2x upsample.  These layers move 64-128 FLOP per byte: they are bound by HBM, and on the
// implicit-GEMM ring kernels (conv_v3_kernel: 128 x 128 tiles, two ring stages for the 2..4 k-steps of a tile, the ring drained at
// every tile's epilogue) the 40 x 40 / 20 x 20 layers ran at 2.0-3.3 TB/s — a tile paid its pipeline fill and its epilogue with
// nothing of its own in flight.  Here, as in conv_pw_kernel (inference, YOLOv5x widths), the reduction is so short
// that a tile has NO k loop over memory:
//
//   * a workgroup is TWO groups of four waves that walk the 32-pixel tiles of the workgroup's slot alternately (group g: tiles g,
//     g + 2, ...), half a round apart: every s_barrier of the workgroup is the "tile arrived" barrier of one group and the "staging
//     written" barrier of the other, so while one group multiplies the other stores and requests — the matrix and the memory halves
//     of a round overlap inside one CU without a second workgroup's LDS;
//   * a tile's input channels arrive by LDS-DMA as one piece per segment (unpadded rows; 16-byte chunks XOR-swizzled with the pixel
//     row on the SOURCE side of the DMA and again on the fragment reads: conflict-free ds_read_b128) into a ring of FOUR tile buffers,
//     two per group: while a group's tile i is multiplied its tile i + 2 is in flight and i + 4 is requested as soon as i's buffer
//     is free.  The waits are counted: s_waitcnt vmcnt(N) with N = the vector-memory instructions issued behind the tile's
//     transfers — every load and store of the loop is a buffer instruction with a range-checked offset, issued unconditionally
//     (inline asm: see pt_dma below), so N is a compile-time constant — and the ring never drains;
//   * the 128 x C weight slice of the workgroup's output-channel group lives in REGISTERS for the whole launch (C / 16 fragments of
//     four registers per wave: 32 output channels x C); the operands of the MFMA are swapped — the weights are the A operand, the
//     pixels the B operand — so a lane of the accumulator holds FOUR CONSECUTIVE OUTPUT CHANNELS of one pixel: the tile goes to the
//     group's staging area as 8-byte stores and leaves as whole 16-byte chunks;
//   * wave w multiplies the 32 pixels by its 32 output channels: C / 16 v_mfma_f32_32x32x16_bf16 with one fragment read each;
//   * epilogues work on the staged chunks: EPI 1 the per-channel BatchNorm partial sums (two registers per thread for the whole
//     launch, reduced once at the end into the slab row of the workgroup), EPI 2 bias / folded BN / SiLU, EPI 4 the same with a
//     residual and / or an accumulating store, EPI 3 the fused BatchNorm-backward reduction of a data gradient.  The operands
//     EPI 3 / 4 take from memory (z, the residual, the output's earlier contents) are requested one tile ahead BY LDS-DMA into a
//     slot of the requesting wave (every lane fetches the 16 bytes its own thread will consume, two slot sets used alternately)
//     and read back behind the wave's own counted wait.  They are never held in registers while in flight: the first form loaded
//     them into registers by inline asm and tied the wait to those registers — and the compiler, free to copy a value it
//     believes defined, copied them to other registers IN FRONT of the wait in one branch of the loop (round 5: one slab row of
//     the fused reduction differed in 1 of ~60 launches on some boxes; tools/race_screen.py).  EPI 3 / 4 are not built for 512 and
//     320 input channels (no LDS left for the slots; pt_plan declines).
// The output-channel groups of a pixel slot are consecutive workgroups of one XCD (ids b, b + 8, ...): they walk the same tiles at
// the same time and the second read of a tile comes from that XCD's L2.  Chosen per layer by the engine's timing
// (yh_conv_desc.algo 13); the same arithmetic as the ring kernels (fp32 accumulation over the channels in the same order).
#include "common.h"
#include <stdlib.h>

namespace {

constexpr int PT_TM = 32;                   // pixels per tile
constexpr int PT_TN = 128;                  // output channels per workgroup
constexpr int PT_NBUF = 4;                  // pixel-tile buffers
constexpr int PT_GT = 256;                  // threads of a group (four waves)
constexpr int PT_NT = 2 * PT_GT;            // two groups, half a round apart
constexpr int PT_CP = PT_TN + 8;            // staging row pitch (bf16 elements)
constexpr unsigned PT_OOB = 0x80000000u;

struct PtK {
    const uint16_t* x0; const uint16_t* x1;
    int ld0, ld1, ups0, ups1;
    unsigned xbytes0, xbytes1;
    const uint16_t* w; unsigned wbytes;
    uint16_t* out0; int ldo0;
    unsigned obytes0;
    const uint16_t* res; int ldr; unsigned rbytes;
    const float* bias; const float* scale; const float* shift;
    int act, accumulate;
    float* stats; int Npad;
    const uint16_t* bnr_z; int bnr_ldz, bnr_C; unsigned zbytes;
    const float* bnr_ws; float* bnr_part;
    int N, M, Ho, Wo, ntiles, gx, gy;
    unsigned long long* stamps;     // diagnostics (yh_pt_set_stamps): [workgroup][wave][8] shader-clock stamps of the workgroup's 9th tile
};

// swizzle term of a row: XORed into the chunk index (rows of 128 B: two rows per 256-byte bank period)
template <int CS> __device__ __forceinline__ int pt_swz(int row) { return (CS * 2) % 256 ? ((row >> 1) & 7) : (row & 15); }

template <int CS0, int CS1>
struct PtCfg {
    static constexpr int CT = CS0 + CS1;
    static constexpr int SUB0 = PT_TM * CS0 * 2, SUB1 = PT_TM * CS1 * 2;     // bytes of the two sub-images of a pixel buffer
    static constexpr int A_BYTES = SUB0 + SUB1;
    static constexpr int NI0 = SUB0 / 1024 / 4, NI1 = SUB1 / 1024 / 4;       // DMA instructions per wave and tile, per segment
    static constexpr int NAI = NI0 + NI1;
    static constexpr int STAGE_OFF = PT_NBUF * A_BYTES;
    static constexpr int STAGE_BYTES = PT_TM * PT_CP * 2;                     // the output tile
    static constexpr int CONST_OFF = STAGE_OFF + 2 * STAGE_BYTES;
    static constexpr int OPS_OFF = CONST_OFF + 3 * PT_TN * 4;                 // EPI 3 / 4: [wave 0..7][set 0..1][4 transfers of 1 KiB]
    static constexpr int OPS_BYTES = 8 * 2 * 4 * 1024;
    static constexpr int SMEM = OPS_OFF;                                      // without / with (SMEM_OPS) the operand slots
    static constexpr int SMEM_OPS = OPS_OFF + OPS_BYTES;
    static_assert((CS0 == 64 || CS0 == 128 || CS0 == 256 || CS0 == 512 || (CS0 == 320 && CS1 == 0)) && (CS1 == 0 || CS1 == CS0) && (CT == 128 || CT == 256 || CT == 512 || CT == 320), "segment widths");
    static_assert((CS0 * 2) % 128 == 0 && (PT_TM * CS0 * 2) % 4096 == 0, "whole 16-byte chunk groups of 8 per row; whole transfers per wave");
    static_assert(NI0 >= 1 && (CS1 == 0 || NI1 >= 1) && SMEM <= 160 * 1024, "LDS budget");
    static_assert(PT_NT * 16 * 4 <= PT_NBUF * A_BYTES, "the final reduction of EPI 3 runs in the pixel buffers");
    static constexpr bool OPS_FIT = SMEM_OPS <= 160 * 1024;
};

__device__ __forceinline__ bf16x8_t pt_lds16(const unsigned char* p) {
    return __builtin_bit_cast(bf16x8_t, *reinterpret_cast<const uint4*>(p));
}
typedef unsigned int pt_u32x4 __attribute__((ext_vector_type(4)));
// The vector-memory traffic of the main loop is spelled as inline asm, invisible to the compiler's s_waitcnt bookkeeping: behind the
// builtin forms it cannot tell the ring's buffers from the staging area (it drained vmcnt to 0 in front of the staging reads of
// every tile) and it does not count the transfers when it waits for a register load, so its counts come out too small.  All waits on
// vmcnt in the loop are the hand-counted ones below; every instruction is issued unconditionally (masked lanes carry an out-of-range
// offset), so the counts are compile-time constants.
// One LDS-DMA wave instruction: 64 lanes x 16 bytes -> LDS bytes [lds, lds + 1024)
// (soff: a scalar byte offset added to every lane's; it is not part of the descriptor's range check, voff is)
__device__ __forceinline__ void pt_dma(unsigned lds, unsigned voff, const __amdgpu_buffer_rsrc_t rs, unsigned soff) {
    // M0 is written in the SAME statement that reads it: the compiler reserves M0 and keeps nothing in it across an asm statement
    // (an "m0" clobber only draws -Winline-asm "clobber list contains reserved registers"; cdna_hip_programming.md §5.7)
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %3 offen lds" :: "s"(lds), "v"(voff), "s"(rs), "s"(soff) : "memory");
}
__device__ __forceinline__ uint4 pt_u4(const pt_u32x4 v) { return make_uint4(v.x, v.y, v.z, v.w); }
// A 16-byte store with a SCALAR tile offset.  Inline asm with its own wait states: behind the builtin form the compiler placed a
// VALU write of the first data register directly behind the store (its hazard table exempts buffer stores of more than 64 bits
// whose soffset is an SGPR from the wait state in front of a write to their data registers), and on gfx950 the store then picked up
// the NEW value in the last lanes of every 16-lane row of its first dword: a box- and timing-dependent handful of wrong output
// chunks (4 rows x 2 channels of four chunks; found by tests/test_gpu_tune_table.py, screened with tools/pt_race.py).  The store
// stays an unconditional vector-memory instruction of the loop: the hand-counted waits include it (NST).
__device__ __forceinline__ void pt_bstore(const __amdgpu_buffer_rsrc_t rs, unsigned voff, unsigned soff, uint4 v) {
    const pt_u32x4 w = {v.x, v.y, v.z, v.w};
    asm volatile("buffer_store_dwordx4 %0, %1, %2, %3 offen\n\ts_nop 1" :: "v"(w), "v"(voff), "s"(rs), "s"(soff) : "memory");
}

// EPI 0: plain store, 1: + BatchNorm partial sums, 2: bias / folded BN / SiLU, 3: data gradient + fused BatchNorm-backward reduction,
// 4: as 2 with a residual and / or an accumulating store (operands from memory)
template <int CS0, int CS1, int EPI>
__global__ __launch_bounds__(PT_NT, 1) void conv_pt_kernel(const PtK p)
{
    using G = PtCfg<CS0, CS1>;
    constexpr int CT = G::CT, A_BYTES = G::A_BYTES, SUB0 = G::SUB0, NI0 = G::NI0, NAI = G::NAI, NKS = G::CT / 16;
    constexpr int TM = PT_TM, TN = PT_TN, NBUF = PT_NBUF, GT = PT_GT, CP = PT_CP;
    constexpr int CPR = TN / 8;                                 // 16-byte chunks per staged row
    constexpr int NOI = TM * CPR / GT;                          // output chunks per thread and tile
    // Vector-memory instructions of a wave per tile, in program order: [wait for the tile's transfers, barrier 1, MFMAs, staging,
    // barrier 2] the transfers of the group's tile after next (NAI), the epilogue's operand loads of its NEXT tile (NLD: two registers
    // per output chunk for EPI 2 / 3), [wait for this tile's operand loads] the output stores (NST).  A wait for X is
    // vmcnt(number of instructions issued behind X): completion is in issue order.
    constexpr int NST = NOI;
    constexpr bool OPS = EPI == 3 || EPI == 4;
    constexpr bool GEN = EPI == 2 || EPI == 4;
    constexpr int NLD = OPS ? 2 * NOI : 0;
    static_assert(NOI == 2 && NBUF == 4, "thread -> (row, chunk) map of the store phase; two groups x two ring buffers");

    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    float* const sConst = reinterpret_cast<float*>(smem + G::CONST_OFF);            // EPI 2: bias | scale | shift ; EPI 3: scale | shift

    const int t = threadIdx.x;
    const int lane = t & 63;
    const int wave8 = __builtin_amdgcn_readfirstlane(t >> 6);
    const int grp = wave8 >> 2, wave = wave8 & 3;               // the group of four waves, the wave inside it
    const int tg = t & (GT - 1);                                // the thread inside its group
    uint16_t* const sC = reinterpret_cast<uint16_t*>(smem + G::STAGE_OFF + grp * G::STAGE_BYTES);
    // one row of workgroups in XCD-major (pixel slot, output-channel group) order: see the header
    const int lin = blockIdx.x, xcd = lin & 7, jj = lin >> 3;
    const int by = jj % p.gy, bx = (jj / p.gy) * 8 + xcd;
    const int n0 = by * TN;
    const int r31 = lane & 31, kq = lane >> 5;

    if (bx >= p.ntiles) {               // a slot of the last XCD round without a tile: its rows of the partial-sum slabs are zeros
        if (EPI == 1 && kq == 0 && grp == 0) {
            const int n = n0 + wave * 32 + r31;
            if (n < p.Npad) { p.stats[((size_t)bx * 2 + 0) * p.Npad + n] = 0.f; p.stats[((size_t)bx * 2 + 1) * p.Npad + n] = 0.f; }
        }
        if (EPI == 3)
            for (int i = t; i < 2 * TN; i += PT_NT) {
                const int which = i / TN, c = i - which * TN;
                if (n0 + c < p.N) p.bnr_part[((size_t)bx * 2 + which) * p.N + n0 + c] = 0.f;
            }
        return;
    }
    // this workgroup's tiles: bx + k * gx, k < K; group g takes k = g, g + 2, ...: J rounds (the last may be empty for group 1)
    const int K = (p.ntiles - bx + p.gx - 1) / p.gx;
    const int J = (K + 1) >> 1;

    // per-channel constants first: their loads are consumed here, ahead of every transfer (a later wait for them would drain the ring)
    if (GEN) {
        for (int i = t; i < 3 * TN; i += PT_NT) {
            const int which = i / TN, c = i - which * TN;
            const float* src = which == 0 ? p.bias : (which == 1 ? p.scale : p.shift);
            sConst[i] = (src && n0 + c < p.N) ? src[n0 + c] : (which == 1 ? 1.f : 0.f);
        }
    }
    if (EPI == 3) {
        for (int i = t; i < 2 * TN; i += PT_NT) {
            const int which = i / TN, c = i - which * TN;
            sConst[i] = (n0 + c < p.N) ? p.bnr_ws[(size_t)which * p.bnr_C + n0 + c] : 0.f;
        }
    }

    const __amdgpu_buffer_rsrc_t rs0 = __builtin_amdgcn_make_buffer_rsrc((void*)p.x0, 0, p.xbytes0, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs1 = __builtin_amdgcn_make_buffer_rsrc((void*)p.x1, 0, p.xbytes1, 0x00020000);
    const __amdgpu_buffer_rsrc_t rso0 = __builtin_amdgcn_make_buffer_rsrc((void*)p.out0, 0, p.obytes0, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsz = __builtin_amdgcn_make_buffer_rsrc((void*)p.bnr_z, 0, p.zbytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsr = __builtin_amdgcn_make_buffer_rsrc((void*)p.res, 0, p.rbytes, 0x00020000);

    // ---- loader geometry.  Instruction h of this wave (h < NI0: segment 0, else segment 1) is instruction ii = h' * 4 + wave of its
    // sub-image and fills the rows ii * RPI .. + RPI of it; the lane writes chunk position q of its row and fetches the source chunk
    // q ^ swz(row)
    auto geom = [&](int h, int& row, unsigned& ch) {
        const bool s1 = h >= NI0;
        const int cs = s1 ? CS1 : CS0;
        const int chr = cs / 8;                                   // 16-byte chunks per row
        const int ii = (s1 ? h - NI0 : h) * 4 + wave;
        const int g = ii * 64 + lane;                             // chunk g of the sub-image: a transfer is 64 consecutive chunks
        row = g / chr;                                            // (rows of 640 B — 320 channels — straddle transfers)
        const int q = g - row * chr;
        ch = (unsigned)((q ^ (s1 ? pt_swz<CS1 ? CS1 : 128>(row) : pt_swz<CS0>(row))) * 16);
    };
    const int HoWo = p.Ho * p.Wo;
    const unsigned lds0 = (unsigned)(size_t)((__attribute__((address_space(3))) unsigned char*)smem);
    // A tile is addressed as (per-lane offset inside the tile, fixed for the whole launch) + (a SCALAR offset of the tile): a handful of
    // instructions per transfer.  Only a ragged last tile, a tile past the end (the ring's dummies) and upsampled segments take the
    // per-lane path (rows past M carry an out-of-range offset: the transfer writes zeros).
    unsigned lvo[NAI];
#pragma unroll
    for (int h = 0; h < NAI; ++h) {
        int row; unsigned ch;
        geom(h, row, ch);
        lvo[h] = (unsigned)row * (unsigned)((h >= NI0 ? p.ld1 : p.ld0) * 2) + ch;
    }
    const bool anyups = (p.ups0 | p.ups1) != 0;
    auto issue_A = [&](int tl, int buf) {
        const int m0 = tl * TM;
        const unsigned base = lds0 + buf * A_BYTES;
        if (m0 + TM <= p.M && !anyups) {
            const unsigned so0 = (unsigned)m0 * (unsigned)(p.ld0 * 2), so1 = (unsigned)m0 * (unsigned)(p.ld1 * 2);
#pragma unroll
            for (int h = 0; h < NAI; ++h) {
                const bool s1 = h >= NI0;
                pt_dma(base + (s1 ? SUB0 : 0) + ((s1 ? h - NI0 : h) * 4 + wave) * 1024, lvo[h], s1 ? rs1 : rs0, s1 ? so1 : so0);
            }
            return;
        }
#pragma unroll
        for (int h = 0; h < NAI; ++h) {
            const bool s1 = h >= NI0;
            int row; unsigned ch;
            geom(h, row, ch);
            const int m = m0 + row;
            unsigned px = (unsigned)m;
            if (s1 ? p.ups1 : p.ups0) {         // nearest-2x upsample: the source pixel of (img, y, x) is (img, y / 2, x / 2) of the half-size map
                const int im = m / HoWo, rem = m - im * HoWo, yy = rem / p.Wo, xx = rem - yy * p.Wo;
                px = (unsigned)((im * (p.Ho >> 1) + (yy >> 1)) * (p.Wo >> 1) + (xx >> 1));
            }
            const unsigned v = (px * (unsigned)((s1 ? p.ld1 : p.ld0) * 2) + ch) | ((m < p.M && m >= 0) ? 0u : PT_OOB);
            pt_dma(base + (s1 ? SUB0 : 0) + ((s1 ? h - NI0 : h) * 4 + wave) * 1024, v, s1 ? rs1 : rs0, 0u);
        }
    };
    // this thread's output chunks: chunk cch (8 channels) of rows (tg >> 4) and (tg >> 4) + 16 of every tile of its group, addressed the
    // same way: (per-thread offset inside the tile; out of range for channels past N) + (scalar offset of the tile); the operands the
    // epilogue takes from memory sit at the same (row, channel) of their tensors
    const int cch = tg & (CPR - 1);
    const int nch = n0 + cch * 8;
    const bool nok = nch < p.N;
    unsigned oc[NOI], ac[NOI];
#pragma unroll
    for (int it = 0; it < NOI; ++it) {
        const unsigned row = (unsigned)((tg >> 4) + it * (GT / CPR));
        oc[it] = nok ? row * (unsigned)(p.ldo0 * 2) + (unsigned)(nch * 2) : PT_OOB;
        ac[it] = !nok ? PT_OOB : (EPI == 3 ? row * (unsigned)(p.bnr_ldz * 2) + (unsigned)(nch * 2)
                                           : (p.res != nullptr ? row * (unsigned)(p.ldr * 2) + (unsigned)(nch * 2) : PT_OOB));
    }
    const unsigned a_ld2 = (unsigned)((EPI == 3 ? p.bnr_ldz : p.ldr) * 2);
    // the thread's offset for a tile: unchanged for a full tile, out of range for the rows past M of a ragged one / of a tile past the end
    auto tile_off = [&](int tl, int it, unsigned c) -> unsigned {
        const int m0 = tl * TM;
        if (m0 + TM <= p.M) return c;
        return m0 + (tg >> 4) + it * (GT / CPR) < p.M ? c : PT_OOB;
    };
    // scalar byte offset of a tile's first row (a tile past the end: 0 — every lane is out of range there and the product could wrap)
    auto tile_so = [&](int tl, unsigned ld2) -> unsigned { return tl < p.ntiles ? (unsigned)(tl * TM) * ld2 : 0u; };
    // epilogue operands from memory per output chunk: a = z (EPI 3) / the residual (EPI 4), o = the output's earlier contents
    // (accumulate), by LDS-DMA into the wave's slot set `set`: transfer 2 it (a) and 2 it + 1 (o) of the set, every lane the 16 bytes
    // its own thread consumes.  An operand that is not in use is still requested, with an out-of-range offset (zeros, no memory
    // access): NLD is a constant.
    const unsigned ops0 = lds0 + G::OPS_OFF + wave8 * (2 * 4 * 1024);
    auto load_ops = [&](int tl, int set) {
        if (!OPS) return;
#pragma unroll
        for (int it = 0; it < NOI; ++it) {
            pt_dma(ops0 + (set * 4 + 2 * it) * 1024, tile_off(tl, it, ac[it]), EPI == 3 ? rsz : rsr, tile_so(tl, a_ld2));
            pt_dma(ops0 + (set * 4 + 2 * it + 1) * 1024, p.accumulate ? tile_off(tl, it, oc[it]) : PT_OOB, rso0, tile_so(tl, (unsigned)(p.ldo0 * 2)));
        }
    };
    // the thread's own 16 bytes of transfer `slot` of a set (read only behind the wave's wait for that set's transfers)
    auto ops_rd = [&](int set, int slot) -> uint4 {
        return *reinterpret_cast<const uint4*>(smem + G::OPS_OFF + wave8 * (2 * 4 * 1024) + (set * 4 + slot) * 1024 + lane * 16);
    };
    auto tile_of = [&](int k) -> int { return bx + k * p.gx; };

    // The wave's 32 x C slice of the weights stays in REGISTERS for the whole launch, as the MFMA fragments themselves (lane: output
    // channel n0 + 32 wave + (lane & 31), chunk 2 ks + (lane >> 5) of its row; rows past Npad are out of range: zeros): a tile then
    // costs one LDS fragment read per MFMA, and no LDS is spent on weights.  Requested first; waited for behind the requests of the
    // group's first two pixel tiles and of the epilogue operands of its first.
    pt_u32x4 wreg[NKS];
    {
        const __amdgpu_buffer_rsrc_t rsw = __builtin_amdgcn_make_buffer_rsrc((void*)p.w, 0, p.wbytes, 0x00020000);
        const unsigned wv = (unsigned)(((n0 + wave * 32 + r31) * CT + kq * 8) * 2);
        // ordinary (compiler-visible) loads: the compiler waits for them where they are first used — in front of the loop, beside the
        // first tiles' transfers, once per workgroup
#pragma unroll
        for (int ks = 0; ks < NKS; ++ks) wreg[ks] = __builtin_amdgcn_raw_buffer_load_b128(rsw, wv + ks * 32, 0, 0);
    }
    issue_A(tile_of(grp), grp);
    issue_A(tile_of(grp + 2), grp + 2);
    load_ops(tile_of(grp), 0);        // epilogue operands of the group's first tile: slot set 0 (tile j of the group: set j & 1)
    YH_LDS_BARRIER();                 // sConst published

    // fragment read offsets: row lane & 31 of the sub-image, chunk 2 ks + (lane >> 5)
    const int f0 = pt_swz<CS0>(r31), f1 = pt_swz<CS1 ? CS1 : 128>(r31);

    float bs_[8], bq_[8];                            // per-thread partial sums of its 8 channels (EPI 1: sum, sum of squares; EPI 3: dz, dz * z)
#pragma unroll
    for (int e = 0; e < 8; ++e) { bs_[e] = 0.f; bq_[e] = 0.f; }

    // The two groups run the same rounds half a round apart: every s_barrier is barrier 1 (the tile's transfers have landed for the
    // whole group; its staging area is free) of one group and barrier 2 (its staging area is written; its pixel buffer is consumed)
    // of the other, so one group's MFMA phase runs beside the other's epilogue on every SIMD.  Group 1 starts with one barrier of its
    // own, group 0 ends with one.
    if (grp == 1) YH_LDS_BARRIER();
    int j = 0;
#define PT_STAMP(I) do { if (p.stamps && j == 4 && lane == 0) p.stamps[((size_t)blockIdx.x * 8 + wave8) * 8 + (I)] = __builtin_amdgcn_s_memtime(); } while (0)
    auto one_tile = [&]() {
        const int cset = j & 1;
        const int k = grp + 2 * j;
        const int tile = tile_of(k);
        const int pb = k & (NBUF - 1);
        PT_STAMP(0);
        // ---- the tile's transfers have landed: behind them were issued (at least) the group's next transfers and the stores of two tiles
        if (j == 0)      YH_VMCNT(NAI);
        else if (j == 1) YH_VMCNT(NAI + NST);
        else             YH_VMCNT(NAI + 2 * NST);
        PT_STAMP(1);
        YH_LDS_BARRIER();
        PT_STAMP(2);
        if (k < K) {
            const unsigned char* const abase = smem + pb * A_BYTES;
            // ---- 32 pixels x this wave's 32 output channels: every pixel fragment of the tile is requested ahead of the MFMAs
            // operands swapped (D = W X^T): a lane holds ONE pixel (lane & 31) and the channels (r & 3) + 8 (r >> 2) + 4 (lane >> 5)
            // of the wave's 32 — four runs of four consecutive channels: 8-byte staging stores
            f32x16_t acc;
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[r] = 0.f;
            constexpr int KC = NKS > 16 ? (NKS % 8 ? 10 : 8) : NKS;   // fragments in registers at a time (512 channels: the weight slice alone is 128 registers)
            static_assert(NKS % KC == 0, "whole fragment groups");
#pragma unroll
            for (int k0 = 0; k0 < NKS; k0 += KC) {
                bf16x8_t af[KC];
#pragma unroll
                for (int u = 0; u < KC; ++u) {
                    const int ks = k0 + u;
                    const bool s1 = ks * 16 >= CS0;
                    const int cl = 2 * (s1 ? ks - CS0 / 16 : ks) + kq;
                    af[u] = s1 ? pt_lds16(abase + SUB0 + r31 * (CS1 * 2) + ((cl ^ f1) * 16))
                               : pt_lds16(abase + r31 * (CS0 * 2) + ((cl ^ f0) * 16));
                }
#pragma unroll
                for (int u = 0; u < KC; ++u) acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, wreg[k0 + u]), af[u], acc, 0, 0, 0);
            }
            PT_STAMP(3);
            // ---- accumulators -> staging (bf16, row-major [pixel][channel])
#pragma unroll
            for (int g4 = 0; g4 < 4; ++g4) {
                const int c = wave * 32 + 8 * g4 + 4 * kq;
                float v[4] = {acc[4 * g4], acc[4 * g4 + 1], acc[4 * g4 + 2], acc[4 * g4 + 3]};
                if (GEN) {
                    const float4 cb = *reinterpret_cast<const float4*>(sConst + c), cs = *reinterpret_cast<const float4*>(sConst + TN + c),
                                 ct = *reinterpret_cast<const float4*>(sConst + 2 * TN + c);
                    v[0] = (v[0] + cb.x) * cs.x + ct.x; v[1] = (v[1] + cb.y) * cs.y + ct.y;
                    v[2] = (v[2] + cb.z) * cs.z + ct.z; v[3] = (v[3] + cb.w) * cs.w + ct.w;
                    if (p.act == YH_ACT_SILU) { v[0] = silu_fast(v[0]); v[1] = silu_fast(v[1]); v[2] = silu_fast(v[2]); v[3] = silu_fast(v[3]); }
                }
                *reinterpret_cast<uint2*>(sC + r31 * CP + c) = make_uint2(pack2(v[0], v[1]), pack2(v[2], v[3]));
            }
        }
        PT_STAMP(4);
        YH_LDS_BARRIER();
        PT_STAMP(5);
        issue_A(tile_of(k + 4), pb);                 // the group's tile after next, into the buffer just consumed (past the end: a dummy of zeros)
        load_ops(tile_of(k + 2), cset ^ 1);
        uint4 opa[NOI], opo[NOI];
        if (OPS) {
            // this tile's operands have landed: behind them were issued (at least) this round's NAI + NLD transfers — from the
            // second round on also the previous tile's stores, which the one count simply waits for as well
            YH_VMCNT(NAI + NLD);
#pragma unroll
            for (int it = 0; it < NOI; ++it) { opa[it] = ops_rd(cset, 2 * it); opo[it] = ops_rd(cset, 2 * it + 1); }
        }
        PT_STAMP(6);
        // ---- whole 16-byte chunks to memory (a round without a tile: every offset is out of range, the stores still count)
#pragma unroll
        for (int it = 0; it < NOI; ++it) {
            const int row = (tg >> 4) + it * (GT / CPR);
            const unsigned oo = tile_off(tile, it, oc[it]);
            uint4 v = *reinterpret_cast<const uint4*>(sC + row * CP + cch * 8);
            if (EPI == 1 && k < K) {        // BatchNorm partial sums of the STORED values (rows past M are exact zeros), per thread and channel
                float f[8];
                unpack8(v, f);
#pragma unroll
                for (int e = 0; e < 8; ++e) { bs_[e] += f[e]; bq_[e] += f[e] * f[e]; }
            }
            if (EPI == 4) {
                const bool addres = p.res != nullptr;
                if (addres || p.accumulate) {
                    float f[8];
                    unpack8(v, f);
                    if (addres) {
                        float g2[8]; unpack8(opa[it], g2);
#pragma unroll
                        for (int e = 0; e < 8; ++e) f[e] += g2[e];
                    }
                    if (p.accumulate) {
                        float g2[8]; unpack8(opo[it], g2);
#pragma unroll
                        for (int e = 0; e < 8; ++e) f[e] += g2[e];
                    }
                    v = pack8(f);
                }
            }
            if (EPI == 3) {
                float g[8], z[8];
                unpack8(v, g);
                if (p.accumulate) {
                    // last writer of a gradient with several contributions: add the earlier ones (bf16, as the generic epilogue does) and
                    // take the BatchNorm-backward sums over the rounded total
                    float g0[8];
                    unpack8(opo[it], g0);
#pragma unroll
                    for (int e = 0; e < 8; ++e) g[e] += g0[e];
                    v = pack8(g);
                    unpack8(v, g);
                }
                unpack8(opa[it], z);
                if (oo != PT_OOB) {
                    float zs[8], zh[8];                  // scale | shift of the producer's BatchNorm for this thread's 8 channels
                    *reinterpret_cast<float4*>(zs) = *reinterpret_cast<const float4*>(sConst + cch * 8);
                    *reinterpret_cast<float4*>(zs + 4) = *reinterpret_cast<const float4*>(sConst + cch * 8 + 4);
                    *reinterpret_cast<float4*>(zh) = *reinterpret_cast<const float4*>(sConst + TN + cch * 8);
                    *reinterpret_cast<float4*>(zh + 4) = *reinterpret_cast<const float4*>(sConst + TN + cch * 8 + 4);
#pragma unroll
                    for (int e = 0; e < 8; ++e) {
                        const float a = z[e] * zs[e] + zh[e];
                        const float sg = sigmoid_fast(a);
                        const float dz = g[e] * (sg * (1.f + a * (1.f - sg)));
                        bs_[e] += dz; bq_[e] += dz * z[e];
                    }
                }
            }
            pt_bstore(rso0, oo, tile_so(tile, (unsigned)(p.ldo0 * 2)), v);
        }
        PT_STAMP(7);
        ++j;
    };
#undef PT_STAMP
    while (j < J) one_tile();
    if (grp == 0) YH_LDS_BARRIER();

    // ---- the partial sums of the workgroup: the two groups hold the same channels (EPI 1) / chunks (EPI 3)
    float* const sRed = reinterpret_cast<float*>(smem);        // the pixel buffers: whatever is still in flight into them is a dummy of a tile past the end
    if (EPI == 1 || EPI == 3) {
        YH_VMCNT(0);
        __syncthreads();
    }
    if (EPI == 1 || EPI == 3) {
#pragma unroll
        for (int e = 0; e < 8; ++e) { sRed[t * 16 + e] = bs_[e]; sRed[t * 16 + 8 + e] = bq_[e]; }
        __syncthreads();
        for (int i = t; i < 2 * TN; i += PT_NT) {
            const int which = i / TN, c = i - which * TN;
            float v = 0.f;
            for (int k2 = c / 8; k2 < PT_NT; k2 += CPR) v += sRed[k2 * 16 + which * 8 + (c & 7)];       // fixed order: deterministic
            if (EPI == 1) { if (n0 + c < p.Npad) p.stats[((size_t)bx * 2 + which) * p.Npad + n0 + c] = v; }
            else if (n0 + c < p.N) p.bnr_part[((size_t)bx * 2 + which) * p.N + n0 + c] = v;
        }
    }
}

struct PtPlan { int cs0, cs1, epi, grid; PtK k; };

bool pt_plan(const yh_conv_desc* d, PtPlan* pl)
{
    if (!d || (d->nseg != 1 && d->nseg != 2)) return false;
    if (d->KH != 1 || d->KW != 1 || d->stride != 1 || d->pad != 0 || d->Ho != d->Hi || d->Wo != d->Wi) return false;
    const int C0 = d->seg[0].C, C1 = d->nseg == 2 ? d->seg[1].C : 0;
    // 320 channels (YOLOv5x's bottlenecks at 1280^2): inference forms only (plain / generic epilogue)
    const bool c320 = C1 == 0 && C0 == 320 && !d->stats && !d->bnr_part;
    if (!(c320 || (C1 == 0 && (C0 == 128 || C0 == 256 || C0 == 512)) || (C1 == C0 && (C0 == 64 || C0 == 128 || C0 == 256)))) return false;
    if (d->N <= 0 || d->Npad < d->N || d->Npad % 128) return false;
    const unsigned long M = (unsigned long)d->B * d->Ho * d->Wo;
    if (M == 0 || M >= (1ul << 31) - (1ul << 20)) return false;          // (tile indices of the ring's dummies past the end stay in range)
    const unsigned long lim = (1ul << 31) - 4096;
    PtK& k = pl->k;
    memset(&k, 0, sizeof(k));
    for (int s = 0; s < d->nseg; ++s) {
        const yh_seg& g = d->seg[s];
        if (!g.ptr || g.ld % 8 || g.ld < g.C || (g.ups != 0 && g.ups != 1)) return false;
        if (g.ups && (d->Hi % 2 || d->Wi % 2)) return false;
        const unsigned long npix = (unsigned long)d->B * (d->Hi >> g.ups) * (d->Wi >> g.ups);
        const unsigned long bytes = ((npix - 1) * g.ld + g.C) * 2;
        if (bytes >= lim) return false;
        if (s == 0) { k.x0 = g.ptr; k.ld0 = g.ld; k.ups0 = g.ups; k.xbytes0 = (unsigned)bytes; }
        else        { k.x1 = g.ptr; k.ld1 = g.ld; k.ups1 = g.ups; k.xbytes1 = (unsigned)bytes; }
    }
    if (d->nseg == 1) { k.x1 = k.x0; k.ld1 = k.ld0; k.ups1 = k.ups0; k.xbytes1 = k.xbytes0; }
    const int Ct = C0 + C1;
    const unsigned long wb = (unsigned long)d->Npad * Ct * 2;
    if (wb >= lim) return false;
    if (d->nsplit < d->N) return false;                  // one destination (the split store of C3's stacked cba1 | cba2 is an inference form)
    const bool generic_na = d->bias || d->scale || d->shift || d->act != YH_ACT_NONE || d->res;
    const bool generic = generic_na || d->accumulate;
    if (generic && d->stats) return false;
    // 512 / 320 channels: no LDS left for the operand slots of the epilogues that read memory (fused reduction, residual, accumulate);
    // at 512 the weight slice alone is 128 registers
    if ((Ct == 512 || Ct == 320) && (d->bnr_part || d->res || d->accumulate)) return false;
    if (d->bnr_part && (generic_na || d->stats || d->mode != YH_CONV_DGRAD || d->N % 8 || !d->bnr_z || !d->bnr_ws || d->bnr_C < d->N || d->bnr_ldz % 8)) return false;
    const unsigned long Nr = (unsigned long)(d->N + 7) / 8 * 8;
    const unsigned long n0r = Nr;
    const unsigned long ob0 = ((M - 1) * d->ld0 + n0r) * 2;
    if (ob0 >= lim || d->ld0 % 8 || (unsigned long)d->ld0 < n0r) return false;
    k.w = d->w; k.wbytes = (unsigned)wb;
    k.out0 = d->out0; k.ldo0 = d->ld0;
    k.obytes0 = (unsigned)ob0;
    k.res = d->res; k.ldr = d->ldr;
    if (d->res) {
        const unsigned long rb = ((M - 1) * d->ldr + n0r) * 2;
        if (rb >= lim || d->ldr % 8) return false;
        k.rbytes = (unsigned)rb;
    }
    k.bias = d->bias; k.scale = d->scale; k.shift = d->shift; k.act = d->act; k.accumulate = d->accumulate;
    k.stats = d->stats; k.Npad = d->Npad;
    k.bnr_z = d->bnr_z; k.bnr_ldz = d->bnr_ldz; k.bnr_C = d->bnr_C; k.bnr_ws = d->bnr_ws; k.bnr_part = d->bnr_part;
    if (d->bnr_part) {
        const unsigned long zb = ((M - 1) * d->bnr_ldz + Nr) * 2;
        if (zb >= lim) return false;
        k.zbytes = (unsigned)zb;
    }
    k.N = d->N; k.M = (int)M; k.Ho = d->Ho; k.Wo = d->Wo;
    k.ntiles = (int)((M + PT_TM - 1) / PT_TM);
    k.gy = (d->N + PT_TN - 1) / PT_TN;
    pl->cs0 = C0; pl->cs1 = C1;
    pl->epi = d->bnr_part ? 3 : ((d->res || d->accumulate) ? 4 : (generic ? 2 : (d->stats ? 1 : 0)));
    // persistent blocks: one resident round (one workgroup of eight waves per CU), pixel slots in multiples of 8
    int cap = (256 / k.gy) & ~7;
    if (cap < 8) cap = 8;
    if (d->grid_cap > 0) cap = (d->grid_cap + 7) & ~7;
    int gx = ((k.ntiles + 7) / 8) * 8;              // whole XCD rounds: a slot past the tiles returns at once
    if (gx > cap) gx = cap;
    k.gx = gx;
    pl->grid = gx * k.gy;
    return true;
}

}  // namespace

static unsigned long long* g_pt_stamps = nullptr;
/* diagnostics: a device buffer of grid x 4 x 8 uint64 that receives shader-clock stamps of every wave's 9th tile (NULL: off) */
extern "C" void yh_pt_set_stamps(void* p) { g_pt_stamps = (unsigned long long*)p; }

int yh_pt_rows(const yh_conv_desc* d)
{
    PtPlan pl;
    return pt_plan(d, &pl) ? pl.k.gx : 0;
}

int yh_pt_run(const yh_conv_desc* d, yh_stream stream, char* name_out, int name_len)
{
    PtPlan pl;
    YH_CHECK_ARG(pt_plan(d, &pl), "yh_conv_igemm: algo 13 (training pointwise kernel) is not eligible for this descriptor");
    if (name_out) { snprintf(name_out, name_len, "conv_pt_kernel<%d, %d, %d>", pl.cs0, pl.cs1, pl.epi); return YH_OK; }
    YH_CHECK_ARG(yh_aligned16(d->seg[0].ptr) && (d->nseg == 1 || yh_aligned16(d->seg[1].ptr)) && yh_aligned16(d->w) && d->out0 && yh_aligned16(d->out0) &&
                 yh_aligned16(d->res) && yh_aligned16(d->bnr_z), "yh_conv_igemm(pt): unaligned operand");
    // timing-only diagnostics (results wrong): YH_PT_ABL bit 0: zero-record input descriptors (every transfer returns zeros without
    // touching memory), bit 1: zero-record output / z / residual descriptors (stores dropped, operand loads return zeros)
    pl.k.stamps = g_pt_stamps;
    static const int abl = [] { const char* e = getenv("YH_PT_ABL"); return e ? atoi(e) : 0; }();
    if (abl & 1) pl.k.xbytes0 = pl.k.xbytes1 = 0;
    if (abl & 2) pl.k.obytes0 = pl.k.zbytes = pl.k.rbytes = 0;
    hipStream_t st = (hipStream_t)stream;
    const dim3 grid(pl.grid), blk(PT_NT);
#define YH_LAUNCH_PT(A_, B_)                                                                                            \
    do {                                                                                                                \
        constexpr int sm = PtCfg<A_, B_>::SMEM, smo = PtCfg<A_, B_>::SMEM_OPS;                                           \
        constexpr bool OK_ = PtCfg<A_, B_>::OPS_FIT;        /* 512 channels: the forms with memory operands are not built (pt_plan) */ \
        constexpr int E3 = OK_ ? 3 : 0, E4 = OK_ ? 4 : 2;                                                                \
        static YhDevOnce attr_set;                                                                                         \
        if (attr_set.need()) {                                                                                                \
            attr_set.set((const void*)conv_pt_kernel<A_, B_, 0>, hipFuncAttributeMaxDynamicSharedMemorySize, sm); \
            attr_set.set((const void*)conv_pt_kernel<A_, B_, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, sm); \
            attr_set.set((const void*)conv_pt_kernel<A_, B_, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, sm); \
            if (OK_) {                                                                                                  \
                attr_set.set((const void*)conv_pt_kernel<A_, B_, E3>, hipFuncAttributeMaxDynamicSharedMemorySize, smo); \
                attr_set.set((const void*)conv_pt_kernel<A_, B_, E4>, hipFuncAttributeMaxDynamicSharedMemorySize, smo); \
            }                                                                                                           \
            attr_set.done();                                                                                             \
        }                                                                                                               \
        YH_CHECK_ARG(OK_ || pl.epi < 3, "yh_conv_igemm(pt): no operand slots for this width");                          \
        switch (pl.epi) {                                                                                               \
        case 0: conv_pt_kernel<A_, B_, 0><<<grid, blk, sm, st>>>(pl.k); break;                                          \
        case 1: conv_pt_kernel<A_, B_, 1><<<grid, blk, sm, st>>>(pl.k); break;                                          \
        case 2: conv_pt_kernel<A_, B_, 2><<<grid, blk, sm, st>>>(pl.k); break;                                          \
        case 3: conv_pt_kernel<A_, B_, E3><<<grid, blk, smo, st>>>(pl.k); break;                                        \
        default: conv_pt_kernel<A_, B_, E4><<<grid, blk, smo, st>>>(pl.k); break;                                       \
        }                                                                                                               \
    } while (0)
    if (pl.cs1 == 0 && pl.cs0 == 320) {
        constexpr int sm = PtCfg<320, 0>::SMEM;
        static YhDevOnce attr_set;      
        if (attr_set.need()) {
            attr_set.set((const void*)conv_pt_kernel<320, 0, 0>, hipFuncAttributeMaxDynamicSharedMemorySize, sm);
            attr_set.set((const void*)conv_pt_kernel<320, 0, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, sm);
            attr_set.done(); 
        }
        YH_CHECK_ARG(pl.epi == 0 || pl.epi == 2, "yh_conv_igemm(pt): 320 channels: plain / bias-BN-SiLU epilogues only");
        if (pl.epi == 2) conv_pt_kernel<320, 0, 2><<<grid, blk, sm, st>>>(pl.k);
        else             conv_pt_kernel<320, 0, 0><<<grid, blk, sm, st>>>(pl.k);
    }
    else if (pl.cs1 == 0) { if (pl.cs0 == 128) YH_LAUNCH_PT(128, 0); else if (pl.cs0 == 256) YH_LAUNCH_PT(256, 0); else YH_LAUNCH_PT(512, 0); }
    else             { if (pl.cs0 == 64) YH_LAUNCH_PT(64, 64); else if (pl.cs0 == 128) YH_LAUNCH_PT(128, 128); else YH_LAUNCH_PT(256, 256); }
#undef YH_LAUNCH_PT
    YH_CHECK_LAUNCH("yh_conv_igemm(pt)");
    return YH_OK;
}
