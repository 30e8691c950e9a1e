// Weight gradient of the NHWC bf16 convolution for the K-heavy layers (input channels a multiple of 32, >= 64 outputs), gfx950:
//
//   dW[n][tap*Ctot + coff_k + c] += sum_m gy[m][n] * X[pixel(m, tap)][c]          (the GEMM gy^T (N x M) * im2col(X) (M x K))
//
// Structure (yh_wgrad_desc.tile_k == 129; round 4):
//   * ONE workgroup of four waves per CU, one wave per SIMD with the whole 512-register file: every wave owns a complete
//     128 x 128 fp32 output tile (256 accumulator registers) over ITS OWN range of pixels, so the waves share nothing in the
//     main loop and there is NO workgroup barrier in it;
//   * every wave feeds a private ring of four LDS stages (16 pixels each: gy [16][128] | X [16][128], 8 KB) by LDS-DMA
//     (buffer_load ... lds, 1 KiB per instruction) and waits only on its own counted vmcnt: three stages stay in flight;
//   * both MFMA operands have the reduction index (pixel) as their slow LDS axis: fragments come from ds_read_b64_tr_b16
//     (transposing read), ONE read per v_mfma_f32_32x32x16_bf16 (16 reads feed 16 MFMAs), requested one k-step ahead;
//     the LDS rows are unpadded (DMA writes lane * 16 B), bank conflicts are avoided by an XOR of the 16-byte chunk index
//     with (pixel & 3) << 2 applied on the SOURCE side of the DMA and again on the reads;
//   * the im2col address of a pixel row is computed by ONE lane per row and handed to the 16 lanes of the row by a DPP
//     quad broadcast (a row's 16 chunks differ by a lane constant);
//   * stream-K: the (tile, 32 pixels) units of the layer are dealt evenly to the workgroups (a workgroup may end one
//     tile and begin the next), so 144 tiles on 256 CUs cost 0.56 tile-times, not one;
//   * the four partial tiles of a workgroup are combined through LDS (reduce-scatter in two exchange rounds of plain 16-byte
//     stores / loads, every wave ends with one quarter) and leave as ONE set of fp32 atomics per workgroup and tile: half the
//     adds of two independent blocks per CU.
// Replaces autograd's conv weight gradient (train_yolov5.py:337 -> utils/layer_tools.py:82-94).
#include "common.h"
#include <stdlib.h>

namespace {

typedef __attribute__((ext_vector_type(4))) short v4s;
typedef __attribute__((ext_vector_type(8))) short v8s;

constexpr int WGS_STG = 4;                           // ring stages per wave
constexpr int WGS_STAGE = 8192;                      // bytes per stage: A [16 px][128 ch] | B [16 px][128 cols]
constexpr int WGS_WAVE_LDS = WGS_STG * WGS_STAGE;    // 32 KB per wave
constexpr int WGS_LDS = 4 * WGS_WAVE_LDS;            // 128 KB per workgroup
constexpr unsigned WGS_OOB = 0x80000000u;

struct WgsK {
    yh_wgrad_desc d;
    unsigned long long* stamps;   // diagnostics (yh_wgs_set_stamps): [workgroup][wave][8] shader-clock stamps of the first segment
    int Ktot;          // columns of a dw row
    int nk;            // 32-pixel work units of the layer (M / 32); a unit is two 16-pixel steps
    int nct, T;        // column tiles (128 im2col columns of the segment: column = tap * C + c) per n-tile, tiles
    int G;             // virtual workgroups
    int mg, mGp, minv; // XCD-aware block map (mg = gcd(T, G), mGp = G / mg, minv = (T / mg)^-1 mod mGp); mg 0: identity
    long U;            // T * nk work units
    unsigned gybytes, xbytes, dwbytes;
    int pw;            // 1x1 / stride 1 / pad 0 layer on a plain segment (the kernel's PW form)
};

__device__ __forceinline__ v4s wgs_tr(const unsigned char* p) {
    return __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) v4s*)(p));
}
// fragment of 32 channels x 16 pixels: pixels 8h .. 8h+3 and 8h+4 .. 8h+7 of the lane's channel
__device__ __forceinline__ bf16x8_t wgs_frag(const unsigned char* p) {
    const v4s lo = wgs_tr(p);
    const v4s hi = wgs_tr(p + 4 * 256);
    const v8s v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
    return __builtin_bit_cast(bf16x8_t, v);
}
template <int I> __device__ __forceinline__ unsigned wgs_quad(unsigned v) {
    return (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, I * 0x55, 0xf, 0xf, false);
}
// One LDS-DMA wave instruction, 64 lanes x 16 bytes -> LDS bytes [lds, lds + 1024), as inline asm: the compiler then neither
// counts it in its s_waitcnt bookkeeping nor treats it as an LDS write (behind the builtin form it drains vmcnt to 0 in front
// of EVERY LDS read: SIInsertWaitcnts cannot tell the slots of the ring apart).  The waits for these transfers are the
// hand-counted YH_VMCNT() of the main loop; soff is excluded from the descriptor's range check, voff is not (0x80000000: zeros).
__device__ __forceinline__ void wgs_dma(unsigned lds, unsigned voff, const __amdgpu_buffer_rsrc_t rs, int soff) {
    // M0 is written in the SAME statement that reads it: the compiler reserves M0 and keeps nothing in it across an asm statement
    // (an "m0" clobber only draws -Winline-asm "clobber list contains reserved registers"; cdna_hip_programming.md §5.7)
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %3 offen lds" :: "s"(lds), "v"(voff), "s"(rs), "s"(soff) : "memory");
}

// PW: 1x1 / stride 1 / pad 0 layer on a plain segment (the im2col row IS the pixel's row: scalar walk)
template <bool PW>
__device__ __forceinline__ void wgs_body(const WgsK& p, const int bidx, unsigned char* const smem)
{
    const yh_wgrad_desc& d = p.d;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    unsigned char* const wbase = smem + wave * WGS_WAVE_LDS;

    // virtual workgroup.  Virtual workgroup v covers units [v U / G, (v + 1) U / G) of the tile-major order, so it STARTS at the
    // fraction frac(v T / G) of a tile's pixel range: workgroups with the same start fraction ("phase") walk the same pixels of
    // different tiles at the same time and read the same gy / X rows.  Hardware workgroups are dealt round-robin to the 8 XCDs by
    // linear id, ONE per CU; XCD x (ids x, x + 8, ...) takes a contiguous range of the phase-sorted order, so equal and
    // neighbouring phases share one L2 and no XCD gets more workgroups than CUs.  With g = gcd(T, G), G' = G / g: phase index
    // ph = (v T / g) mod G' and the g workgroups of a phase are v0 + k G', v0 = ph (T / g)^-1 mod G'.  An exact T x S grid is the
    // case g = T, G' = S (split-major order: the T tiles of a pixel split on one XCD).
    int v = bidx;
    if (p.mg > 0) {
        const int b = bidx, x = b & 7, q = p.G >> 3, r = p.G & 7;
        const int pos = (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + (b >> 3);
        const int ph = pos / p.mg;
        v = (ph * p.minv) % p.mGp + (pos - ph * p.mg) * p.mGp;
    }
    const long u0 = p.U * v / p.G, u1 = p.U * (v + 1) / p.G;

    const __amdgpu_buffer_rsrc_t rsg = __builtin_amdgcn_make_buffer_rsrc((void*)d.gy, 0, p.gybytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsx = __builtin_amdgcn_make_buffer_rsrc((void*)d.seg.ptr, 0, p.xbytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsd = __builtin_amdgcn_make_buffer_rsrc((void*)d.dw, 0, p.dwbytes, 0x00020000);

    // loader geometry: a transfer fills 4 pixel rows x 256 B; this lane's row inside it, its chunk position inside the LDS
    // row and the source chunk that position holds
    const int lrow = lane >> 4;
    const int srcch = (lane & 15) ^ ((lrow & 3) << 2);
    // fragment reads: lane 4q+pp of a 16-lane group supplies row q, columns 4pp .. 4pp+3 of the group's 4 x 16 block
    const int g16 = lane >> 4, q = (lane >> 2) & 3, pp = lane & 3;
    const int rdc = (8 * (g16 >> 1) + q) * 256 + ((g16 & 1) << 5) + ((pp >> 1) << 4) + ((pp & 1) << 3);
    // Physical accumulator [i][j] of wave w holds the logical sub-tile (x, y) = (i ^ 2 hx, j ^ 2 hy), hx = w & 1, hy = w >> 1 (a
    // permutation of the fragment read addresses, free): the quarter a wave keeps in the combine is then ALWAYS the physical
    // [0..1][0..1] and what it gives away [2..3][*] and [0..1][2..3] — the exchange below is straight-line code for every wave.
    const int hx = wave & 1, hy = wave >> 1;
    int rdA[4], rdB[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        rdA[i] = rdc + (((i ^ (hx << 1)) ^ q) << 6);
        rdB[i] = rdc + (((i ^ (hy << 1)) ^ q) << 6) + 4096;
    }

    const int HoWo = d.Ho * d.Wo;
    const int stepH = 16 / d.Wo, stepW = 16 - stepH * d.Wo;
    const int sshift = d.stride - 1;

#define WGS_STAMP(I) do { if (p.stamps && first && lane == 0) p.stamps[((size_t)blockIdx.x * 4 + wave) * 8 + (I)] = __builtin_amdgcn_s_memtime(); } while (0)
    bool first = true;
    long u = u0;
    while (u < u1) {
        WGS_STAMP(0);
        // ---- segment: tile t, units [ka, kb)
        const int t = (int)(u / p.nk);
        const int ka = (int)(u - (long)t * p.nk);
        const long left = u1 - u;
        const int kb = (left < (long)(p.nk - ka)) ? ka + (int)left : p.nk;
        u += kb - ka;
        const int ntile = t / p.nct, ctile = t - ntile * p.nct;
        const int n0 = ntile * 128;
        const int segC = d.seg.C, segld = d.seg.ld, ups = d.seg.ups;
        const int coffk = d.coff_k;
        const int Hs = d.Hi >> ups, Ws = d.Wi >> ups;
        const unsigned pixb = (unsigned)(segld * 2);
        // this lane's 16-byte chunk of a row: im2col column colv = tap * C + channel.  The four lanes of a quad hold 32 consecutive
        // columns, C is a multiple of 32: a quad never straddles a tap, so the lane that OWNS a row (below) computes the row's
        // offset for ITS quad's tap (C = 64: a tile is two taps; C >= 128: one tap or a part of one)
        const int colv = ctile * 128 + srcch * 8;
        const int tapv = colv / segC;
        const int chv = colv - tapv * segC;
        const int khv = tapv / d.KW, kwv = tapv - khv * d.KW;
        const bool tapok = tapv < d.KH * d.KW;
        // this wave's share of the segment
        const int ns = kb - ka, nsb = ns >> 2, nsr = ns & 3;
        const int k0 = 2 * (ka + wave * nsb + (wave < nsr ? wave : nsr));      // in 16-pixel steps
        const int k1 = k0 + 2 * (nsb + (wave < nsr ? 1 : 0));

        unsigned voffA[4], voffB[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            voffA[i] = (n0 + srcch * 8 < d.N) ? (unsigned)(((4 * i + lrow) * d.ldg + n0 + srcch * 8) * 2) : WGS_OOB;
            voffB[i] = tapok ? (unsigned)(((4 * i + lrow) * segld + chv) * 2) : WGS_OOB;
        }
        const unsigned chunkB = (unsigned)(chv * 2);
        // the pixel of the row this lane OWNS (row lrow + 4 * (lane & 3): the quad's lane i owns the row of transfer i)
        int pim = 0, pho = 0, pwo = 0;
        if (!PW && k0 < k1) {
            const int m = 16 * k0 + lrow + 4 * (lane & 3);
            pim = m / HoWo;
            const int rem = m - pim * HoWo;
            pho = rem / d.Wo;
            pwo = rem - pho * d.Wo;
        }

        f32x16_t acc[4][4];
#pragma unroll
        for (int x = 0; x < 4; ++x)
#pragma unroll
            for (int y = 0; y < 4; ++y)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[x][y][r] = 0.f;

        // The im2col offset of the owned row for the stage about to be issued, then the walk to the next stage (16 pixels on:
        // at most one row wrap and one image wrap, Ho * Wo >= 16).  `oob` (scalar) turns a stage past the wave's range into a dummy.
        unsigned boff = 0;
        auto b_addr = [&](unsigned oob) {
            const int hi = (pho << sshift) - d.pad + khv, wi = (pwo << sshift) - d.pad + kwv;
            const bool ok = (unsigned)hi < (unsigned)d.Hi && (unsigned)wi < (unsigned)d.Wi && tapok;
            const unsigned pix = (unsigned)__umul24((unsigned)__umul24(pim, Hs) + (unsigned)(hi >> ups), Ws) + (unsigned)(wi >> ups);
            // a pixel outside the image (its pix is garbage) or a dummy stage: bit 31 puts the offset past the descriptor's range
            boff = (pix * pixb) | (ok ? 0u : WGS_OOB) | oob;
            pwo += stepW; pho += stepH;
            const bool cw = pwo >= d.Wo;
            pwo -= cw ? d.Wo : 0; pho += cw ? 1 : 0;
            const bool ch = pho >= d.Ho;
            pho -= ch ? d.Ho : 0; pim += ch ? 1 : 0;
        };
        // transfer i (0..3: gy rows 4i .. 4i+3, 4..7: X rows) of stage s into the slot at LDS address la
#define WGS_DMA_A(I, LA, SA, OOB_) wgs_dma(LA + (I) * 1024, voffA[I] | (OOB_), rsg, SA)
#define WGS_DMA_B(I, LA, SB, OOB_)                                                          \
        do {                                                                                 \
            if (PW) wgs_dma(LA + 4096 + (I) * 1024, voffB[I] | (OOB_), rsx, SB);             \
            else    wgs_dma(LA + 4096 + (I) * 1024, wgs_quad<I>(boff) + chunkB, rsx, 0);     \
        } while (0)
        // A wave's range is a whole number of 32-pixel units, i.e. an even number of steps: the loop body is two steps with the
        // fragment sets named statically (no runtime-indexed register arrays) and nothing conditional touches the accumulators.
        // Step K multiplies the fragments of stage K (in registers), requests those of stage K+1 and refills the slot of stage K
        // with stage K+4 (a dummy of out-of-range offsets past the wave's range: every step issues 8 transfers and every wait is
        // the same counted vmcnt).  The body is ONE basic block cut into 8 groups of {2 MFMA, 1 fragment = 2 transposing reads,
        // 1 transfer} by scheduling fences: the issue order in the binary is the order written here.
        bf16x8_t fa0[4], fb0[4], fa1[4], fb1[4];
        if (k0 < k1) {
            const unsigned lbase = (unsigned)(size_t)((__attribute__((address_space(3))) unsigned char*)wbase);
#pragma unroll
            for (int s = 0; s < WGS_STG; ++s) {
                const unsigned oob = (k0 + s < k1) ? 0u : WGS_OOB;
                const unsigned la = lbase + s * WGS_STAGE;
                const int sa = (k0 + s) * 16 * d.ldg * 2, sb = (k0 + s) * 16 * segld * 2;
                if (!PW) b_addr(oob);
                WGS_DMA_A(0, la, sa, oob); WGS_DMA_A(1, la, sa, oob); WGS_DMA_A(2, la, sa, oob); WGS_DMA_A(3, la, sa, oob);
                WGS_DMA_B(0, la, sb, oob); WGS_DMA_B(1, la, sb, oob); WGS_DMA_B(2, la, sb, oob); WGS_DMA_B(3, la, sb, oob);
            }
            YH_VMCNT(24);
            WGS_STAMP(1);
#pragma unroll
            for (int x = 0; x < 4; ++x) { fa0[x] = wgs_frag(wbase + rdA[x]); fb0[x] = wgs_frag(wbase + rdB[x]); }
            int slot = 0;
#define WGS_MM(CA, CB, I) acc[(I) >> 2][(I) & 3] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(CA[(I) >> 2], CB[(I) & 3], acc[(I) >> 2][(I) & 3], 0, 0, 0)
#define WGS_STEP(K, CA, CB, NA_, NB_)                                                            \
            {                                                                                    \
                YH_VMCNT(16);                                  /* stage K+1 has landed */        \
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   /* stage K is in registers: its slot is free */ \
                __builtin_amdgcn_sched_barrier(0);                                               \
                const int s4 = (K) + WGS_STG;                                                    \
                const unsigned oob = s4 < k1 ? 0u : WGS_OOB;                                     \
                const unsigned la = lbase + slot * WGS_STAGE;                                    \
                const int sa = s4 * 16 * d.ldg * 2, sb = s4 * 16 * segld * 2;                    \
                slot = (slot + 1) & (WGS_STG - 1);                                               \
                const unsigned char* const rb = wbase + slot * WGS_STAGE;                        \
                WGS_MM(CA, CB, 0); WGS_MM(CA, CB, 1); NA_[0] = wgs_frag(rb + rdA[0]);          WGS_DMA_A(0, la, sa, oob); if (!PW) b_addr(oob); \
                __builtin_amdgcn_sched_barrier(0);                                               \
                WGS_MM(CA, CB, 2); WGS_MM(CA, CB, 3); NB_[0] = wgs_frag(rb + rdB[0]);   WGS_DMA_A(1, la, sa, oob); \
                __builtin_amdgcn_sched_barrier(0);                                               \
                WGS_MM(CA, CB, 4); WGS_MM(CA, CB, 5); NA_[1] = wgs_frag(rb + rdA[1]);          WGS_DMA_A(2, la, sa, oob); \
                __builtin_amdgcn_sched_barrier(0);                                               \
                WGS_MM(CA, CB, 6); WGS_MM(CA, CB, 7); NB_[1] = wgs_frag(rb + rdB[1]);   WGS_DMA_A(3, la, sa, oob); \
                __builtin_amdgcn_sched_barrier(0);                                               \
                WGS_MM(CA, CB, 8); WGS_MM(CA, CB, 9); NA_[2] = wgs_frag(rb + rdA[2]);          WGS_DMA_B(0, la, sb, oob); \
                __builtin_amdgcn_sched_barrier(0);                                               \
                WGS_MM(CA, CB, 10); WGS_MM(CA, CB, 11); NB_[2] = wgs_frag(rb + rdB[2]); WGS_DMA_B(1, la, sb, oob); \
                __builtin_amdgcn_sched_barrier(0);                                               \
                WGS_MM(CA, CB, 12); WGS_MM(CA, CB, 13); NA_[3] = wgs_frag(rb + rdA[3]);        WGS_DMA_B(2, la, sb, oob); \
                __builtin_amdgcn_sched_barrier(0);                                               \
                WGS_MM(CA, CB, 14); WGS_MM(CA, CB, 15); NB_[3] = wgs_frag(rb + rdB[3]); WGS_DMA_B(3, la, sb, oob); \
                __builtin_amdgcn_sched_barrier(0);                                               \
            }
            for (int k = k0; k < k1; k += 2) {
                WGS_STEP(k, fa0, fb0, fa1, fb1)
                WGS_STEP(k + 1, fa1, fb1, fa0, fb0)
            }
#undef WGS_STEP
#undef WGS_MM
            YH_VMCNT(0);       // the dummies behind the last stage
        }
#undef WGS_DMA_A
#undef WGS_DMA_B

        // ---- combine the four partial tiles through LDS, reduce-scatter in two rounds of plain 16-byte stores / loads (LDS float
        // atomics serialise per lane: 150 000 cycles for this exchange).  Round 1: waves w and w^1 split the tile over x (n):
        // each stores the half it gives away into its own ring (32 KB: nothing of it is in flight) and adds the partner's copy of
        // the half it keeps; round 2: w and w^2 split what is left over y (columns).  The kept data is acc[0..1][0..1] in every wave.
        // Every sub-tile is fenced for the scheduler: at most 16 loaded values are live at a time.
        float4* const mine = reinterpret_cast<float4*>(wbase) + lane;
        const float4* const px = reinterpret_cast<const float4*>(smem + (wave ^ 1) * WGS_WAVE_LDS) + lane;
        const float4* const py = reinterpret_cast<const float4*>(smem + (wave ^ 2) * WGS_WAVE_LDS) + lane;
        auto put = [&](int sidx, const f32x16_t& a) {
#pragma unroll
            for (int r4 = 0; r4 < 4; ++r4) mine[(sidx * 4 + r4) * 64] = make_float4(a[4 * r4], a[4 * r4 + 1], a[4 * r4 + 2], a[4 * r4 + 3]);
            __builtin_amdgcn_sched_barrier(0);
        };
        // a += the partner's copy.  The accumulators live in AGPRs and VALU reads none: written as plain C the allocator copies the
        // whole kept half (128 values) to arch VGPRs at the loop exit, and the kernel's STATIC register allocation — what decides
        // which other waves may share the SIMD with this one — grows by them.  So the add is spelled out per element:
        // v_accvgpr_read -> v_add_f32 -> v_accvgpr_write through four temporaries.
        auto add = [&](const float4* from, int sidx, f32x16_t& a) {
            float4 ov[4];
#pragma unroll
            for (int r4 = 0; r4 < 4; ++r4) ov[r4] = from[(sidx * 4 + r4) * 64];
#pragma unroll
            for (int r4 = 0; r4 < 4; ++r4) {
                const float4 o = ov[r4];
                float t0, t1, t2, t3;
                asm volatile("v_accvgpr_read_b32 %4, %0\n\tv_accvgpr_read_b32 %5, %1\n\tv_accvgpr_read_b32 %6, %2\n\tv_accvgpr_read_b32 %7, %3\n\t"
                             "v_add_f32 %4, %4, %8\n\tv_add_f32 %5, %5, %9\n\tv_add_f32 %6, %6, %10\n\tv_add_f32 %7, %7, %11\n\t"
                             "v_accvgpr_write_b32 %0, %4\n\tv_accvgpr_write_b32 %1, %5\n\tv_accvgpr_write_b32 %2, %6\n\tv_accvgpr_write_b32 %3, %7"
                             : "+a"(a[4 * r4]), "+a"(a[4 * r4 + 1]), "+a"(a[4 * r4 + 2]), "+a"(a[4 * r4 + 3]), "=&v"(t0), "=&v"(t1), "=&v"(t2), "=&v"(t3)
                             : "v"(o.x), "v"(o.y), "v"(o.z), "v"(o.w));
            }
            __builtin_amdgcn_sched_barrier(0);
        };
        WGS_STAMP(2);
#pragma unroll
        for (int i = 0; i < 8; ++i) put(i, acc[2 + (i >> 2)][i & 3]);
        __syncthreads();
        WGS_STAMP(3);
#pragma unroll
        for (int i = 0; i < 8; ++i) add(px, i, acc[i >> 2][i & 3]);
        __syncthreads();       // the partner has read my round-1 bytes
#pragma unroll
        for (int i = 0; i < 4; ++i) put(i, acc[i >> 1][2 + (i & 1)]);
        __syncthreads();
#pragma unroll
        for (int i = 0; i < 4; ++i) add(py, i, acc[i >> 1][i & 1]);
        WGS_STAMP(4);

        // ---- this wave's quarter (x = 2 hx + xx, y = 2 hy + yy) -> dw.  C[n][col]: the lane holds column lane & 31, register r
        // holds row (r & 3) + 8 (r >> 2) + 4 (lane >> 5): one wave instruction adds two 128-byte row segments.  Buffer atomics:
        // the row walk is a SCALAR offset (no address arithmetic per add, one offset register), rows past N carry an
        // out-of-range offset instead of an exec mask.
        {
            const int nl = n0 + 64 * hx + 4 * (lane >> 5);
            // the two 32-column groups of this wave's quarter: tap and channel of a group are wave-uniform (32 | C)
            const int col0 = ctile * 128 + 64 * hy, col1 = col0 + 32;
            const int tap0 = col0 / segC, tap1 = col1 / segC;
            const int ntap = d.KH * d.KW;
            const unsigned voff0 = (unsigned)((nl * p.Ktot + tap0 * d.Ctot + coffk + (col0 - tap0 * segC) + (lane & 31)) * 4);
            const unsigned voff1 = (unsigned)((nl * p.Ktot + tap1 * d.Ctot + coffk + (col1 - tap1 * segC) + (lane & 31)) * 4);
            const unsigned dead0 = tap0 < ntap ? 0u : WGS_OOB, dead1 = tap1 < ntap ? 0u : WGS_OOB;    // columns past the last tap
#pragma unroll
            for (int xx = 0; xx < 2; ++xx) {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int ro = 32 * xx + (r & 3) + 8 * (r >> 2);
                    const unsigned rowdead = (nl + ro < d.N) ? 0u : WGS_OOB;
                    const int so = ro * p.Ktot * 4;
                    __builtin_amdgcn_raw_ptr_buffer_atomic_fadd_f32(acc[xx][0][r], rsd, voff0 | dead0 | rowdead, so, 0);
                    __builtin_amdgcn_raw_ptr_buffer_atomic_fadd_f32(acc[xx][1][r], rsd, voff1 | dead1 | rowdead, so, 0);
                    if ((r & 3) == 3) __builtin_amdgcn_sched_barrier(0);
                }
            }
        }
        WGS_STAMP(5);
        if (p.stamps) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
        WGS_STAMP(6);
        __syncthreads();       // the partner has read my round-2 bytes: the areas are rings again
        first = false;
    }
#undef WGS_STAMP
}

template <bool PW>
__global__ __launch_bounds__(256, 1) void conv_wgs_kernel(const WgsK p)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    wgs_body<PW>(p, blockIdx.x, smem);
}

// eligibility of the layer and the launch plan shared by the queries and the launcher
struct WgsPlan { long M; int nk, ntn, nct, T, G, S, grid; bool pw; unsigned long xb; };

bool wgs_seg_ok(const yh_wgrad_desc* d, const yh_seg& sg, int coff, unsigned long* xb)
{
    if (sg.C <= 0 || sg.C % 32 != 0 || sg.ld % 8 != 0 || coff % 8 != 0 || coff < 0 || coff + sg.C > d->Ctot) return false;
    if (sg.ups && (d->Hi % 2 || d->Wi % 2)) return false;
    const unsigned long npix = (unsigned long)d->B * (d->Hi >> sg.ups) * (d->Wi >> sg.ups);
    *xb = ((npix - 1) * sg.ld + sg.C) * 2;
    return *xb < (1ul << 31) - 4096;
}

bool wgs_plan(const yh_wgrad_desc* d, WgsPlan* pl)
{
    if (!d || d->B <= 0 || d->Ho <= 0 || d->Wo <= 0 || d->Hi <= 0 || d->Wi <= 0 || d->N < 64 || d->KH <= 0 || d->KW <= 0) return false;
    if (d->KH > 7 || d->KW > 7 || (d->stride != 1 && d->stride != 2) || d->pad < 0) return false;
    if (d->ldg % 8 != 0 || d->ldg < (d->N + 7) / 8 * 8) return false;
    if (!wgs_seg_ok(d, d->seg, d->coff_k, &pl->xb)) return false;
    if (d->bn_z || d->partial) return false;
    if ((d->Hi + 2 * d->pad - d->KH) / d->stride + 1 != d->Ho || (d->Wi + 2 * d->pad - d->KW) / d->stride + 1 != d->Wo) return false;
    const long M = (long)d->B * d->Ho * d->Wo;
    if (M % 32 != 0 || M >= (1L << 31) - 256 || d->Ho * d->Wo < 16) return false;
    const unsigned long gyb = ((unsigned long)(M - 1) * d->ldg + (d->N + 7) / 8 * 8) * 2;
    if (gyb >= (1ul << 31) - 4096) return false;
    if ((unsigned long)(d->N + 127) * d->KH * d->KW * d->Ctot * 4 >= (1ul << 31)) return false;      // dw is addressed with 32-bit offsets
    pl->M = M;
    pl->nk = (int)(M / 32);
    pl->ntn = (d->N + 127) / 128;
    pl->nct = (d->KH * d->KW * d->seg.C + 127) / 128;
    pl->T = pl->ntn * pl->nct;
    pl->pw = d->KH == 1 && d->KW == 1 && d->stride == 1 && d->pad == 0 && !d->seg.ups;
    int G = d->splits < 1 ? 1 : (d->splits > 4096 ? 4096 : d->splits);
    const long U = (long)pl->T * pl->nk;
    if ((long)G > U) G = (int)U;
    pl->G = G;
    pl->S = (G % pl->T == 0) ? G / pl->T : 0;
    pl->grid = G;
    return true;
}

}  // namespace

static unsigned long long* g_wgs_stamps = nullptr;
/* diagnostics: a device buffer of grid x 4 x 8 uint64 that receives shader-clock stamps of every wave's first segment (NULL: off) */
extern "C" void yh_wgs_set_stamps(void* p) { g_wgs_stamps = (unsigned long long*)p; }

int yh_wgs_ok(const yh_wgrad_desc* d) { WgsPlan pl; return wgs_plan(d, &pl) ? 1 : 0; }
/* tiles (128 out channels x 128 im2col columns) of the layer: workgroups = `splits` are dealt (tile, 32 pixels) units */
int yh_wgs_tiles(const yh_wgrad_desc* d) { WgsPlan pl; return wgs_plan(d, &pl) ? pl.T : 0; }
const char* yh_wgs_name(const yh_wgrad_desc* d)
{
    WgsPlan pl;
    if (!wgs_plan(d, &pl)) return "";
    return pl.pw ? "conv_wgs_kernel<true>" : "conv_wgs_kernel<false>";
}

// kernel parameters of one layer on G workgroups (block map, descriptor ranges, diagnostics)
static void wgs_fill(const yh_wgrad_desc* d, const WgsPlan& pl, int G, WgsK* kp)
{
    WgsK& k = *kp;
    k.d = *d;
    k.stamps = g_wgs_stamps;
    k.Ktot = d->KH * d->KW * d->Ctot;
    k.nk = pl.nk; k.nct = pl.nct; k.T = pl.T; k.G = G;
    {   // phase-sorted block map: see the kernel.  YH_WGS_SKMAP=0 keeps the identity map on stream-K grids (A/B)
        static const int skm = getenv("YH_WGS_SKMAP") ? atoi(getenv("YH_WGS_SKMAP")) : 1;
        int a = pl.T, b = G;
        while (b) { const int t = a % b; a = b; b = t; }
        k.mg = a; k.mGp = G / a; k.minv = 0;
        const int tp = (pl.T / a) % k.mGp;
        for (int i = 1; i < k.mGp; ++i) if ((long)tp * i % k.mGp == 1) { k.minv = i; break; }
        if (G % pl.T != 0 && !skm) k.mg = 0;
    }
    k.U = (long)pl.T * pl.nk;
    k.gybytes = (unsigned)(((unsigned long)(pl.M - 1) * d->ldg + (d->N + 7) / 8 * 8) * 2);
    k.xbytes = (unsigned)pl.xb;
    k.dwbytes = (unsigned)((unsigned long)d->N * d->KH * d->KW * d->Ctot * 4);
    k.pw = pl.pw ? 1 : 0;
    // timing-only diagnostics (results wrong): YH_WGS_ABL bit 0: zero-record operand descriptors (every transfer returns zeros without
    // touching memory: the loop's issue-bound time), bit 1: zero-record dw descriptor (the atomics are dropped by the range check)
    static const int abl = [] { const char* e = getenv("YH_WGS_ABL"); return e ? atoi(e) : 0; }();
    if (abl & 1) k.gybytes = k.xbytes = 0;
    if (abl & 2) k.dwbytes = 0;
}

static void wgs_attrs()
{
    static YhDevOnce attr_set;      
    if (attr_set.need()) {
        attr_set.set((const void*)conv_wgs_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, WGS_LDS);
        attr_set.set((const void*)conv_wgs_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, WGS_LDS);
        attr_set.done(); 
    }
}

int yh_wgs_run(const yh_wgrad_desc* d, yh_stream stream)
{
    WgsPlan pl;
    YH_CHECK_ARG(wgs_plan(d, &pl), "yh_conv_wgrad(tile_k 129): layer not eligible");
    YH_CHECK_ARG(d->gy && yh_aligned16(d->gy) && d->seg.ptr && yh_aligned16(d->seg.ptr) && d->dw,
                 "yh_conv_wgrad(tile_k 129): bad operands");
    WgsK k;
    wgs_fill(d, pl, pl.G, &k);
    wgs_attrs();
    hipStream_t st = (hipStream_t)stream;
    if (pl.pw) conv_wgs_kernel<true><<<dim3(pl.grid), dim3(256), WGS_LDS, st>>>(k);
    else       conv_wgs_kernel<false><<<dim3(pl.grid), dim3(256), WGS_LDS, st>>>(k);
    YH_CHECK_LAUNCH("yh_conv_wgrad(tile_k 129)");
    return YH_OK;
}
