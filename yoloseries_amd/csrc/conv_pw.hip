// 1x1 convolution (pointwise: C3's cba1 / cba2 / cba3 and the bottlenecks' conv_bn_act_1, utils/layer_tools.py:90-114,152-169)
// for the 80- / 160- / 320-channel layers of YOLOv5x at 320 x 320 ... 80 x 80 (1280 x 1280 input), inference epilogues.  These layers move
// 40-160 FLOP per byte: they are bound by HBM, and the implicit-GEMM tiles (128 output channels, 64-channel k-steps, a block-wide
// barrier per k-step) ran them at 2.8 TB/s.  Here the reduction is so short (80 / 160) that a tile has NO k loop over memory:
//
//   * a block (4 waves) owns 128 (C = 80) or 64 (C = 160, 320) consecutive pixels; their C input channels arrive by LDS-DMA as one piece (rows of C/8 chunks + one
//     zero pad chunk: an odd pitch, conflict-free 16-byte fragment reads), double buffered across the tiles of a persistent block;
//   * the 80 x C weight tile of the block's output-channel group stays in LDS for the whole launch;
//   * a wave multiplies 32 / 16 pixels x 80 channels (2 / 1 x 5 tiles of v_mfma_f32_16x16x32_bf16, operands swapped: a lane holds one pixel and
//     four consecutive channels per accumulator) in ceil(C / 32) steps without a barrier; the last step of C = 80 reads the zero pad chunk
//     for the reduction indices past the end;
//   * bias / folded BatchNorm / SiLU in registers, the tile through LDS, whole 160-byte rows to memory.
// Blocks (x, y) and (x, y + 1) — the two output-channel groups of a 160-channel layer — walk the same pixel tiles at the same time and the
// grid width is a multiple of 8, so both sit on one XCD and the second read of a tile comes from its L2.
// Chosen per layer by the engine's timing (yh_conv_desc.algo 10).
#include "common.h"
#include <stdlib.h>

namespace {

struct PwK {
    const uint16_t* x; int ldx;
    const uint16_t* w; int Ktot;
    uint16_t* out0; int ld0;
    uint16_t* out1; int ld1; int nsplit;
    const uint16_t* res; int ldr;
    const float* bias; const float* scale; const float* shift;
    int act, accumulate;
    int N, M, ntiles;
    unsigned wbytes;
};

template <int CIN, int TNC, int TMW>
struct PwCfg {
    static constexpr int CH = CIN / 8;                       // 16-byte chunks per pixel
    static constexpr int PCH = CH + 1;                       // + one zero pad chunk; odd pitch: 16 consecutive rows hit 16 different bank groups
    static constexpr int PITCH = PCH * 16;
    static constexpr int TM = 64 * TMW;                      // pixels per tile: 4 waves x TMW groups of 16
    static constexpr int AINST = (TM * PCH + 63) / 64;       // DMA instructions per pixel tile
    static constexpr int A_BYTES = AINST * 1024;
    static constexpr int TN = TNC * 16;
    static constexpr int WINST = (TN * PCH + 63) / 64;
    static constexpr int W_BYTES = WINST * 1024;
    static constexpr int NKS = (CH + 3) / 4;                 // MFMA steps of 32 reduction indices
    static constexpr int CP = TN + 8;
    static constexpr int CONST_OFF = 2 * A_BYTES + W_BYTES;
    static constexpr int SMEM = CONST_OFF + 3 * TN * 4;
    static_assert(CIN % 16 == 0 && CH % 2 == 0, "even chunk count (the pad chunk makes the pitch odd)");
    static_assert(TM * CP * 2 <= A_BYTES, "output tile must fit a pixel-tile buffer");
    static constexpr int MINB = 2 * SMEM <= 160 * 1024 ? 2 : 1;       // blocks per CU
    static_assert(SMEM <= 160 * 1024, "LDS budget");
};

__device__ __forceinline__ bf16x8_t pw_lds16(const unsigned char* smem, int off) {
    return __builtin_bit_cast(bf16x8_t, *reinterpret_cast<const uint4*>(smem + off));
}

template <int CIN, int TNC, int TMW, int EPI>
__global__ __launch_bounds__(256, (PwCfg<CIN, TNC, TMW>::MINB)) void conv_pw_kernel(const PwK p)
{
    using G = PwCfg<CIN, TNC, TMW>;
    constexpr int CH = G::CH, PCH = G::PCH, PITCH = G::PITCH, TM = G::TM, AINST = G::AINST, A_BYTES = G::A_BYTES;
    constexpr int TN = G::TN, WINST = G::WINST, NKS = G::NKS, CP = G::CP, NT = 256, NWV = 4;
    constexpr int NAI = (AINST + NWV - 1) / NWV, NWI = (WINST + NWV - 1) / NWV;
    constexpr unsigned OOB = 0x80000000u;
    static_assert(EPI == 0 || EPI == 2, "inference epilogues only");

    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    float* sConst = reinterpret_cast<float*>(smem + G::CONST_OFF);                  // [3][TN]: bias | scale | shift

    const int t = threadIdx.x;
    const int lane = t & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int l15 = lane & 15, kq = lane >> 4;
    const int n0 = blockIdx.y * TN;
    const int ldx2 = p.ldx * 2;

    // DMA slot g = inst * 64 + lane -> (row, chunk); the pad chunk and the slots past the tile are never in range
    unsigned arel[NAI];
#pragma unroll
    for (int h = 0; h < NAI; ++h) {
        const int g = (h * NWV + wave) * 64 + lane;
        const int row = g / PCH, c = g - row * PCH;
        arel[h] = (row < TM && c < CH) ? (unsigned)(row * ldx2 + c * 16) : OOB;
    }
    // fragment offsets inside a row: chunk 4 ks + kq, the zero pad chunk for the reduction indices past the end
    int koff[NKS];
#pragma unroll
    for (int ks = 0; ks < NKS; ++ks) koff[ks] = (4 * ks + kq < CH ? 4 * ks + kq : CH) * 16;

    auto issue_A = [&](int tile, int buf) {
        const long m0 = (long)tile * TM;
        const long rows = (long)p.M - m0 < TM ? (long)p.M - m0 : TM;
        // the descriptor covers exactly the tile's rows: a ragged last tile reads zeros behind the last pixel
        const __amdgpu_buffer_rsrc_t rsx = __builtin_amdgcn_make_buffer_rsrc((void*)(p.x + (size_t)m0 * p.ldx), 0,
                                                                             (unsigned)((rows - 1) * ldx2 + CIN * 2), 0x00020000);
#pragma unroll
        for (int h = 0; h < NAI; ++h)
            if (h * NWV + wave < AINST) lds_dma16(rsx, smem + buf * A_BYTES + (h * NWV + wave) * 1024, arel[h], 0);
    };

    int tile = blockIdx.x;
    if (tile < p.ntiles) {
        const __amdgpu_buffer_rsrc_t rsw = __builtin_amdgcn_make_buffer_rsrc((void*)p.w, 0, p.wbytes, 0x00020000);
#pragma unroll
        for (int h = 0; h < NWI; ++h) {
            const int inst = h * NWV + wave;
            const int g = inst * 64 + lane;
            const int row = g / PCH, c = g - row * PCH;
            const unsigned vo = (row < TN && c < CH) ? (unsigned)(((n0 + row) * p.Ktot + c * 8) * 2) : OOB;
            if (inst < WINST) lds_dma16(rsw, smem + 2 * A_BYTES + inst * 1024, vo, 0);
        }
        issue_A(tile, 0);
    }
    if (EPI == 2) {
        for (int i = t; i < 3 * TN; i += NT) {
            const int which = i / TN, c = i - which * TN;
            const float* src = which == 0 ? p.bias : (which == 1 ? p.scale : p.shift);
            sConst[i] = (src && n0 + c < p.N) ? src[n0 + c] : (which == 1 ? 1.f : 0.f);
        }
    }
    __syncthreads();                               // weights, first tile, constants (drains the DMAs: once per block)

    const int wbase = 2 * A_BYTES + l15 * PITCH;
    int pb = 0;
    constexpr int CPR = TN / 8;
    constexpr int NOI = (TM * CPR + NT - 1) / NT;

    for (; tile < p.ntiles; tile += gridDim.x) {
        const bool has_next = tile + (int)gridDim.x < p.ntiles;
        if (has_next) issue_A(tile + gridDim.x, pb ^ 1);                  // a tile's time to land; waited for ahead of this tile's stores
        const long m0 = (long)tile * TM;

        // residual chunks of this thread's output rows: requested now, used behind the activation math
        long orow[NOI];
        uint4 rv[EPI == 2 ? NOI : 1];
#pragma unroll
        for (int it = 0; it < NOI; ++it) {
            const int id = t + it * NT;
            const int row = id / CPR;
            const int n = n0 + (id - row * CPR) * 8;
            orow[it] = (id < TM * CPR && m0 + row < p.M) ? m0 + row : -1;
            if (EPI == 2) {
                rv[it] = make_uint4(0, 0, 0, 0);
                if (p.res != nullptr && orow[it] >= 0 && n < p.nsplit) rv[it] = *reinterpret_cast<const uint4*>(p.res + (size_t)orow[it] * p.ldr + n);
            }
        }

        f32x4_t acc[TMW][TNC];
#pragma unroll
        for (int i = 0; i < TMW; ++i)
#pragma unroll
            for (int j = 0; j < TNC; ++j)
#pragma unroll
                for (int r = 0; r < 4; ++r) acc[i][j][r] = 0.f;
        const int xbase = pb * A_BYTES + (wave * (16 * TMW) + l15) * PITCH;
#pragma unroll
        for (int ks = 0; ks < NKS; ++ks) {
            bf16x8_t xf[TMW], wf[TNC];
#pragma unroll
            for (int i = 0; i < TMW; ++i) xf[i] = pw_lds16(smem, xbase + i * 16 * PITCH + koff[ks]);
#pragma unroll
            for (int j = 0; j < TNC; ++j) wf[j] = pw_lds16(smem, wbase + j * 16 * PITCH + koff[ks]);
#pragma unroll
            for (int i = 0; i < TMW; ++i)
#pragma unroll
                for (int j = 0; j < TNC; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[j], xf[i], acc[i][j], 0, 0, 0);
        }

        // ---- epilogue: bias / folded BatchNorm / SiLU in registers, the tile through the (now idle) pixel buffer, whole rows to memory
        uint16_t* sC = reinterpret_cast<uint16_t*>(smem + pb * A_BYTES);
        YH_LDS_BARRIER();                                                 // every wave has read its fragments
#pragma unroll
        for (int j = 0; j < TNC; ++j) {
            const int cc = j * 16 + 4 * kq;
            float4 cb = make_float4(0.f, 0.f, 0.f, 0.f), cs = make_float4(1.f, 1.f, 1.f, 1.f), ct = cb;
            if (EPI == 2) {
                cb = *reinterpret_cast<const float4*>(sConst + cc);
                cs = *reinterpret_cast<const float4*>(sConst + TN + cc);
                ct = *reinterpret_cast<const float4*>(sConst + 2 * TN + cc);
            }
#pragma unroll
            for (int i = 0; i < TMW; ++i) {
                float v0 = acc[i][j][0], v1 = acc[i][j][1], v2 = acc[i][j][2], v3 = acc[i][j][3];
                if (EPI == 2) {
                    v0 = (v0 + cb.x) * cs.x + ct.x; v1 = (v1 + cb.y) * cs.y + ct.y;
                    v2 = (v2 + cb.z) * cs.z + ct.z; v3 = (v3 + cb.w) * cs.w + ct.w;
                    if (p.act == YH_ACT_SILU) { v0 = silu_fast(v0); v1 = silu_fast(v1); v2 = silu_fast(v2); v3 = silu_fast(v3); }
                }
                *reinterpret_cast<uint2*>(sC + (wave * (16 * TMW) + i * 16 + l15) * CP + cc) = make_uint2(pack2(v0, v1), pack2(v2, v3));
            }
        }
        YH_LDS_BARRIER();
        YH_VMCNT(0);                               // next tile and residual landed; the stores below leave the counter clean
#pragma unroll
        for (int it = 0; it < NOI; ++it) {
            if (orow[it] < 0) continue;
            const int id = t + it * NT;
            const int row = id / CPR;
            const int cch = id - row * CPR;
            const int n = n0 + cch * 8;
            const size_t orw = (size_t)orow[it];
            uint4 v = *reinterpret_cast<const uint4*>(sC + row * CP + cch * 8);
            if (EPI == 2) {
                const bool first = n < p.nsplit;
                uint16_t* dst = first ? p.out0 + orw * p.ld0 + n : p.out1 + orw * p.ld1 + (n - p.nsplit);
                const bool addres = (p.res != nullptr) && first;
                if (addres || p.accumulate) {
                    float f[8];
                    unpack8(v, f);
                    if (addres) {
                        float g2[8]; unpack8(rv[it], g2);
#pragma unroll
                        for (int e = 0; e < 8; ++e) f[e] += g2[e];
                    }
                    if (p.accumulate) {
                        const uint4 ov = *reinterpret_cast<const uint4*>(dst);
                        float g2[8]; unpack8(ov, g2);
#pragma unroll
                        for (int e = 0; e < 8; ++e) f[e] += g2[e];
                    }
                    v = pack8(f);
                }
                *reinterpret_cast<uint4*>(dst) = v;
            } else {
                *reinterpret_cast<uint4*>(p.out0 + orw * p.ld0 + n) = v;
            }
        }
        YH_LDS_BARRIER();                          // output buffer free; the next tile (every wave waited for its own DMAs) visible
        pb ^= 1;
    }
}

struct PwPlan { int gx, gy, epi, cin; PwK k; };

bool pw_plan(const yh_conv_desc* d, PwPlan* pl)
{
    if (d->nseg != 1 || d->seg[0].ups || d->mode != YH_CONV_FWD) return false;
    if (d->KH != 1 || d->KW != 1 || d->stride != 1 || d->pad != 0) return false;
    if (d->stats || d->bnr_part) return false;                       // inference epilogues only
    const int C = d->seg[0].C;
    if ((C != 80 && C != 160 && C != 320) || d->N % 80 || d->N <= 0) return false;
    const unsigned long M = (unsigned long)d->B * d->Ho * d->Wo;
    const int tm = C == 80 ? 128 : 64;
    const unsigned long wb = (unsigned long)d->Npad * C * 2;
    if (M >= (1ul << 31) || wb >= (1ul << 31) || (unsigned long)d->seg[0].ld * 2 * 128 >= (1ul << 31)) return false;
    PwK& k = pl->k;
    k.x = d->seg[0].ptr; k.ldx = d->seg[0].ld;
    k.w = d->w; k.Ktot = C;
    k.out0 = d->out0; k.ld0 = d->ld0; k.out1 = d->out1; k.ld1 = d->ld1; k.nsplit = d->nsplit;
    k.res = d->res; k.ldr = d->ldr;
    k.bias = d->bias; k.scale = d->scale; k.shift = d->shift;
    k.act = d->act; k.accumulate = d->accumulate;
    k.N = d->N; k.M = (int)M;
    k.ntiles = (int)((M + tm - 1) / tm);
    k.wbytes = (unsigned)wb;
    const bool generic = d->bias || d->scale || d->shift || d->act != YH_ACT_NONE || d->res || d->accumulate || d->nsplit < d->N;
    pl->epi = generic ? 2 : 0;
    pl->cin = C;
    pl->gy = d->N / 80;
    int cap = ((C == 320 ? 256 : 512) / pl->gy) & ~7;    // two blocks per CU (one with 320 channels); a multiple of 8: the output-channel groups of a tile share an XCD
    if (cap < 8) cap = 8;
    if (d->grid_cap > 0) cap = d->grid_cap;
    pl->gx = k.ntiles < cap ? k.ntiles : cap;
    return true;
}

}  // namespace

int yh_pw_rows(const yh_conv_desc* d)
{
    PwPlan pl;
    return pw_plan(d, &pl) ? pl.gx : 0;
}

int yh_pw_run(const yh_conv_desc* d, yh_stream stream, char* name_out, int name_len)
{
    PwPlan pl;
    YH_CHECK_ARG(pw_plan(d, &pl), "yh_conv_igemm: algo 10 (pointwise kernel) is not eligible for this descriptor");
    if (name_out) { snprintf(name_out, name_len, "conv_pw_kernel<%d, 5, %d, %d>", pl.cin, pl.cin == 80 ? 2 : 1, pl.epi); return YH_OK; }
    hipStream_t st = (hipStream_t)stream;
    const dim3 grid(pl.gx, pl.gy), blk(256);
#define YH_LAUNCH_PW(CIN_, TMW_)                                                                                           \
    do {                                                                                                               \
        constexpr int sm = PwCfg<CIN_, 5, TMW_>::SMEM;                                                                    \
        static YhDevOnce attr_set;                                                                                        \
        if (attr_set.need()) {                                                                                               \
            attr_set.set((const void*)conv_pw_kernel<CIN_, 5, TMW_, 0>, hipFuncAttributeMaxDynamicSharedMemorySize, sm); \
            attr_set.set((const void*)conv_pw_kernel<CIN_, 5, TMW_, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, sm); \
            attr_set.done();                                                                                            \
        }                                                                                                              \
        if (pl.epi == 2) conv_pw_kernel<CIN_, 5, TMW_, 2><<<grid, blk, sm, st>>>(pl.k);                                      \
        else             conv_pw_kernel<CIN_, 5, TMW_, 0><<<grid, blk, sm, st>>>(pl.k);                                      \
    } while (0)
    if (pl.cin == 80) YH_LAUNCH_PW(80, 2); else if (pl.cin == 160) YH_LAUNCH_PW(160, 1); else YH_LAUNCH_PW(320, 1);
#undef YH_LAUNCH_PW
    YH_CHECK_LAUNCH("yh_conv_igemm(pw)");
    return YH_OK;
}
