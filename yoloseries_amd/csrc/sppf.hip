// FastSPP's chain of three 5x5 / stride-1 / pad-2 max-pools (utils/layer_tools.py:282-288: x2 = mp(x1), x3 = mp(x2), x4 = mp(x3))
// in ONE launch per direction.  A block owns one image x 32 channels (16 / 8 on larger maps): the whole map (<= 480 / 960 / 1920
// pixels: 20 x 20 at 640^2 input, 40 x 40 at 1280^2) lives in LDS, so x1 is read from HBM once, the three pools run separably (row maxima, then column maxima: 10 comparisons per output
// instead of 25 — the single-pool kernel is VALU-bound on its 25-tap scan) and a pool's output is the next pool's input without a
// round trip.  Results are bit-identical to three yh_maxpool5_fwd / yh_maxpool5_bwd launches: same arg-max (first maximum in
// row-major window order; a NaN takes over and the last NaN wins, as the one-pass scan does — the separable scan provably ends on
// the same element), same fp32 summation order and bf16 rounding points in the backward.
#include "common.h"

namespace {

constexpr int SP_NT = 512;
constexpr int SP_MAXPX = 480;        // map pixels with 32 channels per block (4 chunks of 16 bytes per pixel): 64 KB of LDS at 20 x 20, two blocks per CU;
                                     // twice / four times the pixels with 16 / 8 channels per block (template parameter SP_CH)

// one step of the scan: the current best (value, position) against a new value
__device__ __forceinline__ void scan_step(float f, int pos, bool& any, float& best, int& bi) {
    if (!any || f > best || f != f) { best = f; bi = pos; }
    any = true;
}

template <int SP_CH>
__global__ __launch_bounds__(SP_NT) void sppf_pool3_fwd_kernel(const uint16_t* __restrict__ x, int ldx, int H, int W, int C,
                                                               uint16_t* __restrict__ o1, uint16_t* __restrict__ o2, uint16_t* __restrict__ o3, int ldo,
                                                               int8_t* __restrict__ i1, int8_t* __restrict__ i2, int8_t* __restrict__ i3)
{
    constexpr int SP_CPP = SP_CH / 8;    // 16-byte chunks per pixel
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int HW = H * W;
    uint16_t* sV0 = reinterpret_cast<uint16_t*>(smem);                 // [HW][SP_CH] current pool input
    uint16_t* sV1 = sV0 + (size_t)HW * SP_CH;                           // [HW][SP_CH] row maxima
    uint8_t* sI = reinterpret_cast<uint8_t*>(sV1 + (size_t)HW * SP_CH); // [HW][SP_CH] column offset (0..4) of each row maximum
    const int b = blockIdx.x, c0 = blockIdx.y * SP_CH;
    const int t = threadIdx.x, chunk = t % SP_CPP;
    const bool cok = c0 + chunk * 8 < C;
    const size_t img = (size_t)b * HW;
    for (int px = t / SP_CPP; px < HW; px += SP_NT / SP_CPP) {
        uint4 v = make_uint4(0, 0, 0, 0);
        if (cok) v = *reinterpret_cast<const uint4*>(x + (img + px) * ldx + c0 + chunk * 8);
        *reinterpret_cast<uint4*>(sV0 + px * SP_CH + chunk * 8) = v;
    }
    __syncthreads();
    for (int pool = 0; pool < 3; ++pool) {
        uint16_t* out = pool == 0 ? o1 : (pool == 1 ? o2 : o3);
        int8_t* idx = pool == 0 ? i1 : (pool == 1 ? i2 : i3);
        // rows: maximum over the columns j-2 .. j+2 of the pixel's own row, first maximum in column order
        for (int px = t / SP_CPP; px < HW; px += SP_NT / SP_CPP) {
            const int i = px / W, j = px - i * W;
            float best[8]; int bi[8]; bool any[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) { best[e] = -INFINITY; bi[e] = 0; any[e] = false; }
#pragma unroll
            for (int dj = 0; dj < 5; ++dj) {
                const int jj = j - 2 + dj;
                if (jj < 0 || jj >= W) continue;
                float f[8];
                unpack8(*reinterpret_cast<const uint4*>(sV0 + (i * W + jj) * SP_CH + chunk * 8), f);
#pragma unroll
                for (int e = 0; e < 8; ++e) scan_step(f[e], dj, any[e], best[e], bi[e]);
            }
            *reinterpret_cast<uint4*>(sV1 + px * SP_CH + chunk * 8) = pack8(best);
            uint2 pk;
            pk.x = (uint32_t)bi[0] | ((uint32_t)bi[1] << 8) | ((uint32_t)bi[2] << 16) | ((uint32_t)bi[3] << 24);
            pk.y = (uint32_t)bi[4] | ((uint32_t)bi[5] << 8) | ((uint32_t)bi[6] << 16) | ((uint32_t)bi[7] << 24);
            *reinterpret_cast<uint2*>(sI + px * SP_CH + chunk * 8) = pk;
        }
        __syncthreads();
        // columns: maximum over the rows i-2 .. i+2 of the row maxima, first row wins; arg-max = row offset * 5 + that row's column offset
        for (int px = t / SP_CPP; px < HW; px += SP_NT / SP_CPP) {
            const int i = px / W, j = px - i * W;
            float best[8]; int bi[8]; bool any[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) { best[e] = -INFINITY; bi[e] = 0; any[e] = false; }
#pragma unroll
            for (int di = 0; di < 5; ++di) {
                const int ii = i - 2 + di;
                if (ii < 0 || ii >= H) continue;
                float f[8];
                unpack8(*reinterpret_cast<const uint4*>(sV1 + (ii * W + j) * SP_CH + chunk * 8), f);
                const uint2 pk = *reinterpret_cast<const uint2*>(sI + (ii * W + j) * SP_CH + chunk * 8);
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const int cj = (int)(((e < 4 ? pk.x : pk.y) >> (8 * (e & 3))) & 0xff);
                    scan_step(f[e], di * 5 + cj, any[e], best[e], bi[e]);
                }
            }
            const uint4 ov = pack8(best);
            *reinterpret_cast<uint4*>(sV0 + px * SP_CH + chunk * 8) = ov;          // the next pool's input (all row passes are done)
            if (cok) {
                *reinterpret_cast<uint4*>(out + (img + px) * ldo + c0 + chunk * 8) = ov;
                if (idx) {
                    uint2 pk;
                    pk.x = (uint32_t)bi[0] | ((uint32_t)bi[1] << 8) | ((uint32_t)bi[2] << 16) | ((uint32_t)bi[3] << 24);
                    pk.y = (uint32_t)bi[4] | ((uint32_t)bi[5] << 8) | ((uint32_t)bi[6] << 16) | ((uint32_t)bi[7] << 24);
                    *reinterpret_cast<uint2*>(idx + (img + px) * C + c0 + chunk * 8) = pk;
                }
            }
        }
        __syncthreads();
    }
}

// backward of the chain: t = g(x4);  g(x3)' = g(x3) + P3^T t;  g(x2)' = g(x2) + P2^T g(x3)';  g(x1) += P1^T g(x2)'  — the intermediate
// sums are rounded to bf16 where the three single launches store them, and only g(x1) is written back
template <int SP_CH>
__global__ __launch_bounds__(SP_NT) void sppf_pool3_bwd_kernel(const uint16_t* __restrict__ g1, const uint16_t* __restrict__ g2, const uint16_t* __restrict__ g3,
                                                               int ldg, const int8_t* __restrict__ i1, const int8_t* __restrict__ i2, const int8_t* __restrict__ i3,
                                                               int H, int W, int C, uint16_t* __restrict__ gx, int ldx, int acc)
{
    constexpr int SP_CPP = SP_CH / 8;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int HW = H * W;
    uint16_t* sT = reinterpret_cast<uint16_t*>(smem);                  // [HW][SP_CH] gradient of the current pool's output
    uint16_t* sN = sT + (size_t)HW * SP_CH;                             // [HW][SP_CH] gradient of its input (next stage's sT)
    uint8_t* sI = reinterpret_cast<uint8_t*>(sN + (size_t)HW * SP_CH);  // [HW][SP_CH] arg-max of the current pool
    const int b = blockIdx.x, c0 = blockIdx.y * SP_CH;
    const int t = threadIdx.x, chunk = t % SP_CPP;
    const bool cok = c0 + chunk * 8 < C;
    const size_t img = (size_t)b * HW;
    for (int px = t / SP_CPP; px < HW; px += SP_NT / SP_CPP) {
        uint4 v = make_uint4(0, 0, 0, 0);
        if (cok) v = *reinterpret_cast<const uint4*>(g3 + (img + px) * ldg + c0 + chunk * 8);
        *reinterpret_cast<uint4*>(sT + px * SP_CH + chunk * 8) = v;
    }
    for (int stage = 2; stage >= 0; --stage) {
        const int8_t* idx = stage == 2 ? i3 : (stage == 1 ? i2 : i1);
        for (int px = t / SP_CPP; px < HW; px += SP_NT / SP_CPP) {
            uint2 pk = make_uint2(0xffffffffu, 0xffffffffu);
            if (cok) pk = *reinterpret_cast<const uint2*>(idx + (img + px) * C + c0 + chunk * 8);
            *reinterpret_cast<uint2*>(sI + px * SP_CH + chunk * 8) = pk;
        }
        __syncthreads();
        uint16_t* cur = (stage & 1) ? sN : sT;          // stage 2: sT -> sN, stage 1: sN -> sT, stage 0: sT -> global
        uint16_t* nxt = (stage & 1) ? sT : sN;
        if (stage == 2) { cur = sT; nxt = sN; }
        for (int px = t / SP_CPP; px < HW; px += SP_NT / SP_CPP) {
            const int h = px / W, w = px - h * W;
            float s[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) s[e] = 0.f;
#pragma unroll
            for (int i = 0; i < 5; ++i) {
                const int oh = h + 2 - i;
                if (oh < 0 || oh >= H) continue;
#pragma unroll
                for (int j = 0; j < 5; ++j) {
                    const int ow = w + 2 - j;
                    if (ow < 0 || ow >= W) continue;
                    const int op = oh * W + ow;
                    const uint2 pk = *reinterpret_cast<const uint2*>(sI + op * SP_CH + chunk * 8);
                    float g[8];
                    unpack8(*reinterpret_cast<const uint4*>(cur + op * SP_CH + chunk * 8), g);
                    const int want = i * 5 + j;
#pragma unroll
                    for (int e = 0; e < 8; ++e) {
                        const int k = (int)(((e < 4 ? pk.x : pk.y) >> (8 * (e & 3))) & 0xff);
                        if (k == want) s[e] += g[e];
                    }
                }
            }
            // + the gradient this tensor already holds (its direct use by cba2's data gradient), rounded to bf16 like the stored tensor
            const uint16_t* own = stage == 2 ? g2 : (stage == 1 ? g1 : gx);
            const int ldo = stage == 0 ? ldx : ldg;
            if (cok && (stage > 0 || acc)) {
                float f[8];
                unpack8(*reinterpret_cast<const uint4*>(own + (img + px) * ldo + c0 + chunk * 8), f);
#pragma unroll
                for (int e = 0; e < 8; ++e) s[e] += f[e];
            }
            const uint4 ov = pack8(s);
            if (stage > 0) *reinterpret_cast<uint4*>(nxt + px * SP_CH + chunk * 8) = ov;
            else if (cok) *reinterpret_cast<uint4*>(gx + (img + px) * ldx + c0 + chunk * 8) = ov;
        }
        __syncthreads();
    }
}

// channels per block for a map of HW pixels (the LDS footprint HW * ch * 5 bytes stays below 77 KB); 0 = map too large
int sppf_ch(int HW) { return HW <= SP_MAXPX ? 32 : (HW <= 2 * SP_MAXPX ? 16 : (HW <= 4 * SP_MAXPX ? 8 : 0)); }
size_t sppf_smem(int HW, int ch) { return (size_t)HW * ch * (2 + 2 + 1); }

}  // namespace

/* 1 if the fused SPPF kernels take this map (else the caller runs three yh_maxpool5_* launches) */
extern "C" int yh_sppf_pool3_ok(int H, int W, int C) { return H > 0 && W > 0 && sppf_ch(H * W) > 0 && C > 0 && C % 8 == 0; }

extern "C" int yh_sppf_pool3_fwd(const yh_bf16* x, int ldx, int B, int H, int W, int C, yh_bf16* o1, yh_bf16* o2, yh_bf16* o3, int ldo,
                                 int8_t* i1, int8_t* i2, int8_t* i3, yh_stream stream)
{
    YH_CHECK_ARG(yh_sppf_pool3_ok(H, W, C) && B > 0, "yh_sppf_pool3_fwd: map of more than 1920 pixels / bad dims (use yh_maxpool5_fwd)");
    YH_CHECK_ARG(x && o1 && o2 && o3 && yh_aligned16(x) && yh_aligned16(o1) && yh_aligned16(o2) && yh_aligned16(o3) && ldx % 8 == 0 && ldo % 8 == 0,
                 "yh_sppf_pool3_fwd: null / unaligned slices");
    YH_CHECK_ARG((i1 == nullptr) == (i2 == nullptr) && (i2 == nullptr) == (i3 == nullptr), "yh_sppf_pool3_fwd: all three arg-max buffers or none");
    const int ch = sppf_ch(H * W);
    const size_t sm = sppf_smem(H * W, ch);
    static YhDevOnce attr;
    if (attr.need()) {
        const int mx = (int)sppf_smem(SP_MAXPX, 32);
        attr.set((const void*)sppf_pool3_fwd_kernel<32>, hipFuncAttributeMaxDynamicSharedMemorySize, mx);
        attr.set((const void*)sppf_pool3_fwd_kernel<16>, hipFuncAttributeMaxDynamicSharedMemorySize, mx);
        attr.set((const void*)sppf_pool3_fwd_kernel<8>, hipFuncAttributeMaxDynamicSharedMemorySize, mx);
        attr.done();
    }
    const dim3 grid(B, (C + ch - 1) / ch), blk(SP_NT);
    if (ch == 32)      sppf_pool3_fwd_kernel<32><<<grid, blk, sm, (hipStream_t)stream>>>(x, ldx, H, W, C, o1, o2, o3, ldo, i1, i2, i3);
    else if (ch == 16) sppf_pool3_fwd_kernel<16><<<grid, blk, sm, (hipStream_t)stream>>>(x, ldx, H, W, C, o1, o2, o3, ldo, i1, i2, i3);
    else               sppf_pool3_fwd_kernel<8><<<grid, blk, sm, (hipStream_t)stream>>>(x, ldx, H, W, C, o1, o2, o3, ldo, i1, i2, i3);
    YH_CHECK_LAUNCH("yh_sppf_pool3_fwd");
    return YH_OK;
}

extern "C" int yh_sppf_pool3_bwd(const yh_bf16* g1, const yh_bf16* g2, const yh_bf16* g3, int ldg, const int8_t* i1, const int8_t* i2, const int8_t* i3,
                                 int B, int H, int W, int C, yh_bf16* gx, int ldx, int accumulate, yh_stream stream)
{
    YH_CHECK_ARG(yh_sppf_pool3_ok(H, W, C) && B > 0, "yh_sppf_pool3_bwd: map of more than 1920 pixels / bad dims (use yh_maxpool5_bwd)");
    YH_CHECK_ARG(g1 && g2 && g3 && gx && i1 && i2 && i3 && yh_aligned16(g1) && yh_aligned16(g2) && yh_aligned16(g3) && yh_aligned16(gx) &&
                 ldg % 8 == 0 && ldx % 8 == 0, "yh_sppf_pool3_bwd: null / unaligned slices");
    const int ch = sppf_ch(H * W);
    const size_t sm = sppf_smem(H * W, ch);
    static YhDevOnce attr;
    if (attr.need()) {
        const int mx = (int)sppf_smem(SP_MAXPX, 32);
        attr.set((const void*)sppf_pool3_bwd_kernel<32>, hipFuncAttributeMaxDynamicSharedMemorySize, mx);
        attr.set((const void*)sppf_pool3_bwd_kernel<16>, hipFuncAttributeMaxDynamicSharedMemorySize, mx);
        attr.set((const void*)sppf_pool3_bwd_kernel<8>, hipFuncAttributeMaxDynamicSharedMemorySize, mx);
        attr.done();
    }
    const dim3 grid(B, (C + ch - 1) / ch), blk(SP_NT);
    if (ch == 32)      sppf_pool3_bwd_kernel<32><<<grid, blk, sm, (hipStream_t)stream>>>(g1, g2, g3, ldg, i1, i2, i3, H, W, C, gx, ldx, accumulate);
    else if (ch == 16) sppf_pool3_bwd_kernel<16><<<grid, blk, sm, (hipStream_t)stream>>>(g1, g2, g3, ldg, i1, i2, i3, H, W, C, gx, ldx, accumulate);
    else               sppf_pool3_bwd_kernel<8><<<grid, blk, sm, (hipStream_t)stream>>>(g1, g2, g3, ldg, i1, i2, i3, H, W, C, gx, ldx, accumulate);
    YH_CHECK_LAUNCH("yh_sppf_pool3_bwd");
    return YH_OK;
}
