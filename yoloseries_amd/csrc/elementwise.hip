// HBM-bound NHWC bf16 kernels around the convolutions: training-mode BatchNorm
// statistics finalisation, BN+SiLU apply and its two-pass backward
// (utils/layer_tools.py:87-91), SPPF max-pool (utils/layer_tools.py:270-288),
// nearest-upsample gradient, and the stem's space-to-depth input transform.
// Every global access is a 16-byte chunk of 8 channels; reductions are
// deterministic (per-block partial slabs combined in a fixed order, fp64).
#include "common.h"
#include <stdlib.h>

namespace {

constexpr int EW_THREADS = 256;

__device__ __forceinline__ uint4 ld_nt(const uint16_t* p) {
    typedef __attribute__((ext_vector_type(4))) unsigned int u4;
    u4 v = __builtin_nontemporal_load(reinterpret_cast<const u4*>(p));
    return make_uint4(v[0], v[1], v[2], v[3]);
}

// ---------------------------------------------------------------- BN finalize
// Deterministic column sums of a [nblk][nwhich][ld] partial slab.  These kernels sit on the layer chain's critical path
// (conv -> finalize -> apply) and move little data, so they are built for latency: one block = FIN_CPB channels
// (a quarter wave reads 64 contiguous bytes of a slab row), 32 row groups (8 waves x 4 quarter waves) with FIN_U independent
// row loads in flight per lane, fixed-order combine through LDS in fp64.  C / 16 blocks spread the slab over the CUs
// (the 64-channel, 4-in-flight version took 7-24 us per launch, 114 launches per train step).
constexpr int FIN_CPB = 16;
constexpr int FIN_RG = 32;             // 512 threads: measured against 1024 / 256 / 128 threads and 8 / 32 channels per block on the train step
constexpr int FIN_NT = FIN_CPB * FIN_RG;
#ifndef YH_FIN_U
#define YH_FIN_U 8
#endif
constexpr int FIN_U = YH_FIN_U;          // independent row loads in flight per lane

template <int NW, int RG = FIN_RG>
__device__ __forceinline__ void slab_colsum(const float* __restrict__ slab, int nblk, int ld, int C, int c, double* out /*NW*/)
{
    __shared__ double sred[RG][NW][FIN_CPB];
    const int lc = threadIdx.x & (FIN_CPB - 1), rg = threadIdx.x / FIN_CPB;
    double acc[NW];
#pragma unroll
    for (int w = 0; w < NW; ++w) acc[w] = 0.0;
    if (c < C) {
        // branch-free batches (rows past the end are clamped and masked): a predicated load per row makes the compiler wait for
        // each one before the next is issued, which was most of the 6 us these launches took
        for (int b = rg; b < nblk; b += FIN_U * RG) {
            float v[FIN_U][NW];
#pragma unroll
            for (int u = 0; u < FIN_U; ++u) {
                const int r = b + RG * u;
                const size_t rc = (size_t)(r < nblk ? r : nblk - 1);
#pragma unroll
                for (int w = 0; w < NW; ++w) v[u][w] = slab[(rc * NW + w) * ld + c];
            }
#pragma unroll
            for (int u = 0; u < FIN_U; ++u)
#pragma unroll
                for (int w = 0; w < NW; ++w) acc[w] += (b + RG * u < nblk) ? (double)v[u][w] : 0.0;
        }
    }
#pragma unroll
    for (int w = 0; w < NW; ++w) sred[rg][w][lc] = acc[w];
    __syncthreads();
    if (rg == 0) {
#pragma unroll
        for (int w = 0; w < NW; ++w) {
            double s = 0.0;
            for (int i = 0; i < RG; ++i) s += sred[i][w][lc];
            out[w] = s;
        }
    }
}

template <int RG = FIN_RG>
__device__ __forceinline__ void bn_finalize_body(int cb, const float* __restrict__ stats, int nblk, int ldstat, int C, double count,
                                                 const float* __restrict__ gamma, const float* __restrict__ beta,
                                                 float* running_mean, float* running_var, int64_t* num_batches,
                                                 float eps, float momentum, float* ws)
{
    const int c = cb * FIN_CPB + (threadIdx.x & (FIN_CPB - 1));
    // the per-channel parameters are fetched BEFORE the column sums, so their latency hides under the slab's
    const bool fin = threadIdx.x < FIN_CPB && c < C;
    float g = 0.f, bt = 0.f, rm = 0.f, rv = 0.f;
    if (fin) {
        g = gamma[c]; bt = beta[c];
        if (running_mean) { rm = running_mean[c]; rv = running_var[c]; }
    }
    int64_t nb = 0;
    if (fin && c == 0 && num_batches) nb = *num_batches;
    double sq[2];
    slab_colsum<2, RG>(stats, nblk, ldstat, C, c, sq);
    if (threadIdx.x >= FIN_CPB) return;
    if (c == 0 && num_batches) *num_batches = nb + 1;
    if (c >= C) return;
    double mean = sq[0] / count;
    double var = sq[1] / count - mean * mean;
    if (var < 0.0) var = 0.0;
    float invstd = (float)(1.0 / sqrt(var + (double)eps));
    float scale = g * invstd;
    ws[c] = scale;
    ws[C + c] = bt - (float)mean * scale;
    ws[2 * C + c] = (float)mean;
    ws[3 * C + c] = invstd;
    if (running_mean) {
        double unbiased = count > 1.0 ? var * count / (count - 1.0) : var;
        running_mean[c] = (1.f - momentum) * rm + momentum * (float)mean;
        running_var[c] = (1.f - momentum) * rv + momentum * (float)unbiased;
    }
}

__global__ __launch_bounds__(FIN_NT) void bn_finalize_kernel(const float* __restrict__ stats, int nblk, int ldstat, int C, double count,
                                   const float* __restrict__ gamma, const float* __restrict__ beta,
                                   float* running_mean, float* running_var, int64_t* num_batches,
                                   float eps, float momentum, float* ws)
{
    bn_finalize_body(blockIdx.x, stats, nblk, ldstat, C, count, gamma, beta, running_mean, running_var, num_batches, eps, momentum, ws);
}

__global__ void bn_fold_kernel(const float* gamma, const float* beta, const float* rm, const float* rv,
                               float eps, int C, float* scale, float* shift)
{
    int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    float s = gamma[c] / sqrtf(rv[c] + eps);
    scale[c] = s;
    shift[c] = beta[c] - rm[c] * s;
}

// BatchNorm in evaluation mode inside a differentiable forward (model.eval() with gradients enabled, as torch runs
// nn.BatchNorm2d then): the constants of the BN+SiLU passes come from the running statistics; ws = scale | shift | mean | invstd
__global__ void bn_frozen_kernel(const float* gamma, const float* beta, const float* rm, const float* rv, float eps, int C, float* ws)
{
    int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    const float invstd = (float)(1.0 / sqrt((double)rv[c] + (double)eps));
    const float s = gamma[c] * invstd;
    ws[c] = s;
    ws[C + c] = beta[c] - rm[c] * s;
    ws[2 * C + c] = rm[c];
    ws[3 * C + c] = invstd;
}

// every BatchNorm of a network in ONE launch: block b folds item b (a table in device memory)
__global__ void bn_fold_batch_kernel(const yh_bn_fold_item* items)
{
    const yh_bn_fold_item it = items[blockIdx.x];
    for (int c = threadIdx.x; c < it.C; c += blockDim.x) {
        float s = it.gamma[c] / sqrtf(it.rv[c] + it.eps);
        it.scale[c] = s;
        it.shift[c] = it.beta[c] - it.rm[c] * s;
    }
}

// ---------------------------------------------------------------- BN+SiLU apply
// Grid-stride over 16-byte chunks with a stride that is a multiple of the chunks per row, so a thread
// keeps its 8 channels for the whole pass and the per-channel constants live in registers.
__global__ void bn_silu_apply_kernel(const uint16_t* __restrict__ y, int ldy, const float* __restrict__ ws, int C, int cpr,
                                     long M, uint16_t* __restrict__ out, int ldo,
                                     const uint16_t* __restrict__ res, int ldr)
{
    const long T = (long)gridDim.x * blockDim.x;
    const long rstep = T / cpr;
    const long gid = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (gid >= rstep * cpr) return;
    long m = gid / cpr;
    const int c = (int)(gid - m * cpr) * 8;
    float sc[8], sh[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) { sc[e] = ws[c + e]; sh[e] = ws[C + c + e]; }
    for (; m < M; m += rstep) {
        uint4 v = *reinterpret_cast<const uint4*>(y + m * ldy + c);
        float f[8];
        unpack8(v, f);
#pragma unroll
        for (int e = 0; e < 8; ++e) f[e] = silu_fast(f[e] * sc[e] + sh[e]);
        if (res) {
            uint4 rv = *reinterpret_cast<const uint4*>(res + m * ldr + c);
            float g[8];
            unpack8(rv, g);
            // residual is added to the bf16-rounded activation, as x += x_res does on stored tensors
#pragma unroll
            for (int e = 0; e < 8; ++e) f[e] = bf_round(f[e]) + g[e];
        }
        *reinterpret_cast<uint4*>(out + m * ldo + c) = pack8(f);
    }
}

// Stacked ConvBnAct layers (one conv, several BatchNorms over consecutive channel ranges of its output: C3's cba1 | cba2): ONE pass
// over the whole rows of y instead of one per part (a part's slice is half a row: 64-byte pieces of 128-byte lines).  A thread's
// 8 channels lie in one part; it picks that part's constants and destination once.
static_assert(sizeof(yh_bn_part) == 128, "yh_bn_part layout is part of the C ABI (yoloseries_amd/_lib.py: BnPart)");
struct PartsK { yh_bn_part p[YH_BN_MAX_PARTS]; int cend[YH_BN_MAX_PARTS]; int bend[YH_BN_MAX_PARTS]; int n; };   // cend: channels, bend: finalize blocks (cumulative)

// part and block-within-part of a finalize block (block-uniform)
__device__ __forceinline__ int fin_part_of(const PartsK& P, int blk, int& cb)
{
    int k = 0, b0 = 0;
#pragma unroll
    for (int i = 0; i < YH_BN_MAX_PARTS - 1; ++i)
        if (i + 1 < P.n && blk >= P.bend[i]) { k = i + 1; b0 = P.bend[i]; }
    cb = blk - b0;
    return k;
}

__global__ __launch_bounds__(FIN_NT) void bn_finalize_parts_kernel(const PartsK P, double count)
{
    int cb;
    const yh_bn_part& q = P.p[fin_part_of(P, blockIdx.x, cb)];
    bn_finalize_body(cb, q.slab, q.nblk, q.ldslab, q.C, count, q.gamma, q.beta, q.running_mean, q.running_var, q.num_batches,
                     q.eps, q.momentum, q.ws);
}

__device__ __forceinline__ int part_of(const PartsK& P, int c, int& c_in)
{
    int k = 0, c0 = 0;
#pragma unroll
    for (int i = 0; i < YH_BN_MAX_PARTS - 1; ++i)
        if (i + 1 < P.n && c >= P.cend[i]) { k = i + 1; c0 = P.cend[i]; }
    c_in = c - c0;
    return k;
}

__global__ void bn_silu_apply_parts_kernel(const uint16_t* __restrict__ y, int ldy, const PartsK P, int cpr, long M)
{
    const long T = (long)gridDim.x * blockDim.x;
    const long rstep = T / cpr;
    const long gid = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (gid >= rstep * cpr) return;
    long m = gid / cpr;
    const int c = (int)(gid - m * cpr) * 8;
    int cp;
    const int k = part_of(P, c, cp);
    const float* __restrict__ ws = P.p[k].ws;
    const int Cp = P.p[k].C;
    uint16_t* __restrict__ out = P.p[k].out + cp;
    const int ldo = P.p[k].ldo;
    float sc[8], sh[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) { sc[e] = ws[cp + e]; sh[e] = ws[Cp + cp + e]; }
    for (; m < M; m += rstep) {
        uint4 v = *reinterpret_cast<const uint4*>(y + m * ldy + c);
        float f[8];
        unpack8(v, f);
#pragma unroll
        for (int e = 0; e < 8; ++e) f[e] = silu_fast(f[e] * sc[e] + sh[e]);
        *reinterpret_cast<uint4*>(out + m * ldo) = pack8(f);
    }
}

// ---------------------------------------------------------------- column reductions
// Block b owns rows [b*rpb, (b+1)*rpb).  Thread (rg, cch) accumulates 8 channels over
// rows rg, rg+RG, ...; row groups are combined through LDS.
constexpr int RED_THREADS = 512;

template <int MODE>   // 0: BN+SiLU backward sums (gz, gz*xhat) ; 1: plain column sum of g
__global__ __launch_bounds__(RED_THREADS, 4) void col_reduce_kernel(const uint16_t* __restrict__ ga, int ldga,
                                                                 const uint16_t* __restrict__ y, int ldy,
                                                                 const float* __restrict__ ws, int C, int cpr,
                                                                 long M, long rpb, float* __restrict__ part)
{
    __shared__ float sP[RED_THREADS * 16];
    const int t = threadIdx.x;
    const int RG = RED_THREADS / cpr;
    const int rg = t / cpr;
    const int cch = t - rg * cpr;
    const int c = cch * 8;
    float a1[8], a2[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) { a1[e] = 0.f; a2[e] = 0.f; }
    if (rg < RG) {
        // accumulates sum(dz) and sum(dz*y); the finalize turns the second into sum(dz*xhat) = is*(sum(dz*y) - mu*sum(dz))
        float sc[8], sh[8];
        if (MODE == 0) {
#pragma unroll
            for (int e = 0; e < 8; ++e) { sc[e] = ws[c + e]; sh[e] = ws[C + c + e]; }
        }
        auto accum = [&](const uint4& gv, const uint4& yv) {
            float g[8];
            unpack8(gv, g);
            if (MODE == 0) {
                float yy[8];
                unpack8(yv, yy);
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    float z = yy[e] * sc[e] + sh[e];
                    float sg = sigmoid_fast(z);
                    float gz = g[e] * (sg * (1.f + z * (1.f - sg)));
                    a1[e] += gz; a2[e] += gz * yy[e];
                }
            } else {
#pragma unroll
                for (int e = 0; e < 8; ++e) a1[e] += g[e];
            }
        };
        // Blocks sweep the tensor together: block b takes the row groups (RG rows each) b, b+grid, b+2*grid, ...,
        // so the chip reads one moving window (like the apply passes) and the blocks' shares differ by at most
        // one group; four groups in flight per thread.  The assignment is fixed -> deterministic sums.
        const long G = (M + RG - 1) / RG;
        const long nb = gridDim.x;
        long gi = blockIdx.x;
        for (; gi + 3 * nb < G - 1; gi += 4 * nb) {
            uint4 gv[4], yv[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const long m = (gi + u * nb) * RG + rg;
                gv[u] = *reinterpret_cast<const uint4*>(ga + m * ldga + c);
                if (MODE == 0) yv[u] = *reinterpret_cast<const uint4*>(y + m * ldy + c);
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) accum(gv[u], yv[u]);
        }
        {   // remainder: at most four groups are left for this block; their loads are issued together as well
            uint4 gv[4], yv[4];
            bool ok[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const long m = (gi + u * nb) * RG + rg;
                ok[u] = m < M;
                gv[u] = make_uint4(0, 0, 0, 0);
                yv[u] = gv[u];
                if (ok[u]) {
                    gv[u] = *reinterpret_cast<const uint4*>(ga + m * ldga + c);
                    if (MODE == 0) yv[u] = *reinterpret_cast<const uint4*>(y + m * ldy + c);
                }
            }
#pragma unroll
            for (int u = 0; u < 4; ++u)
                if (ok[u]) accum(gv[u], yv[u]);
        }
    }
    // sP layout [rg][2][C]  (RG*2*C = RED_THREADS/cpr*2*cpr*8 <= RED_THREADS*16 floats)
    if (rg < RG) {
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            sP[(rg * 2 + 0) * C + c + e] = a1[e];
            sP[(rg * 2 + 1) * C + c + e] = a2[e];
        }
    }
    __syncthreads();
    for (int i = t; i < 2 * C; i += RED_THREADS) {
        int which = i / C;
        int cc = i - which * C;
        float s = 0.f;
        for (int r = 0; r < RG; ++r) s += sP[(r * 2 + which) * C + cc];
        part[((size_t)blockIdx.x * 2 + which) * C + cc] = s;
    }
}

__device__ __forceinline__ void bn_bwd_finalize_body(int cb, const float* __restrict__ part, int nblk, int C, double M,
                                                     const float* __restrict__ ws, float* dgamma, float* dbeta, float* coef)
{
    const int c = cb * FIN_CPB + (threadIdx.x & (FIN_CPB - 1));
    float mu = 0.f, is = 0.f;           // fetched before the column sums (latency hidden under the slab's)
    if (threadIdx.x < FIN_CPB && c < C) { mu = ws[2 * C + c]; is = ws[3 * C + c]; }
    double s[2];
    slab_colsum<2>(part, nblk, C, C, c, s);
    if (threadIdx.x >= FIN_CPB || c >= C) return;
    s[1] = (double)is * (s[1] - (double)mu * s[0]);      // sum(dz*y) -> sum(dz*xhat)
    if (dbeta) dbeta[c] = (float)s[0];
    if (dgamma) dgamma[c] = (float)s[1];
    if (coef) { coef[c] = (float)(s[0] / M); coef[C + c] = (float)(s[1] / M); }
}

__global__ __launch_bounds__(FIN_NT) void bn_bwd_finalize_kernel(const float* __restrict__ part, int nblk, int C, double M,
                                       const float* __restrict__ ws, float* dgamma, float* dbeta, float* coef)
{
    bn_bwd_finalize_body(blockIdx.x, part, nblk, C, M, ws, dgamma, dbeta, coef);
}

__global__ __launch_bounds__(FIN_NT) void bn_bwd_finalize_parts_kernel(const PartsK P, double M)
{
    int cb;
    const yh_bn_part& q = P.p[fin_part_of(P, blockIdx.x, cb)];
    bn_bwd_finalize_body(cb, q.slab, q.nblk, q.C, M, q.ws, q.dgamma, q.dbeta, q.coef);
}

__global__ __launch_bounds__(FIN_NT) void colsum_finalize_kernel(const float* __restrict__ part, int nblk, int C, float* out)
{
    const int c = blockIdx.x * FIN_CPB + (threadIdx.x & (FIN_CPB - 1));
    double s[2];
    slab_colsum<2>(part, nblk, C, C, c, s);
    if (threadIdx.x >= FIN_CPB || c >= C) return;
    out[c] = (float)s[0];
}

// gz = gamma*is*(dz - c1 - xhat*c2) with dz = g*silu'(z), xhat = (y-mu)*is, written per channel as
// gz = A*dz + Bc*y + D (constants in registers, see bn_silu_apply_kernel for the striding).
__global__ void bn_silu_bwd_apply_kernel(const uint16_t* __restrict__ ga, int ldga, const uint16_t* __restrict__ y, int ldy,
                                         const float* __restrict__ ws, const float* __restrict__ gamma,
                                         const float* __restrict__ coef, int C, int cpr, long M,
                                         uint16_t* __restrict__ gy, int ldgy, uint16_t* gres, int ldgres, int gres_acc)
{
    const long T = (long)gridDim.x * blockDim.x;
    const long rstep = T / cpr;
    const long gid = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (gid >= rstep * cpr) return;
    long m = gid / cpr;
    const int c = (int)(gid - m * cpr) * 8;
    float sc[8], sh[8], A[8], Bc[8], D[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        sc[e] = ws[c + e]; sh[e] = ws[C + c + e];
        const float mu = ws[2 * C + c + e], is = ws[3 * C + c + e];
        const float gi = gamma[c + e] * is;
        const float c1 = coef[c + e], c2 = coef[C + c + e];
        A[e] = gi;
        Bc[e] = -gi * is * c2;
        D[e] = gi * (mu * is * c2 - c1);
    }
    for (; m < M; m += rstep) {
        // last readers of both tensors: streamed (non-temporal), they should not displace the gz rows written below,
        // which the weight- and data-gradient kernels read next
        uint4 gv = ld_nt(ga + m * ldga + c);
        uint4 yv = ld_nt(y + m * ldy + c);
        float g[8], yy[8], o[8];
        unpack8(gv, g);
        unpack8(yv, yy);
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            float z = yy[e] * sc[e] + sh[e];
            float sg = sigmoid_fast(z);
            float dz = g[e] * (sg * (1.f + z * (1.f - sg)));
            o[e] = A[e] * dz + (Bc[e] * yy[e] + D[e]);
        }
        *reinterpret_cast<uint4*>(gy + m * ldgy + c) = pack8(o);
        if (gres) {
            uint16_t* dst = gres + m * ldgres + c;
            if (gres_acc) {
                uint4 ov = *reinterpret_cast<const uint4*>(dst);
                float f[8];
                unpack8(ov, f);
#pragma unroll
                for (int e = 0; e < 8; ++e) f[e] += g[e];
                *reinterpret_cast<uint4*>(dst) = pack8(f);
            } else {
                *reinterpret_cast<uint4*>(dst) = gv;
            }
        }
    }
}

// bn_silu_bwd_apply over all parts of a stacked layer in one pass (see bn_silu_apply_parts_kernel): a part brings its own incoming
// gradient tensor; gz of all parts is one buffer.
__global__ void bn_silu_bwd_apply_parts_kernel(const uint16_t* __restrict__ y, int ldy, const PartsK P, int cpr, long M,
                                               uint16_t* __restrict__ gy, int ldgy)
{
    const long T = (long)gridDim.x * blockDim.x;
    const long rstep = T / cpr;
    const long gid = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (gid >= rstep * cpr) return;
    long m = gid / cpr;
    const int c = (int)(gid - m * cpr) * 8;
    int cp;
    const int k = part_of(P, c, cp);
    const float* __restrict__ ws = P.p[k].ws;
    const float* __restrict__ gamma = P.p[k].gamma;
    const float* __restrict__ coef = P.p[k].coef;
    const int Cp = P.p[k].C;
    const uint16_t* __restrict__ ga = P.p[k].ga + cp;
    const int ldga = P.p[k].ldga;
    float sc[8], sh[8], A[8], Bc[8], D[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        sc[e] = ws[cp + e]; sh[e] = ws[Cp + cp + e];
        const float mu = ws[2 * Cp + cp + e], is = ws[3 * Cp + cp + e];
        const float gi = gamma[cp + e] * is;
        const float c1 = coef[cp + e], c2 = coef[Cp + cp + e];
        A[e] = gi;
        Bc[e] = -gi * is * c2;
        D[e] = gi * (mu * is * c2 - c1);
    }
    for (; m < M; m += rstep) {
        uint4 gv = ld_nt(ga + m * ldga);
        uint4 yv = ld_nt(y + m * ldy + c);
        float g[8], yy[8], o[8];
        unpack8(gv, g);
        unpack8(yv, yy);
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            float z = yy[e] * sc[e] + sh[e];
            float sg = sigmoid_fast(z);
            float dz = g[e] * (sg * (1.f + z * (1.f - sg)));
            o[e] = A[e] * dz + (Bc[e] * yy[e] + D[e]);
        }
        *reinterpret_cast<uint4*>(gy + m * ldgy + c) = pack8(o);
    }
}

// ---------------------------------------------------------------- max-pool 5x5 s1 p2
__global__ void maxpool5_fwd_kernel(const uint16_t* __restrict__ x, int ldx, int B, int H, int W, int cpr,
                                    uint16_t* __restrict__ out, int ldo, int8_t* __restrict__ idx)
{
    long total = (long)B * H * W * cpr;
    for (long id = (long)blockIdx.x * blockDim.x + threadIdx.x; id < total; id += (long)gridDim.x * blockDim.x) {
        long pix = id / cpr;
        int c = (int)(id - pix * cpr) * 8;
        int w = (int)(pix % W);
        long t2 = pix / W;
        int h = (int)(t2 % H);
        long b = t2 / H;
        float best[8];
        int bi[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) { best[e] = -INFINITY; bi[e] = 0; }
        // a window row's five loads are issued together from clamped addresses (the pass is latency bound: 25 dependent round
        // trips otherwise); positions outside the image are skipped when comparing.  Scan order (i, j) as before: first maximum wins.
        bool any = false;
#pragma unroll
        for (int i = 0; i < 5; ++i) {
            const int hh = h - 2 + i;
            const bool rok = hh >= 0 && hh < H;
            const int hc = hh < 0 ? 0 : (hh >= H ? H - 1 : hh);
            uint4 v[5];
#pragma unroll
            for (int j = 0; j < 5; ++j) {
                const int ww = w - 2 + j;
                const int wc = ww < 0 ? 0 : (ww >= W ? W - 1 : ww);
                v[j] = *reinterpret_cast<const uint4*>(x + ((b * H + hc) * W + wc) * ldx + c);
            }
#pragma unroll
            for (int j = 0; j < 5; ++j) {
                const int ww = w - 2 + j;
                if (!(rok && ww >= 0 && ww < W)) continue;
                float f[8];
                unpack8(v[j], f);
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    if (!any || f[e] > best[e] || f[e] != f[e]) { best[e] = f[e]; bi[e] = i * 5 + j; }
                }
                any = true;
            }
        }
        *reinterpret_cast<uint4*>(out + pix * ldo + c) = pack8(best);
        if (idx) {
            uint2 pk;
            pk.x = (uint32_t)bi[0] | ((uint32_t)bi[1] << 8) | ((uint32_t)bi[2] << 16) | ((uint32_t)bi[3] << 24);
            pk.y = (uint32_t)bi[4] | ((uint32_t)bi[5] << 8) | ((uint32_t)bi[6] << 16) | ((uint32_t)bi[7] << 24);
            *reinterpret_cast<uint2*>(idx + pix * (cpr * 8) + c) = pk;
        }
    }
}

__global__ void maxpool5_bwd_kernel(const uint16_t* __restrict__ gout, int ldgo, const int8_t* __restrict__ idx,
                                    int B, int H, int W, int cpr, uint16_t* __restrict__ gin, int ldgi, int acc)
{
    long total = (long)B * H * W * cpr;
    const int C = cpr * 8;
    for (long id = (long)blockIdx.x * blockDim.x + threadIdx.x; id < total; id += (long)gridDim.x * blockDim.x) {
        long pix = id / cpr;
        int c = (int)(id - pix * cpr) * 8;
        int w = (int)(pix % W);
        long t2 = pix / W;
        int h = (int)(t2 % H);
        long b = t2 / H;
        float s[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) s[e] = 0.f;
#pragma unroll
        for (int i = 0; i < 5; ++i) {          // the five (index, gradient) pairs of a row in flight together, see the forward
            const int oh = h + 2 - i;
            const bool rok = oh >= 0 && oh < H;
            const int hc = oh < 0 ? 0 : (oh >= H ? H - 1 : oh);
            uint2 pk[5];
            uint4 gv[5];
#pragma unroll
            for (int j = 0; j < 5; ++j) {
                const int ow = w + 2 - j;
                const int wc = ow < 0 ? 0 : (ow >= W ? W - 1 : ow);
                const long op = (b * H + hc) * W + wc;
                pk[j] = *reinterpret_cast<const uint2*>(idx + op * C + c);
                gv[j] = *reinterpret_cast<const uint4*>(gout + op * ldgo + c);
            }
#pragma unroll
            for (int j = 0; j < 5; ++j) {
                const int ow = w + 2 - j;
                if (!(rok && ow >= 0 && ow < W)) continue;
                float g[8];
                unpack8(gv[j], g);
                const int want = i * 5 + j;
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    int k = (int)(((e < 4 ? pk[j].x : pk[j].y) >> (8 * (e & 3))) & 0xff);
                    if (k == want) s[e] += g[e];
                }
            }
        }
        uint16_t* dst = gin + pix * ldgi + c;
        if (acc) {
            uint4 ov = *reinterpret_cast<const uint4*>(dst);
            float f[8];
            unpack8(ov, f);
#pragma unroll
            for (int e = 0; e < 8; ++e) s[e] += f[e];
        }
        *reinterpret_cast<uint4*>(dst) = pack8(s);
    }
}

__global__ void upsample2_bwd_kernel(const uint16_t* __restrict__ ghi, int ldh, int B, int Hlo, int Wlo, int cpr,
                                     uint16_t* __restrict__ glo, int ldl, int acc)
{
    long total = (long)B * Hlo * Wlo * cpr;
    for (long id = (long)blockIdx.x * blockDim.x + threadIdx.x; id < total; id += (long)gridDim.x * blockDim.x) {
        long pix = id / cpr;
        int c = (int)(id - pix * cpr) * 8;
        int w = (int)(pix % Wlo);
        long t2 = pix / Wlo;
        int h = (int)(t2 % Hlo);
        long b = t2 / Hlo;
        float s[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) s[e] = 0.f;
#pragma unroll
        for (int dy = 0; dy < 2; ++dy)
#pragma unroll
            for (int dx = 0; dx < 2; ++dx) {
                long hp = ((b * (2 * Hlo) + 2 * h + dy) * (2L * Wlo) + 2 * w + dx);
                uint4 v = *reinterpret_cast<const uint4*>(ghi + hp * ldh + c);
                float g[8];
                unpack8(v, g);
#pragma unroll
                for (int e = 0; e < 8; ++e) s[e] += g[e];
            }
        uint16_t* dst = glo + pix * ldl + c;
        if (acc) {
            uint4 ov = *reinterpret_cast<const uint4*>(dst);
            float f[8];
            unpack8(ov, f);
#pragma unroll
            for (int e = 0; e < 8; ++e) s[e] += f[e];
        }
        *reinterpret_cast<uint4*>(dst) = pack8(s);
    }
}

// ---------------------------------------------------------------- input transform
__global__ void input_s2d_kernel(const float* __restrict__ x, int B, int Cin, int H, int W, uint16_t* __restrict__ out)
{
    const int H2 = H / 2, W2 = W / 2;
    long total = (long)B * H2 * W2;
    for (long id = (long)blockIdx.x * blockDim.x + threadIdx.x; id < total; id += (long)gridDim.x * blockDim.x) {
        int w2 = (int)(id % W2);
        long t2 = id / W2;
        int h2 = (int)(t2 % H2);
        long b = t2 / H2;
        float f[16];
#pragma unroll
        for (int e = 0; e < 16; ++e) f[e] = 0.f;
        for (int c = 0; c < Cin; ++c) {
#pragma unroll
            for (int dy = 0; dy < 2; ++dy) {
                const float2 v = *reinterpret_cast<const float2*>(x + ((b * Cin + c) * H + 2 * h2 + dy) * (long)W + 2 * w2);
                f[(dy * 2 + 0) * Cin + c] = v.x;
                f[(dy * 2 + 1) * Cin + c] = v.y;
            }
        }
        uint4* o = reinterpret_cast<uint4*>(out + id * 16);
        o[0] = pack8(f);
        o[1] = pack8(f + 8);
    }
}

__global__ void fill_u32_kernel(uint32_t* p, uint32_t v, long n)
{
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) p[i] = v;
}

// YH_FIN_LDS_PAD=<bytes>: dynamic LDS the finalize launches ask for without using it, so that their workgroups do not fit on a CU beside a
// weight-gradient workgroup (128 KiB of the 160) and go to the CUs it leaves free (experiment: profiles/r04_step_experiments.txt, t)
inline unsigned fin_lds_pad() { static const unsigned v = [] { const char* e = getenv("YH_FIN_LDS_PAD"); return e ? (unsigned)atoi(e) : 0u; }(); return v; }

inline int ew_grid(long nthreads) {
    long g = (nthreads + EW_THREADS - 1) / EW_THREADS;
    if (g > 256 * 8) g = 256 * 8;          // 8 blocks of 256 threads per CU: measured against 4 / 6 / 10 / 16 / 32 and 128- / 512-thread blocks on the train step
    if (g < 1) g = 1;
    return (int)g;
}

}  // namespace

extern "C" int yh_ew_blocks(int64_t M) {
    long b = (M + 255) / 256;
    if (b > 512) b = 512;
    if (b < 1) b = 1;
    return (int)b;
}

extern "C" int yh_bn_finalize(const float* stats, int nblk, int ldstat, int C, int64_t count,
                              const float* gamma, const float* beta, float* running_mean, float* running_var,
                              int64_t* num_batches, float eps, float momentum, float* ws, yh_stream stream)
{
    YH_CHECK_ARG(stats && gamma && beta && ws && nblk > 0 && C > 0 && count > 0 && ldstat >= C, "yh_bn_finalize: bad args");
    hipLaunchKernelGGL(bn_finalize_kernel, dim3((C + FIN_CPB - 1) / FIN_CPB), dim3(FIN_NT), fin_lds_pad(), (hipStream_t)stream,
                       stats, nblk, ldstat, C, (double)count, gamma, beta, running_mean, running_var, num_batches, eps, momentum, ws);
    YH_CHECK_LAUNCH("yh_bn_finalize");
    return YH_OK;
}

extern "C" int yh_bn_fold(const float* gamma, const float* beta, const float* rm, const float* rv,
                          float eps, int C, float* scale, float* shift, yh_stream stream)
{
    YH_CHECK_ARG(gamma && beta && rm && rv && scale && shift && C > 0, "yh_bn_fold: bad args");
    hipLaunchKernelGGL(bn_fold_kernel, dim3((C + 127) / 128), dim3(128), 0, (hipStream_t)stream, gamma, beta, rm, rv, eps, C, scale, shift);
    YH_CHECK_LAUNCH("yh_bn_fold");
    return YH_OK;
}

extern "C" int yh_bn_frozen(const float* gamma, const float* beta, const float* rm, const float* rv, float eps, int C, float* ws, yh_stream stream)
{
    YH_CHECK_ARG(gamma && beta && rm && rv && ws && C > 0, "yh_bn_frozen: bad args");
    hipLaunchKernelGGL(bn_frozen_kernel, dim3((C + 127) / 128), dim3(128), 0, (hipStream_t)stream, gamma, beta, rm, rv, eps, C, ws);
    YH_CHECK_LAUNCH("yh_bn_frozen");
    return YH_OK;
}

extern "C" int yh_bn_fold_batch(const yh_bn_fold_item* items_dev, int nitems, yh_stream stream)
{
    YH_CHECK_ARG(items_dev && nitems > 0, "yh_bn_fold_batch: bad args");
    hipLaunchKernelGGL(bn_fold_batch_kernel, dim3(nitems), dim3(256), 0, (hipStream_t)stream, items_dev);
    YH_CHECK_LAUNCH("yh_bn_fold_batch");
    return YH_OK;
}

#define YH_CHECK_SLICE(name, p, ld, C) \
    YH_CHECK_ARG((p) && yh_aligned16(p) && (ld) % 8 == 0 && (ld) >= (C), name ": slice null/unaligned (ld=%d C=%d)", (int)(ld), (int)(C))

extern "C" int yh_bn_silu_apply(const yh_bf16* y, int ldy, const float* ws, int C, int64_t M,
                                yh_bf16* out, int ldo, const yh_bf16* res, int ldr, yh_stream stream)
{
    YH_CHECK_ARG(C > 0 && C % 8 == 0 && C <= 2048 && M > 0 && ws && yh_aligned16(ws), "yh_bn_silu_apply: bad C/M/ws");
    YH_CHECK_SLICE("yh_bn_silu_apply", y, ldy, C);
    YH_CHECK_SLICE("yh_bn_silu_apply", out, ldo, C);
    if (res) YH_CHECK_SLICE("yh_bn_silu_apply", res, ldr, C);
    int cpr = C / 8;
    long nch = (long)M * cpr;
    hipLaunchKernelGGL(bn_silu_apply_kernel, dim3(ew_grid(nch)), dim3(EW_THREADS), 0, (hipStream_t)stream,
                       y, ldy, ws, C, cpr, (long)M, out, ldo, res, ldr);
    YH_CHECK_LAUNCH("yh_bn_silu_apply");
    return YH_OK;
}

extern "C" int yh_bn_silu_bwd_reduce(const yh_bf16* ga, int ldga, const yh_bf16* y, int ldy,
                                     const float* ws, int C, int64_t M, float* part, yh_stream stream)
{
    YH_CHECK_ARG(C > 0 && C % 8 == 0 && C <= 2048 && M > 0 && ws && part, "yh_bn_silu_bwd_reduce: bad args");
    YH_CHECK_SLICE("yh_bn_silu_bwd_reduce", ga, ldga, C);
    YH_CHECK_SLICE("yh_bn_silu_bwd_reduce", y, ldy, C);
    int nblk = yh_ew_blocks(M);
    long rpb = (M + nblk - 1) / nblk;
    hipLaunchKernelGGL((col_reduce_kernel<0>), dim3(nblk), dim3(RED_THREADS), 0, (hipStream_t)stream,
                       ga, ldga, y, ldy, ws, C, C / 8, (long)M, rpb, part);
    YH_CHECK_LAUNCH("yh_bn_silu_bwd_reduce");
    return YH_OK;
}

extern "C" int yh_bn_bwd_finalize(const float* part, int nblk, int C, int64_t M, const float* ws,
                                  float* dgamma, float* dbeta, float* coef, yh_stream stream)
{
    YH_CHECK_ARG(part && ws && nblk > 0 && C > 0 && M > 0, "yh_bn_bwd_finalize: bad args");
    hipLaunchKernelGGL(bn_bwd_finalize_kernel, dim3((C + FIN_CPB - 1) / FIN_CPB), dim3(FIN_NT), fin_lds_pad(), (hipStream_t)stream,
                       part, nblk, C, (double)M, ws, dgamma, dbeta, coef);
    YH_CHECK_LAUNCH("yh_bn_bwd_finalize");
    return YH_OK;
}

extern "C" int yh_bn_silu_bwd_apply(const yh_bf16* ga, int ldga, const yh_bf16* y, int ldy,
                                    const float* ws, const float* gamma, const float* coef,
                                    int C, int64_t M, yh_bf16* gy, int ldgy,
                                    yh_bf16* gres, int ldgres, int gres_accumulate, yh_stream stream)
{
    YH_CHECK_ARG(C > 0 && C % 8 == 0 && C <= 2048 && M > 0 && ws && gamma && coef, "yh_bn_silu_bwd_apply: bad args");
    YH_CHECK_SLICE("yh_bn_silu_bwd_apply", ga, ldga, C);
    YH_CHECK_SLICE("yh_bn_silu_bwd_apply", y, ldy, C);
    YH_CHECK_SLICE("yh_bn_silu_bwd_apply", gy, ldgy, C);
    if (gres) YH_CHECK_SLICE("yh_bn_silu_bwd_apply", gres, ldgres, C);
    int cpr = C / 8;
    long nch = (long)M * cpr;
    hipLaunchKernelGGL(bn_silu_bwd_apply_kernel, dim3(ew_grid(nch)), dim3(EW_THREADS), 0, (hipStream_t)stream,
                       ga, ldga, y, ldy, ws, gamma, coef, C, cpr, (long)M, gy, ldgy, gres, ldgres, gres_accumulate);
    YH_CHECK_LAUNCH("yh_bn_silu_bwd_apply");
    return YH_OK;
}

// mode: 0 forward pass, 1 backward pass, 2 forward finalize, 3 backward finalize
static int parts_pack(const char* who, const yh_bn_part* parts, int nparts, int mode, PartsK* P, int* Ctot)
{
    YH_CHECK_ARG(parts && nparts >= 1 && nparts <= YH_BN_MAX_PARTS, "%s: 1..%d parts", who, YH_BN_MAX_PARTS);
    int c = 0, nb = 0;
    for (int i = 0; i < nparts; ++i) {
        const yh_bn_part& q = parts[i];
        YH_CHECK_ARG(q.C > 0 && q.C % 8 == 0 && q.ws && yh_aligned16(q.ws), "%s: part %d: bad C / ws", who, i);
        if (mode == 2) {
            YH_CHECK_ARG(q.slab && q.nblk > 0 && q.ldslab >= q.C && q.gamma && q.beta, "%s: part %d: bad slab / gamma / beta", who, i);
        } else if (mode == 3) {
            YH_CHECK_ARG(q.slab && q.nblk > 0, "%s: part %d: bad slab", who, i);
        } else if (mode == 1) {
            YH_CHECK_ARG(q.gamma && q.coef, "%s: part %d: gamma / coef missing", who, i);
            YH_CHECK_ARG(q.ga && yh_aligned16(q.ga) && q.ldga % 8 == 0 && q.ldga >= q.C, "%s: part %d: ga null/unaligned (ld=%d C=%d)", who, i, q.ldga, q.C);
        } else {
            YH_CHECK_ARG(q.out && yh_aligned16(q.out) && q.ldo % 8 == 0 && q.ldo >= q.C, "%s: part %d: out null/unaligned (ld=%d C=%d)", who, i, q.ldo, q.C);
        }
        c += q.C;
        nb += (q.C + FIN_CPB - 1) / FIN_CPB;
        P->p[i] = q;
        P->cend[i] = c;
        P->bend[i] = nb;
    }
    for (int i = nparts; i < YH_BN_MAX_PARTS; ++i) { P->p[i] = parts[nparts - 1]; P->cend[i] = c; P->bend[i] = nb; }
    P->n = nparts;
    *Ctot = c;
    YH_CHECK_ARG(c <= 2048, "%s: more than 2048 channels", who);
    return YH_OK;
}

extern "C" int yh_bn_finalize_parts(const yh_bn_part* parts, int nparts, int64_t count, yh_stream stream)
{
    PartsK P;
    int C = 0;
    const int rc = parts_pack("yh_bn_finalize_parts", parts, nparts, 2, &P, &C);
    if (rc != YH_OK) return rc;
    YH_CHECK_ARG(count > 0, "yh_bn_finalize_parts: bad count");
    hipLaunchKernelGGL(bn_finalize_parts_kernel, dim3(P.bend[nparts - 1]), dim3(FIN_NT), fin_lds_pad(), (hipStream_t)stream, P, (double)count);
    YH_CHECK_LAUNCH("yh_bn_finalize_parts");
    return YH_OK;
}

extern "C" int yh_bn_bwd_finalize_parts(const yh_bn_part* parts, int nparts, int64_t M, yh_stream stream)
{
    PartsK P;
    int C = 0;
    const int rc = parts_pack("yh_bn_bwd_finalize_parts", parts, nparts, 3, &P, &C);
    if (rc != YH_OK) return rc;
    YH_CHECK_ARG(M > 0, "yh_bn_bwd_finalize_parts: bad M");
    hipLaunchKernelGGL(bn_bwd_finalize_parts_kernel, dim3(P.bend[nparts - 1]), dim3(FIN_NT), fin_lds_pad(), (hipStream_t)stream, P, (double)M);
    YH_CHECK_LAUNCH("yh_bn_bwd_finalize_parts");
    return YH_OK;
}

extern "C" int yh_bn_silu_apply_parts(const yh_bf16* y, int ldy, int64_t M, const yh_bn_part* parts, int nparts, yh_stream stream)
{
    PartsK P;
    int C = 0;
    const int rc = parts_pack("yh_bn_silu_apply_parts", parts, nparts, 0, &P, &C);
    if (rc != YH_OK) return rc;
    YH_CHECK_ARG(M > 0, "yh_bn_silu_apply_parts: bad M");
    YH_CHECK_SLICE("yh_bn_silu_apply_parts", y, ldy, C);
    const int cpr = C / 8;
    hipLaunchKernelGGL(bn_silu_apply_parts_kernel, dim3(ew_grid((long)M * cpr)), dim3(EW_THREADS), 0, (hipStream_t)stream, y, ldy, P, cpr, (long)M);
    YH_CHECK_LAUNCH("yh_bn_silu_apply_parts");
    return YH_OK;
}

extern "C" int yh_bn_silu_bwd_apply_parts(const yh_bf16* y, int ldy, int64_t M, const yh_bn_part* parts, int nparts,
                                          yh_bf16* gy, int ldgy, yh_stream stream)
{
    PartsK P;
    int C = 0;
    const int rc = parts_pack("yh_bn_silu_bwd_apply_parts", parts, nparts, 1, &P, &C);
    if (rc != YH_OK) return rc;
    YH_CHECK_ARG(M > 0, "yh_bn_silu_bwd_apply_parts: bad M");
    YH_CHECK_SLICE("yh_bn_silu_bwd_apply_parts", y, ldy, C);
    YH_CHECK_SLICE("yh_bn_silu_bwd_apply_parts", gy, ldgy, C);
    const int cpr = C / 8;
    hipLaunchKernelGGL(bn_silu_bwd_apply_parts_kernel, dim3(ew_grid((long)M * cpr)), dim3(EW_THREADS), 0, (hipStream_t)stream,
                       y, ldy, P, cpr, (long)M, gy, ldgy);
    YH_CHECK_LAUNCH("yh_bn_silu_bwd_apply_parts");
    return YH_OK;
}

extern "C" int yh_colsum(const yh_bf16* g, int ldg, int C, int64_t M, float* part, float* out, yh_stream stream)
{
    YH_CHECK_ARG(C > 0 && C % 8 == 0 && C <= 2048 && M > 0 && part && out, "yh_colsum: bad args");
    YH_CHECK_SLICE("yh_colsum", g, ldg, C);
    int nblk = yh_ew_blocks(M);
    long rpb = (M + nblk - 1) / nblk;
    hipLaunchKernelGGL((col_reduce_kernel<1>), dim3(nblk), dim3(RED_THREADS), 0, (hipStream_t)stream,
                       g, ldg, (const uint16_t*)nullptr, 0, (const float*)nullptr, C, C / 8, (long)M, rpb, part);
    hipLaunchKernelGGL(colsum_finalize_kernel, dim3((C + FIN_CPB - 1) / FIN_CPB), dim3(FIN_NT), 0, (hipStream_t)stream, part, nblk, C, out);
    YH_CHECK_LAUNCH("yh_colsum");
    return YH_OK;
}

extern "C" int yh_maxpool5_fwd(const yh_bf16* x, int ldx, int B, int H, int W, int C,
                               yh_bf16* out, int ldo, int8_t* idx, yh_stream stream)
{
    YH_CHECK_ARG(B > 0 && H > 0 && W > 0 && C > 0 && C % 8 == 0, "yh_maxpool5_fwd: bad dims");
    YH_CHECK_SLICE("yh_maxpool5_fwd", x, ldx, C);
    YH_CHECK_SLICE("yh_maxpool5_fwd", out, ldo, C);
    if (idx) YH_CHECK_ARG((((uintptr_t)idx) & 7) == 0, "yh_maxpool5_fwd: idx unaligned");
    long n = (long)B * H * W * (C / 8);
    hipLaunchKernelGGL(maxpool5_fwd_kernel, dim3(ew_grid(n)), dim3(EW_THREADS), 0, (hipStream_t)stream, x, ldx, B, H, W, C / 8, out, ldo, idx);
    YH_CHECK_LAUNCH("yh_maxpool5_fwd");
    return YH_OK;
}

extern "C" int yh_maxpool5_bwd(const yh_bf16* gout, int ldgo, const int8_t* idx, int B, int H, int W, int C,
                               yh_bf16* gin, int ldgi, int accumulate, yh_stream stream)
{
    YH_CHECK_ARG(B > 0 && H > 0 && W > 0 && C > 0 && C % 8 == 0 && idx, "yh_maxpool5_bwd: bad dims");
    YH_CHECK_SLICE("yh_maxpool5_bwd", gout, ldgo, C);
    YH_CHECK_SLICE("yh_maxpool5_bwd", gin, ldgi, C);
    long n = (long)B * H * W * (C / 8);
    hipLaunchKernelGGL(maxpool5_bwd_kernel, dim3(ew_grid(n)), dim3(EW_THREADS), 0, (hipStream_t)stream, gout, ldgo, idx, B, H, W, C / 8, gin, ldgi, accumulate);
    YH_CHECK_LAUNCH("yh_maxpool5_bwd");
    return YH_OK;
}

extern "C" int yh_upsample2_bwd(const yh_bf16* ghi, int ldh, int B, int Hlo, int Wlo, int C,
                                yh_bf16* glo, int ldl, int accumulate, yh_stream stream)
{
    YH_CHECK_ARG(B > 0 && Hlo > 0 && Wlo > 0 && C > 0 && C % 8 == 0, "yh_upsample2_bwd: bad dims");
    YH_CHECK_SLICE("yh_upsample2_bwd", ghi, ldh, C);
    YH_CHECK_SLICE("yh_upsample2_bwd", glo, ldl, C);
    long n = (long)B * Hlo * Wlo * (C / 8);
    hipLaunchKernelGGL(upsample2_bwd_kernel, dim3(ew_grid(n)), dim3(EW_THREADS), 0, (hipStream_t)stream, ghi, ldh, B, Hlo, Wlo, C / 8, glo, ldl, accumulate);
    YH_CHECK_LAUNCH("yh_upsample2_bwd");
    return YH_OK;
}

extern "C" int yh_input_s2d(const float* x, int B, int Cin, int H, int W, yh_bf16* out, yh_stream stream)
{
    YH_CHECK_ARG(x && out && B > 0 && Cin > 0 && Cin <= 4 && H % 2 == 0 && W % 2 == 0, "yh_input_s2d: bad dims (Cin<=4, even H/W)");
    YH_CHECK_ARG((((uintptr_t)x) & 7) == 0 && yh_aligned16(out), "yh_input_s2d: unaligned pointers");
    long n = (long)B * (H / 2) * (W / 2);
    hipLaunchKernelGGL(input_s2d_kernel, dim3(ew_grid(n)), dim3(EW_THREADS), 0, (hipStream_t)stream, x, B, Cin, H, W, out);
    YH_CHECK_LAUNCH("yh_input_s2d");
    return YH_OK;
}

extern "C" int yh_fill_u32(void* p, uint32_t v, int64_t n_words, yh_stream stream)
{
    YH_CHECK_ARG(p && n_words >= 0 && (((uintptr_t)p) & 3) == 0, "yh_fill_u32: bad args");
    if (n_words == 0) return YH_OK;
    hipLaunchKernelGGL(fill_u32_kernel, dim3(ew_grid(n_words)), dim3(EW_THREADS), 0, (hipStream_t)stream, (uint32_t*)p, v, (long)n_words);
    YH_CHECK_LAUNCH("yh_fill_u32");
    return YH_OK;
}
