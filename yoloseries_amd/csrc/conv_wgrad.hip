// Weight gradient of the NHWC bf16 convolution on gfx950 MFMA:
//
//   dW[n][col] += sum_{m in split} gy[m][n] * A[m][col]        col = tap*Ctot + coff_k + c
//
// i.e. the GEMM  gy^T (N x M)  *  im2col(X) (M x K): both operands have the reduction index m
// (output pixel) as their slow memory axis.  Tiles are staged in their natural [pixel][channel]
// layout with 16-byte ds_write_b128 (row pitch == 64/192 B mod 256 B) and the MFMA fragments — which
// need 8 consecutive PIXELS per lane — are formed by ds_read_b64_tr_b16 (the CDNA4 transposing LDS
// read), so no element-wise transposition is ever executed.  Operands are fetched with raw buffer
// loads: per-lane offsets are fixed per thread, the walk along the pixels is a scalar offset, rows past
// the split and padding taps carry an out-of-range offset (hardware zero fill, no branches).
// The pixel reduction is split over blockIdx.y; partial tiles are added with fp32 atomics (two
// 128-byte row segments per wave instruction).  Tilings:
//   wide    (K <= 384: stem, 3x3x32 and most 1x1 layers) one block covers ALL im2col columns, so gy
//           is read once and the 9 taps of a pixel are gathered by the same block (L1/L2 hits);
//   general 128 x 128 (8 waves) or 64 x 128 tiles, 64 pixels per k-step, one column tile per block.
// Replaces autograd's conv weight gradient (train_yolov5.py:337).
#include "common.h"
#include <stdlib.h>

namespace {

typedef __attribute__((ext_vector_type(4))) short v4s;
typedef __attribute__((ext_vector_type(8))) short v8s;
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4_t;

struct WgK {
    yh_wgrad_desc d;
    int M, Ktot, Kseg, rows_per_split, ctiles, pointwise;
    unsigned gybytes, xbytes, zbytes;
    // split-M partial tiles leave the kernel either as fp32 atomics into dw (part == nullptr) or as plain stores into the
    // workspace part[split][pn][pk], summed afterwards in split order by wgrad_reduce_kernel (deterministic; the L2 atomic
    // unit — one 4-byte add per channel and clock — is what bounds the atomic form)
    float* part;
    int pn, pk;
};

// transposing read: 16-lane group reads a 4(row) x 16(col) block of 16-bit elements; lane i of the group gets
// column i of the 4 rows.  Lane 4q+p supplies the address of row q, columns 4p..4p+3.
__device__ __forceinline__ v4s tr_read(const uint16_t* p) {
    return __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) v4s*)(p));
}

constexpr int wg_pitch(int cols) { return (cols % 64 == 32) ? cols : cols + 32; }

// WN x WC waves, each wave computes (TNW*32) x (TCW*32); TK pixels per k-step
// PF2: two tiles in flight per thread (see the register sets below): +7..12 % on the stem and the general (K > 384) tilings;
// off where the second set would spill and on the 1x1 tilings, which already stream at 5.4 TB/s (measured 2 % slower with it)
// FBN: the A operand is formed in the loader from (ga, z) — the BatchNorm+SiLU backward apply of a layer without a data gradient
// (yh_wgrad_desc.bn_*): one chunk column per thread, its 8 channels' constants in registers
template <int WN, int WC, int TNW, int TCW, int TK, int MINW, bool PF2 = false, bool FBN = false>
__global__ __launch_bounds__(WN * WC * 64, MINW) void conv_wgrad_kernel(const WgK p)
{
    constexpr int NT = WN * WC * 64;              // threads per block
    constexpr int TN = WN * TNW * 32;             // out-channel rows of the tile
    constexpr int TCOLS = WC * TCW * 32;          // im2col columns of the tile
    // LDS pitches in elements: pitch bytes == 64 or 192 (mod 256) puts the 4 rows of a transposing read on
    // disjoint 64-byte bank windows (conflict-free ds_read_b64_tr_b16)
    constexpr int PA = wg_pitch(TN);
    constexpr int PB = wg_pitch(TCOLS);
    static_assert(((PA * 2) % 256 == 64 || (PA * 2) % 256 == 192) && ((PB * 2) % 256 == 64 || (PB * 2) % 256 == 192), "bad LDS pitch");
    constexpr int TPR = NT / TK;                  // threads per pixel row
    constexpr int ACH = (TN / 8 + TPR - 1) / TPR; // A chunks per thread
    constexpr int BCH = (TCOLS / 8) / TPR;        // B chunks per thread
    static_assert((TCOLS / 8) % TPR == 0, "column chunks must divide over the row's threads");
    constexpr unsigned OOB = 0x80000000u;

    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    uint16_t* sA = reinterpret_cast<uint16_t*>(smem);           // [2][TK][PA]
    uint16_t* sB = sA + 2 * TK * PA;                             // [2][TK][PB]

    const yh_wgrad_desc& d = p.d;
    const int t = threadIdx.x;
    const int lane = t & 63;
    const int wave = t >> 6;
    const int wn = wave / WC, wc = wave % WC;
    // XCD-aware block -> (tile, split) map: workgroups are dealt round-robin to the 8 XCDs by linear id, so the
    // tiles of one pixel split (which re-read the same gy / X rows) are given to ONE XCD, back to back, and share
    // its L2 instead of each XCD fetching the rows again (gridDim.y is padded to a multiple of 8).
    const int T = gridDim.x;
    const int lin = blockIdx.y * T + blockIdx.x;
    const int seq = lin >> 3;
    const int tile = seq % T;
    const int split = (seq / T) * 8 + (lin & 7);
    const int ntile = tile / p.ctiles;
    const int ctile = tile - ntile * p.ctiles;
    const int n0 = ntile * TN, col0 = ctile * TCOLS;
    const long mb0 = (long)split * p.rows_per_split;
    if (mb0 >= p.M) return;                         // padding split (uniform for the block)
    const int mbeg = (int)mb0;
    const int mend = (int)min((long)p.M, mb0 + p.rows_per_split);
    const int nkt = (mend - mbeg + TK - 1) / TK;

    const int row = t / TPR;          // pixel row within the k-step
    const int sub = t - row * TPR;
    const int HoWo = d.Ho * d.Wo;
    const int ups = d.seg.ups;
    const int Hs = d.Hi >> ups, Ws = d.Wi >> ups;
    const int C = d.seg.C;

    const __amdgpu_buffer_rsrc_t rsg = __builtin_amdgcn_make_buffer_rsrc((void*)d.gy, 0, p.gybytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsx = __builtin_amdgcn_make_buffer_rsrc((void*)d.seg.ptr, 0, p.xbytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsz = __builtin_amdgcn_make_buffer_rsrc((void*)(FBN ? d.bn_z : d.gy), 0, FBN ? p.zbytes : p.gybytes, 0x00020000);

    // per-thread constants
    unsigned voffA[ACH];
#pragma unroll
    for (int j = 0; j < ACH; ++j) {
        const int ch = (sub + TPR * j) * 8;
        voffA[j] = (ch < TN && n0 + ch < d.N) ? (unsigned)((row * d.ldg + n0 + ch) * 2) : OOB;
    }
    // FBN: gz = A*dz + (Bc*z + D), dz = ga*silu'(z*sc + sh) (bn_silu_bwd_apply_kernel, elementwise.hip): the five per-channel
    // constants of the block's TN channels live in LDS behind the tile buffers: sCst[5][TN] = sc | sh | A | Bc | D
    float* const sCst = reinterpret_cast<float*>(sB + 2 * TK * PB);
    unsigned voffZ[ACH];
#pragma unroll
    for (int j = 0; j < ACH; ++j) voffZ[j] = OOB;
    if (FBN) {
#pragma unroll
        for (int j = 0; j < ACH; ++j) {
            const int ch = (sub + TPR * j) * 8;
            voffZ[j] = (ch < TN && n0 + ch < d.N) ? (unsigned)((row * d.bn_ldz + n0 + ch) * 2) : OOB;
        }
        for (int c = t; c < TN; c += NT) {
            const int cc = n0 + c < d.N ? n0 + c : 0;
            const float mu = d.bn_ws[2 * d.N + cc], is = d.bn_ws[3 * d.N + cc];
            const float gi = d.bn_gamma[cc] * is;
            const float c1 = d.bn_coef[cc], c2 = d.bn_coef[d.N + cc];
            sCst[c] = d.bn_ws[cc]; sCst[TN + c] = d.bn_ws[d.N + cc];
            sCst[2 * TN + c] = gi;
            sCst[3 * TN + c] = -gi * is * c2;
            sCst[4 * TN + c] = gi * (mu * is * c2 - c1);
        }
        __syncthreads();
    }
    int bkh[BCH], bkw[BCH];
    unsigned bcoff[BCH];              // byte offset of the chunk's channel inside a pixel (OOB when the column is padding)
#pragma unroll
    for (int j = 0; j < BCH; ++j) {
        const int col = col0 + (sub + TPR * j) * 8;
        const bool ok = col < p.Kseg;
        const int tap = ok ? col / C : 0;
        bcoff[j] = ok ? (unsigned)((col - tap * C) * 2) : OOB;
        bkh[j] = tap / d.KW;
        bkw[j] = tap - bkh[j] * d.KW;
        if (p.pointwise && ok) bcoff[j] += (unsigned)(row * d.seg.ld * 2);
    }

    f32x16_t acc[TNW][TCW];
#pragma unroll
    for (int i = 0; i < TNW; ++i)
#pragma unroll
        for (int j = 0; j < TCW; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    // pixel of this thread's row in the first k-step, and the (row, column) advance of one k-step of TK pixels
    int pim, pho, pwo;
    {
        const int m = mbeg + row;
        pim = m / HoWo;
        const int rem = m - pim * HoWo;
        pho = rem / d.Wo;
        pwo = rem - pho * d.Wo;
    }
    const int stepH = TK / d.Wo, stepW = TK - stepH * d.Wo;

    // two register sets: the tile of k-step kt+2 is requested while tile kt is multiplied and tile kt+1 (requested one step
    // earlier) moves to LDS — two tiles in flight per thread; the layers this kernel serves are HBM bound
    u32x4_t ra0[ACH], rb0[BCH], ra1[ACH], rb1[BCH];
    u32x4_t rz0[ACH], rz1[ACH];
    bool live0 = false, live1 = false;           // FBN: this thread's row of the tile lies inside the split
    auto load_tile = [&](int kt, u32x4_t (&ra)[ACH], u32x4_t (&rb)[BCH], u32x4_t (&rz)[ACH], bool& live) {
        const int mb = mbeg + kt * TK;             // scalar
        const bool mv = mb + row < mend;
        const unsigned sg = (unsigned)mb * (unsigned)(d.ldg * 2);
#pragma unroll
        for (int j = 0; j < ACH; ++j) ra[j] = __builtin_amdgcn_raw_buffer_load_b128(rsg, mv ? voffA[j] : OOB, sg, FBN ? 2 : 0);
        if (FBN) {
            const unsigned sz = (unsigned)mb * (unsigned)(d.bn_ldz * 2);
#pragma unroll
            for (int j = 0; j < ACH; ++j) rz[j] = __builtin_amdgcn_raw_buffer_load_b128(rsz, mv ? voffZ[j] : OOB, sz, 2);
            live = mv;
        }
        if (p.pointwise) {
            const unsigned sx = (unsigned)mb * (unsigned)(d.seg.ld * 2);
#pragma unroll
            for (int j = 0; j < BCH; ++j) rb[j] = __builtin_amdgcn_raw_buffer_load_b128(rsx, mv ? bcoff[j] : OOB, sx, 2 /* nt: last reader of the layer input */);
        } else {
            // this thread's pixel (im, ho, wo) is carried from k-step to k-step (tiles are visited in order): no divisions
            const int im = pim;
            const int hb = mv ? pho * d.stride - d.pad : -(1 << 28);
            const int wb = mv ? pwo * d.stride - d.pad : -(1 << 28);
            pwo += stepW; pho += stepH;
            if (pwo >= d.Wo) { pwo -= d.Wo; ++pho; }
            while (pho >= d.Ho) { pho -= d.Ho; ++pim; }          // tiny grids: a k-step may span several images
#pragma unroll
            for (int j = 0; j < BCH; ++j) {
                const int hi = hb + bkh[j], wi = wb + bkw[j];
                const bool ok = hi >= 0 && wi >= 0 && hi < d.Hi && wi < d.Wi && bcoff[j] != OOB;
                const unsigned pix = (unsigned)((im * Hs + (hi >> ups)) * Ws + (wi >> ups));
                rb[j] = __builtin_amdgcn_raw_buffer_load_b128(rsx, ok ? pix * (unsigned)(d.seg.ld * 2) + bcoff[j] : OOB, 0, 0);
            }
        }
    };
    uint16_t* const stA = sA + row * PA + sub * 8;
    uint16_t* const stB = sB + row * PB + sub * 8;
    auto store_tile = [&](int buf, const u32x4_t (&ra)[ACH], const u32x4_t (&rb)[BCH], const u32x4_t (&rz)[ACH], const bool live) {
        if (FBN) {
#pragma unroll
            for (int j = 0; j < ACH; ++j) {
                const int ch = (sub + TPR * j) * 8;
                if (ch >= TN) continue;
                float g[8], z[8], o[8], cst[5][8];
                unpack8(make_uint4(ra[j].x, ra[j].y, ra[j].z, ra[j].w), g);
                unpack8(make_uint4(rz[j].x, rz[j].y, rz[j].z, rz[j].w), z);
#pragma unroll
                for (int q = 0; q < 5; ++q) {
                    const float4 lo = *reinterpret_cast<const float4*>(sCst + q * TN + ch);
                    const float4 hi = *reinterpret_cast<const float4*>(sCst + q * TN + ch + 4);
                    cst[q][0] = lo.x; cst[q][1] = lo.y; cst[q][2] = lo.z; cst[q][3] = lo.w;
                    cst[q][4] = hi.x; cst[q][5] = hi.y; cst[q][6] = hi.z; cst[q][7] = hi.w;
                }
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const float a = z[e] * cst[0][e] + cst[1][e];
                    const float sg = sigmoid_fast(a);
                    const float dz = g[e] * (sg * (1.f + a * (1.f - sg)));
                    o[e] = cst[2][e] * dz + (cst[3][e] * z[e] + cst[4][e]);
                }
                // rows past the split / padded channels were loaded as zeros: dz = 0, but Bc*0 + D is not — they stay zero
                const uint4 v = (live && voffZ[j] != OOB) ? pack8(o) : make_uint4(0, 0, 0, 0);
                *reinterpret_cast<u32x4_t*>(stA + buf * TK * PA + TPR * j * 8) = u32x4_t{v.x, v.y, v.z, v.w};
            }
        } else
#pragma unroll
        for (int j = 0; j < ACH; ++j)
            if ((sub + TPR * j) * 8 < TN) *reinterpret_cast<u32x4_t*>(stA + buf * TK * PA + TPR * j * 8) = ra[j];
#pragma unroll
        for (int j = 0; j < BCH; ++j) *reinterpret_cast<u32x4_t*>(stB + buf * TK * PB + TPR * j * 8) = rb[j];
    };

    // fragment addressing for the transposing reads (see tr_read)
    const int g16 = lane >> 4, i16 = lane & 15;
    const int q = i16 >> 2, pp = i16 & 3;
    const int frag_row = 8 * (g16 >> 1) + q;               // + 16*ks + 4*r
    const int frag_col = 16 * (g16 & 1) + 4 * pp;          // + 32*tile
    const uint16_t* const fa = sA + frag_row * PA + wn * TNW * 32 + frag_col;
    const uint16_t* const fb = sB + frag_row * PB + wc * TCW * 32 + frag_col;

    auto multiply = [&](int buf) {
        const uint16_t* a = fa + buf * TK * PA;
        const uint16_t* b = fb + buf * TK * PB;
#pragma unroll
        for (int ks = 0; ks < TK / 16; ++ks) {
            bf16x8_t af[TNW], bfr[TCW];
#pragma unroll
            for (int i = 0; i < TNW; ++i) {
                const uint16_t* base = a + ks * 16 * PA + i * 32;
                const v4s lo = tr_read(base);
                const v4s hi = tr_read(base + 4 * PA);
                const v8s v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
                af[i] = __builtin_bit_cast(bf16x8_t, v);
            }
#pragma unroll
            for (int j = 0; j < TCW; ++j) {
                const uint16_t* base = b + ks * 16 * PB + j * 32;
                const v4s lo = tr_read(base);
                const v4s hi = tr_read(base + 4 * PB);
                const v8s v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
                bfr[j] = __builtin_bit_cast(bf16x8_t, v);
            }
#pragma unroll
            for (int i = 0; i < TNW; ++i)
#pragma unroll
                for (int j = 0; j < TCW; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i], bfr[j], acc[i][j], 0, 0, 0);
        }
    };

    // tile j travels through register set j & 1 into LDS buffer j & 1
    load_tile(0, ra0, rb0, rz0, live0);
    store_tile(0, ra0, rb0, rz0, live0);
    if (PF2) {
        if (nkt > 1) load_tile(1, ra1, rb1, rz1, live1);
        __syncthreads();
        for (int kt = 0; kt < nkt; kt += 2) {
            if (kt + 2 < nkt) load_tile(kt + 2, ra0, rb0, rz0, live0);
            multiply(0);
            if (kt + 1 < nkt) store_tile(1, ra1, rb1, rz1, live1);
            __syncthreads();
            if (kt + 1 >= nkt) break;
            if (kt + 3 < nkt) load_tile(kt + 3, ra1, rb1, rz1, live1);
            multiply(1);
            if (kt + 2 < nkt) store_tile(0, ra0, rb0, rz0, live0);
            __syncthreads();
        }
    } else {
        __syncthreads();
        for (int kt = 0; kt < nkt; ++kt) {
            const int buf = kt & 1;
            const bool more = (kt + 1) < nkt;
            if (more) load_tile(kt + 1, ra0, rb0, rz0, live0);
            multiply(buf);
            if (more) store_tile(buf ^ 1, ra0, rb0, rz0, live0);
            __syncthreads();
        }
    }

    // C[n][col]: lane holds column (lane&31), rows (r&3)+8*(r>>2)+4*(lane>>5)
    if (p.part) {
        float* const pb = p.part + (size_t)split * p.pn * p.pk;
#pragma unroll
        for (int j = 0; j < TCW; ++j) {
            const int col = col0 + (wc * TCW + j) * 32 + (lane & 31);
            if (col >= p.Kseg) continue;
#pragma unroll
            for (int i = 0; i < TNW; ++i) {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int n = n0 + (wn * TNW + i) * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                    if (n < d.N) __builtin_nontemporal_store(acc[i][j][r], pb + (size_t)n * p.pk + col);
                }
            }
        }
        return;
    }
#pragma unroll
    for (int j = 0; j < TCW; ++j) {
        const int col = col0 + (wc * TCW + j) * 32 + (lane & 31);
        if (col >= p.Kseg) continue;
        const int tap = col / C;
        const int cc = col - tap * C;
        float* base = d.dw + (size_t)tap * d.Ctot + d.coff_k + cc;
#pragma unroll
        for (int i = 0; i < TNW; ++i) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int n = n0 + (wn * TNW + i) * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                if (n < d.N) atomicAdd(base + (size_t)n * p.Ktot, acc[i][j][r]);
            }
        }
    }
}

typedef float f32x4_nt __attribute__((ext_vector_type(4)));
__device__ __forceinline__ float4 ld4nt(const float* p) {
    const f32x4_nt v = __builtin_nontemporal_load(reinterpret_cast<const f32x4_nt*>(p));
    return make_float4(v.x, v.y, v.z, v.w);
}

// dw[n][tap*Ctot + coff_k + c] += sum over the splits (in split order) of part[s][n][col].  A block sums 64 groups of 4 columns;
// its SG wave-rows take the splits s = sg, sg + SG, ... and meet in LDS (fixed order: the result does not depend on timing).
__global__ __launch_bounds__(1024) void wgrad_reduce_kernel(const float* __restrict__ part, int splits, int pn, int pk, int N, int Kseg,
                                                            int C, int Ctot, int coff_k, int Ktot, float* __restrict__ dw)
{
    __shared__ float4 sred[16][64];
    const int SG = blockDim.y;
    const int kq = Kseg >> 2;                              // float4 groups per row
    const long item = (long)blockIdx.x * 64 + threadIdx.x;
    const bool ok = item < (long)N * kq;
    const int n = ok ? (int)(item / kq) : 0;
    const int col = ok ? (int)(item - (long)n * kq) * 4 : 0;
    const float* src = part + (size_t)n * pk + col;
    const size_t sstride = (size_t)pn * pk;
    float4 a = make_float4(0.f, 0.f, 0.f, 0.f);
    if (ok) {
        int s = threadIdx.y;
        for (; s + 3 * SG < splits; s += 4 * SG) {
            const float4 v0 = ld4nt((src + (size_t)s * sstride));
            const float4 v1 = ld4nt((src + (size_t)(s + SG) * sstride));
            const float4 v2 = ld4nt((src + (size_t)(s + 2 * SG) * sstride));
            const float4 v3 = ld4nt((src + (size_t)(s + 3 * SG) * sstride));
            a.x += v0.x; a.y += v0.y; a.z += v0.z; a.w += v0.w;
            a.x += v1.x; a.y += v1.y; a.z += v1.z; a.w += v1.w;
            a.x += v2.x; a.y += v2.y; a.z += v2.z; a.w += v2.w;
            a.x += v3.x; a.y += v3.y; a.z += v3.z; a.w += v3.w;
        }
        for (; s < splits; s += SG) {
            const float4 v0 = ld4nt((src + (size_t)s * sstride));
            a.x += v0.x; a.y += v0.y; a.z += v0.z; a.w += v0.w;
        }
    }
    sred[threadIdx.y][threadIdx.x] = a;
    __syncthreads();
    if (threadIdx.y == 0 && ok) {
        for (int g = 1; g < SG; ++g) {
            const float4 v = sred[g][threadIdx.x];
            a.x += v.x; a.y += v.y; a.z += v.z; a.w += v.w;
        }
        const int tap = col / C;
        float4* dst = reinterpret_cast<float4*>(dw + (size_t)n * Ktot + (size_t)tap * Ctot + coff_k + (col - tap * C));
        float4 o = *dst;
        o.x += a.x; o.y += a.y; o.z += a.z; o.w += a.w;
        *dst = o;
    }
}

template <int WN, int WC, int TNW, int TCW, int TK>
constexpr size_t wg_smem() {
    // tiles [2][TK][PA + PB] + (fused BatchNorm backward) the per-channel constants [5][TN]
    return (size_t)2 * TK * (wg_pitch(WN * TNW * 32) + wg_pitch(WC * TCW * 32)) * 2 + 5 * WN * TNW * 32 * 4;
}

// instantiation used for a layer (N out channels, Kseg im2col columns of the segment)
// tile_k == 128 asks for the general 128-column tiling on a layer that would get a wide one (64 x Kseg tiles): measured 17-29 %
// faster on 1x1 layers with >= 128 channels on both sides (fewer re-reads of gy / x per tile, half the blocks per output element)
bool wg_wide(int Kseg, int tile_k) { return Kseg <= 384 && !(tile_k == 128 && Kseg >= 128); }
int wg_config(int N, int Kseg, int tile_k)
{
    if (wg_wide(Kseg, tile_k)) {
        if (N <= 32) return Kseg <= 256 ? 0 : 1;
        return Kseg <= 128 ? 2 : (Kseg <= 256 ? 3 : 4);
    }
    return N <= 64 ? 5 : 6;
}
const char* const wg_names[7] = {
    "conv_wgrad_kernel<1, 4, 1, 2, 32, 3, true, false>", "conv_wgrad_kernel<1, 4, 1, 3, 32, 3, false, false>", "conv_wgrad_kernel<1, 4, 2, 1, 32, 4, false, false>",
    "conv_wgrad_kernel<1, 4, 2, 2, 32, 3, false, false>", "conv_wgrad_kernel<1, 4, 2, 3, 32, 2, false, false>", "conv_wgrad_kernel<2, 2, 1, 2, 64, 3, true, false>",
    "conv_wgrad_kernel<4, 2, 1, 2, 64, 4, true, false>"};

}  // namespace

/* name of the kernel instantiation yh_conv_wgrad launches for a layer, as profilers print it */
extern "C" const char* yh_conv_wgrad_kernel_name2(int N, int Kseg, int tile_k)
{
    const int c = wg_config(N, Kseg, tile_k);
    if (c == 6 && tile_k == 32) return "conv_wgrad_kernel<4, 2, 1, 2, 32, 2, true, false>";
    if (c == 6 && tile_k == 35) return "conv_wgrad_kernel<2, 2, 2, 2, 32, 2, false, false>";
    return (c == 0 && Kseg <= 160) ? "conv_wgrad_kernel<1, 5, 1, 1, 32, 3, true, false>" : wg_names[c];
}
extern "C" const char* yh_conv_wgrad_kernel_name(int N, int Kseg) { return yh_conv_wgrad_kernel_name2(N, Kseg, 0); }
/* 1 if the patch form (tile_k 40, conv_wgp_kernel) applies to this descriptor */
extern "C" int yh_conv_wgrad_patch_ok(const yh_wgrad_desc* d) { return d ? yh_wgp_ok(d) : 0; }
/* tile_k 129 (conv_wgs_kernel): number of 128 x 128 tiles of the layer, 0 when the form does not apply; its profiler name */
extern "C" int yh_conv_wgrad_wave_tiles(const yh_wgrad_desc* d) { return d ? yh_wgs_tiles(d) : 0; }
extern "C" const char* yh_conv_wgrad_wave_name(const yh_wgrad_desc* d) { return d ? yh_wgs_name(d) : ""; }

/* tile the kernel will use for a layer: rows (out channels) x im2col columns per block; used by the host to size `splits` */
extern "C" int yh_conv_wgrad_tiles2(int N, int Kseg, int tile_k)
{
    if (wg_wide(Kseg, tile_k)) return N <= 32 ? (N + 31) / 32 : (N + 63) / 64;
    const int tn = N <= 64 ? 64 : 128;
    return ((N + tn - 1) / tn) * ((Kseg + 127) / 128);
}
extern "C" int yh_conv_wgrad_tiles(int N, int Kseg) { return yh_conv_wgrad_tiles2(N, Kseg, 0); }

static int conv_wgrad_launch(const yh_wgrad_desc* d, yh_stream stream);

// pixels per k-step / effective split count of a launch (shared by the launcher and the workspace query)
static void wg_split_plan(const yh_wgrad_desc* d, long M, int* tk_out, int* rps_out, int* splits_out)
{
    const int Kseg = d->KH * d->KW * d->seg.C;
    const bool wide = wg_wide(Kseg, d->tile_k);
    const int cfg = wg_config(d->N, Kseg, d->tile_k);
    const bool tk64 = wide && d->tile_k == 64 && cfg <= 3;
    const int TK = ((wide && !tk64) || (cfg == 6 && (d->tile_k == 32 || d->tile_k == 35))) ? 32 : 64;
    int splits = d->splits < 1 ? 1 : d->splits;
    int rps = (int)((M + splits - 1) / splits);
    rps = ((rps + TK - 1) / TK) * TK;
    splits = (int)((M + rps - 1) / rps);
    *tk_out = TK; *rps_out = rps; *splits_out = splits;
}

/* bytes of workspace (yh_wgrad_desc.partial) a launch with these dims / splits needs for the plain-store partial tiles */
extern "C" size_t yh_conv_wgrad_ws_bytes(const yh_wgrad_desc* d)
{
    if (!d || d->B <= 0 || d->Ho <= 0 || d->Wo <= 0 || d->N <= 0 || d->seg.C <= 0 || d->KH <= 0 || d->KW <= 0) return 0;
    const long M = (long)d->B * d->Ho * d->Wo;
    int tk, rps, splits;
    wg_split_plan(d, M, &tk, &rps, &splits);
    const size_t pn = ((size_t)d->N + 7) / 8 * 8, pk = (size_t)d->KH * d->KW * d->seg.C;
    return (size_t)splits * pn * pk * sizeof(float);
}

/* The kernels address gy and the input segment with 32-bit buffer offsets.  A launch whose operands reach 2 GiB is split over the
 * batch: every part is a launch of its own on a sub-range of images (pointers advanced, dw accumulated by the same atomics). */
extern "C" int yh_conv_wgrad(const yh_wgrad_desc* d, yh_stream stream)
{
    YH_CHECK_ARG(d != nullptr && d->B > 0 && d->Ho > 0 && d->Wo > 0 && d->Hi > 0 && d->Wi > 0, "yh_conv_wgrad: null desc / bad dims");
    const unsigned long gy_img = (unsigned long)d->Ho * d->Wo * d->ldg * 2;
    const unsigned long x_img = (unsigned long)(d->Hi >> d->seg.ups) * (d->Wi >> d->seg.ups) * d->seg.ld * 2;
    const unsigned long lim = (1ul << 31) - 4096;
    if (d->partial) YH_CHECK_ARG(yh_aligned16(d->partial) && d->partial_bytes >= yh_conv_wgrad_ws_bytes(d), "yh_conv_wgrad: workspace too small / unaligned");
    if (d->tile_k == 129 && yh_wgs_ok(d)) return yh_wgs_run(d, stream);   // wave-private tiles + stream-K (conv_wgs.hip), else the forms below
    if (d->tile_k == 40 && yh_wgp_ok(d)) {          // patch form (conv_wgp.hip) where it is eligible, else the im2col form below
        YH_CHECK_ARG(d->gy && yh_aligned16(d->gy) && d->ldg % 8 == 0 && d->seg.ptr && yh_aligned16(d->seg.ptr) && d->seg.ld % 8 == 0 && d->dw &&
                     d->coff_k % 8 == 0 && d->coff_k + d->seg.C <= d->Ctot, "yh_conv_wgrad: bad operands");
        return yh_wgp_run(d, stream);
    }
    if (gy_img * d->B < lim && x_img * d->B < lim) return conv_wgrad_launch(d, stream);
    const unsigned long per = gy_img > x_img ? gy_img : x_img;
    YH_CHECK_ARG(per < lim, "yh_conv_wgrad: a single image needs a 2 GiB operand");
    const int chunk = (int)(lim / per);
    const int parts = (d->B + chunk - 1) / chunk;
    for (int b0 = 0; b0 < d->B; b0 += chunk) {
        yh_wgrad_desc p = *d;
        p.B = d->B - b0 < chunk ? d->B - b0 : chunk;
        p.gy = d->gy + (size_t)b0 * (gy_img / 2);
        p.seg.ptr = d->seg.ptr + (size_t)b0 * (x_img / 2);
        p.splits = (d->splits + parts - 1) / parts;
        const int rc = conv_wgrad_launch(&p, stream);
        if (rc != YH_OK) return rc;
    }
    return YH_OK;
}

static int conv_wgrad_launch(const yh_wgrad_desc* d, yh_stream stream)
{
    YH_CHECK_ARG(d != nullptr, "yh_conv_wgrad: null desc");
    YH_CHECK_ARG(d->gy && yh_aligned16(d->gy) && d->ldg % 8 == 0, "yh_conv_wgrad: gy null/unaligned");
    YH_CHECK_ARG(d->N > 0 && (d->N % 8 == 0 || d->ldg >= ((d->N + 7) / 8) * 8), "yh_conv_wgrad: N=%d needs ldg padded to 8", d->N);
    YH_CHECK_ARG(d->seg.ptr && yh_aligned16(d->seg.ptr) && d->seg.C > 0 && d->seg.C % 8 == 0 && d->seg.ld % 8 == 0, "yh_conv_wgrad: segment misaligned");
    YH_CHECK_ARG(d->coff_k % 8 == 0 && d->coff_k + d->seg.C <= d->Ctot, "yh_conv_wgrad: bad channel offset");
    YH_CHECK_ARG(d->stride == 1 || d->stride == 2, "yh_conv_wgrad: bad stride");
    YH_CHECK_ARG(d->KH > 0 && d->KW > 0 && d->KH <= 7 && d->KW <= 7, "yh_conv_wgrad: bad kernel size");
    YH_CHECK_ARG((d->Hi + 2 * d->pad - d->KH) / d->stride + 1 == d->Ho && (d->Wi + 2 * d->pad - d->KW) / d->stride + 1 == d->Wo,
                 "yh_conv_wgrad: geometry mismatch");
    YH_CHECK_ARG(d->dw != nullptr && d->splits >= 1, "yh_conv_wgrad: dw null / bad splits");
    if (d->seg.ups) YH_CHECK_ARG(d->Hi % 2 == 0 && d->Wi % 2 == 0, "yh_conv_wgrad: upsampled segment needs even dims");
    long M = (long)d->B * d->Ho * d->Wo;
    YH_CHECK_ARG(M < (1L << 31) - 256, "yh_conv_wgrad: too many pixels");
    WgK k;
    k.d = *d;
    k.M = (int)M;
    k.Ktot = d->KH * d->KW * d->Ctot;
    k.Kseg = d->KH * d->KW * d->seg.C;          // im2col columns of THIS segment
    const unsigned long gyb = ((unsigned long)(M - 1) * d->ldg + ((d->N + 7) / 8) * 8) * 2;
    const unsigned long npix = (unsigned long)d->B * (d->Hi >> d->seg.ups) * (d->Wi >> d->seg.ups);
    const unsigned long xb = ((npix - 1) * d->seg.ld + d->seg.C) * 2;
    YH_CHECK_ARG(gyb < (1ul << 31) && xb < (1ul << 31), "yh_conv_wgrad: operands of 2 GiB or more are not supported");
    k.gybytes = (unsigned)gyb;
    k.xbytes = (unsigned)xb;
    k.pointwise = (d->KH == 1 && d->KW == 1 && d->stride == 1 && d->pad == 0 && !d->seg.ups) ? 1 : 0;
    hipStream_t st = (hipStream_t)stream;
    const bool wide = wg_wide(k.Kseg, d->tile_k);
    // pixels per k-step: 32 for the wide tilings (more resident blocks), 64 when asked for (d->tile_k == 64: half the barriers and
    // twice the loads in flight per thread; the engine times both per layer) and for the general tiles
    const int cfg = wg_config(d->N, k.Kseg, d->tile_k);
    const bool tk64 = wide && d->tile_k == 64 && cfg <= 3;
    int TK, rps, splits;
    wg_split_plan(d, M, &TK, &rps, &splits);
    k.rows_per_split = rps;
    YH_CHECK_ARG(splits <= 65528, "yh_conv_wgrad: too many splits");
    // a single split needs no reduction: its tile goes straight to dw (the atomic form is then a plain add per element)
    k.part = (d->partial && splits > 1) ? d->partial : nullptr;
    k.pn = (d->N + 7) / 8 * 8;
    k.pk = k.Kseg;
#define YH_WG(WN_, WC_, TNW_, TCW_, TK_, MINW_, NT_, PF2_)                                                      \
    do {                                                                                                        \
        dim3 grid((NT_) * k.ctiles, (splits + 7) / 8 * 8);                                                      \
        conv_wgrad_kernel<WN_, WC_, TNW_, TCW_, TK_, MINW_, PF2_><<<grid, dim3(WN_ * WC_ * 64), wg_smem<WN_, WC_, TNW_, TCW_, TK_>(), st>>>(k); \
    } while (0)
    // the same with the BatchNorm-backward apply fused into the A-operand loader (bn_z): the wide tilings a stem layer gets
#define YH_WGF(WN_, WC_, TNW_, TCW_, TK_, MINW_, NT_, PF2_)                                                     \
    do {                                                                                                        \
        dim3 grid((NT_) * k.ctiles, (splits + 7) / 8 * 8);                                                      \
        if (fbn) conv_wgrad_kernel<WN_, WC_, TNW_, TCW_, TK_, MINW_, PF2_, true><<<grid, dim3(WN_ * WC_ * 64), wg_smem<WN_, WC_, TNW_, TCW_, TK_>(), st>>>(k); \
        else     conv_wgrad_kernel<WN_, WC_, TNW_, TCW_, TK_, MINW_, PF2_><<<grid, dim3(WN_ * WC_ * 64), wg_smem<WN_, WC_, TNW_, TCW_, TK_>(), st>>>(k); \
    } while (0)
    const bool fbn = d->bn_z != nullptr;
    if (fbn) {
        YH_CHECK_ARG(wide && (cfg == 0 || cfg == 3), "yh_conv_wgrad: the fused BatchNorm backward (bn_z) needs a wide tiling with <= 256 im2col columns "
                     "(N <= 32: <= 256 columns; N > 32: 129..256 columns)");
        YH_CHECK_ARG(yh_aligned16(d->bn_z) && d->bn_ldz % 8 == 0 && d->bn_ldz >= d->N && d->bn_ws && d->bn_gamma && d->bn_coef && d->N % 8 == 0,
                     "yh_conv_wgrad: bad fused-BatchNorm operands");
        const unsigned long zb = ((unsigned long)(M - 1) * d->bn_ldz + d->N) * 2;
        YH_CHECK_ARG(zb < (1ul << 31), "yh_conv_wgrad: bn_z of 2 GiB or more is not supported");
        k.zbytes = (unsigned)zb;
    } else k.zbytes = 0;
    k.ctiles = wide ? 1 : (k.Kseg + 127) / 128;
    switch (cfg) {
    case 0:
        if (k.Kseg <= 160) { if (tk64) YH_WGF(1, 5, 1, 1, 64, 3, (d->N + 31) / 32, true); else YH_WGF(1, 5, 1, 1, 32, 3, (d->N + 31) / 32, true); }  // stem: 5 waves x one 32-column tile (144 of 160 used)
        else               { if (tk64) YH_WGF(1, 4, 1, 2, 64, 2, (d->N + 31) / 32, true); else YH_WGF(1, 4, 1, 2, 32, 3, (d->N + 31) / 32, true); }
        break;
    case 1: if (tk64) YH_WG(1, 4, 1, 3, 64, 2, (d->N + 31) / 32, false); else YH_WG(1, 4, 1, 3, 32, 3, (d->N + 31) / 32, false); break;
    case 2: if (tk64) YH_WG(1, 4, 2, 1, 64, 2, (d->N + 63) / 64, false); else YH_WG(1, 4, 2, 1, 32, 4, (d->N + 63) / 64, false); break;
    case 3: if (tk64) YH_WGF(1, 4, 2, 2, 64, 2, (d->N + 63) / 64, false); else YH_WGF(1, 4, 2, 2, 32, 3, (d->N + 63) / 64, false); break;
    case 4: YH_WG(1, 4, 2, 3, 32, 2, (d->N + 63) / 64, false); break;
    case 5: YH_WG(2, 2, 1, 2, 64, 3, (d->N + 63) / 64, true); break;
    default:
        // 8 waves of 32 x 64, 64-pixel k-steps: 82 KB of LDS, ONE block per CU whose every k-step ends in a block-wide barrier.
        // tile_k == 32: the same waves on 32-pixel k-steps (41 KB: two blocks per CU, one block's barrier is covered by the other's
        // MFMAs: +10 % on the 3x3 layers of YOLOv5s at 40 x 40); tile_k == 35: FOUR waves of 64 x 64 on 32-pixel k-steps (two
        // transposing reads per MFMA instead of three: +15 % at 20 x 20).  Measured and dropped: three blocks per CU (equal), the
        // 4-wave tile with two register sets in flight (spills: half the speed).  The engine times the three per layer.
        if (d->tile_k == 32)      YH_WG(4, 2, 1, 2, 32, 2, (d->N + 127) / 128, true);
        else if (d->tile_k == 35) YH_WG(2, 2, 2, 2, 32, 2, (d->N + 127) / 128, false);
        else                      YH_WG(4, 2, 1, 2, 64, 4, (d->N + 127) / 128, true);
        break;
    }
#undef YH_WG
#undef YH_WGF
    YH_CHECK_LAUNCH("yh_conv_wgrad");
    if (k.part) {
        const long items = (long)d->N * (k.Kseg / 4);
        int sg = 1;
        while (sg < 16 && sg * 8 < splits) sg *= 2;
        wgrad_reduce_kernel<<<dim3((unsigned)((items + 63) / 64)), dim3(64, sg), 0, st>>>(k.part, splits, k.pn, k.pk, d->N, k.Kseg, d->seg.C, d->Ctot,
                                                                                        d->coff_k, k.Ktot, d->dw);
        YH_CHECK_LAUNCH("yh_conv_wgrad(reduce)");
    }
    return YH_OK;
}
