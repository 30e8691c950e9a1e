// Weight gradient of the NHWC bf16 convolution on gfx950 MFMA:
//
//   dW[n][col] += sum_{m in split} gy[m][n] * A[m][col]        col = tap*Ctot + coff_k + c
//
// i.e. the GEMM  gy^T (N x M)  *  im2col(X) (M x K): both operands have the reduction index m
// (output pixel) as their slow memory axis.  Tiles are staged in their natural [pixel][channel]
// layout with 16-byte ds_write_b128 (row pitch == 64 B mod 256 B) and the MFMA fragments — which
// need 8 consecutive PIXELS per lane — are formed by ds_read_b64_tr_b16 (the CDNA4 transposing LDS
// read), so no element-wise transposition is ever executed.  The pixel reduction is split over
// blockIdx.y; partial tiles are added with fp32 atomics (two 128-byte row segments per wave
// instruction).  Two tilings:
//   wide    (K <= 384: stem, 3x3x32 and most 1x1 layers) one block covers ALL im2col columns, so gy
//           is read once and the 9 taps of a pixel are gathered by the same block (L1/L2 hits);
//   general 64 x 64 tiles, 64 pixels per k-step.
// Replaces autograd's conv weight gradient (train_yolov5.py:337).
#include "common.h"

namespace {

typedef __attribute__((ext_vector_type(4))) short v4s;

struct WgK {
    yh_wgrad_desc d;
    int M, Ktot, Kseg, rows_per_split, ctiles;
};

// transposing read: 16-lane group reads a 4(row) x 16(col) block of 16-bit elements; lane i of the group gets
// column i of the 4 rows.  Lane 4q+p supplies the address of row q, columns 4p..4p+3.
__device__ __forceinline__ v4s tr_read(const uint16_t* p) {
    return __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) v4s*)(p));
}

constexpr int wg_pitch(int cols) { return (cols % 64 == 32) ? cols : cols + 32; }

// WN x WC waves, each wave computes (TNW*32) x (TCW*32); TK pixels per k-step
template <int WN, int WC, int TNW, int TCW, int TK>
__global__ __launch_bounds__(256) void conv_wgrad_kernel(const WgK p)
{
    constexpr int TN = WN * TNW * 32;             // out-channel rows of the tile
    constexpr int TCOLS = WC * TCW * 32;          // im2col columns of the tile
    // LDS pitches in elements: pitch bytes == 64 or 192 (mod 256) puts the 4 rows of a transposing read on
    // disjoint 64-byte bank windows (conflict-free ds_read_b64_tr_b16)
    constexpr int PA = wg_pitch(TN);
    constexpr int PB = wg_pitch(TCOLS);
    static_assert(((PA * 2) % 256 == 64 || (PA * 2) % 256 == 192) && ((PB * 2) % 256 == 64 || (PB * 2) % 256 == 192), "bad LDS pitch");
    constexpr int TPR = 256 / TK;                 // threads per pixel row
    constexpr int ACH = (TN / 8 + TPR - 1) / TPR; // A chunks per thread
    constexpr int BCH = (TCOLS / 8) / TPR;        // B chunks per thread
    static_assert((TCOLS / 8) % TPR == 0, "column chunks must divide over the row's threads");

    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    uint16_t* sA = reinterpret_cast<uint16_t*>(smem);           // [2][TK][PA]
    uint16_t* sB = sA + 2 * TK * PA;                             // [2][TK][PB]

    const yh_wgrad_desc& d = p.d;
    const int t = threadIdx.x;
    const int lane = t & 63;
    const int wave = t >> 6;
    const int wn = wave / WC, wc = wave % WC;
    const int ntile = blockIdx.x / p.ctiles;
    const int ctile = blockIdx.x - ntile * p.ctiles;
    const int n0 = ntile * TN, col0 = ctile * TCOLS;
    const int mbeg = blockIdx.y * p.rows_per_split;
    const int mend = min(p.M, mbeg + p.rows_per_split);
    if (mbeg >= mend) return;
    const int nkt = (mend - mbeg + TK - 1) / TK;

    const int row = t / TPR;          // pixel row within the k-step
    const int sub = t - row * TPR;
    const int HoWo = d.Ho * d.Wo;
    const int ups = d.seg.ups;
    const int Hs = d.Hi >> ups, Ws = d.Wi >> ups;
    const int C = d.seg.C;

    // per-thread column chunks: (tap, channel) is fixed for the whole reduction
    int bkh[BCH], bkw[BCH], bc[BCH];
    bool bok[BCH];
#pragma unroll
    for (int j = 0; j < BCH; ++j) {
        const int col = col0 + (sub + TPR * j) * 8;
        bok[j] = col < p.Kseg;
        const int tap = bok[j] ? col / C : 0;
        bc[j] = col - tap * C;
        bkh[j] = tap / d.KW;
        bkw[j] = tap - bkh[j] * d.KW;
    }

    f32x16_t acc[TNW][TCW];
#pragma unroll
    for (int i = 0; i < TNW; ++i)
#pragma unroll
        for (int j = 0; j < TCW; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    uint4 ra[ACH], rb[BCH];
    auto load_tile = [&](int kt) {
        const int m = mbeg + kt * TK + row;
        const bool mv = m < mend;
#pragma unroll
        for (int j = 0; j < ACH; ++j) {
            const int ch = (sub + TPR * j) * 8;
            uint4 v = make_uint4(0, 0, 0, 0);
            if (mv && ch < TN && n0 + ch < d.N) v = *reinterpret_cast<const uint4*>(d.gy + (size_t)m * d.ldg + n0 + ch);
            ra[j] = v;
        }
        int im = 0, hb = 0, wb = 0;
        if (mv) {
            im = m / HoWo;
            const int rem = m - im * HoWo;
            const int ho = rem / d.Wo;
            hb = ho * d.stride - d.pad;
            wb = (rem - ho * d.Wo) * d.stride - d.pad;
        }
#pragma unroll
        for (int j = 0; j < BCH; ++j) {
            uint4 v = make_uint4(0, 0, 0, 0);
            const int hi = hb + bkh[j], wi = wb + bkw[j];
            if (mv && bok[j] && hi >= 0 && wi >= 0 && hi < d.Hi && wi < d.Wi) {
                const size_t pix = ((size_t)im * Hs + (hi >> ups)) * Ws + (wi >> ups);
                v = *reinterpret_cast<const uint4*>(d.seg.ptr + pix * d.seg.ld + bc[j]);
            }
            rb[j] = v;
        }
    };
    auto store_tile = [&](int buf) {
        uint16_t* a = sA + buf * TK * PA + row * PA;
        uint16_t* b = sB + buf * TK * PB + row * PB;
#pragma unroll
        for (int j = 0; j < ACH; ++j) {
            const int ch = (sub + TPR * j) * 8;
            if (ch < TN) *reinterpret_cast<uint4*>(a + ch) = ra[j];
        }
#pragma unroll
        for (int j = 0; j < BCH; ++j) *reinterpret_cast<uint4*>(b + (sub + TPR * j) * 8) = rb[j];
    };

    // fragment addressing for the transposing reads (see tr_read)
    const int g16 = lane >> 4, i16 = lane & 15;
    const int q = i16 >> 2, pp = i16 & 3;
    const int frag_row = 8 * (g16 >> 1) + q;               // + 16*ks + 4*r
    const int frag_col = 16 * (g16 & 1) + 4 * pp;          // + 32*tile

    load_tile(0);
    store_tile(0);
    __syncthreads();
    for (int kt = 0; kt < nkt; ++kt) {
        const int buf = kt & 1;
        const bool more = (kt + 1) < nkt;
        if (more) load_tile(kt + 1);
        const uint16_t* a = sA + buf * TK * PA;
        const uint16_t* b = sB + buf * TK * PB;
#pragma unroll
        for (int ks = 0; ks < TK / 16; ++ks) {
            bf16x8_t af[TNW], bfr[TCW];
            const int r0 = ks * 16 + frag_row;
#pragma unroll
            for (int i = 0; i < TNW; ++i) {
                const uint16_t* base = a + r0 * PA + (wn * TNW + i) * 32 + frag_col;
                v4s lo = tr_read(base);
                v4s hi = tr_read(base + 4 * PA);
                typedef __attribute__((ext_vector_type(8))) short v8s;
                v8s v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
                af[i] = __builtin_bit_cast(bf16x8_t, v);
            }
#pragma unroll
            for (int j = 0; j < TCW; ++j) {
                const uint16_t* base = b + r0 * PB + (wc * TCW + j) * 32 + frag_col;
                v4s lo = tr_read(base);
                v4s hi = tr_read(base + 4 * PB);
                typedef __attribute__((ext_vector_type(8))) short v8s;
                v8s v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
                bfr[j] = __builtin_bit_cast(bf16x8_t, v);
            }
#pragma unroll
            for (int i = 0; i < TNW; ++i)
#pragma unroll
                for (int j = 0; j < TCW; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i], bfr[j], acc[i][j], 0, 0, 0);
        }
        if (more) store_tile(buf ^ 1);
        __syncthreads();
    }

    // C[n][col]: lane holds column (lane&31), rows (r&3)+8*(r>>2)+4*(lane>>5)
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

template <int WN, int WC, int TNW, int TCW, int TK>
constexpr size_t wg_smem() {
    return (size_t)2 * TK * (wg_pitch(WN * TNW * 32) + wg_pitch(WC * TCW * 32)) * 2;
}

}  // namespace

extern "C" int yh_conv_wgrad(const yh_wgrad_desc* d, yh_stream stream)
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
    hipStream_t st = (hipStream_t)stream;
    const bool wide = k.Kseg <= 384;
    const int TK = wide ? 32 : 64;
    int splits = d->splits;
    int rps = (int)((M + splits - 1) / splits);
    rps = ((rps + TK - 1) / TK) * TK;
    k.rows_per_split = rps;
    splits = (int)((M + rps - 1) / rps);
    YH_CHECK_ARG(splits <= 65535, "yh_conv_wgrad: too many splits");
    if (wide) {
        if (d->N <= 32 && k.Kseg <= 256) {
            k.ctiles = 1;
            dim3 grid(((d->N + 31) / 32) * k.ctiles, splits);
            conv_wgrad_kernel<1, 4, 1, 2, 32><<<grid, dim3(256), wg_smem<1, 4, 1, 2, 32>(), st>>>(k);
        } else if (d->N <= 32) {
            k.ctiles = 1;
            dim3 grid(((d->N + 31) / 32) * k.ctiles, splits);
            conv_wgrad_kernel<1, 4, 1, 3, 32><<<grid, dim3(256), wg_smem<1, 4, 1, 3, 32>(), st>>>(k);
        } else if (k.Kseg <= 128) {
            k.ctiles = 1;
            dim3 grid(((d->N + 63) / 64) * k.ctiles, splits);
            conv_wgrad_kernel<1, 4, 2, 1, 32><<<grid, dim3(256), wg_smem<1, 4, 2, 1, 32>(), st>>>(k);
        } else if (k.Kseg <= 256) {
            k.ctiles = 1;
            dim3 grid(((d->N + 63) / 64) * k.ctiles, splits);
            conv_wgrad_kernel<1, 4, 2, 2, 32><<<grid, dim3(256), wg_smem<1, 4, 2, 2, 32>(), st>>>(k);
        } else {
            k.ctiles = 1;
            dim3 grid(((d->N + 63) / 64) * k.ctiles, splits);
            conv_wgrad_kernel<1, 4, 2, 3, 32><<<grid, dim3(256), wg_smem<1, 4, 2, 3, 32>(), st>>>(k);
        }
    } else {
        k.ctiles = (k.Kseg + 63) / 64;
        dim3 grid(((d->N + 63) / 64) * k.ctiles, splits);
        conv_wgrad_kernel<2, 2, 1, 1, 64><<<grid, dim3(256), wg_smem<2, 2, 1, 1, 64>(), st>>>(k);
    }
    YH_CHECK_LAUNCH("yh_conv_wgrad");
    return YH_OK;
}
