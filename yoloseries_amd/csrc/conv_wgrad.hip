// Weight gradient of the NHWC bf16 convolution on gfx950 MFMA:
//
//   dW[n][tap*Ctot + coff_k + c] += sum_{m in split} gy[m][n] * X[src(m,tap)][c]
//
// GEMM view: rows = output channels n, cols = input channels c (one tap per block),
// reduction = output pixels m, split over blockIdx.z; partial results are added with
// fp32 atomics (two 128-byte row segments per wave instruction, the shape the memory
// side executes at full atomic rate).  Both operands have the reduction index as
// their slow memory axis, so the 16-byte row chunks loaded from HBM are transposed
// while being written to LDS ([channel][pixel], 8 x ds_write_b16), after which the
// MFMA fragments are the same conflict-free ds_read_b128 as in conv_igemm.hip.
// Replaces autograd's conv weight gradient (train_yolov5.py:337).
#include "common.h"

namespace {

constexpr int TK = 32;     // pixels per k-tile
constexpr int TT = 64;     // tile: 64 out-channels x 64 in-channels
constexpr int LP = 40;     // LDS row pitch in elements (80 B)

struct WgK {
    yh_wgrad_desc d;
    int M, Ktot, rows_per_split, ctiles;
};

__global__ __launch_bounds__(256) void conv_wgrad_kernel(const WgK p)
{
    __shared__ __attribute__((aligned(16))) uint16_t sA[2][TT * LP];   // [n][m]
    __shared__ __attribute__((aligned(16))) uint16_t sB[2][TT * LP];   // [c][m]

    const yh_wgrad_desc& d = p.d;
    const int t = threadIdx.x;
    const int lane = t & 63;
    const int wave = t >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int ntile = blockIdx.x / p.ctiles;
    const int ctile = blockIdx.x - ntile * p.ctiles;
    const int n0 = ntile * TT, c0 = ctile * TT;
    const int tap = blockIdx.y;
    const int kh = tap / d.KW, kw = tap - kh * d.KW;
    const int mbeg = blockIdx.z * p.rows_per_split;
    const int mend = min(p.M, mbeg + p.rows_per_split);
    if (mbeg >= mend) return;
    const int nkt = (mend - mbeg + TK - 1) / TK;

    const int row = t >> 3;        // pixel within k-tile (0..31)
    const int ch = (t & 7) * 8;    // channel chunk start within tile
    const int HoWo = d.Ho * d.Wo;
    const int ups = d.seg.ups;
    const int Hs = d.Hi >> ups, Ws = d.Wi >> ups;
    const bool nvalid = (n0 + ch) < d.N;       // N is a multiple of 8 or padded buffer, checked on host
    const bool cvalid = (c0 + ch) < d.seg.C;

    f32x16_t acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;

    uint4 ra, rb;
    auto load_tile = [&](int kt) {
        int m = mbeg + kt * TK + row;
        ra = make_uint4(0, 0, 0, 0);
        rb = make_uint4(0, 0, 0, 0);
        if (m < mend) {
            if (nvalid) ra = *reinterpret_cast<const uint4*>(d.gy + (size_t)m * d.ldg + n0 + ch);
            if (cvalid) {
                int im = m / HoWo;
                int rem = m - im * HoWo;
                int ho = rem / d.Wo;
                int wo = rem - ho * d.Wo;
                int hi = ho * d.stride - d.pad + kh;
                int wi = wo * d.stride - d.pad + kw;
                if (hi >= 0 && wi >= 0 && hi < d.Hi && wi < d.Wi) {
                    size_t pix = ((size_t)im * Hs + (hi >> ups)) * Ws + (wi >> ups);
                    rb = *reinterpret_cast<const uint4*>(d.seg.ptr + pix * d.seg.ld + c0 + ch);
                }
            }
        }
    };
    auto store_tile = [&](int buf) {
        uint16_t* a = sA[buf];
        uint16_t* b = sB[buf];
        const uint32_t wa[4] = {ra.x, ra.y, ra.z, ra.w};
        const uint32_t wb[4] = {rb.x, rb.y, rb.z, rb.w};
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            a[(ch + 2 * j) * LP + row] = (uint16_t)(wa[j] & 0xffffu);
            a[(ch + 2 * j + 1) * LP + row] = (uint16_t)(wa[j] >> 16);
            b[(ch + 2 * j) * LP + row] = (uint16_t)(wb[j] & 0xffffu);
            b[(ch + 2 * j + 1) * LP + row] = (uint16_t)(wb[j] >> 16);
        }
    };

    load_tile(0);
    store_tile(0);
    __syncthreads();
    for (int kt = 0; kt < nkt; ++kt) {
        const int buf = kt & 1;
        const bool more = (kt + 1) < nkt;
        if (more) load_tile(kt + 1);
        const uint16_t* a = sA[buf];
        const uint16_t* b = sB[buf];
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            const int koff = (ks * 2 + (lane >> 5)) * 8;
            uint4 va = *reinterpret_cast<const uint4*>(a + (wm * 32 + (lane & 31)) * LP + koff);
            uint4 vb = *reinterpret_cast<const uint4*>(b + (wn * 32 + (lane & 31)) * LP + koff);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, va),
                                                          __builtin_bit_cast(bf16x8_t, vb), acc, 0, 0, 0);
        }
        if (more) store_tile(buf ^ 1);
        __syncthreads();
    }

    const int c = c0 + wn * 32 + (lane & 31);
    if (c < d.seg.C) {
        float* base = d.dw + (size_t)tap * d.Ctot + d.coff_k + c;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            int n = n0 + wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
            if (n < d.N) atomicAdd(base + (size_t)n * p.Ktot, acc[r]);
        }
    }
}

}  // namespace

extern "C" int yh_conv_wgrad(const yh_wgrad_desc* d, yh_stream stream)
{
    YH_CHECK_ARG(d != nullptr, "yh_conv_wgrad: null desc");
    YH_CHECK_ARG(d->gy && yh_aligned16(d->gy) && d->ldg % 8 == 0, "yh_conv_wgrad: gy null/unaligned");
    YH_CHECK_ARG(d->N > 0 && (d->N % 8 == 0 || d->ldg >= ((d->N + 7) / 8) * 8), "yh_conv_wgrad: N=%d needs ldg padded to 8", d->N);
    YH_CHECK_ARG(d->seg.ptr && yh_aligned16(d->seg.ptr) && d->seg.C % 8 == 0 && d->seg.ld % 8 == 0, "yh_conv_wgrad: segment misaligned");
    YH_CHECK_ARG(d->coff_k % 8 == 0 && d->coff_k + d->seg.C <= d->Ctot, "yh_conv_wgrad: bad channel offset");
    YH_CHECK_ARG(d->stride == 1 || d->stride == 2, "yh_conv_wgrad: bad stride");
    YH_CHECK_ARG((d->Hi + 2 * d->pad - d->KH) / d->stride + 1 == d->Ho && (d->Wi + 2 * d->pad - d->KW) / d->stride + 1 == d->Wo,
                 "yh_conv_wgrad: geometry mismatch");
    YH_CHECK_ARG(d->dw != nullptr && d->splits >= 1, "yh_conv_wgrad: dw null / bad splits");
    if (d->seg.ups) YH_CHECK_ARG(d->Hi % 2 == 0 && d->Wi % 2 == 0, "yh_conv_wgrad: upsampled segment needs even dims");
    long M = (long)d->B * d->Ho * d->Wo;
    YH_CHECK_ARG(M < (1L << 31) - 64, "yh_conv_wgrad: too many pixels");
    WgK k;
    k.d = *d;
    k.M = (int)M;
    k.Ktot = d->KH * d->KW * d->Ctot;
    int rps = (int)((M + d->splits - 1) / d->splits);
    rps = ((rps + TK - 1) / TK) * TK;
    k.rows_per_split = rps;
    int splits = (int)((M + rps - 1) / rps);
    int ntiles = (d->N + TT - 1) / TT;
    k.ctiles = (d->seg.C + TT - 1) / TT;
    dim3 grid(ntiles * k.ctiles, d->KH * d->KW, splits), block(256);
    YH_CHECK_ARG(grid.z <= 65535 && grid.y <= 65535, "yh_conv_wgrad: grid too large");
    hipLaunchKernelGGL(conv_wgrad_kernel, grid, block, 0, (hipStream_t)stream, k);
    YH_CHECK_LAUNCH("yh_conv_wgrad");
    return YH_OK;
}
