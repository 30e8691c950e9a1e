// Implicit-GEMM convolution for gfx950: NHWC bf16 activations, packed bf16 weights
// [Npad][taps*Ctot], fp32 accumulate on v_mfma_f32_32x32x16_bf16.
//
//   out[m][n] = sum_{tap,c} X[src(m,tap)][c] * W[n][tap*Ctot + c]      m = (img,ho,wo)
//
// One kernel serves the forward conv (ConvBnAct / Detect, utils/layer_tools.py:82-94,
// :454-470) and the data gradient (transposed gather, YH_CONV_DGRAD); the input may be
// a virtual channel-concat of two buffers, one of them read through a nearest-2x
// upsample (the neck joins of models/normal/yolov5s.py:101-114 are never materialised).
//
// Tile: BM=128 pixels x BN in {32,64,128} channels x BK=32, 256 threads = 4 waves.
// A/B tiles are register-staged (global_load_dwordx4 -> ds_write_b128) so that padding
// taps can be zero-filled; LDS rows are padded to 80 B which makes the ds_read_b128
// fragment reads conflict-free.  Double-buffered LDS, next tile's global loads are
// issued before the MFMAs of the current one.  The epilogue transposes through LDS
// so every global store is a 16-byte row chunk, and optionally emits per-channel
// sum / sum-of-squares partials (training-mode BatchNorm statistics) per block.
#include "common.h"
#include <stdlib.h>

#ifndef YH_CONV_ABLATE
#define YH_CONV_ABLATE 0      // timing builds only (make ablate ABL=mask): compile parts of the main loop out
#endif

namespace {

constexpr int BM = 128;
constexpr int BK = 32;
constexpr int LDSP = 40;   // bf16 elements per LDS row (32 + 8 pad = 80 bytes)

typedef unsigned int u32x4_nt __attribute__((ext_vector_type(4)));
__device__ __forceinline__ uint4 ld_nt16(const uint16_t* p) {
    u32x4_nt v = __builtin_nontemporal_load(reinterpret_cast<const u32x4_nt*>(p));
    return make_uint4(v.x, v.y, v.z, v.w);
}
__device__ __forceinline__ void st_nt16(uint16_t* p, uint4 v) {
    u32x4_nt w = {v.x, v.y, v.z, v.w};
    __builtin_nontemporal_store(w, reinterpret_cast<u32x4_nt*>(p));
}

struct ConvK {
    yh_conv_desc d;
    int M, Ctot, Ktot, nkt, mtiles;
    int sa, sb, sc, sdshift;
    // stride-2 data gradient: the output pixels are processed in 4 parity classes (blockIdx.z); a class only
    // touches the taps whose parity matches, so no MFMA work is spent on structurally-zero taps
    int cls, Hc, Wc;
    int fast;      // every input segment has a multiple of 32 channels: tap-major incremental loader
    int v2;        // lean buffer-load kernel eligible
    unsigned segbytes[2], wbytes;   // addressable bytes from seg[i].ptr / w (hardware range check zero-fills beyond)
    unsigned long segbytes64[2];    // the same, unclamped: conv_v3_kernel re-bases its input descriptors at every tile
    int pointwise; // 1x1 / stride 1 / no upsample: input pixel == output pixel
    int xgx, xgy;  // conv_v3_kernel launched as ONE row of xgx * xgy workgroups in XCD-major (m-tile slot, n-tile) order (0: 2-D grid)
    int dbg;       // YH_CONV_DBG kernel-selection switches for A/B timing: 16 generic kernel instead of v2, 64 32-channel
                   // k-steps only, 256 no stem kernel (the ablation masks are compile-time: make ablate ABL=mask)
};

// Per-block partial sums leave the kernel as one fp32 slab row per block (deterministic fixed-order reduction by
// yh_bn_finalize / yh_bn_bwd_finalize).
__device__ __forceinline__ void put_stat(const yh_conv_desc& d, size_t blk, int which, int n, float v)
{
    d.stats[(blk * 2 + which) * d.Npad + n] = v;
}
__device__ __forceinline__ void put_bnr(const yh_conv_desc& d, size_t row, int which, int n, float v)
{
    d.bnr_part[(row * 2 + which) * d.N + n] = v;
}

// Stride-2 data gradients run as four parity classes of output pixels (2i+ph, 2j+pw) that touch only their structurally
// non-zero taps: 1, 2, 2 and 4 taps for a 3x3 kernel.  blockIdx.z enumerates (class, slot): a class owns as many slots as it
// has taps (KH*KW slots in all) and its m-tiles are dealt over slots x gridDim.x blocks, so every block of the launch carries
// the same work (one z per class left the one-tap class idle 3/4 of the time).
__device__ __forceinline__ void cls_slot(const yh_conv_desc& d, int z, int& ph, int& pw, int& slot, int& nslots)
{
    int acc = 0;
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        const int cph = c >> 1, cpw = c & 1;
        const int kh0 = (cph + d.pad) & 1, kw0 = (cpw + d.pad) & 1;
        const int n = ((d.KH - kh0 + 1) / 2) * ((d.KW - kw0 + 1) / 2);
        if (z < acc + n || c == 3) { ph = cph; pw = cpw; slot = z - acc; nslots = n; return; }
        acc += n;
    }
}

template <int BN, int WM, int WN, bool FAST, int MINW>
__global__ __launch_bounds__(256, MINW) void conv_igemm_kernel(const ConvK p)
{
    constexpr int TM = BM / (WM * 32);
    constexpr int TN = BN / (WN * 32);
    constexpr int NBL = (BN * 4 + 255) / 256;       // B chunks per thread
    constexpr int CP = BN + 8;                      // sC row pitch (elements)
    constexpr int MAIN_BYTES = (2 * (BM + BN) * LDSP * 2) > (BM * CP * 2) ? (2 * (BM + BN) * LDSP * 2) : (BM * CP * 2);

    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    uint16_t* sA = reinterpret_cast<uint16_t*>(smem);            // [2][BM][LDSP]
    uint16_t* sB = sA + 2 * BM * LDSP;                            // [2][BN][LDSP]
    uint16_t* sC = reinterpret_cast<uint16_t*>(smem);            // [BM][CP] (aliases sA/sB)
    float* sStat = reinterpret_cast<float*>(smem + MAIN_BYTES);  // [WM][2][BN]
    int* sPix = reinterpret_cast<int*>(smem + MAIN_BYTES + WM * 2 * BN * 4);   // [BM] output pixel of each tile row (cls mode)

    const yh_conv_desc& d = p.d;
    const int t = threadIdx.x;
    const int lane = t & 63;
    const int wave = t >> 6;
    const int wm = wave / WN;
    const int wn = wave % WN;
    const int kc = t & 3;
    const int rowA = t >> 2;
    const int n0 = blockIdx.y * BN;
    const int HoWo = d.Ho * d.Wo;
    const int sdmask = (1 << p.sdshift) - 1;
    // parity class of this block (cls mode): output pixels (2i+ph, 2j+pw); taps kh = kh0 + 2a, kw = kw0 + 2b
    int ph = 0, pw = 0, zslot = 0, zslots = 1;
    if (p.cls) cls_slot(d, blockIdx.z, ph, pw, zslot, zslots);
    const int kh0 = (ph + d.pad) & 1, kw0 = (pw + d.pad) & 1;
    const int nkw = p.cls ? (d.KW - kw0 + 1) / 2 : d.KW;
    const int nkh = p.cls ? (d.KH - kh0 + 1) / 2 : d.KH;
    const int Keff = nkh * nkw * p.Ctot;
    const int nkt = p.cls ? (Keff + BK - 1) / BK : p.nkt;
    const int HcWc = p.Hc * p.Wc;

    float run_s = 0.f, run_q = 0.f;

    for (int mt = blockIdx.x + gridDim.x * zslot; mt < p.mtiles; mt += gridDim.x * zslots) {
        const int m0 = mt * BM;
        int hb[2], wb[2], img[2];
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            int m = m0 + rowA + 64 * i;
            if (m < p.M) {
                int im, ho, wo;
                if (p.cls) {
                    im = m / HcWc;
                    int rem = m - im * HcWc;
                    int ii = rem / p.Wc;
                    ho = 2 * ii + ph; wo = 2 * (rem - ii * p.Wc) + pw;
                } else {
                    im = m / HoWo;
                    int rem = m - im * HoWo;
                    ho = rem / d.Wo;
                    wo = rem - ho * d.Wo;
                }
                img[i] = im; hb[i] = ho * p.sa + p.sc; wb[i] = wo * p.sa + p.sc;
                if (p.cls && kc == 0) sPix[rowA + 64 * i] = (im * d.Ho + ho) * d.Wo + wo;
            } else {
                img[i] = 0; hb[i] = -(1 << 28); wb[i] = -(1 << 28);
            }
        }

        f32x16_t acc[TM][TN];
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

        uint4 ra[2], rb[NBL];

        // ---- loader state: k-tiles are walked tap-major.  On the fast path (every segment a multiple of 32
        // channels) a k-tile never straddles a tap or a segment, so the per-row pixel offsets and validity are
        // recomputed only when the tap changes and a k-tile costs a pointer increment.
        int ld_tap = 0, ld_cb = 0;
        int pix0[2] = {0, 0}, pix1[2] = {0, 0};
        bool okr[2] = {false, false};
        int kcol_base = 0;
        const int ncb = p.Ctot >> 5;
        auto tap_setup = [&](int tapl) {
            int kh = tapl / nkw;
            int kw = tapl - kh * nkw;
            if (p.cls) { kh = kh0 + 2 * kh; kw = kw0 + 2 * kw; }
            kcol_base = (kh * d.KW + kw) * p.Ctot;
            const int u0 = d.seg[0].ups, u1 = d.seg[1].ups;
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int hn = hb[i] + kh * p.sb;
                const int wn_ = wb[i] + kw * p.sb;
                bool ok = hn >= 0 && wn_ >= 0 && (((hn | wn_) & sdmask) == 0);
                const int hs = hn >> p.sdshift, ws = wn_ >> p.sdshift;
                ok = ok && hs < d.Hi && ws < d.Wi;
                okr[i] = ok;
                pix0[i] = (img[i] * (d.Hi >> u0) + (hs >> u0)) * (d.Wi >> u0) + (ws >> u0);
                pix1[i] = (img[i] * (d.Hi >> u1) + (hs >> u1)) * (d.Wi >> u1) + (ws >> u1);
            }
        };
        auto load_fast = [&]() {
            if (ld_cb == 0) tap_setup(ld_tap);
            const int c = ld_cb * 32 + kc * 8;
            const bool s1 = d.nseg > 1 && c >= d.seg[0].C;
            const uint16_t* sp = s1 ? d.seg[1].ptr : d.seg[0].ptr;
            const int sld = s1 ? d.seg[1].ld : d.seg[0].ld;
            const int cc = s1 ? c - d.seg[0].C : c;
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                uint4 v = make_uint4(0, 0, 0, 0);
                if (okr[i] && !(YH_CONV_ABLATE & 1)) v = *reinterpret_cast<const uint4*>(sp + (size_t)(s1 ? pix1[i] : pix0[i]) * sld + cc);
                ra[i] = v;
            }
            const uint16_t* wp = d.w + (size_t)n0 * p.Ktot + kcol_base + c;
#pragma unroll
            for (int i = 0; i < NBL; ++i) {
                const int id = t + i * 256;
                uint4 v = make_uint4(0, 0, 0, 0);
                if (id < BN * 4 && !(YH_CONV_ABLATE & 8)) v = *reinterpret_cast<const uint4*>(wp + (size_t)(id >> 2) * p.Ktot);
                rb[i] = v;
            }
            if (++ld_cb == ncb) { ld_cb = 0; ++ld_tap; }
        };
        auto load_generic = [&](int kt) {
            const int k = kt * BK + kc * 8;
            const bool kvalid = k < Keff;
            int tap = 0, c = 0;
            if (kvalid) { tap = k / p.Ctot; c = k - tap * p.Ctot; }
            int kh = tap / nkw;
            int kw = tap - kh * nkw;
            if (p.cls) { kh = kh0 + 2 * kh; kw = kw0 + 2 * kw; }
            const int kcol = (kh * d.KW + kw) * p.Ctot + c;      // column in the packed weight rows
            const int sidx = (d.nseg > 1 && c >= d.seg[0].C) ? 1 : 0;
            const yh_seg& sg = d.seg[sidx];
            const int cc = c - (sidx ? d.seg[0].C : 0);
            const int ups = sg.ups;
            const int Hs = d.Hi >> ups, Ws = d.Wi >> ups;
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                int hn = hb[i] + kh * p.sb;
                int wn_ = wb[i] + kw * p.sb;
                bool ok = kvalid && hn >= 0 && wn_ >= 0 && (((hn | wn_) & sdmask) == 0);
                int hs = hn >> p.sdshift, ws = wn_ >> p.sdshift;
                ok = ok && hs < d.Hi && ws < d.Wi;
                hs >>= ups; ws >>= ups;
                uint4 v = make_uint4(0, 0, 0, 0);
                if (ok) {
                    size_t pix = ((size_t)img[i] * Hs + hs) * Ws + ws;
                    v = *reinterpret_cast<const uint4*>(sg.ptr + pix * sg.ld + cc);
                }
                ra[i] = v;
            }
#pragma unroll
            for (int i = 0; i < NBL; ++i) {
                int id = t + i * 256;
                uint4 v = make_uint4(0, 0, 0, 0);
                if (id < BN * 4 && kvalid) {
                    int n = id >> 2;
                    v = *reinterpret_cast<const uint4*>(d.w + (size_t)(n0 + n) * p.Ktot + kcol);
                }
                rb[i] = v;
            }
        };
        auto load_tile = [&](int kt) {
            if constexpr (FAST) load_fast(); else load_generic(kt);
        };
        auto store_tile = [&](int buf) {
            uint16_t* a = sA + buf * BM * LDSP;
            uint16_t* b = sB + buf * BN * LDSP;
#pragma unroll
            for (int i = 0; i < 2; ++i)
                *reinterpret_cast<uint4*>(a + (rowA + 64 * i) * LDSP + kc * 8) = ra[i];
#pragma unroll
            for (int i = 0; i < NBL; ++i) {
                int id = t + i * 256;
                if (id < BN * 4) *reinterpret_cast<uint4*>(b + (id >> 2) * LDSP + kc * 8) = rb[i];
            }
        };

        load_tile(0);
        store_tile(0);
        __syncthreads();

        for (int kt = 0; kt < nkt; ++kt) {
            const int buf = kt & 1;
            const bool more = (kt + 1) < nkt;
            if (more) load_tile(kt + 1);
            const uint16_t* a = sA + buf * BM * LDSP;
            const uint16_t* b = sB + buf * BN * LDSP;
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                bf16x8_t af[TM], bfr[TN];
                const int koff = (ks * 2 + (lane >> 5)) * 8;
#pragma unroll
                for (int i = 0; i < TM; ++i) {
                    int r = wm * (TM * 32) + i * 32 + (lane & 31);
                    uint4 v = *reinterpret_cast<const uint4*>(a + r * LDSP + koff);
                    af[i] = __builtin_bit_cast(bf16x8_t, v);
                }
#pragma unroll
                for (int j = 0; j < TN; ++j) {
                    int r = wn * (TN * 32) + j * 32 + (lane & 31);
                    uint4 v = *reinterpret_cast<const uint4*>(b + r * LDSP + koff);
                    bfr[j] = __builtin_bit_cast(bf16x8_t, v);
                }
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j)
                        if (!(YH_CONV_ABLATE & 4)) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i], bfr[j], acc[i][j], 0, 0, 0);
            }
            if (more) store_tile(buf ^ 1);
            __syncthreads();
        }

        // ---- epilogue: registers -> LDS (bf16, transposed to row-major) ----
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const int c = wn * (TN * 32) + j * 32 + (lane & 31);
            const int n = n0 + c;
            const bool nv = n < d.N;
            const float bs = (d.bias && nv) ? d.bias[n] : 0.f;
            const float scl = (d.scale && nv) ? d.scale[n] : 1.f;
            const float sft = (d.shift && nv) ? d.shift[n] : 0.f;
            float s = 0.f, q = 0.f;
#pragma unroll
            for (int i = 0; i < TM; ++i) {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    int row = wm * (TM * 32) + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                    float v = acc[i][j][r] + bs;
                    v = v * scl + sft;
                    if (d.act == YH_ACT_SILU) v = silu_fast(v);
                    uint16_t hb16 = f2bf(v);
                    if (!(YH_CONV_ABLATE & 2)) sC[row * CP + c] = hb16;
                    float vr = (m0 + row < p.M) ? bf2f(hb16) : 0.f;   // rows past M carry only the bias
                    s += vr; q += vr * vr;
                }
            }
            if (d.stats) {
                s += __shfl_xor(s, 32, 64);
                q += __shfl_xor(q, 32, 64);
                if (lane < 32) {
                    sStat[(wm * 2 + 0) * BN + c] = s;
                    sStat[(wm * 2 + 1) * BN + c] = q;
                }
            }
        }
        __syncthreads();

        constexpr int CPR = BN / 8;                 // chunks per row
        constexpr int NCH = BM * CPR / 256;         // chunks per thread
#pragma unroll
        for (int i = 0; i < NCH; ++i) {
            int id = t + i * 256;
            int row = id / CPR;
            int cch = id - row * CPR;
            int m = m0 + row;
            int n = n0 + cch * 8;
            if (m < p.M && n < d.N) {
                uint4 v = *reinterpret_cast<const uint4*>(sC + row * CP + cch * 8);
                uint16_t* dst;
                bool first = n < d.nsplit;
                const size_t orow = p.cls ? (size_t)sPix[row] : (size_t)m;
                if (first) dst = d.out0 + orow * d.ld0 + n;
                else       dst = d.out1 + orow * d.ld1 + (n - d.nsplit);
                const bool addres = (d.res != nullptr) && first;
                if (addres || d.accumulate) {
                    float f[8];
                    unpack8(v, f);
                    if (addres) {
                        uint4 rv = *reinterpret_cast<const uint4*>(d.res + orow * d.ldr + n);
                        float g[8]; unpack8(rv, g);
#pragma unroll
                        for (int e = 0; e < 8; ++e) f[e] += g[e];
                    }
                    if (d.accumulate) {
                        uint4 ov = *reinterpret_cast<const uint4*>(dst);
                        float g[8]; unpack8(ov, g);
#pragma unroll
                        for (int e = 0; e < 8; ++e) f[e] += g[e];
                    }
                    v = pack8(f);
                }
                *reinterpret_cast<uint4*>(dst) = v;
            }
        }
        if (d.stats && t < BN) {
#pragma unroll
            for (int w = 0; w < WM; ++w) {
                run_s += sStat[(w * 2 + 0) * BN + t];
                run_q += sStat[(w * 2 + 1) * BN + t];
            }
        }
        __syncthreads();
    }

    if (d.stats && t < BN) {
        put_stat(d, blockIdx.x, 0, n0 + t, run_s);
        put_stat(d, blockIdx.x, 1, n0 + t, run_q);
    }
}


// ------------------------------------------------------------------------------------------------
// v2: the same tiling, written for a minimal instruction count per k-tile.  Operands are fetched with raw
// buffer loads: the per-lane byte offset of a tile row is computed once per TAP (not per k-tile), padding taps
// and rows past M simply carry an out-of-range offset (the hardware range check returns zeros, no branches),
// and the walk along the channels / along K is a SCALAR offset.  Eligible when every segment has a multiple
// of 32 channels and every operand is smaller than 2 GiB (host check).
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4_t;

#ifdef YH_CONV_STAMPS
// timing build only (make stamps): per-phase s_memtime totals of block 0's four waves
__device__ long long g_stamps[32];
#endif

template <int BN, int WM, int WN, int MINW, int EPI, int BKT>     // block = WM x WN waves   // BKT: channels per k-step (32 | 64); EPI 0: plain store, 1: + BatchNorm partial sums, 2: generic epilogue
__global__ __launch_bounds__(WM * WN * 64, MINW) void conv_v2_kernel(const ConvK p)
{
    constexpr int NT = WM * WN * 64;                // threads per block
    constexpr int TM = BM / (WM * 32);
    constexpr int TN = BN / (WN * 32);
    constexpr int LDSPX = BKT + 8;                  // LDS row pitch in elements: 80 B / 144 B rows, conflict-free ds_read_b128
    constexpr int CHR = BKT / 8;                    // 16-byte chunks per tile row
    constexpr int RPP = NT / CHR;                   // tile rows covered by one pass of the block's threads
    constexpr int NA = BM / RPP;                    // A chunks per thread
    constexpr int NBL = (BN * CHR + NT - 1) / NT;   // B chunks per thread
    constexpr int CP = BN + 8;
    constexpr int MAIN_BYTES = (2 * (BM + BN) * LDSPX * 2) > (BM * CP * 2) ? (2 * (BM + BN) * LDSPX * 2) : (BM * CP * 2);
    constexpr unsigned OOB = 0x80000000u;

    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    uint16_t* sA = reinterpret_cast<uint16_t*>(smem);
    uint16_t* sB = sA + 2 * BM * LDSPX;
    uint16_t* sC = reinterpret_cast<uint16_t*>(smem);
    float* sStat = reinterpret_cast<float*>(smem + MAIN_BYTES);
    int* sPix = reinterpret_cast<int*>(smem + MAIN_BYTES + WM * 2 * BN * 4);

    const yh_conv_desc& d = p.d;
    const int t = threadIdx.x;
    const int lane = t & 63;
    const int wave = t >> 6;
    const int wm = wave / WN;
    const int wn = wave % WN;
    const int kc = t % CHR;
    const int rowA = t / CHR;
    const int n0 = blockIdx.y * BN;
    const int HoWo = d.Ho * d.Wo;
    const int sdmask = (1 << p.sdshift) - 1;
    // channel blocks are walked per segment: a segment's ragged last block (channel count not a multiple of the k-step) is masked
    // to zero on the activation side, and the next segment starts a block of its own at its first channel (80 + 80: 3 + 3 blocks)
    const int C0 = d.seg[0].C;
    const int ncb0 = d.nseg > 1 ? (C0 + BKT - 1) / BKT : 0;
    const int ncb = d.nseg > 1 ? ncb0 + (p.Ctot - C0 + BKT - 1) / BKT : (p.Ctot + BKT - 1) / BKT;
    const int HcWc = p.Hc * p.Wc;

    const __amdgpu_buffer_rsrc_t rs0 = __builtin_amdgcn_make_buffer_rsrc((void*)d.seg[0].ptr, 0, p.segbytes[0], 0x00020000);
    const __amdgpu_buffer_rsrc_t rs1 = __builtin_amdgcn_make_buffer_rsrc((void*)(d.nseg > 1 ? d.seg[1].ptr : d.seg[0].ptr), 0,
                                                                          d.nseg > 1 ? p.segbytes[1] : p.segbytes[0], 0x00020000);
    const __amdgpu_buffer_rsrc_t rsw = __builtin_amdgcn_make_buffer_rsrc((void*)d.w, 0, p.wbytes, 0x00020000);

    // per-thread constants: weight row offsets, LDS addresses
    unsigned voffB[NBL];
#pragma unroll
    for (int j = 0; j < NBL; ++j) {
        const int id = t + j * NT;
        voffB[j] = id < BN * CHR ? (unsigned)(((n0 + id / CHR) * p.Ktot + kc * 8) * 2) : OOB;
    }
    const int ldsA0 = rowA * LDSPX + kc * 8;         // + RPP*LDSPX per further row of this thread
    const int ldsB0 = rowA * LDSPX + kc * 8;         // row (t + j*NT)/CHR == rowA + j*RPP

    float run_s = 0.f, run_q = 0.f;
    // EPI 3 (data gradient that is the ONLY writer of a ConvBnAct output's gradient): the BatchNorm+SiLU backward
    // reduction of that producer is taken here, from the tile that is being stored anyway: dz = g * silu'(bn(z)),
    // partial sums of dz and dz*z per channel.  A thread always handles the same 8-channel chunk (NT % (BN/8) == 0).
    float bs_[8], bq_[8];
    if (EPI == 3) {
#pragma unroll
        for (int e = 0; e < 8; ++e) { bs_[e] = 0.f; bq_[e] = 0.f; }
        for (int i = t; i < 2 * BN; i += NT) {          // scale | shift of the block's channels -> LDS (sStat is free in this mode)
            const int which = i / BN, c = i - which * BN;
            sStat[i] = (n0 + c < d.N) ? d.bnr_ws[(size_t)which * d.bnr_C + n0 + c] : 0.f;
        }
        __syncthreads();
    }
#ifdef YH_CONV_STAMPS
    long long st_load = 0, st_mma = 0, st_store = 0, st_bar = 0, st_pro = 0, st_epi = 0, st_n = 0;
    const long long st_begin = __builtin_amdgcn_s_memtime();
#define STAMP(var) do { const long long now_ = __builtin_amdgcn_s_memtime(); var += now_ - st_t; st_t = now_; } while (0)
#else
#define STAMP(var) do { } while (0)
#endif

    // Stride-2 data gradient: a block takes the four parity classes of a pixel region one after the other (1, 2, 2 and 4 taps for
    // a 3x3 kernel), so the gy rows the classes share come out of its XCD's L2 after the first fetch and the two 64-byte halves
    // of the 128-byte lines that neighbouring classes write (a class owns every other pixel) meet in L2 before they are evicted.
    // Layers with fewer regions than CUs keep the classes in blockIdx.z instead ((class, slot) pairs, see cls_slot): more blocks.
    const bool zcls = p.cls && gridDim.z > 1;
    int zph = 0, zpw = 0, zslot = 0, zslots = 1;
    if (zcls) cls_slot(d, blockIdx.z, zph, zpw, zslot, zslots);
    const int csh = (p.cls && !zcls) ? 2 : 0;
    // With a multiple of 32 blocks the four classes of a region go to four blocks of ONE XCD at the same time instead (blocks b,
    // b+8, b+16, b+24 form a group; the class rotates per step so that every block carries the same work): the shared gy rows are
    // still in that XCD's L2 when the other three classes ask for them (a lone block comes back to them after ~8 MB of other
    // blocks' traffic: PMC showed gy fetched 4.6 times).  +3 / +7 / +21 % on the YOLOv5s stage-1 / 2 / 3 layers.
    const bool grp = p.cls && !zcls && (gridDim.x % 32 == 0);
    const int g_q = (blockIdx.x >> 3) & 3;
    const int g_idx = (blockIdx.x & 7) + 8 * (blockIdx.x >> 5);
    const int g_n = gridDim.x >> 2;
    for (int it = 0;; ++it) {
        const int mt = grp ? g_idx + g_n * it : blockIdx.x + gridDim.x * (zcls ? zslot + zslots * it : (it >> csh));
        if (mt >= p.mtiles) break;
#ifdef YH_CONV_STAMPS
        long long st_t = __builtin_amdgcn_s_memtime();
#endif
        const int gc = (g_q + it) & 3;
        const int ph = zcls ? zph : (p.cls ? ((grp ? gc >> 1 : it >> 1) & 1) : 0), pw = zcls ? zpw : (p.cls ? ((grp ? gc : it) & 1) : 0);
        const int kh0 = (ph + d.pad) & 1, kw0 = (pw + d.pad) & 1;
        const int nkw = p.cls ? (d.KW - kw0 + 1) / 2 : d.KW;
        const int nkh = p.cls ? (d.KH - kh0 + 1) / 2 : d.KH;
        const int nkt = nkh * nkw * ncb;
        const int m0 = mt * BM;
        int hb[NA], wb[NA], img[NA];
        unsigned voff0[NA], voff1[NA];
#pragma unroll
        for (int i = 0; i < NA; ++i) {
            const int m = m0 + rowA + RPP * i;
            voff0[i] = OOB; voff1[i] = OOB;
            img[i] = 0; hb[i] = -(1 << 28); wb[i] = -(1 << 28);
            if (m < p.M) {
                if (p.pointwise) {
                    voff0[i] = (unsigned)m * (unsigned)(d.seg[0].ld * 2) + kc * 16;
                    voff1[i] = (unsigned)m * (unsigned)(d.seg[1].ld * 2) + kc * 16;
                } else {
                    int im, ho, wo;
                    if (p.cls) {
                        im = m / HcWc;
                        const int rem = m - im * HcWc;
                        const int ii = rem / p.Wc;
                        ho = 2 * ii + ph; wo = 2 * (rem - ii * p.Wc) + pw;
                    } else {
                        im = m / HoWo;
                        const int rem = m - im * HoWo;
                        ho = rem / d.Wo;
                        wo = rem - ho * d.Wo;
                    }
                    img[i] = im; hb[i] = ho * p.sa + p.sc; wb[i] = wo * p.sa + p.sc;
                    if (p.cls && kc == 0) sPix[rowA + RPP * i] = (im * d.Ho + ho) * d.Wo + wo;
                }
            }
        }

        f32x16_t acc[TM][TN];
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

        u32x4_t ra[NA], rb[NBL];
        int ld_tap = 0, ld_cb = 0, kcol_base = 0;

        auto tap_setup = [&](int tapl) {
            int kh = tapl / nkw;
            int kw = tapl - kh * nkw;
            if (p.cls) { kh = kh0 + 2 * kh; kw = kw0 + 2 * kw; }
            kcol_base = (kh * d.KW + kw) * p.Ctot;
            if (p.pointwise) return;
            const int u0 = d.seg[0].ups, u1 = d.seg[1].ups;
#pragma unroll
            for (int i = 0; i < NA; ++i) {
                const int hn = hb[i] + kh * p.sb;
                const int wn_ = wb[i] + kw * p.sb;
                bool ok = hn >= 0 && wn_ >= 0 && (((hn | wn_) & sdmask) == 0);
                const int hs = hn >> p.sdshift, ws = wn_ >> p.sdshift;
                ok = ok && hs < d.Hi && ws < d.Wi;
                const unsigned pix0 = (unsigned)((img[i] * (d.Hi >> u0) + (hs >> u0)) * (d.Wi >> u0) + (ws >> u0));
                const unsigned pix1 = (unsigned)((img[i] * (d.Hi >> u1) + (hs >> u1)) * (d.Wi >> u1) + (ws >> u1));
                voff0[i] = ok ? pix0 * (unsigned)(d.seg[0].ld * 2) + kc * 16 : OOB;
                voff1[i] = ok ? pix1 * (unsigned)(d.seg[1].ld * 2) + kc * 16 : OOB;
            }
        };
        auto load_tile = [&]() {
            if (ld_cb == 0) tap_setup(ld_tap);
            const bool s1 = d.nseg > 1 && ld_cb >= ncb0;     // wave-uniform
            const int c = s1 ? C0 + (ld_cb - ncb0) * BKT : ld_cb * BKT;      // channel of the concatenated input the block starts at
            // channels of this thread's chunk that lie past the segment (last block of a channel count that is not a
            // multiple of the k-step) read as zero; the weight columns they meet are finite, so they add nothing
            const bool cok = c + kc * 8 < (s1 || d.nseg == 1 ? p.Ctot : C0);
#if YH_CONV_ABLATE
            if (YH_CONV_ABLATE & 1) {
#pragma unroll
                for (int i = 0; i < NA; ++i) ra[i] = u32x4_t{0, 0, 0, 0};
            } else
#endif
            if (s1) {
                const int so = (c - C0) * 2;
#pragma unroll
                for (int i = 0; i < NA; ++i) ra[i] = __builtin_amdgcn_raw_buffer_load_b128(rs1, cok ? voff1[i] : OOB, so, 0);
            } else {
                const int so = c * 2;
#pragma unroll
                for (int i = 0; i < NA; ++i) ra[i] = __builtin_amdgcn_raw_buffer_load_b128(rs0, cok ? voff0[i] : OOB, so, 0);
            }
            const int sw = (kcol_base + c) * 2;
#if YH_CONV_ABLATE
            if (YH_CONV_ABLATE & 8) {
#pragma unroll
                for (int j = 0; j < NBL; ++j) rb[j] = u32x4_t{0, 0, 0, 0};
            } else
#endif
#pragma unroll
            for (int j = 0; j < NBL; ++j) rb[j] = __builtin_amdgcn_raw_buffer_load_b128(rsw, voffB[j], sw, 0);
            if (++ld_cb == ncb) { ld_cb = 0; ++ld_tap; }
        };
        auto store_tile = [&](int buf) {
            uint16_t* a = sA + buf * BM * LDSPX + ldsA0;
            uint16_t* b = sB + buf * BN * LDSPX + ldsB0;
#pragma unroll
            for (int i = 0; i < NA; ++i) *reinterpret_cast<u32x4_t*>(a + i * RPP * LDSPX) = ra[i];
#pragma unroll
            for (int j = 0; j < NBL; ++j)
                if (t + j * NT < BN * CHR) *reinterpret_cast<u32x4_t*>(b + j * RPP * LDSPX) = rb[j];
        };

        load_tile();
        store_tile(0);
        __syncthreads();
        STAMP(st_pro);
        for (int kt = 0; kt < nkt; ++kt) {
            const int buf = kt & 1;
            const bool more = (kt + 1) < nkt;
            if (more) load_tile();
            STAMP(st_load);
            const uint16_t* a = sA + buf * BM * LDSPX;
            const uint16_t* b = sB + buf * BN * LDSPX;
#pragma unroll
            for (int ks = 0; ks < BKT / 16; ++ks) {
                bf16x8_t af[TM], bfr[TN];
                const int koff = (ks * 2 + (lane >> 5)) * 8;
#if YH_CONV_ABLATE
                if (YH_CONV_ABLATE & 64) {
#pragma unroll
                    for (int i = 0; i < TM; ++i) af[i] = __builtin_bit_cast(bf16x8_t, ra[0]);
#pragma unroll
                    for (int j = 0; j < TN; ++j) bfr[j] = __builtin_bit_cast(bf16x8_t, rb[0]);
                } else {
#endif
#pragma unroll
                for (int i = 0; i < TM; ++i)
                    af[i] = __builtin_bit_cast(bf16x8_t, *reinterpret_cast<const uint4*>(a + (wm * (TM * 32) + i * 32 + (lane & 31)) * LDSPX + koff));
#pragma unroll
                for (int j = 0; j < TN; ++j)
                    bfr[j] = __builtin_bit_cast(bf16x8_t, *reinterpret_cast<const uint4*>(b + (wn * (TN * 32) + j * 32 + (lane & 31)) * LDSPX + koff));
#if YH_CONV_ABLATE
                }
#endif
#if YH_CONV_ABLATE
                if (YH_CONV_ABLATE & 4) {
#pragma unroll
                    for (int i = 0; i < TM; ++i)
#pragma unroll
                        for (int j = 0; j < TN; ++j) acc[i][j][0] += (float)af[i][0] * (float)bfr[j][0];
                } else
#endif
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i], bfr[j], acc[i][j], 0, 0, 0);
            }
            STAMP(st_mma);
#if YH_CONV_ABLATE
            if (more && !(YH_CONV_ABLATE & 128)) store_tile(buf ^ 1);
            STAMP(st_store);
            if (!(YH_CONV_ABLATE & 256)) __syncthreads();
#else
            if (more) store_tile(buf ^ 1);
            STAMP(st_store);
            __syncthreads();
#endif
            STAMP(st_bar);
#ifdef YH_CONV_STAMPS
            ++st_n;
#endif
        }

        // ---- epilogue
#if YH_CONV_ABLATE
        if (YH_CONV_ABLATE & 2) {
            float sink = 0.f;
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j)
#pragma unroll
                    for (int r = 0; r < 16; ++r) sink += acc[i][j][r];
            if (sink == 12345.678f) d.out0[0] = 1;
            continue;
        }
#endif
        // EPI 3: the producer's z chunks this thread will need in the read-out are requested now, so that their latency
        // hides behind the accumulator -> LDS transposition and its barrier
        constexpr int CPRz = BN / 8;
        constexpr int NCHz = BM * CPRz / NT;
        uint4 zpre[EPI == 3 ? NCHz : 1];
        if (EPI == 3) {
#pragma unroll
            for (int i = 0; i < NCHz; ++i) {
                const int id = t + i * NT;
                const int row = id / CPRz;
                const int m = m0 + row;
                const int n = n0 + (id - row * CPRz) * 8;
                zpre[i] = make_uint4(0, 0, 0, 0);
                if (m < p.M && n < d.N) {
                    const size_t orow = p.cls ? (size_t)sPix[row] : (size_t)m;
                    zpre[i] = ld_nt16(d.bnr_z + orow * d.bnr_ldz + n);          // last reader of z
                }
            }
        }
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const int c = wn * (TN * 32) + j * 32 + (lane & 31);
            float bs = 0.f, scl = 1.f, sft = 0.f;
            if (EPI == 2) {
                const int n = n0 + c;
                const bool nv = n < d.N;
                bs = (d.bias && nv) ? d.bias[n] : 0.f;
                scl = (d.scale && nv) ? d.scale[n] : 1.f;
                sft = (d.shift && nv) ? d.shift[n] : 0.f;
            }
            float s = 0.f, q = 0.f;
#pragma unroll
            for (int i = 0; i < TM; ++i) {
                uint16_t* dst = sC + (wm * (TM * 32) + i * 32 + 4 * (lane >> 5)) * CP + c;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    float v = acc[i][j][r];
                    if (EPI == 2) {
                        v = (v + bs) * scl + sft;
                        if (d.act == YH_ACT_SILU) v = silu_fast(v);
                    }
                    dst[((r & 3) + 8 * (r >> 2)) * CP] = f2bf(v);
                    if (EPI == 1) { s += v; q += v * v; }      // rows past M and padded channels are exact zeros
                }
            }
            if (EPI == 1) {
                s += __shfl_xor(s, 32, 64);
                q += __shfl_xor(q, 32, 64);
                if (lane < 32) {
                    sStat[(wm * 2 + 0) * BN + c] = s;
                    sStat[(wm * 2 + 1) * BN + c] = q;
                }
            }
        }
        __syncthreads();

        constexpr int CPR = BN / 8;
        constexpr int NCH = BM * CPR / NT;
#pragma unroll
        for (int i = 0; i < NCH; ++i) {
            const int id = t + i * NT;
            const int row = id / CPR;
            const int cch = id - row * CPR;
            const int m = m0 + row;
            const int n = n0 + cch * 8;
            if (m < p.M && n < d.N) {
                uint4 v = *reinterpret_cast<const uint4*>(sC + row * CP + cch * 8);
                const size_t orow = p.cls ? (size_t)sPix[row] : (size_t)m;
                if (EPI == 2) {
                    uint16_t* dst;
                    const bool first = n < d.nsplit;
                    if (first) dst = d.out0 + orow * d.ld0 + n;
                    else       dst = d.out1 + orow * d.ld1 + (n - d.nsplit);
                    const bool addres = (d.res != nullptr) && first;
                    if (addres || d.accumulate) {
                        float f[8];
                        unpack8(v, f);
                        if (addres) {
                            uint4 rv = *reinterpret_cast<const uint4*>(d.res + orow * d.ldr + n);
                            float g[8]; unpack8(rv, g);
#pragma unroll
                            for (int e = 0; e < 8; ++e) f[e] += g[e];
                        }
                        if (d.accumulate) {
                            uint4 ov = *reinterpret_cast<const uint4*>(dst);
                            float g[8]; unpack8(ov, g);
#pragma unroll
                            for (int e = 0; e < 8; ++e) f[e] += g[e];
                        }
                        v = pack8(f);
                    }
                    *reinterpret_cast<uint4*>(dst) = v;
                } else {
                    if (EPI == 3 && d.accumulate) {
                        // last writer of a gradient with several contributions: add the earlier ones (bf16, as the generic
                        // epilogue does) and take the BatchNorm-backward sums over the rounded total
                        const uint4 ov = *reinterpret_cast<const uint4*>(d.out0 + orow * d.ld0 + n);
                        float f[8], g0[8];
                        unpack8(v, f);
                        unpack8(ov, g0);
#pragma unroll
                        for (int e = 0; e < 8; ++e) f[e] += g0[e];
                        v = pack8(f);
                    }
                    // EPI 3 streams its output (non-temporal): the gradient is read next by a whole-tensor pass, and the stride-2
                    // classes write 64-byte halves of lines — kept in L2 they cost a line fill each and push out the gy rows the
                    // classes share (PMC on the YOLOv5s stage-1 layer: 1.99 -> 1.39 GB fetched; +1.5 % on the train step)
                    if (EPI == 3) st_nt16(d.out0 + orow * d.ld0 + n, v);
                    else *reinterpret_cast<uint4*>(d.out0 + orow * d.ld0 + n) = v;
                    if (EPI == 3) {
                        const uint4 zv = zpre[i];
                        float g[8], z[8];
                        unpack8(v, g);
                        unpack8(zv, z);
                        const float4 s0 = *reinterpret_cast<const float4*>(sStat + cch * 8);
                        const float4 s1 = *reinterpret_cast<const float4*>(sStat + cch * 8 + 4);
                        const float4 h0 = *reinterpret_cast<const float4*>(sStat + BN + cch * 8);
                        const float4 h1 = *reinterpret_cast<const float4*>(sStat + BN + cch * 8 + 4);
                        const float sc[8] = {s0.x, s0.y, s0.z, s0.w, s1.x, s1.y, s1.z, s1.w};
                        const float sh[8] = {h0.x, h0.y, h0.z, h0.w, h1.x, h1.y, h1.z, h1.w};
#pragma unroll
                        for (int e = 0; e < 8; ++e) {
                            const float a = z[e] * sc[e] + sh[e];
                            const float sg = sigmoid_fast(a);
                            const float dz = g[e] * (sg * (1.f + a * (1.f - sg)));
                            bs_[e] += dz; bq_[e] += dz * z[e];
                        }
                    }
                }
            }
        }
        if (EPI == 1 && t < BN) {
#pragma unroll
            for (int w = 0; w < WM; ++w) {
                run_s += sStat[(w * 2 + 0) * BN + t];
                run_q += sStat[(w * 2 + 1) * BN + t];
            }
        }
        __syncthreads();
        STAMP(st_epi);
    }
#ifdef YH_CONV_STAMPS
    if (blockIdx.x == 0 && blockIdx.y == 0 && blockIdx.z == 0 && lane == 0) {
        long long* o = g_stamps + wave * 8;
        o[0] = st_load; o[1] = st_mma; o[2] = st_store; o[3] = st_bar; o[4] = st_pro; o[5] = st_epi; o[6] = st_n;
        o[7] = __builtin_amdgcn_s_memtime() - st_begin;
    }
#endif
#undef STAMP

    if (EPI == 1 && t < BN) {
        put_stat(d, blockIdx.x, 0, n0 + t, run_s);
        put_stat(d, blockIdx.x, 1, n0 + t, run_q);
    }
    if (EPI == 3) {
        constexpr int CPR2 = BN / 8;
        float* sRed = reinterpret_cast<float*>(smem);          // [NT][16], aliases the (now idle) tile buffers
        __syncthreads();
#pragma unroll
        for (int e = 0; e < 8; ++e) { sRed[t * 16 + e] = bs_[e]; sRed[t * 16 + 8 + e] = bq_[e]; }
        __syncthreads();
        const size_t rowi = (size_t)blockIdx.z * gridDim.x + blockIdx.x;
        for (int i = t; i < 2 * BN; i += NT) {
            const int which = i / BN, c = i - which * BN;
            float v = 0.f;
            for (int j = c / 8; j < NT; j += CPR2) v += sRed[j * 16 + which * 8 + (c & 7)];
            if (n0 + c < d.N) put_bnr(d, rowi, which, n0 + c, v);
        }
    }
}

// ------------------------------------------------------------------------------------------------
// v3: the main loop restructured around LDS-DMA.  A/B tiles go global -> LDS directly (buffer_load ... lds, 1 KiB per
// wave instruction, no VGPR staging and no ds_write phase) into a ring of STG stages; a wave waits only for ITS OWN
// transfers of the stage it is about to read (counted s_waitcnt vmcnt(N): the younger stages stay in flight across the
// barrier), then ONE raw s_barrier per k-step both publishes that stage and frees the stage consumed one step earlier,
// which is refilled at once.  Wave tile 64 x 64 (four MFMAs per four fragment reads: half the LDS read traffic per
// MFMA of the 32 x 64 wave tile of v2), block tile BMT x BN = 256 x 128 (8 waves) or 128 x 128 / 128 x 64 (4 waves).
// LDS rows are unpadded (an LDS-DMA instruction writes 1 KiB contiguously: lane l -> base + 16 l); bank conflicts are
// avoided by an XOR swizzle of the 16-byte chunks applied on the SOURCE side (the lane that fills LDS chunk position q
// of row r fetches global chunk q ^ f(r)) and again on the fragment reads.  Padding taps / rows past M carry an
// out-of-range offset: the range check of the buffer descriptor makes the DMA write zeros.
// Eligible when every segment has a multiple of BKT channels (host check); everything else runs on v2.
template <int CHR> __device__ __forceinline__ int swz_f(int row) { return CHR == 8 ? ((row >> 1) & 7) : ((row >> 2) & 3); }

// TL (tail, BKT 64 only): the channel count is a multiple of 16 but not of 64 (YOLOv5m / v5x widths: 96, 80, 160, 320 + 160 ...):
// the last channel block of every tap runs (C % 64) / 16 of its four 16-channel sub-steps.  Its rows are still fetched whole
// (128 bytes: the bytes behind the last channel are the next pixel's / the next tap's, zeros past the end of the buffer)
// and never read as fragments.
template <int BMT, int BN, int WM, int WN, int BKT, int STG, int EPI, bool TL = false>
__global__ __launch_bounds__(WM * WN * 64, 2) void conv_v3_kernel(const ConvK p)
{
    static_assert(!TL || BKT == 64, "tail blocks: 64-channel k-steps only");
    constexpr int NWV = WM * WN;
    constexpr int NT = NWV * 64;
    constexpr int TM = BMT / (WM * 32);
    constexpr int TN = BN / (WN * 32);
    constexpr int ROWB = BKT * 2;                   // bytes per tile row
    constexpr int CHR = BKT / 8;                    // 16-byte chunks per tile row
    constexpr int RPI = 1024 / ROWB;                // tile rows one LDS-DMA wave instruction fills
    constexpr int NA = BMT / (RPI * NWV);           // A instructions per wave and stage
    constexpr int NB = BN / (RPI * NWV);            // B instructions per wave and stage
    constexpr int LPS = NA + NB;                    // LDS-DMA instructions per wave and stage
    constexpr int STAGE_BYTES = (BMT + BN) * ROWB;
    constexpr int CP = BN + 8;
    constexpr int RING_BYTES = STG * STAGE_BYTES;
    constexpr int MAIN_BYTES = RING_BYTES > (BMT * CP * 2) ? RING_BYTES : (BMT * CP * 2);
    constexpr unsigned OOB = 0x80000000u;
    static_assert(NA >= 1 && NB >= 1 && BMT % (RPI * NWV) == 0 && BN % (RPI * NWV) == 0, "tile / wave count mismatch");
    static_assert(STG >= 2 && STG <= 4, "2..4 stages");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    uint16_t* sC = reinterpret_cast<uint16_t*>(smem);
    float* sStat = reinterpret_cast<float*>(smem + MAIN_BYTES);
    int* sPix = reinterpret_cast<int*>(smem + MAIN_BYTES + WM * 2 * BN * 4);

    const yh_conv_desc& d = p.d;
    const int t = threadIdx.x;
    const int lane = t & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int wm = wave / WN;
    const int wn = wave % WN;
    // the gy output-channel tiles of an m-tile read the same input rows: as grid rows they are gx workgroups apart (other XCDs, other
    // times: every one fetches the rows again — PMC on YOLOv5x stage-2 conv, 3x3 / s2, 3 channel tiles: 30 GB fetched for a 4.2 GB
    // input); launched as one row in XCD-major (slot, channel tile) order they share an XCD's L2 (halo_block_map)
    int bx = blockIdx.x, by = blockIdx.y, gdx = gridDim.x;
    if (p.xgx > 0) {
        const int nb = p.xgx * p.xgy, lin = blockIdx.x, x = lin & 7, q = nb >> 3, r = nb & 7;
        const int vid = (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + (lin >> 3);
        bx = vid / p.xgy; by = vid - bx * p.xgy; gdx = p.xgx;
    }
    const int n0 = by * BN;
    const int HoWo = d.Ho * d.Wo;
    const int sdmask = (1 << p.sdshift) - 1;
    int ph = 0, pw = 0, zslot = 0, zslots = 1;
    if (p.cls) cls_slot(d, blockIdx.z, ph, pw, zslot, zslots);
    const int kh0 = (ph + d.pad) & 1, kw0 = (pw + d.pad) & 1;
    const int nkw = p.cls ? (d.KW - kw0 + 1) / 2 : d.KW;
    const int nkh = p.cls ? (d.KH - kh0 + 1) / 2 : d.KH;
    const int ncb = TL ? (p.Ctot + BKT - 1) / BKT : p.Ctot / BKT;
    const int tailn = TL ? (p.Ctot % BKT) / 16 : BKT / 16;
    const int nkt = nkh * nkw * ncb;
    const int HcWc = p.Hc * p.Wc;
    const int C0 = d.seg[0].C;
    const int mtiles = (p.M + BMT - 1) / BMT;

    const __amdgpu_buffer_rsrc_t rsw = __builtin_amdgcn_make_buffer_rsrc((void*)d.w, 0, p.wbytes, 0x00020000);
    // Input descriptors are re-based at every tile (to the first pixel row of a pointwise tile, else to the first image the
    // tile touches), so the 32-bit lane offsets stay tile-relative and an input tensor may be larger than the 2 GiB one
    // descriptor can address (the host checks that the images one tile spans fit: conv_v3_span_ok).
    const unsigned long pimg0 = (unsigned long)(d.Hi >> d.seg[0].ups) * (d.Wi >> d.seg[0].ups) * (unsigned long)(d.seg[0].ld * 2);
    const unsigned long pimg1 = (unsigned long)(d.Hi >> d.seg[1].ups) * (d.Wi >> d.seg[1].ups) * (unsigned long)(d.seg[1].ld * 2);
    auto rebased = [](const void* ptr, unsigned long off, unsigned long total) {
        const unsigned long rem = total - off;
        return __builtin_amdgcn_make_buffer_rsrc((void*)(reinterpret_cast<const char*>(ptr) + off), 0,
                                                 rem > 0x7fffffffUL ? 0x7fffffffu : (unsigned)rem, 0x00020000);
    };

    // loader geometry: instruction i of this wave fills tile rows (i*NWV + wave)*RPI .. +RPI; this lane's row / chunk
    const int lrow = lane / CHR, lq = lane % CHR;
    unsigned voffB[NB];
#pragma unroll
    for (int j = 0; j < NB; ++j) {
        const int row = (j * NWV + wave) * RPI + lrow;
        voffB[j] = (unsigned)(((n0 + row) * p.Ktot + ((lq ^ swz_f<CHR>(row)) * 8)) * 2);
    }
    unsigned chA[NA];                                // byte offset of this lane's (swizzled) source chunk inside the k-block
#pragma unroll
    for (int i = 0; i < NA; ++i) chA[i] = (unsigned)((lq ^ swz_f<CHR>((i * NWV + wave) * RPI + lrow)) * 16);

    // fragment reads: rows of a wave tile differ by multiples of 32, so the swizzle term depends on the lane only
    const int fx = swz_f<CHR>(lane & 31);
    int koff[BKT / 16];
#pragma unroll
    for (int ks = 0; ks < BKT / 16; ++ks) koff[ks] = ((ks * 2 + (lane >> 5)) ^ fx) * 16;
    const int rdA0 = (wm * (TM * 32) + (lane & 31)) * ROWB;                 // + i*32*ROWB
    const int rdB0 = BMT * ROWB + (wn * (TN * 32) + (lane & 31)) * ROWB;    // + j*32*ROWB

    float run_s = 0.f, run_q = 0.f;
    float bs_[8], bq_[8];
    if (EPI == 3) {
#pragma unroll
        for (int e = 0; e < 8; ++e) { bs_[e] = 0.f; bq_[e] = 0.f; }
        for (int i = t; i < 2 * BN; i += NT) {
            const int which = i / BN, c = i - which * BN;
            sStat[i] = (n0 + c < d.N) ? d.bnr_ws[(size_t)which * d.bnr_C + n0 + c] : 0.f;
        }
        __syncthreads();
    }

    for (int mt = bx + gdx * zslot; mt < mtiles; mt += gdx * zslots) {
        const int m0 = mt * BMT;
        const int im0 = p.pointwise ? 0 : m0 / (p.cls ? HcWc : HoWo);          // first image of the tile
        const unsigned long off0 = p.pointwise ? (unsigned long)m0 * (unsigned long)(d.seg[0].ld * 2) : (unsigned long)im0 * pimg0;
        const unsigned long off1 = p.pointwise ? (unsigned long)m0 * (unsigned long)(d.seg[1].ld * 2) : (unsigned long)im0 * pimg1;
        const __amdgpu_buffer_rsrc_t rs0 = rebased(d.seg[0].ptr, off0, p.segbytes64[0]);
        const __amdgpu_buffer_rsrc_t rs1 = rebased(d.seg[1].ptr, off1, p.segbytes64[1]);
        int hb[NA], wb[NA], img[NA];          // img: relative to im0
        unsigned voff0[NA], voff1[NA];
#pragma unroll
        for (int i = 0; i < NA; ++i) {
            const int row = (i * NWV + wave) * RPI + lrow;
            const int m = m0 + row;
            voff0[i] = OOB; voff1[i] = OOB;
            img[i] = 0; hb[i] = -(1 << 28); wb[i] = -(1 << 28);
            if (m < p.M) {
                if (p.pointwise) {
                    voff0[i] = (unsigned)row * (unsigned)(d.seg[0].ld * 2) + chA[i];
                    voff1[i] = (unsigned)row * (unsigned)(d.seg[1].ld * 2) + chA[i];
                } else {
                    int im, ho, wo;
                    if (p.cls) {
                        im = m / HcWc;
                        const int rem = m - im * HcWc;
                        const int ii = rem / p.Wc;
                        ho = 2 * ii + ph; wo = 2 * (rem - ii * p.Wc) + pw;
                    } else {
                        im = m / HoWo;
                        const int rem = m - im * HoWo;
                        ho = rem / d.Wo;
                        wo = rem - ho * d.Wo;
                    }
                    img[i] = im - im0; hb[i] = ho * p.sa + p.sc; wb[i] = wo * p.sa + p.sc;
                    if (p.cls && lq == 0) sPix[row] = (im * d.Ho + ho) * d.Wo + wo;
                }
            }
        }

        f32x16_t acc[TM][TN];
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

        int ld_tap = 0, ld_cb = 0, kcol_base = 0;
        auto tap_setup = [&](int tapl) {
            int kh = tapl / nkw;
            int kw = tapl - kh * nkw;
            if (p.cls) { kh = kh0 + 2 * kh; kw = kw0 + 2 * kw; }
            kcol_base = (kh * d.KW + kw) * p.Ctot;
            if (p.pointwise) return;
            const int u0 = d.seg[0].ups, u1 = d.seg[1].ups;
#pragma unroll
            for (int i = 0; i < NA; ++i) {
                const int hn = hb[i] + kh * p.sb;
                const int wn_ = wb[i] + kw * p.sb;
                bool ok = hn >= 0 && wn_ >= 0 && (((hn | wn_) & sdmask) == 0);
                const int hs = hn >> p.sdshift, ws = wn_ >> p.sdshift;
                ok = ok && hs < d.Hi && ws < d.Wi;
                const unsigned pix0 = (unsigned)((img[i] * (d.Hi >> u0) + (hs >> u0)) * (d.Wi >> u0) + (ws >> u0));
                const unsigned pix1 = (unsigned)((img[i] * (d.Hi >> u1) + (hs >> u1)) * (d.Wi >> u1) + (ws >> u1));
                voff0[i] = ok ? pix0 * (unsigned)(d.seg[0].ld * 2) + chA[i] : OOB;
                voff1[i] = ok ? pix1 * (unsigned)(d.seg[1].ld * 2) + chA[i] : OOB;
            }
        };
        // one stage = NA + NB LDS-DMA instructions of this wave
        auto issue = [&](int slot) {
            if (ld_cb == 0) tap_setup(ld_tap);
            const int c = ld_cb * BKT;
            const bool s1 = d.nseg > 1 && c >= C0;           // wave-uniform
            unsigned char* sa = smem + slot * STAGE_BYTES + wave * (RPI * ROWB);
            if (s1) {
                const int so = (c - C0) * 2;
#pragma unroll
                for (int i = 0; i < NA; ++i)
                    lds_dma16(rs1, sa + i * (NWV * RPI * ROWB), voff1[i], so);
            } else {
                const int so = c * 2;
#pragma unroll
                for (int i = 0; i < NA; ++i)
                    lds_dma16(rs0, sa + i * (NWV * RPI * ROWB), voff0[i], so);
            }
            const int sw = (kcol_base + c) * 2;
            unsigned char* sb = sa + BMT * ROWB;
#pragma unroll
            for (int j = 0; j < NB; ++j)
                lds_dma16(rsw, sb + j * (NWV * RPI * ROWB), voffB[j], sw);
            if (++ld_cb == ncb) { ld_cb = 0; ++ld_tap; }
        };

#pragma unroll
        for (int s = 0; s < STG - 1; ++s)
            if (s < nkt) issue(s);
        int slot = 0, islot = STG - 1, rd_cb = 0;
        for (int kt = 0; kt < nkt; ++kt) {
            // this wave's transfers of stage kt have landed when at most the younger stages are outstanding
            const int younger = nkt - 1 - kt;
            if (STG == 2 || younger == 0) YH_VMCNT(0);
            else if (STG == 3 || younger == 1) YH_VMCNT(LPS);
            else YH_VMCNT(2 * LPS);
            __builtin_amdgcn_s_barrier();          // stage kt complete for every wave; stage kt-1 no longer read by anyone
            // The 8-wave tiles: a wave issues 6-8 transfers per step (~100 cycles of issue each, no MFMA of that wave meanwhile).  Both
            // waves of a SIMD (w and w + 4) doing so right behind the barrier leaves the matrix pipe idle for that long: waves 4-7
            // issue theirs behind their first sub-step — which every channel block has, also a tail block — beside their partners'
            // MFMAs (the slot they fill was freed by the barrier; a wave still issues once per step, in order: the counted waits
            // hold).  Isolated +3...+11 % on both 8-wave tiles (profiles/r06_step_experiments.txt j).
            constexpr bool W8 = NWV == 8;
            const bool more = kt + STG - 1 < nkt;
            if (more && (!W8 || wave < NWV / 2)) issue(islot);
            const unsigned char* sbase = smem + slot * STAGE_BYTES;
            int ksn = BKT / 16;                    // wave-uniform: sub-steps of this channel block
            if (TL) {
                if (rd_cb + 1 == ncb) { ksn = tailn; rd_cb = 0; } else ++rd_cb;
            }
#pragma unroll
            for (int ks = 0; ks < BKT / 16; ++ks) {
                if (TL && ks >= ksn) continue;
                bf16x8_t af[TM], bfr[TN];
#pragma unroll
                for (int i = 0; i < TM; ++i)
                    af[i] = __builtin_bit_cast(bf16x8_t, *reinterpret_cast<const uint4*>(sbase + rdA0 + i * (32 * ROWB) + koff[ks]));
#pragma unroll
                for (int j = 0; j < TN; ++j)
                    bfr[j] = __builtin_bit_cast(bf16x8_t, *reinterpret_cast<const uint4*>(sbase + rdB0 + j * (32 * ROWB) + koff[ks]));
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i], bfr[j], acc[i][j], 0, 0, 0);
                if (W8 && ks == 0 && more && wave >= NWV / 2) issue(islot);
            }
            slot = slot + 1 == STG ? 0 : slot + 1;
            islot = islot + 1 == STG ? 0 : islot + 1;
        }
        __syncthreads();                           // nothing in flight (last wait was vmcnt(0)); ring -> epilogue buffer

        // ---- epilogue (as v2): accumulators -> LDS (bf16, row-major) -> 16-byte global stores
        // EPI 3 reads the producer's raw output z of every output chunk: requested ahead of the accumulator conversion — except on the
        // 256 x 256 tile (16 chunks per thread: 64 registers beside 128 accumulators spill), which loads z where it is used
        constexpr int CPRz = BN / 8;
        constexpr int NCHz = BMT * CPRz / NT;
        constexpr bool ZPRE = NCHz <= 8;
        uint4 zpre[(EPI == 3 && ZPRE) ? NCHz : 1];
        if (EPI == 3 && ZPRE) {
#pragma unroll
            for (int i = 0; i < NCHz; ++i) {
                const int id = t + i * NT;
                const int row = id / CPRz;
                const int m = m0 + row;
                const int n = n0 + (id - row * CPRz) * 8;
                zpre[i] = make_uint4(0, 0, 0, 0);
                if (m < p.M && n < d.N) {
                    const size_t orow = p.cls ? (size_t)sPix[row] : (size_t)m;
                    zpre[i] = *reinterpret_cast<const uint4*>(d.bnr_z + orow * d.bnr_ldz + n);
                }
            }
        }
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const int c = wn * (TN * 32) + j * 32 + (lane & 31);
            float bs = 0.f, scl = 1.f, sft = 0.f;
            if (EPI == 2) {
                const int n = n0 + c;
                const bool nv = n < d.N;
                bs = (d.bias && nv) ? d.bias[n] : 0.f;
                scl = (d.scale && nv) ? d.scale[n] : 1.f;
                sft = (d.shift && nv) ? d.shift[n] : 0.f;
            }
            float s = 0.f, q = 0.f;
#pragma unroll
            for (int i = 0; i < TM; ++i) {
                uint16_t* dst = sC + (wm * (TM * 32) + i * 32 + 4 * (lane >> 5)) * CP + c;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    float v = acc[i][j][r];
                    if (EPI == 2) {
                        v = (v + bs) * scl + sft;
                        if (d.act == YH_ACT_SILU) v = silu_fast(v);
                    }
                    dst[((r & 3) + 8 * (r >> 2)) * CP] = f2bf(v);
                    if (EPI == 1) { s += v; q += v * v; }
                }
            }
            if (EPI == 1) {
                s += __shfl_xor(s, 32, 64);
                q += __shfl_xor(q, 32, 64);
                if (lane < 32) {
                    sStat[(wm * 2 + 0) * BN + c] = s;
                    sStat[(wm * 2 + 1) * BN + c] = q;
                }
            }
        }
        __syncthreads();

        constexpr int CPR = BN / 8;
        constexpr int NCH = BMT * CPR / NT;
        uint4 zq[(EPI == 3 && !ZPRE) ? 4 : 1];       // the 256 x 256 tile: z of the next four chunks, requested together
#pragma unroll
        for (int i = 0; i < NCH; ++i) {
            if (EPI == 3 && !ZPRE && (i & 3) == 0) {
                // (the stores of the previous chunks may alias z for all the compiler knows: loads left inside the chunk's own
                // iteration were issued one by one behind them — sixteen dependent round trips per tile)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const int id2 = t + (i + g) * NT;
                    const int row2 = id2 / CPR;
                    const int m2 = m0 + row2;
                    const int n2 = n0 + (id2 - row2 * CPR) * 8;
                    zq[g] = make_uint4(0, 0, 0, 0);
                    if (m2 < p.M && n2 < d.N) {
                        const size_t orow2 = p.cls ? (size_t)sPix[row2] : (size_t)m2;
                        zq[g] = *reinterpret_cast<const uint4*>(d.bnr_z + orow2 * d.bnr_ldz + n2);
                    }
                }
            }
            const int id = t + i * NT;
            const int row = id / CPR;
            const int cch = id - row * CPR;
            const int m = m0 + row;
            const int n = n0 + cch * 8;
            if (m < p.M && n < d.N) {
                uint4 v = *reinterpret_cast<const uint4*>(sC + row * CP + cch * 8);
                const size_t orow = p.cls ? (size_t)sPix[row] : (size_t)m;
                if (EPI == 2) {
                    uint16_t* dst;
                    const bool first = n < d.nsplit;
                    if (first) dst = d.out0 + orow * d.ld0 + n;
                    else       dst = d.out1 + orow * d.ld1 + (n - d.nsplit);
                    const bool addres = (d.res != nullptr) && first;
                    if (addres || d.accumulate) {
                        float f[8];
                        unpack8(v, f);
                        if (addres) {
                            uint4 rv = *reinterpret_cast<const uint4*>(d.res + orow * d.ldr + n);
                            float g[8]; unpack8(rv, g);
#pragma unroll
                            for (int e = 0; e < 8; ++e) f[e] += g[e];
                        }
                        if (d.accumulate) {
                            uint4 ov = *reinterpret_cast<const uint4*>(dst);
                            float g[8]; unpack8(ov, g);
#pragma unroll
                            for (int e = 0; e < 8; ++e) f[e] += g[e];
                        }
                        v = pack8(f);
                    }
                    *reinterpret_cast<uint4*>(dst) = v;
                } else {
                    if (EPI == 3 && d.accumulate) {
                        // last writer of a gradient with several contributions: add the earlier ones (bf16, as the generic
                        // epilogue does) and take the BatchNorm-backward sums over the rounded total
                        const uint4 ov = *reinterpret_cast<const uint4*>(d.out0 + orow * d.ld0 + n);
                        float f[8], g0[8];
                        unpack8(v, f);
                        unpack8(ov, g0);
#pragma unroll
                        for (int e = 0; e < 8; ++e) f[e] += g0[e];
                        v = pack8(f);
                    }
                    *reinterpret_cast<uint4*>(d.out0 + orow * d.ld0 + n) = v;
                    if (EPI == 3) {
                        const uint4 zv = ZPRE ? zpre[ZPRE ? i : 0] : zq[ZPRE ? 0 : (i & 3)];
                        float g[8], z[8];
                        unpack8(v, g);
                        unpack8(zv, z);
                        const float4 s0 = *reinterpret_cast<const float4*>(sStat + cch * 8);
                        const float4 s1 = *reinterpret_cast<const float4*>(sStat + cch * 8 + 4);
                        const float4 h0 = *reinterpret_cast<const float4*>(sStat + BN + cch * 8);
                        const float4 h1 = *reinterpret_cast<const float4*>(sStat + BN + cch * 8 + 4);
                        const float sc[8] = {s0.x, s0.y, s0.z, s0.w, s1.x, s1.y, s1.z, s1.w};
                        const float sh[8] = {h0.x, h0.y, h0.z, h0.w, h1.x, h1.y, h1.z, h1.w};
#pragma unroll
                        for (int e = 0; e < 8; ++e) {
                            const float a = z[e] * sc[e] + sh[e];
                            const float sg = sigmoid_fast(a);
                            const float dz = g[e] * (sg * (1.f + a * (1.f - sg)));
                            bs_[e] += dz; bq_[e] += dz * z[e];
                        }
                    }
                }
            }
        }
        if (EPI == 1 && t < BN) {
#pragma unroll
            for (int w = 0; w < WM; ++w) {
                run_s += sStat[(w * 2 + 0) * BN + t];
                run_q += sStat[(w * 2 + 1) * BN + t];
            }
        }
        __syncthreads();
    }

    if (EPI == 1 && t < BN) {
        put_stat(d, bx, 0, n0 + t, run_s);
        put_stat(d, bx, 1, n0 + t, run_q);
    }
    if (EPI == 3) {
        constexpr int CPR2 = BN / 8;
        float* sRed = reinterpret_cast<float*>(smem);
        __syncthreads();
#pragma unroll
        for (int e = 0; e < 8; ++e) { sRed[t * 16 + e] = bs_[e]; sRed[t * 16 + 8 + e] = bq_[e]; }
        __syncthreads();
        const size_t rowi = (size_t)blockIdx.z * gdx + bx;
        for (int i = t; i < 2 * BN; i += NT) {
            const int which = i / BN, c = i - which * BN;
            float v = 0.f;
            for (int j = c / 8; j < NT; j += CPR2) v += sRed[j * 16 + which * 8 + (c & 7)];
            if (n0 + c < d.N) put_bnr(d, rowi, which, n0 + c, v);
        }
    }
}

// ------------------------------------------------------------------------------------------------
// halo kernel: 3x3 / stride 1 / pad 1 convolutions (forward and data gradient) without the im2col redundancy on the
// global -> LDS path.  The CU's intake (LDS-DMA through the texture path, ~35-45 B/clk in practice) is what bounds the
// tiles of conv_v3: a 256 x 128 tile needs 48 KB per 64-channel k-step.  Here a block owns a 2-D tile of TH x TW output
// pixels (TH*TW <= 256) of ONE image and stages, per 64-channel block, the (TH+2) x (TW+2) input patch ONCE; the nine taps
// are nine k-steps that read their A fragments from the same patch at a shifted pixel (patch row + dy*(TW+2) + dx), so only
// the weight tile (16 KB) is streamed per k-step: ~22 KB per k-step instead of 48.  Patch rows are 128 bytes, chunk-swizzled
// by the patch row index on the source side of the DMA (as in conv_v3); the patch of the next channel block is prefetched
// in six parts behind the taps 0..5 of the current one; weight tiles go through a 3-stage ring.  MFMA rows that are not an
// output pixel of the tile (ragged tiles) read a zero line, so their accumulators are exact zeros for the statistics.
struct HaloGeom { int TH, TW, PW, NP, tiles_x, tiles_y, gx, gy, rowmajor;
                  unsigned long long* stamps; };   // diagnostics (yh_halo_set_stamps): [workgroup][wave][8] cycle sums of the workgroup's 2nd tile

static unsigned long long* g_halo_stamps = nullptr;

// Block -> (persistent tile slot bx, output-channel tile by) of the halo kernels, launched as ONE row of gx * gy workgroups.  The gy
// channel tiles of a pixel tile stage the SAME input patch: as grid rows (by = blockIdx.y) their linear ids are gx apart, i.e. they
// land on different XCDs at different times and every one fetches the patch again (PMC, YOLOv5x at 1280^2: 2.78x the algorithmic
// bytes on the 320-channel layers).  Workgroups are dealt round-robin to the 8 XCDs by linear id: XCD x takes a contiguous range of
// the order (slot, channel tile), so the gy readers of a patch sit on one XCD, dispatched back to back, and share its L2.
__device__ __forceinline__ void halo_block_map(const HaloGeom& hg, int& bx, int& by)
{
    if (hg.rowmajor) { by = blockIdx.x / hg.gx; bx = blockIdx.x - by * hg.gx; return; }     // YH_HALO_MAP=0: the round-3 order (A/B)
    const int nb = hg.gx * hg.gy, lin = blockIdx.x, x = lin & 7, q = nb >> 3, r = nb & 7;
    const int vid = (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + (lin >> 3);
    bx = vid / hg.gy;
    by = vid - bx * hg.gy;
}

// TL (tail): the channel count is a multiple of 16 but not of 64 (YOLOv5m / v5x widths: 96, 80, 160): the last channel block
// runs only (C % 64) / 16 of its four 16-channel sub-steps.  Its DMA still fetches whole 128-byte rows (the bytes behind the
// last channel belong to the next pixel / the next tap's weights, or are zero-filled past the end of the buffer): they are
// never read as fragments.
template <int BN, int EPI, bool TL>
__global__ __launch_bounds__(512, 2) void conv_halo_kernel(const ConvK p, const HaloGeom hg)
{
    constexpr int BMT = 256, WM = 4, WN = 2, BKT = 64, STG = 4;
    constexpr int NWV = WM * WN, NT = NWV * 64;
    constexpr int TM = 2, TN = BN / (WN * 32);
    constexpr int ROWB = BKT * 2, CHR = 8, RPI = 8;
    constexpr int NB = BN / (RPI * NWV);             // weight-tile DMA instructions per wave and k-step (2 | 1)
    constexpr int NPW = 6;                           // patch DMA instructions per wave and channel block (<= 6*8*8 = 384 patch rows)
    constexpr int PATCH_ROWS = 352;                  // >= NP*8, see conv_halo_geom
    constexpr int PATCH_BYTES = PATCH_ROWS * ROWB;   // 45056
    constexpr int BST_BYTES = BN * ROWB;
    constexpr int ZERO_OFF = 2 * PATCH_BYTES + STG * BST_BYTES;       // 128 zero bytes
    constexpr int MAIN_BYTES = ZERO_OFF + 128;
    constexpr int CP = BN + 8;                       // epilogue buffer: 128 rows x CP bf16 inside the idle patch buffer
    static_assert(128 * CP * 2 <= PATCH_BYTES, "epilogue half tile must fit a patch buffer");
    constexpr unsigned OOB = 0x80000000u;

    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    float* sConst = reinterpret_cast<float*>(smem + MAIN_BYTES);            // [3][BN]: bias | scale | shift (EPI 2), scale | shift (EPI 3)
    int* sPix = reinterpret_cast<int*>(smem + MAIN_BYTES + 3 * BN * 4);     // [BMT] output pixel of each tile row, -1 = none

    const yh_conv_desc& d = p.d;
    const int t = threadIdx.x;
    const int lane = t & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int wm = wave / WN, wn = wave % WN;
    int bx, by;
    halo_block_map(hg, bx, by);
    const int n0 = by * BN;
    const int ncb = (p.Ctot + BKT - 1) / BKT;
    const int tailn = TL ? (p.Ctot % BKT) / 16 : BKT / 16;   // 16-channel sub-steps of the last channel block
    const int H = d.Ho, W = d.Wo;                    // stride 1, pad 1: input grid == output grid
    const int TH = hg.TH, TW = hg.TW, PW = hg.PW, NP = hg.NP;
    const int tiles_per_img = hg.tiles_x * hg.tiles_y;
    const int ntiles = d.B * tiles_per_img;
    const int ldx2 = d.seg[0].ld * 2;
    const bool dgrad = d.mode == YH_CONV_DGRAD;
#if YH_CONV_ABLATE & 1
    const bool mtile_never = d.B < 0;               // timing build: DMA issue compiled in but never taken
#endif

    const __amdgpu_buffer_rsrc_t rs0 = __builtin_amdgcn_make_buffer_rsrc((void*)d.seg[0].ptr, 0, p.segbytes[0], 0x00020000);
    const __amdgpu_buffer_rsrc_t rsw = __builtin_amdgcn_make_buffer_rsrc((void*)d.w, 0, p.wbytes, 0x00020000);

    const int lrow = lane >> 3, lq = lane & 7;
    unsigned voffB[NB];
#pragma unroll
    for (int j = 0; j < NB; ++j) {
        const int row = (j * NWV + wave) * RPI + lrow;
        voffB[j] = (unsigned)(((n0 + row) * p.Ktot + ((lq ^ swz_f<CHR>(row)) * 8)) * 2);
    }
    const int hsel = lane >> 5;
    int rdBk[BKT / 16];                               // weight fragment offsets inside a ring slot, per 16-channel sub-step
#pragma unroll
    for (int ks = 0; ks < BKT / 16; ++ks)
        rdBk[ks] = 2 * PATCH_BYTES + (wn * (TN * 32) + (lane & 31)) * ROWB + (((ks * 2 + hsel) ^ swz_f<CHR>(lane & 31)) << 4);

    if (t < 32) reinterpret_cast<unsigned*>(smem + ZERO_OFF)[t] = 0u;
    if (EPI == 2) {
        for (int i = t; i < 3 * BN; i += NT) {
            const int which = i / BN, c = i - which * BN;
            const float* src = which == 0 ? d.bias : (which == 1 ? d.scale : d.shift);
            sConst[i] = (src && n0 + c < d.N) ? src[n0 + c] : (which == 1 ? 1.f : 0.f);
        }
    }
    if (EPI == 3) {
        for (int i = t; i < 2 * BN; i += NT) {
            const int which = i / BN, c = i - which * BN;
            sConst[i] = (n0 + c < d.N) ? d.bnr_ws[(size_t)which * d.bnr_C + n0 + c] : 0.f;
        }
    }
    // per-thread running sums over the 8 channels of the chunk this thread stores (EPI 1: sum, sum of squares of the stored
    // bf16 values; EPI 3: sum dz, sum dz*z): a thread always handles the same channel chunk (NT % (BN/8) == 0)
    float bs_[8], bq_[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) { bs_[e] = 0.f; bq_[e] = 0.f; }
    __syncthreads();

    // tile -> origin, patch-loader offsets (instruction i of this wave fills patch rows (i*8 + wave)*8 .. +8)
    auto patch_offsets = [&](int tile, unsigned (&vo)[NPW]) {
        const int img = tile / tiles_per_img;
        const int trem = tile - img * tiles_per_img;
        const int tyi = trem / hg.tiles_x;
        const int y0 = tyi * TH, x0 = (trem - tyi * hg.tiles_x) * TW;
#pragma unroll
        for (int i = 0; i < NPW; ++i) {
            const int pr = (i * NWV + wave) * RPI + lrow;
            const int py = pr / PW, px = pr - py * PW;
            const int y = y0 - 1 + py, x = x0 - 1 + px;
            const bool ok = pr < (TH + 2) * PW && y >= 0 && y < H && x >= 0 && x < W;
            vo[i] = ok ? (unsigned)((img * H + y) * W + x) * (unsigned)ldx2 + (unsigned)((lq ^ swz_f<CHR>(pr)) * 16) : OOB;
        }
    };
    auto issue_patch_part = [&](const unsigned (&vo)[NPW], int part, int cblk, int pbuf) -> int {   // one instruction of this wave
        const int inst = part * NWV + wave;
        if (inst >= NP) return 0;
        unsigned char* dst = smem + pbuf * PATCH_BYTES + inst * (RPI * ROWB);
#pragma unroll
        for (int i = 0; i < NPW; ++i)
            if (i == part) lds_dma16(rs0, dst, vo[i], cblk * (BKT * 2));
        return 1;
    };
    auto issue_B = [&](int kt2, int slot) {           // weight tile of k-step kt2 (the stream repeats every nkt steps: tile independent)
        const int cblk = kt2 / 9, tap = kt2 - cblk * 9;
        const int sw = (tap * p.Ctot + cblk * BKT) * 2;
        unsigned char* sb = smem + 2 * PATCH_BYTES + slot * BST_BYTES + wave * (RPI * ROWB);
#pragma unroll
        for (int j = 0; j < NB; ++j) lds_dma16(rsw, sb + j * (NWV * RPI * ROWB), voffB[j], sw);
    };

    int tile = bx;
    unsigned voffP[NPW], voffN[NPW];
    int slot = 0, islot = STG - 1, pb = 0;
    if (tile < ntiles) {
        patch_offsets(tile, voffP);
#pragma unroll
        for (int part = 0; part < NPW; ++part) issue_patch_part(voffP, part, 0, 0);
        issue_B(0, 0);
        issue_B(1, 1);                               // nkt >= 9
        issue_B(2, 2);
        YH_VMCNT(NB);                                // patch and the weight tiles of steps 0, 1 have landed; the third stays in flight
    }
    for (; tile < ntiles; tile += hg.gx) {
        const bool has_next = tile + hg.gx < ntiles;
        const int img = tile / tiles_per_img;
        const int trem = tile - img * tiles_per_img;
        const int tyi = trem / hg.tiles_x;
        const int y0 = tyi * TH, x0 = (trem - tyi * hg.tiles_x) * TW;
        // fragment rows of this lane: MFMA row -> tile pixel -> patch row of the tap (dy, dx) = (0, 0)
        int pr0[TM];
        bool inv[TM];
#pragma unroll
        for (int i = 0; i < TM; ++i) {
            const int r = wm * (TM * 32) + i * 32 + (lane & 31);
            const int ty = r / TW, tx = r - ty * TW;
            inv[i] = !(ty < TH && y0 + ty < H && x0 + tx < W);
            pr0[i] = ty * PW + tx;
        }
        for (int r = t; r < BMT; r += NT) {
            const int ty = r / TW, tx = r - ty * TW;
            sPix[r] = (ty < TH && y0 + ty < H && x0 + tx < W) ? ((img * H + y0 + ty) * W + x0 + tx) : -1;
        }

        f32x16_t acc[TM][TN];
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

        // k loop: channel blocks outside, the nine taps fully unrolled inside (tap, its (dy, dx), the patch part and the weight
        // column of the step two ahead are compile-time: the per-step scalar / vector overhead is what bounds this loop, not MFMA)
        int prev_group = 0;
        bf16x8_t afN[TM], bfN[TN];
        for (int cblk = 0; cblk < ncb; ++cblk) {
            const bool last_blk = cblk + 1 == ncb;
            const int ksn = (TL && last_blk) ? tailn : BKT / 16;      // wave-uniform
            if (last_blk && has_next) patch_offsets(tile + hg.gx, voffN);
#pragma unroll
            for (int tap = 0; tap < 9; ++tap) {
                const int kt = cblk * 9 + tap;
                // all DMA groups except the one issued at the previous step must have landed (step 0: waited before the tile)
#if YH_CONV_ABLATE & 1
                YH_VMCNT(0);
#else
                if (kt > 0) YH_VMCNT_SW(prev_group);
#endif
#if !(YH_CONV_ABLATE & 256)
                __builtin_amdgcn_s_barrier();
#endif
                prev_group = 0;
#if YH_CONV_ABLATE & 1
                if (mtile_never)
#endif
                {
                    // weight tile of step kt + 2 (wrapping into the next tile's stream)
                    const int tapB = (tap + STG - 1) % 9;
                    const int cblkB = cblk + ((tap + STG - 1) >= 9 ? 1 : 0);
                    if (cblkB < ncb || has_next) {
                        const int cb2 = cblkB < ncb ? cblkB : 0;
                        const int sw = (tapB * p.Ctot + cb2 * BKT) * 2;
                        unsigned char* sb = smem + 2 * PATCH_BYTES + islot * BST_BYTES + wave * (RPI * ROWB);
#pragma unroll
                        for (int j = 0; j < NB; ++j) lds_dma16(rsw, sb + j * (NWV * RPI * ROWB), voffB[j], sw);
                        prev_group = NB;
                    }
                    if (tap < NPW && !(YH_CONV_ABLATE & 512)) {
                        if (!last_blk) prev_group += issue_patch_part(voffP, tap, cblk + 1, pb ^ 1);
                        else if (has_next) prev_group += issue_patch_part(voffN, tap, 0, pb ^ 1);
                    }
                }
                // fragments of this step: sub-step 0 was requested during the previous step (A from the resident patch, B from a ring
                // slot that had landed one barrier earlier: the ring is one stage deeper than the steps in flight need), so the
                // MFMAs start right behind the barrier; only the first step of a tile loads them here
                bf16x8_t af[TM], bfr[TN];
                if (tap == 0 && cblk == 0) {
                    int tapoff0 = dgrad ? 2 * PW + 2 : 0;
                    asm volatile("" : "+s"(tapoff0));
#pragma unroll
                    for (int i = 0; i < TM; ++i) {
                        const int pr = pr0[i] + tapoff0;
                        const int fx = swz_f<CHR>(pr);
                        const int ab = inv[i] ? ZERO_OFF : pb * PATCH_BYTES + pr * ROWB + (((fx & 1) ^ hsel) << 4);
                        afN[i] = __builtin_bit_cast(bf16x8_t, *reinterpret_cast<const uint4*>(smem + ab + ((inv[i] ? 0 : (fx >> 1)) << 5)));
                    }
#pragma unroll
                    for (int j = 0; j < TN; ++j)
                        bfN[j] = __builtin_bit_cast(bf16x8_t, *reinterpret_cast<const uint4*>(smem + slot * BST_BYTES + rdBk[0] + j * (32 * ROWB)));
                }
                const int kh = tap / 3, kw = tap % 3;
                int tapoff = dgrad ? (2 - kh) * PW + (2 - kw) : kh * PW + kw;            // scalar
                // keep this step's address arithmetic inside the step: hoisted to the top of the unrolled block it costs 36 live VGPRs
                asm volatile("" : "+s"(tapoff));
                const int sbase = slot * BST_BYTES;
                // A: patch row of this lane's pixel for this tap; rows outside the tile read the 128-byte zero line.  The chunk
                // swizzle (kc ^ f(row)) << 4 with kc = 2 ks + h splits into a per-step term (h ^ f0) << 4 and ((ks ^ f12) << 5)
                int abase[TM], af12[TM];
#pragma unroll
                for (int i = 0; i < TM; ++i) {
                    const int pr = pr0[i] + tapoff;
                    const int fx = swz_f<CHR>(pr);
                    abase[i] = inv[i] ? ZERO_OFF : pb * PATCH_BYTES + pr * ROWB + (((fx & 1) ^ hsel) << 4);
                    af12[i] = inv[i] ? 0 : (fx >> 1);
                }
#pragma unroll
                for (int ks = 0; ks < BKT / 16; ++ks) {
                    if (TL && ks >= ksn) continue;
                    if (ks == 0) {
#pragma unroll
                        for (int i = 0; i < TM; ++i) af[i] = afN[i];
#pragma unroll
                        for (int j = 0; j < TN; ++j) bfr[j] = bfN[j];
                    } else {
#pragma unroll
                        for (int i = 0; i < TM; ++i)
                            af[i] = __builtin_bit_cast(bf16x8_t, *reinterpret_cast<const uint4*>(smem + abase[i] + ((ks ^ af12[i]) << 5)));
#pragma unroll
                        for (int j = 0; j < TN; ++j)
                            bfr[j] = __builtin_bit_cast(bf16x8_t, *reinterpret_cast<const uint4*>(smem + sbase + rdBk[ks] + j * (32 * ROWB)));
                    }
                    if (ks == (TL ? 0 : BKT / 16 - 2) && !(tap == 8 && last_blk)) {
                        // request sub-step 0 of the NEXT step (next tap of this channel block, or tap 0 of the next block in the other
                        // patch buffer, whose parts landed before this step's barrier)
                        const int ntap = tap == 8 ? 0 : tap + 1;
                        const int nkh = ntap / 3, nkw = ntap % 3;
                        int ntapoff = dgrad ? (2 - nkh) * PW + (2 - nkw) : nkh * PW + nkw;
                        asm volatile("" : "+s"(ntapoff));
                        const int npb = tap == 8 ? (pb ^ 1) : pb;
                        const int nslot = slot + 1 == STG ? 0 : slot + 1;
#pragma unroll
                        for (int i = 0; i < TM; ++i) {
                            const int pr = pr0[i] + ntapoff;
                            const int fx = swz_f<CHR>(pr);
                            const int ab = inv[i] ? ZERO_OFF : npb * PATCH_BYTES + pr * ROWB + (((fx & 1) ^ hsel) << 4);
                            afN[i] = __builtin_bit_cast(bf16x8_t, *reinterpret_cast<const uint4*>(smem + ab + ((inv[i] ? 0 : (fx >> 1)) << 5)));
                        }
#pragma unroll
                        for (int j = 0; j < TN; ++j)
                            bfN[j] = __builtin_bit_cast(bf16x8_t, *reinterpret_cast<const uint4*>(smem + nslot * BST_BYTES + rdBk[0] + j * (32 * ROWB)));
                    }
                    // operands swapped (D = W x X^T): a lane then holds ONE pixel (lane & 31) and, per register group, four
                    // consecutive channels: the accumulators pack to 8-byte LDS stores in the epilogue
#pragma unroll
                    for (int i = 0; i < TM; ++i)
#pragma unroll
                        for (int j = 0; j < TN; ++j)
                            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bfr[j], af[i], acc[i][j], 0, 0, 0);
                }
                slot = slot + 1 == STG ? 0 : slot + 1;
                islot = islot + 1 == STG ? 0 : islot + 1;
            }
            pb ^= 1;
        }
        // the next tile's patch and first weight tile must have landed before its step 0 (its second weight tile stays in flight);
        // the epilogue works in the patch buffer the next tile does NOT use
        if (has_next) { YH_VMCNT(NB); } else { YH_VMCNT(0); }
        uint16_t* sC = reinterpret_cast<uint16_t*>(smem + (pb ^ 1) * PATCH_BYTES);
        constexpr int CPR = BN / 8;
        constexpr int NCH = 128 * CPR / NT;
        static_assert((128 * CPR) % NT == 0, "chunks must divide over the threads");
#pragma unroll
        for (int ph = 0; ph < 2; ++ph) {
            YH_LDS_BARRIER();                         // previous phase's readers done (ph 0: every wave is out of the k loop)
            if ((wm >> 1) == ph) {
#pragma unroll
                for (int j = 0; j < TN; ++j)
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        const int cc = wn * (TN * 32) + j * 32 + 8 * g + 4 * hsel;
                        float4 cb = make_float4(0.f, 0.f, 0.f, 0.f), cs = make_float4(1.f, 1.f, 1.f, 1.f), ct = cb;
                        if (EPI == 2) {
                            cb = *reinterpret_cast<const float4*>(sConst + cc);
                            cs = *reinterpret_cast<const float4*>(sConst + BN + cc);
                            ct = *reinterpret_cast<const float4*>(sConst + 2 * BN + cc);
                        }
#pragma unroll
                        for (int i = 0; i < TM; ++i) {
                            float v0 = acc[i][j][4 * g], v1 = acc[i][j][4 * g + 1], v2 = acc[i][j][4 * g + 2], v3 = acc[i][j][4 * g + 3];
                            if (EPI == 2) {
                                v0 = (v0 + cb.x) * cs.x + ct.x; v1 = (v1 + cb.y) * cs.y + ct.y;
                                v2 = (v2 + cb.z) * cs.z + ct.z; v3 = (v3 + cb.w) * cs.w + ct.w;
                                if (d.act == YH_ACT_SILU) { v0 = silu_fast(v0); v1 = silu_fast(v1); v2 = silu_fast(v2); v3 = silu_fast(v3); }
                            }
                            const int row = (wm & 1) * (TM * 32) + i * 32 + (lane & 31);
                            *reinterpret_cast<uint2*>(sC + row * CP + cc) = make_uint2(pack2(v0, v1), pack2(v2, v3));
                        }
                    }
            }
            YH_LDS_BARRIER();
            uint4 zpre[EPI == 3 ? NCH : 1];
            if (EPI == 3) {
#pragma unroll
                for (int i = 0; i < NCH; ++i) {
                    const int id = t + i * NT;
                    const int row = id / CPR;
                    const int n = n0 + (id - row * CPR) * 8;
                    const int orow = sPix[ph * 128 + row];
                    zpre[i] = make_uint4(0, 0, 0, 0);
                    if (orow >= 0 && n < d.N) zpre[i] = *reinterpret_cast<const uint4*>(d.bnr_z + (size_t)orow * d.bnr_ldz + n);
                }
            }
#pragma unroll
            for (int i = 0; i < NCH; ++i) {
                const int id = t + i * NT;
                const int row = id / CPR;
                const int cch = id - row * CPR;
                const int n = n0 + cch * 8;
                const int orow_i = sPix[ph * 128 + row];
                if (orow_i >= 0 && n < d.N) {
                    const size_t orow = (size_t)orow_i;
                    uint4 v = *reinterpret_cast<const uint4*>(sC + row * CP + cch * 8);
                    if (EPI == 2) {
                        uint16_t* dst;
                        const bool first = n < d.nsplit;
                        if (first) dst = d.out0 + orow * d.ld0 + n;
                        else       dst = d.out1 + orow * d.ld1 + (n - d.nsplit);
                        const bool addres = (d.res != nullptr) && first;
                        if (addres || d.accumulate) {
                            float f[8];
                            unpack8(v, f);
                            if (addres) {
                                uint4 rv = *reinterpret_cast<const uint4*>(d.res + orow * d.ldr + n);
                                float g2[8]; unpack8(rv, g2);
#pragma unroll
                                for (int e = 0; e < 8; ++e) f[e] += g2[e];
                            }
                            if (d.accumulate) {
                                uint4 ov = *reinterpret_cast<const uint4*>(dst);
                                float g2[8]; unpack8(ov, g2);
#pragma unroll
                                for (int e = 0; e < 8; ++e) f[e] += g2[e];
                            }
                            v = pack8(f);
                        }
                        *reinterpret_cast<uint4*>(dst) = v;
                    } else {
                        if (EPI == 3 && d.accumulate) {
                            const uint4 ov = *reinterpret_cast<const uint4*>(d.out0 + orow * d.ld0 + n);
                            float f[8], g0[8];
                            unpack8(v, f);
                            unpack8(ov, g0);
#pragma unroll
                            for (int e = 0; e < 8; ++e) f[e] += g0[e];
                            v = pack8(f);
                        }
                            *reinterpret_cast<uint4*>(d.out0 + orow * d.ld0 + n) = v;
                        if (EPI == 1) {
                            float f[8];
                            unpack8(v, f);
#pragma unroll
                            for (int e = 0; e < 8; ++e) { bs_[e] += f[e]; bq_[e] += f[e] * f[e]; }
                        }
                        if (EPI == 3) {
                            const uint4 zv = zpre[i];
                            float g2[8], z[8];
                            unpack8(v, g2);
                            unpack8(zv, z);
                            const float4 s0 = *reinterpret_cast<const float4*>(sConst + cch * 8);
                            const float4 s1 = *reinterpret_cast<const float4*>(sConst + cch * 8 + 4);
                            const float4 h0 = *reinterpret_cast<const float4*>(sConst + BN + cch * 8);
                            const float4 h1 = *reinterpret_cast<const float4*>(sConst + BN + cch * 8 + 4);
                            const float sc[8] = {s0.x, s0.y, s0.z, s0.w, s1.x, s1.y, s1.z, s1.w};
                            const float sh[8] = {h0.x, h0.y, h0.z, h0.w, h1.x, h1.y, h1.z, h1.w};
#pragma unroll
                            for (int e = 0; e < 8; ++e) {
                                const float a = z[e] * sc[e] + sh[e];
                                const float sg = sigmoid_fast(a);
                                const float dz = g2[e] * (sg * (1.f + a * (1.f - sg)));
                                bs_[e] += dz; bq_[e] += dz * z[e];
                            }
                        }
                    }
                }
            }
        }
        YH_LDS_BARRIER();                             // epilogue buffer and pixel table free for the next tile
#pragma unroll
        for (int i = 0; i < NPW; ++i) voffP[i] = voffN[i];
    }

    // per-block partial sums (EPI 1: statistics row of this block; EPI 3: BatchNorm-backward slab row)
    if (EPI == 1 || EPI == 3) {
        constexpr int CPR2 = BN / 8;
        YH_VMCNT(0);
        float* sRed = reinterpret_cast<float*>(smem);          // [NT][16]
        __syncthreads();
#pragma unroll
        for (int e = 0; e < 8; ++e) { sRed[t * 16 + e] = bs_[e]; sRed[t * 16 + 8 + e] = bq_[e]; }
        __syncthreads();
        for (int i = t; i < 2 * BN; i += NT) {
            const int which = i / BN, c = i - which * BN;
            float v = 0.f;
            for (int j = c / 8; j < NT; j += CPR2) v += sRed[j * 16 + which * 8 + (c & 7)];
            if (EPI == 1) put_stat(d, bx, which, n0 + c, v);
            else if (n0 + c < d.N) put_bnr(d, bx, which, n0 + c, v);
        }
    }
}

// ------------------------------------------------------------------------------------------------
#ifndef YH_H160_STG
#define YH_H160_STG 3          // weight-ring stages of conv_halo160_kernel (A/B builds: -DYH_H160_STG=4 -DYH_H160_ROWS=304)
#endif
#ifndef YH_H160_SPLIT_ISSUE
#define YH_H160_SPLIT_ISSUE 1
#endif
#ifndef YH_H160_ROWS
#define YH_H160_ROWS 328       // rows of one patch buffer
#endif
// 160-wide variant of the halo kernel for the YOLOv5x widths (N = 160, 320, ...): the 128-wide tiles above compute 256
// channels for 160.  Same structure (2-D pixel tile, patch staged once per 64-channel block and double buffered, nine taps =
// nine k-steps, 4-stage weight ring fed by LDS-DMA), but the wave tile is 64 pixels x 80 channels on
// v_mfma_f32_16x16x32_bf16 (4 x 5 tiles of 16 x 16, nine 16-byte fragment reads per twenty MFMAs) so that 4 x 2 waves cover
// 256 x 160 exactly.  A THREE-stage weight ring (60 KB) leaves 2 x 328 patch rows: the 18 x 18 patch of a 16 x 16 pixel tile fits,
// which tiles the 160 x 160 / 80 x 80 maps of YOLOv5x at 1280^2 without a ragged edge (round 5; the four-stage ring left
// 304 rows -> 23 x 10 tiles, 0.89 / 0.83 of the MFMA rows useful).
// Inference epilogues only (EPI 0 plain, EPI 2 bias / folded BN + SiLU / residual / split destination): YOLOv5x is
// not a training configuration of this build.  Operands swapped as above (D = W x X^T): a lane holds one pixel and four
// consecutive channels per accumulator.
struct H160True { static constexpr bool value = true; };
struct H160False { static constexpr bool value = false; };
typedef unsigned int h160_u32x4 __attribute__((ext_vector_type(4)));
// a 16-byte LDS read the CALLER waits for (h160_wait_frags) before the first use
template <int OFF> __device__ __forceinline__ h160_u32x4 h160_lds16(unsigned addr) {
    h160_u32x4 v;
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "n"(OFF) : "memory");
    return v;
}
// s_waitcnt lgkmcnt(0) tied to the nine fragment registers it makes valid: their uses cannot be scheduled ahead of it
__device__ __forceinline__ void h160_wait_frags(h160_u32x4 (&xf)[4], h160_u32x4 (&wf)[5]) {
    asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(xf[0]), "+v"(xf[1]), "+v"(xf[2]), "+v"(xf[3]), "+v"(wf[0]), "+v"(wf[1]), "+v"(wf[2]), "+v"(wf[3]), "+v"(wf[4]) :: "memory");
}
__device__ __forceinline__ void h160_wait_frags2(h160_u32x4 (&xf)[4], h160_u32x4 (&wf)[5], f32x4_t& last) {
    asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(xf[0]), "+v"(xf[1]), "+v"(xf[2]), "+v"(xf[3]), "+v"(wf[0]), "+v"(wf[1]), "+v"(wf[2]), "+v"(wf[3]), "+v"(wf[4]), "+v"(last) :: "memory");
}
template <int EPI, bool TL>
__global__ __launch_bounds__(512, 1) void conv_halo160_kernel(const ConvK p, const HaloGeom hg)
{
    constexpr int BN = 160, BMT = 256, WN = 2, BKT = 64, STG = YH_H160_STG;
    constexpr int NWV = 8, NT = 512;
    constexpr int TMR = 4, TNC = 5;                  // 16-pixel / 16-channel tiles per wave (64 x 80)
    constexpr int ROWB = BKT * 2, CHR = 8, RPI = 8;
    // BN / RPI = 20 weight-tile DMA instructions per k-step: 3 for waves 0..3, 2 for waves 4..7
    constexpr int PATCH_ROWS = YH_H160_ROWS;          // multiple of 8: a DMA instruction writes 8 rows
    constexpr int NPW = (PATCH_ROWS / 8 + NWV - 1) / NWV;   // patch DMA instructions per wave and channel block (one per tap step: <= 9)
    static_assert(PATCH_ROWS % 8 == 0 && NPW <= 9 && STG >= 3 && STG <= 4, "patch parts ride on the tap steps");
    constexpr int PATCH_BYTES = PATCH_ROWS * ROWB;   // 38912
    constexpr int BST_BYTES = BN * ROWB;             // 20480
    constexpr int ZERO_OFF = 2 * PATCH_BYTES + STG * BST_BYTES;
    constexpr int MAIN_BYTES = ZERO_OFF + 128;
    constexpr int CP = BN + 8;                       // epilogue buffer: 64 rows x CP bf16 inside the idle patch buffer
    static_assert(64 * CP * 2 <= PATCH_BYTES, "epilogue quarter tile must fit a patch buffer");
    static_assert(EPI == 0 || EPI == 2, "inference epilogues only");
    constexpr unsigned OOB = 0x80000000u;

    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const unsigned lds0 = (unsigned)(size_t)((__attribute__((address_space(3))) unsigned char*)smem);
    float* sConst = reinterpret_cast<float*>(smem + MAIN_BYTES);            // [3][BN]: bias | scale | shift
    int* sPix = reinterpret_cast<int*>(smem + MAIN_BYTES + 3 * BN * 4);     // [BMT] output pixel of each tile row, -1 = none

    const yh_conv_desc& d = p.d;
    const int t = threadIdx.x;
    const int lane = t & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int wm = wave / WN, wn = wave % WN;
    int bx, by;
    halo_block_map(hg, bx, by);
    const int n0 = by * BN;
    const int ncb = TL ? (p.Ctot + BKT - 1) / BKT : p.Ctot / BKT;
    const int H = d.Ho, W = d.Wo;
    const int TH = hg.TH, TW = hg.TW, PW = hg.PW, NP = hg.NP;
    const int tiles_per_img = hg.tiles_x * hg.tiles_y;
    const int ntiles = d.B * tiles_per_img;
    const int ldx2 = d.seg[0].ld * 2;
    const bool dgrad = d.mode == YH_CONV_DGRAD;
    const int nbw = wave < 4 ? 3 : 2;                // weight DMA instructions of this wave per k-step

    const __amdgpu_buffer_rsrc_t rs0 = __builtin_amdgcn_make_buffer_rsrc((void*)d.seg[0].ptr, 0, p.segbytes[0], 0x00020000);
    const __amdgpu_buffer_rsrc_t rsw = __builtin_amdgcn_make_buffer_rsrc((void*)d.w, 0, p.wbytes, 0x00020000);

    const int lrow = lane >> 3, lq = lane & 7;
    unsigned voffB[3];
#pragma unroll
    for (int j = 0; j < 3; ++j) {
        const int row = (j * NWV + wave) * RPI + lrow;             // j = 2: rows 128 .. 159 (waves 0..3 only)
        voffB[j] = (unsigned)(((n0 + row) * p.Ktot + ((lq ^ swz_f<CHR>(row)) * 8)) * 2);
    }
    // fragment geometry of v_mfma_f32_16x16x32_bf16: lane l supplies row / column (l & 15), k = 8 (l >> 4) .. +8
    const int l15 = lane & 15, kq = lane >> 4;
    int rdW[2];                                       // weight fragment offset inside a ring slot per 32-channel sub-step (+ j * 16 rows)
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
        rdW[ks] = 2 * PATCH_BYTES + (wn * 80 + l15) * ROWB + (((ks * 4 + kq) ^ swz_f<CHR>(l15)) << 4);
    // rows of a wave's channel tiles differ by multiples of 16: swz_f(row) = (row >> 1) & 7 then differs by 8 j mod 8 = 0, so the
    // lane's swizzle term is the same for every tile j

    if (t < 32) reinterpret_cast<unsigned*>(smem + ZERO_OFF)[t] = 0u;
    if (EPI == 2) {
        for (int i = t; i < 3 * BN; i += NT) {
            const int which = i / BN, c = i - which * BN;
            const float* src = which == 0 ? d.bias : (which == 1 ? d.scale : d.shift);
            sConst[i] = (src && n0 + c < d.N) ? src[n0 + c] : (which == 1 ? 1.f : 0.f);
        }
    }
    __syncthreads();

    auto patch_offsets = [&](int tile, unsigned (&vo)[NPW]) {
        const int img = tile / tiles_per_img;
        const int trem = tile - img * tiles_per_img;
        const int tyi = trem / hg.tiles_x;
        const int y0 = tyi * TH, x0 = (trem - tyi * hg.tiles_x) * TW;
#pragma unroll
        for (int i = 0; i < NPW; ++i) {
            const int pr = (i * NWV + wave) * RPI + lrow;
            const int py = pr / PW, px = pr - py * PW;
            const int y = y0 - 1 + py, x = x0 - 1 + px;
            const bool ok = pr < (TH + 2) * PW && y >= 0 && y < H && x >= 0 && x < W;
            vo[i] = ok ? (unsigned)((img * H + y) * W + x) * (unsigned)ldx2 + (unsigned)((lq ^ swz_f<CHR>(pr)) * 16) : OOB;
        }
    };
    auto issue_patch_part = [&](const unsigned (&vo)[NPW], int part, int cblk, int pbuf) -> int {
        const int inst = part * NWV + wave;
        if (inst >= NP) return 0;
        unsigned char* dst = smem + pbuf * PATCH_BYTES + inst * (RPI * ROWB);
#pragma unroll
        for (int i = 0; i < NPW; ++i)
            if (i == part) lds_dma16(rs0, dst, vo[i], cblk * (BKT * 2));
        return 1;
    };
    auto issue_W = [&](int tap, int cblk, int slot) {
        const int sw = (tap * p.Ctot + cblk * BKT) * 2;
        unsigned char* sb = smem + 2 * PATCH_BYTES + slot * BST_BYTES + wave * (RPI * ROWB);
        lds_dma16(rsw, sb, voffB[0], sw);
        lds_dma16(rsw, sb + NWV * RPI * ROWB, voffB[1], sw);
        if (wave < 4) lds_dma16(rsw, sb + 2 * NWV * RPI * ROWB, voffB[2], sw);
    };

    int tile = bx;
    unsigned voffP[NPW];             // patch offsets of the tile whose patch parts are being requested: this tile's; from its last
                                     // channel block on, the next tile's (the current ones are dead by then)
    int slot = 0, islot = STG - 1, pb = 0;
    if (tile < ntiles) {
        patch_offsets(tile, voffP);
#pragma unroll
        for (int part = 0; part < NPW; ++part) issue_patch_part(voffP, part, 0, 0);
        issue_W(0, 0, 0);
        issue_W(1, 0, 1);                             // nine taps per channel block: steps 0..STG-2 are taps 0.. of block 0
        if (STG == 4) issue_W(2, 0, 2);
        if (wave < 4) { YH_VMCNT(3); } else { YH_VMCNT(2); }   // patch + all weight tiles but the last one issued have landed
    }
    for (; tile < ntiles; tile += hg.gx) {
        const bool has_next = tile + hg.gx < ntiles;
        const int img = tile / tiles_per_img;
        const int trem = tile - img * tiles_per_img;
        const int tyi = trem / hg.tiles_x;
        const int y0 = tyi * TH, x0 = (trem - tyi * hg.tiles_x) * TW;
        int pr0[TMR];
        bool inv[TMR];
#pragma unroll
        for (int i = 0; i < TMR; ++i) {
            const int r = wm * 64 + i * 16 + l15;
            const int ty = r / TW, tx = r - ty * TW;
            inv[i] = !(ty < TH && y0 + ty < H && x0 + tx < W);
            pr0[i] = ty * PW + tx;
        }
        for (int r = t; r < BMT; r += NT) {
            const int ty = r / TW, tx = r - ty * TW;
            sPix[r] = (ty < TH && y0 + ty < H && x0 + tx < W) ? ((img * H + y0 + ty) * W + x0 + tx) : -1;
        }

        f32x4_t acc[TMR][TNC];
#pragma unroll
        for (int i = 0; i < TMR; ++i)
#pragma unroll
            for (int j = 0; j < TNC; ++j)
#pragma unroll
                for (int r = 0; r < 4; ++r) acc[i][j][r] = 0.f;

        int prev_group = 0;
        h160_u32x4 xfA[TMR], wfA[TNC], xfB[TMR], wfB[TNC];
        bool pendB = false;                           // wave-uniform: a second sub-step waits in xfB / wfB
        const bool stamp = hg.stamps != nullptr && tile == bx + hg.gx;        // wave-uniform
        unsigned long long tq0 = 0, tq1, tq2, s_wait = 0, s_issue = 0, s_mfma = 0, t_begin = 0;
        if (stamp) t_begin = __builtin_amdgcn_s_memtime();
        // one 64-channel block = nine tap steps.  TWO: the block has both 32-channel sub-steps (all but the tail block of a TL
        // kernel, whose channel count ends in a half block) — a compile-time property of the step's code: with the pipeline
        // state in run-time flags the compiler fenced every step's reads and MFMAs into separate regions
        auto do_block = [&](auto two_tag, const int cblk) {
            constexpr bool TWO = decltype(two_tag)::value;
            const bool last_blk = cblk + 1 == ncb;
            if (last_blk && has_next) patch_offsets(tile + hg.gx, voffP);
#pragma unroll
            for (int tap = 0; tap < 9; ++tap) {
                const int kt = cblk * 9 + tap;
                // all DMA groups except the one issued at the previous step must have landed (step 0: waited before the tile)
                if (stamp) tq0 = __builtin_amdgcn_s_memtime();
                if (kt > 0) YH_VMCNT_SW(prev_group);
                __builtin_amdgcn_s_barrier();
                if (stamp) { tq1 = __builtin_amdgcn_s_memtime(); s_wait += tq1 - tq0; }
                prev_group = 0;
                auto issue_step = [&]() {
                    // weight tile of step kt + STG - 1 (wrapping into the next tile's stream)
                    const int tapB = (tap + STG - 1) % 9;
                    const int cblkB = cblk + ((tap + STG - 1) >= 9 ? 1 : 0);
                    if (cblkB < ncb || has_next) {
                        issue_W(tapB, cblkB < ncb ? cblkB : 0, islot);
                        prev_group = nbw;
                    }
                    if (tap < NPW && !(YH_CONV_ABLATE & 512)) {
                        if (!last_blk) prev_group += issue_patch_part(voffP, tap, cblk + 1, pb ^ 1);
                        else if (has_next) prev_group += issue_patch_part(voffP, tap, 0, pb ^ 1);
                    }
                };
                // The two waves of a SIMD (w and w + 4) come out of the barrier together: waves 0..3 request the step's transfers
                // first, waves 4..7 after their first sub-step, so one of the pair issues MFMAs while the other one issues transfers
                // (stamps: the transfers of a step cost ~390 cycles of issue during which the SIMD ran no MFMA)
                const bool issue_first = !YH_H160_SPLIT_ISSUE || wave < 4;
                if (issue_first) issue_step();
                if (stamp) { tq2 = __builtin_amdgcn_s_memtime(); s_issue += tq2 - tq1; }
                const int kh = tap / 3, kw = tap % 3;
                int tapoff = dgrad ? (2 - kh) * PW + (2 - kw) : kh * PW + kw;            // scalar
                asm volatile("" : "+s"(tapoff));
                const int sbase = slot * BST_BYTES;
                int abase[TMR], afx[TMR];
#pragma unroll
                for (int i = 0; i < TMR; ++i) {
                    const int pr = pr0[i] + tapoff;
                    abase[i] = inv[i] ? ZERO_OFF : pb * PATCH_BYTES + pr * ROWB;
                    afx[i] = inv[i] ? -1 : swz_f<CHR>(pr);
                }
                // Software pipeline over the sub-steps (round 5; stamps showed two exposed LDS round trips per step — both waves of a
                // SIMD leave the barrier together and waited for their fragments at the same time): the fragments of this step's
                // FIRST sub-step are requested, the MFMAs of the PREVIOUS step's second sub-step (fragments already in registers: its
                // ring slot may be refilled by now) run while they arrive, then the second sub-step's fragments are requested behind
                // the first sub-step's MFMAs and wait in registers for the next step.
                // The fragment reads are inline asm, invisible to the compiler's s_waitcnt bookkeeping (it answered every read that
                // crosses a loop iteration with lgkmcnt(0) right behind it); the waits are the explicit ones below, tied to the
                // registers they make valid.
                auto read_frags = [&](int ks, h160_u32x4 (&xf)[TMR], h160_u32x4 (&wf)[TNC]) {
                    const unsigned wb = lds0 + (unsigned)(sbase + rdW[ks]);
                    wf[0] = h160_lds16<0>(wb); wf[1] = h160_lds16<16 * ROWB>(wb); wf[2] = h160_lds16<32 * ROWB>(wb);
                    wf[3] = h160_lds16<48 * ROWB>(wb); wf[4] = h160_lds16<64 * ROWB>(wb);
#pragma unroll
                    for (int i = 0; i < TMR; ++i) {
                        const int off = afx[i] < 0 ? (kq << 4) : (((ks * 4 + kq) ^ afx[i]) << 4);
                        xf[i] = h160_lds16<0>(lds0 + (unsigned)(abase[i] + off));
                    }
                };
                auto mfma_row = [&](int i, const h160_u32x4 (&xf)[TMR], const h160_u32x4 (&wf)[TNC]) {
#pragma unroll
                    for (int j = 0; j < TNC; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, wf[j]), __builtin_bit_cast(bf16x8_t, xf[i]), acc[i][j], 0, 0, 0);
                };
                read_frags(0, xfA, wfA);
                // a pending second sub-step: inside a block the previous step's (TWO), at a block's first step what the flag says
                if (tap == 0 ? pendB : TWO) {
#pragma unroll
                    for (int i = 0; i < TMR; ++i) mfma_row(i, xfB, wfB);
                }
                if (!issue_first) issue_step();
                h160_wait_frags2(xfA, wfA, acc[TMR - 1][TNC - 1]);            // arrived during the MFMAs above (a tile's first step: exposed once)
                mfma_row(0, xfA, wfA);
                if (TWO) read_frags(1, xfB, wfB);
#pragma unroll
                for (int i = 1; i < TMR; ++i) mfma_row(i, xfA, wfA);
                // the second sub-step's fragments land behind those 15 MFMAs, before the next barrier lets the slot be refilled
                if (TWO) h160_wait_frags2(xfB, wfB, acc[TMR - 1][TNC - 1]);
                if (tap == 8) pendB = TWO;
                // the second sub-step's fragments have landed (behind 20 MFMAs) before the next barrier lets the slot be refilled
                if (stamp) s_mfma += __builtin_amdgcn_s_memtime() - tq2;
                slot = slot + 1 == STG ? 0 : slot + 1;
                islot = islot + 1 == STG ? 0 : islot + 1;
            }
            pb ^= 1;
        };
        const int nfull = TL ? ncb - 1 : ncb;         // TL: the channel count ends in a half block (C % 64 == 32)
        for (int cblk = 0; cblk < nfull; ++cblk) do_block(H160True{}, cblk);
        if (TL) do_block(H160False{}, ncb - 1);
        if (pendB) {
#pragma unroll
            for (int i = 0; i < TMR; ++i)
#pragma unroll
                for (int j = 0; j < TNC; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, wfB[j]), __builtin_bit_cast(bf16x8_t, xfB[i]), acc[i][j], 0, 0, 0);
        }
        unsigned long long t_loop = 0, t_act = 0;
        if (stamp) t_loop = __builtin_amdgcn_s_memtime();
        // the next tile's patch and all its weight tiles issued so far but the last must have landed before its step 0;
        // the epilogue works in the patch buffer the next tile does NOT use
        if (has_next) { if (wave < 4) { YH_VMCNT(3); } else { YH_VMCNT(2); } } else { YH_VMCNT(0); }
        uint16_t* sC = reinterpret_cast<uint16_t*>(smem + (pb ^ 1) * PATCH_BYTES);
        constexpr int CPR = BN / 8;                   // 20 chunks of 8 channels per row
        // bias / folded BatchNorm / SiLU of the whole tile first, by all eight waves at once: inside the phases below only the two
        // waves of one pixel row-group are active, and the activation (80 values per lane, two quarter-rate transcendentals each)
        // would run on two of the four SIMDs at a time
        if (EPI == 2) {
#pragma unroll
            for (int j = 0; j < TNC; ++j) {
                const int cc = wn * 80 + j * 16 + 4 * kq;
                const float4 cb = *reinterpret_cast<const float4*>(sConst + cc);
                const float4 cs = *reinterpret_cast<const float4*>(sConst + BN + cc);
                const float4 ct = *reinterpret_cast<const float4*>(sConst + 2 * BN + cc);
#pragma unroll
                for (int i = 0; i < TMR; ++i) {
                    float v0 = (acc[i][j][0] + cb.x) * cs.x + ct.x, v1 = (acc[i][j][1] + cb.y) * cs.y + ct.y;
                    float v2 = (acc[i][j][2] + cb.z) * cs.z + ct.z, v3 = (acc[i][j][3] + cb.w) * cs.w + ct.w;
                    if (d.act == YH_ACT_SILU) { v0 = silu_fast(v0); v1 = silu_fast(v1); v2 = silu_fast(v2); v3 = silu_fast(v3); }
                    acc[i][j][0] = v0; acc[i][j][1] = v1; acc[i][j][2] = v2; acc[i][j][3] = v3;
                }
            }
        }
        if (stamp) t_act = __builtin_amdgcn_s_memtime();
#pragma unroll 1
        for (int ph = 0; ph < 4; ++ph) {
            YH_LDS_BARRIER();                         // previous phase's readers done (ph 0: every wave is out of the k loop)
            if (wm == ph) {
#pragma unroll
                for (int j = 0; j < TNC; ++j) {
                    const int cc = wn * 80 + j * 16 + 4 * kq;
#pragma unroll
                    for (int i = 0; i < TMR; ++i)
                        *reinterpret_cast<uint2*>(sC + (i * 16 + l15) * CP + cc) =
                            make_uint2(pack2(acc[i][j][0], acc[i][j][1]), pack2(acc[i][j][2], acc[i][j][3]));
                }
            }
            YH_LDS_BARRIER();
            for (int id = t; id < 64 * CPR; id += NT) {
                const int row = id / CPR;
                const int cch = id - row * CPR;
                const int n = n0 + cch * 8;
                const int orow_i = sPix[ph * 64 + row];
                if (orow_i >= 0 && n < d.N) {
                    const size_t orow = (size_t)orow_i;
                    uint4 v = *reinterpret_cast<const uint4*>(sC + row * CP + cch * 8);
                    if (EPI == 2) {
                        uint16_t* dst;
                        const bool first = n < d.nsplit;
                        if (first) dst = d.out0 + orow * d.ld0 + n;
                        else       dst = d.out1 + orow * d.ld1 + (n - d.nsplit);
                        const bool addres = (d.res != nullptr) && first;
                        if (addres || d.accumulate) {
                            float f[8];
                            unpack8(v, f);
                            if (addres) {
                                uint4 rv = *reinterpret_cast<const uint4*>(d.res + orow * d.ldr + n);
                                float g2[8]; unpack8(rv, g2);
#pragma unroll
                                for (int e = 0; e < 8; ++e) f[e] += g2[e];
                            }
                            if (d.accumulate) {
                                uint4 ov = *reinterpret_cast<const uint4*>(dst);
                                float g2[8]; unpack8(ov, g2);
#pragma unroll
                                for (int e = 0; e < 8; ++e) f[e] += g2[e];
                            }
                            v = pack8(f);
                        }
                        *reinterpret_cast<uint4*>(dst) = v;
                    } else {
                        *reinterpret_cast<uint4*>(d.out0 + orow * d.ld0 + n) = v;
                    }
                }
            }
        }
        YH_LDS_BARRIER();                             // epilogue buffer and pixel table free for the next tile
        if (stamp && lane == 0) {
            unsigned long long* o = hg.stamps + ((size_t)blockIdx.x * NWV + wave) * 8;
            const unsigned long long t_end = __builtin_amdgcn_s_memtime();
            o[0] = s_wait; o[1] = s_issue; o[2] = s_mfma; o[3] = t_loop - t_begin; o[4] = t_act - t_loop; o[5] = t_end - t_act; o[6] = t_end - t_begin; o[7] = 1;
        }
    }
}

constexpr size_t conv_halo160_smem_bytes() { return 2 * YH_H160_ROWS * 128 + YH_H160_STG * 160 * 128 + 128 + 3 * 160 * 4 + 256 * 4; }
static_assert(conv_halo160_smem_bytes() <= 160 * 1024, "LDS budget of conv_halo160_kernel");

template <int BN>
constexpr size_t conv_halo_smem_bytes() {
    return 2 * 352 * 128 + 4 * (size_t)BN * 128 + 128 + 3 * BN * 4 + 256 * 4;
}

// tile geometry of the halo kernel for an H x W map: TH x TW output pixels (<= 256) whose (TH+2) x (TW+2) patch fits 352 rows,
// chosen to waste the fewest MFMA rows (ragged tiles and TH*TW < 256)
bool conv_halo_geom_search(int H, int W, int max_rows, HaloGeom* g);
// the search below is ~200 divisions: its result per map size is kept (the planner runs for every launch).
// max_rows: patch rows one buffer holds (352 with the 64- / 128-channel weight ring, 324 next to the 160-channel one)
bool conv_halo_geom(int H, int W, HaloGeom* g, int max_rows = 352)
{
    struct Memo { int H, W, R; bool ok; HaloGeom g; };
    static thread_local Memo memo[8];
    static thread_local int next = 0;
    for (int i = 0; i < 8; ++i)
        if (memo[i].H == H && memo[i].W == W && memo[i].R == max_rows && H > 0) { *g = memo[i].g; return memo[i].ok; }
    Memo& m = memo[next];
    next = (next + 1) & 7;
    m.H = H; m.W = W; m.R = max_rows;
    m.ok = conv_halo_geom_search(H, W, max_rows, &m.g);
    *g = m.g;
    return m.ok;
}
bool conv_halo_geom_search(int H, int W, int max_rows, HaloGeom* g)
{
    double best = 0.0;
    bool found = false;
    for (int tw = 4; tw <= 64; ++tw) {
        if (tw > W + 3) break;
        const int th = 256 / tw;
        if (th < 1) continue;
        for (int th2 = th; th2 >= (th > 2 ? th - 2 : 1); --th2) {
            const int pw = tw + 2, ph = th2 + 2;
            if (pw * ph > max_rows) continue;
            const int tx = (W + tw - 1) / tw, ty = (H + th2 - 1) / th2;
            const double eff = (double)H * W / ((double)tx * ty * 256.0);
            if (eff > best + 1e-9) {
                best = eff; found = true;
                g->TH = th2; g->TW = tw; g->PW = pw; g->NP = (pw * ph + 7) / 8; g->tiles_x = tx; g->tiles_y = ty;
            }
        }
    }
    return found && best >= 0.6;
}

template <int BMT, int BN, int WM, int BKT, int STG>
constexpr size_t conv3_smem_bytes() {
    size_t a = (size_t)STG * (BMT + BN) * BKT * 2;
    size_t c = (size_t)BMT * (BN + 8) * 2;
    return (a > c ? a : c) + WM * 2 * BN * 4 + BMT * 4;
}

// ------------------------------------------------------------------------------------------------
// Stem ("focus") kernel: 3x3 / stride 1 / pad 1 on the 16-channel space-to-depth image, 32 output channels
// (utils/layer_tools.py:82-94 applied to models/normal/yolov5s.py's 6x6/s2 stem).  K per tap is exactly one
// v_mfma_f32_32x32x16_bf16, the whole weight matrix (9 fragments) lives in registers, and a WAVE works alone on strips
// of 32 consecutive output pixels of one image row: 9 coalesced 1-KB loads (one per tap, fragments straight from
// global memory in MFMA layout, hardware zero fill at the borders), 9 MFMAs, no LDS staging, no barriers.
// The MFMA runs with swapped operands (D = W x X^T): a lane then holds 4 consecutive CHANNELS of one pixel per
// accumulator group, lanes l and l+32 exchange halves with v_permlane32_swap into 16-byte chunks, and the strip leaves
// through a wave-private LDS strip in memory order (1 KiB of consecutive chunks per store instruction).  BatchNorm statistics are per-lane running sums over all strips of the wave, reduced once at the end.
// The layer moves 630 MB for 0.6 GFLOP/MB: it is HBM bound.
struct StemTag {};
constexpr int STEM_BAND = 8;          // image rows per XCD band of the XCD-aware strip order

// NTL = output-channel tiles of 32 (v5s 32 -> 1; v5m 48 / v5l 64 -> 2; v5x 80 -> 3, inference only: the running sums of
// three tiles do not fit the register file): the nine fragments of every tile stay in registers, a strip's nine input
// fragments are loaded once and multiplied with each tile in turn.
template <int EPI, int NTL>    // 0 plain, 1 + BatchNorm partial sums, 2 folded BN (scale, shift) + SiLU
__global__ __launch_bounds__(256, NTL == 3 ? 1 : 2) void conv_stem_kernel(const ConvK p)
{
    constexpr unsigned OOB = 0x80000000u;
    static_assert(EPI != 1 || NTL <= 2, "statistics: at most two channel tiles");
    __shared__ float sRed[EPI == 1 ? 4 : 1][2][32 * NTL];
    __shared__ __attribute__((aligned(16))) float sConst[2][32 * NTL];           // EPI 2: scale | shift
    // the two-tile training form (64 channels: whole 128-byte rows, 254 registers) keeps its direct stores: staged it spills and gains nothing
    constexpr bool STAGED = !(EPI == 1 && NTL == 2);
    constexpr int STEM_SP = 32 * NTL + 8;                                        // pitch (elements) of a staged output row
    __shared__ __attribute__((aligned(16))) uint16_t sOut[STAGED ? 4 * 32 * STEM_SP : 8];      // [wave][32 pixels][STEM_SP]
    const yh_conv_desc& d = p.d;
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int r = lane & 31, h = lane >> 5;
    const int W = d.Wo, H = d.Ho;
    const int ldx = d.seg[0].ld * 2;                  // bytes per input pixel
    const __amdgpu_buffer_rsrc_t rsx = __builtin_amdgcn_make_buffer_rsrc((void*)d.seg[0].ptr, 0, p.segbytes[0], 0x00020000);

    // weights: fragment of tap t for this lane = W[n = 32 nt + r][t*16 + 8h .. +8] (rows past N are zero in the packed image)
    bf16x8_t wf[NTL][9];
#pragma unroll
    for (int nt = 0; nt < NTL; ++nt)
#pragma unroll
        for (int tp = 0; tp < 9; ++tp)
            wf[nt][tp] = __builtin_bit_cast(bf16x8_t, *reinterpret_cast<const uint4*>(d.w + (size_t)(nt * 32 + r) * p.Ktot + tp * 16 + 8 * h));

    if (EPI == 2) {
        for (int i = t; i < 2 * 32 * NTL; i += 256) {
            const int which = i / (32 * NTL), c = i - which * (32 * NTL);
            sConst[which][c] = c < d.N ? (which == 0 ? d.scale[c] : d.shift[c]) : 0.f;
        }
        __syncthreads();
    }
    // channels of this lane after the MFMA: tile nt, group g (0..3) -> 32 nt + 8g + 4h + (0..3)
    float ssum[EPI == 1 ? NTL : 1][16], ssq[EPI == 1 ? NTL : 1][16];
    if (EPI == 1) {
#pragma unroll
        for (int nt = 0; nt < NTL; ++nt)
#pragma unroll
            for (int e = 0; e < 16; ++e) { ssum[nt][e] = 0.f; ssq[nt][e] = 0.f; }
    }

    const int nstrip = p.M >> 5;                      // W % 32 == 0: strips never cross an image row
    const int wstride = gridDim.x * 4;
    const int spr = W >> 5;                           // strips per row

    u32x4_t xa[2][9];
    auto issue = [&](int s, int set) __attribute__((always_inline)) {
        const int rowid = s / spr;                    // img*H + ho
        const int wo0 = (s - rowid * spr) << 5;
        const int ho = rowid % H;
        const unsigned base = (unsigned)((rowid * W + wo0 + r) * ldx + h * 16);
        const bool lok = (wo0 + r) > 0, rok = (wo0 + r) < W - 1;
#pragma unroll
        for (int kh = 0; kh < 3; ++kh) {
            const bool row_ok = (ho + kh - 1) >= 0 && (ho + kh - 1) < H;        // wave-uniform
#pragma unroll
            for (int kw = 0; kw < 3; ++kw) {
                const bool ok = row_ok && (kw == 0 ? lok : (kw == 2 ? rok : true));
                const unsigned vo = base + (unsigned)(((kh - 1) * W + (kw - 1)) * ldx);
                if (set == 0) xa[0][kh * 3 + kw] = __builtin_amdgcn_raw_buffer_load_b128(rsx, ok ? vo : OOB, 0, 0);
                else          xa[1][kh * 3 + kw] = __builtin_amdgcn_raw_buffer_load_b128(rsx, ok ? vo : OOB, 0, 0);
            }
        }
    };
    auto compute = [&](int s, int set) __attribute__((always_inline)) {
#pragma unroll
        for (int nt = 0; nt < NTL; ++nt) {
            f32x16_t acc;
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[e] = 0.f;
#pragma unroll
            for (int tp = 0; tp < 9; ++tp)
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[nt][tp], __builtin_bit_cast(bf16x8_t, set == 0 ? xa[0][tp] : xa[1][tp]), acc, 0, 0, 0);
            // acc[g*4 + e]: pixel r, channel 32 nt + 8g + 4h + e
            uint32_t pk[4][2];
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                float v[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = acc[g * 4 + e];
                if (EPI == 2) {
                    const float4 sc = *reinterpret_cast<const float4*>(&sConst[0][nt * 32 + 8 * g + 4 * h]);
                    const float4 sh = *reinterpret_cast<const float4*>(&sConst[1][nt * 32 + 8 * g + 4 * h]);
                    v[0] = v[0] * sc.x + sh.x; v[1] = v[1] * sc.y + sh.y; v[2] = v[2] * sc.z + sh.z; v[3] = v[3] * sc.w + sh.w;
                    if (d.act == YH_ACT_SILU) {
#pragma unroll
                        for (int e = 0; e < 4; ++e) v[e] = silu_fast(v[e]);
                    }
                }
                pk[g][0] = pack2(v[0], v[1]);
                pk[g][1] = pack2(v[2], v[3]);
                if (EPI == 1) {                           // statistics of the stored (bf16-rounded) values
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const float q = __uint_as_float(e & 1 ? (pk[g][e >> 1] & 0xffff0000u) : (pk[g][e >> 1] << 16));
                        ssum[EPI == 1 ? nt : 0][g * 4 + e] += q; ssq[EPI == 1 ? nt : 0][g * 4 + e] += q * q;
                    }
                }
            }
            // lanes l / l+32: lower half keeps its channels 8g..8g+3 of g = 0, 2 and receives 8g+4..8g+7 from the upper
            // half; the upper half ends up with the 16-byte chunks of g = 1, 3
            uint4 c01, c23;
            {
                auto a0 = __builtin_amdgcn_permlane32_swap(pk[0][0], pk[1][0], false, false);
                auto a1 = __builtin_amdgcn_permlane32_swap(pk[0][1], pk[1][1], false, false);
                auto b0 = __builtin_amdgcn_permlane32_swap(pk[2][0], pk[3][0], false, false);
                auto b1 = __builtin_amdgcn_permlane32_swap(pk[2][1], pk[3][1], false, false);
                c01 = make_uint4(a0[0], a1[0], a0[1], a1[1]);
                c23 = make_uint4(b0[0], b1[0], b0[1], b1[1]);
            }
            // through a wave-private LDS strip [32 pixels][N channels]: stored from the registers, an instruction writes 32 bytes of
            // every pixel row (2.3 TB/s of writes measured: the layer is all writes); read back in memory order, an instruction
            // writes 1 KiB of consecutive 16-byte chunks
            if constexpr (STAGED) {
                uint16_t* srow = sOut + (wave * 32 + r) * STEM_SP + nt * 32 + 8 * h;
                *reinterpret_cast<uint4*>(srow) = c01;                             // channels 32nt +  0..7  (h = 0) /  8..15 (h = 1)
                *reinterpret_cast<uint4*>(srow + 16) = c23;                        // channels 32nt + 16..23 (h = 0) / 24..31 (h = 1)
            } else {
                uint16_t* dst = d.out0 + ((size_t)s * 32 + r) * d.ld0 + nt * 32 + 8 * h;
                *reinterpret_cast<uint4*>(dst) = c01;
                if (nt * 32 + 16 < d.N) *reinterpret_cast<uint4*>(dst + 16) = c23;
            }
        }
        if constexpr (!STAGED) return;
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                         // LDS executes a wave's accesses in order
        const int cpp = d.N >> 3;                                                  // 16-byte chunks per pixel
        const size_t m0 = (size_t)s * 32;
#pragma unroll
        for (int it = 0; it < (32 * 32 * NTL / 8 + 63) / 64; ++it) {
            const int q = it * 64 + lane;
            const int px = q / cpp, ck = q - px * cpp;
            if (px < 32)
                *reinterpret_cast<uint4*>(d.out0 + (m0 + px) * d.ld0 + ck * 8) = *reinterpret_cast<const uint4*>(sOut + (wave * 32 + px) * STEM_SP + ck * 8);
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                         // read before the next strip overwrites it
    };

    // Strip order.  Workgroups are dealt round-robin to the 8 XCDs by id, and a strip needs the input rows above and below its
    // own: with strips simply dealt to waves in order, vertically adjacent strips land on different XCDs and every XCD fetches
    // every input row (PMC: 3.2 - 4.2x the input).  With p.xgx set (whole bands of STEM_BAND rows per XCD: host check) XCD x works
    // through the bands x, x + 8, ... — its waves walk a band row by row together, so the halo rows are L2 hits.
    auto strip_of = [&](int u) __attribute__((always_inline)) {      // u: position in this wave's XCD-local (or global) order
        if (!p.xgx) return u;
        const int per_band = STEM_BAND * spr;
        const int bl = u / per_band, within = u - bl * per_band;
        return ((bl << 3) + ((int)blockIdx.x & 7)) * per_band + within;
    };
    const int ustride = p.xgx ? (int)(gridDim.x >> 3) * 4 : wstride;
    const int ulim = p.xgx ? nstrip >> 3 : nstrip;
    int u = p.xgx ? (int)(blockIdx.x >> 3) * 4 + wave : (int)blockIdx.x * 4 + wave;
    if (u < ulim) issue(strip_of(u), 0);
    for (; u < ulim; u += 2 * ustride) {
        const int u1 = u + ustride, u2 = u + 2 * ustride;
        if (u1 < ulim) issue(strip_of(u1), 1);
        compute(strip_of(u), 0);
        if (u1 < ulim) {
            if (u2 < ulim) issue(strip_of(u2), 0);
            compute(strip_of(u1), 1);
        }
    }

    if (EPI == 1) {
        // lane holds 16 channels x its pixel column per tile: sum over the 32 pixel lanes of each half, then over the waves
#pragma unroll
        for (int nt = 0; nt < (EPI == 1 ? NTL : 1); ++nt)
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                float a = ssum[nt][e], q = ssq[nt][e];
#pragma unroll
                for (int o = 16; o > 0; o >>= 1) { a += __shfl_xor(a, o, 64); q += __shfl_xor(q, o, 64); }
                if (r == 0) { const int ch = nt * 32 + 8 * (e >> 2) + 4 * h + (e & 3); sRed[wave][0][ch] = a; sRed[wave][1][ch] = q; }
            }
        __syncthreads();
        if (t < 64 * NTL) {
            const int which = t / (32 * NTL), ch = t - which * (32 * NTL);
            const float v = sRed[0][which][ch] + sRed[1][which][ch] + sRed[2][which][ch] + sRed[3][which][ch];
            if (ch < d.Npad) put_stat(d, blockIdx.x, which, ch, v);
        }
    }
}

// YH_CONV_DBG (kernel-selection switches for A/B timing) is read once per process: the planner runs for every launch
int conv_dbg_mask() {
    static const int mask = [] { const char* e = getenv("YH_CONV_DBG"); return e ? atoi(e) : 0; }();
    return mask;
}
bool stem_eligible(const yh_conv_desc* d)
{
    if (d->mode != YH_CONV_FWD || d->nseg != 1 || d->seg[0].C != 16 || d->seg[0].ups) return false;
    if (d->KH != 3 || d->KW != 3 || d->stride != 1 || d->pad != 1) return false;
    if (d->N < 32 || d->N > 96 || d->N % 16 || d->Npad < ((d->N + 31) / 32) * 32) return false;     // 1..3 channel tiles of 32
    if (d->stats && d->N > 64) return false;                                                        // statistics: at most two tiles
    if (d->Wo % 32 || d->bias || d->res || d->accumulate || d->nsplit < d->N) return false;
    if ((d->scale == nullptr) != (d->shift == nullptr)) return false;
    if (d->stats && d->scale) return false;
    if (!d->scale && d->act != YH_ACT_NONE) return false;
    const unsigned long npix = (unsigned long)d->B * d->Hi * d->Wi;
    if (((npix - 1) * d->seg[0].ld + 16) * 2 >= (1ul << 31) || (long)d->B * d->Ho * d->Wo >= (1L << 31) - 64) return false;
    if (conv_dbg_mask() & 256) return false;
    return true;
}
constexpr int STEM_BLOCKS = 256 * 2;

template <int BN, int WM, int WN, int BKT = 32>
constexpr size_t conv_smem_bytes() {
    size_t a = 2 * (BM + BN) * (BKT + 8) * 2;
    size_t c = BM * (BN + 8) * 2;
    return (a > c ? a : c) + WM * 2 * BN * 4 + BM * 4;
}

// can this descriptor run on the buffer-load kernel (conv_v2_kernel)?  Shared by the grid planner and the launcher.
// `rebased`: the caller re-bases its input descriptors at every tile of <= 256 output rows (conv_v3_kernel), so only the images
// one tile spans have to fit the 2 GiB a descriptor addresses; otherwise the whole segment has to.
static bool conv_buf_ok(const yh_conv_desc* d, bool rebased)
{
    if (d->nseg < 1 || d->nseg > 2) return false;
    if (conv_dbg_mask() & 16) return false;
    if ((long)d->B * d->Hi * d->Wi >= (1L << 31) || (long)d->B * d->Ho * d->Wo >= (1L << 31)) return false;
    int Ctot = 0;
    for (int s2 = 0; s2 < d->nseg; ++s2) {
        const yh_seg& g = d->seg[s2];
        const unsigned long npix = (unsigned long)d->B * (d->Hi >> g.ups) * (d->Wi >> g.ups);
        if (!rebased) {
            if (((npix - 1) * g.ld + g.C) * 2 >= (1ul << 31)) return false;  // buffer descriptors address < 2 GiB
        } else {
            // a tile of 256 rows starts in image im0 and ends at most 256 / (rows per image) + 1 images later; the rows per
            // image are Ho*Wo, or a quarter of that for the parity classes of a stride-2 data gradient
            const unsigned long rows_img = (unsigned long)(d->Ho / 2 > 0 ? d->Ho / 2 : 1) * (d->Wo / 2 > 0 ? d->Wo / 2 : 1);
            const unsigned long span = 256 / rows_img + 2;
            const unsigned long pimg = (unsigned long)(d->Hi >> g.ups) * (d->Wi >> g.ups) * g.ld * 2;
            if (span * pimg + 256ul * g.ld * 2 >= (1ul << 31)) return false;
        }
        Ctot += g.C;
    }
    if ((unsigned long)d->Npad * d->KH * d->KW * Ctot * 2 >= (1ul << 31)) return false;
    const bool generic = d->bias || d->scale || d->shift || d->act != YH_ACT_NONE || d->res || d->accumulate || d->nsplit < d->N;
    if (generic && d->stats) return false;                                // statistics of an affine / activated output
    return true;
}
bool conv_v2_ok(const yh_conv_desc* d) { return conv_buf_ok(d, false); }

// output-channel tile (a 96-wide tile for the v5m / v5x widths was measured: +2 % on v5m training, -3 % on v5x inference)
int pick_bn(int N) { return N <= 32 ? 32 : (N <= 64 ? 64 : 128); }

// channels per k-step: 64 (128-byte tile rows: whole cache lines per row, half the barriers) when every input
// segment has a multiple of 64 channels (128-wide output tiles only), else 32
int pick_bkt(const yh_conv_desc* d, int bn) {
    if (bn != 128) return 32;              // measured: the narrower tiles lose more from the lower residency than they gain
    if (d->nseg < 1 || d->nseg > 2) return 32;
    for (int s = 0; s < d->nseg; ++s) if (d->seg[s].C % 64) return 32;
    if (d->tile_k == 32) return 32;
    if (conv_dbg_mask() & 64) return 32;
    return 64;
}

int conv_v3_bkt(const yh_conv_desc* d);
// LDS-DMA kernel (conv_v3_kernel) variant for this descriptor: 0 = none (v2 / generic kernel), 1 = 256 x 128 tile (8 waves),
// 2 = 128 x 128 (4 waves), 3 = 128 x 64 (4 waves), 4 = 256 x 256 (8 waves of 128 x 64: half the LDS-DMA bytes per MFMA of variant 1,
// whose 48 KB per k-step need 48 B/clk of the CU's 64 B/clk intake at the full MFMA rate).
// d->algo: 0 library default, 1 force v2, 2..4 = variant 1..3 when eligible, 14 = variant 4 when eligible.
int conv_v3_variant(const yh_conv_desc* d)
{
    if (d->algo == 1 || d->algo == 5 || d->algo == 6 || stem_eligible(d) || !conv_buf_ok(d, true)) return 0;
    if (conv_dbg_mask() & 512) return 0;
    if (conv_v3_bkt(d) == 0) return 0;
    if (d->N <= 32) return 0;
    if (d->tile_n == 32) return 0;
    const bool generic = d->bias || d->scale || d->shift || d->act != YH_ACT_NONE || d->res || d->accumulate || d->nsplit < d->N;
    if (generic && d->stats) return 0;
    if (d->algo >= 2 && d->algo <= 4) {
        const int v = d->algo - 1;
        return v;
    }
    if (d->algo == 14) {
        // 256 x 256 tile (variant 4): whole 64-channel blocks in every segment, N a multiple of 256, 64-channel k-steps; else the default
        bool ok = conv_v3_bkt(d) == 64 && d->tile_k != 32 && d->N % 256 == 0;
        for (int s = 0; s < d->nseg; ++s) ok = ok && d->seg[s].C % 64 == 0;
        if (ok) return 4;
    }
    // default: the big tile for K-heavy layers with enough pixel tiles to fill the chip, else v2
    const long M = (long)d->B * d->Ho * d->Wo;
    int Ctot = 0;
    for (int s = 0; s < d->nseg; ++s) Ctot += d->seg[s].C;
    const long K = (long)d->KH * d->KW * Ctot;
    if (d->N > 64 && K >= 512 && M >= 256L * 192) return 1;
    return 0;
}
// halo kernel (conv_halo_kernel): 3x3 / stride 1 / pad 1, one input segment with >= 64 channels in a multiple of 16, N > 32.
// d->algo: 5 forces it when eligible; 0 (library default) takes it when the tile geometry wastes < 25 % of the MFMA rows
bool conv_halo_ok(const yh_conv_desc* d, HaloGeom* g)
{
    if (d->algo != 0 && d->algo != 5) return false;
    if (conv_dbg_mask() & 1024) return false;
    if (d->KH != 3 || d->KW != 3 || d->stride != 1 || d->pad != 1 || d->nseg != 1 || d->seg[0].ups) return false;
    if (d->seg[0].C % 16 || d->seg[0].C < 64 || d->N <= 32 || d->tile_n == 32) return false;
    if (d->Ho != d->Hi || d->Wo != d->Wi) return false;
    if (!conv_v2_ok(d)) return false;
    const bool generic = d->bias || d->scale || d->shift || d->act != YH_ACT_NONE || d->res || d->accumulate || d->nsplit < d->N;
    if (generic && d->stats) return false;
    if ((long)d->B * d->Ho * d->Wo >= (1L << 31)) return false;
    HaloGeom gg;
    if (!conv_halo_geom(d->Ho, d->Wo, &gg)) return false;
    if (d->algo == 0) {
        const double eff = (double)d->Ho * d->Wo / ((double)gg.tiles_x * gg.tiles_y * 256.0);
        if (eff < 0.75 || (long)d->B * gg.tiles_x * gg.tiles_y < 128) return false;
    }
    if (g) *g = gg;
    return true;
}

// channels per k-step of the LDS-DMA kernel: 0 = not eligible.  64 needs whole 64-channel blocks in every segment but the
// last, whose channel count may be any multiple of 16 (the "tail" variant of the kernel); 32 needs multiples of 32 everywhere.
// Where both work, 64 is the default (half the barriers) and d->tile_k == 32 selects the short steps.
// 160-wide halo kernel (conv_halo160_kernel): only on request (d->algo == 6; the engine times it where it is eligible):
// 3x3 / stride 1 / pad 1, one segment with >= 64 channels in a multiple of 32, N a multiple of 160, inference epilogues
bool conv_halo160_ok(const yh_conv_desc* d, HaloGeom* g)
{
    if (d->algo != 6 || stem_eligible(d)) return false;
    if (d->KH != 3 || d->KW != 3 || d->stride != 1 || d->pad != 1 || d->nseg != 1 || d->seg[0].ups) return false;
    if (d->seg[0].C % 32 || d->seg[0].C < 64 || d->N % 160 || d->stats || d->bnr_part) return false;
    if (d->Ho != d->Hi || d->Wo != d->Wi) return false;
    if (!conv_v2_ok(d)) return false;
    if ((long)d->B * d->Ho * d->Wo >= (1L << 31)) return false;
    HaloGeom gg;
    if (!conv_halo_geom(d->Ho, d->Wo, &gg, YH_H160_ROWS == 328 ? 324 : YH_H160_ROWS)) return false;
    if (g) *g = gg;
    return true;
}

int conv_v3_bkt(const yh_conv_desc* d) {
    bool ok64 = true, ok32 = true;
    int Ctot = 0;
    for (int s = 0; s < d->nseg; ++s) {
        const int C = d->seg[s].C;
        Ctot += C;
        if (C % 32) ok32 = false;
        if (s + 1 < d->nseg ? (C % 64 != 0) : (C % 16 != 0)) ok64 = false;
    }
    if (Ctot < 64) ok64 = false;
    if (ok64 && !(ok32 && d->tile_k == 32)) return 64;
    return ok32 ? 32 : 0;
}

void conv_grid(const yh_conv_desc* d, int* gx, int* gy, int* bn) {
    long M = (long)d->B * d->Ho * d->Wo;
    if (stem_eligible(d)) {
        const long blocks = (M / 32 + 3) / 4;
        *gx = (int)(blocks < STEM_BLOCKS ? blocks : STEM_BLOCKS); *gy = 1; *bn = 32;
        return;
    }
    HaloGeom hgm;
    if (conv_halo160_ok(d, &hgm)) {
        const int nt = d->N / 160;
        const long ntiles = (long)d->B * hgm.tiles_x * hgm.tiles_y;
        int cap = 256 / nt;
        if (cap < 1) cap = 1;
        if (d->grid_cap > 0) cap = d->grid_cap;
        *gx = (int)(ntiles < cap ? ntiles : cap); *gy = nt; *bn = 160;
        return;
    }
    if (!stem_eligible(d) && conv_halo_ok(d, &hgm)) {
        const int b = d->N <= 64 ? 64 : 128;
        const int nt = (d->N + b - 1) / b;
        const long ntiles = (long)d->B * hgm.tiles_x * hgm.tiles_y;
        int cap = 256 / nt;
        if (cap < 1) cap = 1;
        if (d->grid_cap > 0) cap = d->grid_cap;
        *gx = (int)(ntiles < cap ? ntiles : cap); *gy = nt; *bn = b;
        return;
    }
    if (const int v3 = conv_v3_variant(d)) {
        const int bmt = (v3 == 1 || v3 == 4) ? 256 : 128;
        const int b = v3 == 3 ? 64 : (v3 == 4 ? 256 : 128);
        const int nt = (d->N + b - 1) / b;
        const int bkt = conv_v3_bkt(d);
        const int occ = (v3 == 1 || v3 == 4) ? 1 : (v3 == 2 ? 2 : (bkt == 64 ? 2 : 3));
        const bool cls = d->mode == YH_CONV_DGRAD && d->stride == 2 && d->Ho % 2 == 0 && d->Wo % 2 == 0 && d->KH >= 2 && d->KW >= 2 && !d->stats;
        const long Mc = cls ? M / 4 : M;
        const int mt = (int)((Mc + bmt - 1) / bmt);
        int cap = (256 * occ) / (nt * (cls ? d->KH * d->KW : 1));
        if (cap < 1) cap = 1;
        if (d->grid_cap > 0) cap = d->grid_cap;
        *gx = mt < cap ? mt : cap; *gy = nt; *bn = b;
        return;
    }
    int mtiles = (int)((M + BM - 1) / BM);
    int b = (d->tile_n == 32 || d->tile_n == 64 || d->tile_n == 128) ? d->tile_n : pick_bn(d->N);
    int nt = (d->N + b - 1) / b;
    // persistent blocks: exactly one resident wave of blocks (256 CUs x blocks/CU of this instantiation), so there is
    // no partially filled second round; also bounds the BatchNorm partial-sum rows the finalize kernel reduces
    const int occ = pick_bkt(d, b) == 64 ? (b == 32 ? 3 : 2) : (b == 32 ? 4 : (b == 64 ? 3 : 2));
    int cap = (256 * occ) / nt;
    cap = (cap / 8) * 8;
    if (cap < 8) cap = 8;
    if (d->grid_cap > 0) cap = d->grid_cap;
    int g = mtiles < cap ? mtiles : cap;
    *gx = g; *gy = nt; *bn = b;
}

// Blocks per (parity class, slot) of a stride-2 data gradient on the register-staged / generic kernels.  The nine (class, slot)
// blocks with the same blockIdx.x walk the same pixel regions at the same pace and read the same gy rows (each gy pixel feeds
// nine taps spread over the four classes): with a multiple of 8 blocks per slot they share an XCD (workgroups go round-robin
// to the 8 XCDs by linear id) and the rows are fetched from HBM once per XCD-L2 instead of once per class (measured on the
// YOLOv5s stage-1 / stage-2 layers: +10 % / +7 %).  Rounded DOWN so that the launch still fits one resident wave of blocks.
int cls_inner_blocks(int gx) { return gx >= 32 ? gx / 32 * 32 : gx; }          // whole groups of 4 blocks x 8 XCDs (conv_v2_kernel)
constexpr int CLS_INNER_MIN_TILES = 256;          // conv_v2_kernel: regions per class from which a block walks the four classes itself
int cls_blocks_per_slot(int gx_total, int zslots, long mtiles_cls)
{
    int g = (gx_total + zslots - 1) / zslots;
    if ((g % 8) * 8 <= g) g = (g / 8) * 8;          // at most an eighth of the blocks given up for it
    if (g > mtiles_cls) g = (int)mtiles_cls;
    return g < 1 ? 1 : g;
}

}  // namespace

// the planning helpers are called on half-filled descriptors (sizing, tuning): answer 0 instead of dividing by a zero dimension
static bool conv_desc_plannable(const yh_conv_desc* d) {
    return d && (d->nseg == 1 || d->nseg == 2) && d->B > 0 && d->Ho > 0 && d->Wo > 0 && d->Hi > 0 && d->Wi > 0 && d->KH > 0 && d->KW > 0 &&
           (d->stride == 1 || d->stride == 2) && d->N > 0 && d->seg[0].C > 0;
}
extern "C" int yh_conv_stat_blocks(const yh_conv_desc* d) {
    if (!conv_desc_plannable(d)) return 0;
    if (d->algo == 8) { const int r8 = yh_p3_rows(d); if (r8 > 0) return r8; }
    if (d->algo == 13) { const int r13 = yh_pt_rows(d); if (r13 > 0) return r13; }
    int gx, gy, bn;
    conv_grid(d, &gx, &gy, &bn);
    return gx;
}

namespace {
// validates, plans and (name_out == nullptr) launches; with name_out only the instantiation's name is produced
int conv_run(const yh_conv_desc* d, yh_stream stream, char* name_out, int name_len)
{
    YH_CHECK_ARG(d != nullptr, "yh_conv_igemm: null desc");
    YH_CHECK_ARG(d->nseg == 1 || d->nseg == 2, "yh_conv_igemm: nseg must be 1 or 2 (got %d)", d->nseg);
    YH_CHECK_ARG(d->mode == YH_CONV_FWD || d->mode == YH_CONV_DGRAD, "yh_conv_igemm: bad mode");
    YH_CHECK_ARG(d->stride == 1 || d->stride == 2, "yh_conv_igemm: stride must be 1 or 2");
    YH_CHECK_ARG(d->B > 0 && d->Ho > 0 && d->Wo > 0 && d->Hi > 0 && d->Wi > 0, "yh_conv_igemm: bad dims");
    YH_CHECK_ARG(d->KH > 0 && d->KW > 0 && d->KH <= 7 && d->KW <= 7, "yh_conv_igemm: bad kernel size");
    int Ctot = 0;
    for (int s = 0; s < d->nseg; ++s) {
        const yh_seg& g = d->seg[s];
        YH_CHECK_ARG(g.ptr && yh_aligned16(g.ptr), "yh_conv_igemm: seg %d pointer null/unaligned", s);
        YH_CHECK_ARG(g.C > 0 && g.C % 8 == 0 && g.ld % 8 == 0 && g.ld >= g.C, "yh_conv_igemm: seg %d C=%d ld=%d must be multiples of 8", s, g.C, g.ld);
        YH_CHECK_ARG(g.ups == 0 || g.ups == 1, "yh_conv_igemm: seg %d bad ups", s);
        if (g.ups) YH_CHECK_ARG(d->Hi % 2 == 0 && d->Wi % 2 == 0, "yh_conv_igemm: upsampled segment needs even Hi/Wi");
        Ctot += g.C;
    }
    YH_CHECK_ARG(d->w && yh_aligned16(d->w), "yh_conv_igemm: weights null/unaligned");
    YH_CHECK_ARG(d->N > 0 && d->Npad >= d->N && d->Npad % 128 == 0, "yh_conv_igemm: N=%d Npad=%d (Npad must be a multiple of 128)", d->N, d->Npad);
    YH_CHECK_ARG(d->out0 && yh_aligned16(d->out0) && d->ld0 % 8 == 0, "yh_conv_igemm: out0 null/unaligned");
    YH_CHECK_ARG(d->nsplit > 0 && d->nsplit % 8 == 0 || d->nsplit >= d->N, "yh_conv_igemm: nsplit must be a multiple of 8");
    if (d->nsplit < d->N) YH_CHECK_ARG(d->out1 && yh_aligned16(d->out1) && d->ld1 % 8 == 0, "yh_conv_igemm: out1 null/unaligned");
    if (d->res) YH_CHECK_ARG(yh_aligned16(d->res) && d->ldr % 8 == 0, "yh_conv_igemm: res unaligned");
    if (d->mode == YH_CONV_FWD) {
        YH_CHECK_ARG((d->Hi + 2 * d->pad - d->KH) / d->stride + 1 == d->Ho && (d->Wi + 2 * d->pad - d->KW) / d->stride + 1 == d->Wo,
                     "yh_conv_igemm: fwd geometry mismatch Hi=%d Ho=%d k=%d s=%d p=%d", d->Hi, d->Ho, d->KH, d->stride, d->pad);
    } else {
        YH_CHECK_ARG((d->Ho + 2 * d->pad - d->KH) / d->stride + 1 == d->Hi && (d->Wo + 2 * d->pad - d->KW) / d->stride + 1 == d->Wi,
                     "yh_conv_igemm: dgrad geometry mismatch Ho=%d Hi=%d k=%d s=%d p=%d", d->Ho, d->Hi, d->KH, d->stride, d->pad);
    }
    long M = (long)d->B * d->Ho * d->Wo;
    YH_CHECK_ARG(M < (1L << 31) - BM, "yh_conv_igemm: too many output pixels");
    if (d->algo >= 7 && d->algo <= 13) {      // stride-2 data-gradient kernel (conv_dg2.hip) / 3x3 patch kernel (conv_p3.hip) / 80-channel halo kernel (conv_h80.hip) /
                                              // pointwise kernel (conv_pw.hip) where eligible, else the library default
        if (d->algo == 7 && yh_dg2_rows(d) > 0) return yh_dg2_run(d, stream, name_out, name_len);
        if (d->algo == 8 && yh_p3_rows(d) > 0) return yh_p3_run(d, stream, name_out, name_len);
        if (d->algo == 9 && yh_h80_rows(d) > 0) return yh_h80_run(d, stream, name_out, name_len);
        if (d->algo == 10 && yh_pw_rows(d) > 0) return yh_pw_run(d, stream, name_out, name_len);
        if (d->algo == 12 && yh_c80_rows(d) > 0) return yh_c80_run(d, stream, name_out, name_len);
        if (d->algo == 13 && yh_pt_rows(d) > 0) return yh_pt_run(d, stream, name_out, name_len);
        yh_conv_desc d0 = *d;
        d0.algo = 0;
        return conv_run(&d0, stream, name_out, name_len);
    }

    ConvK k;
    k.d = *d;
    k.M = (int)M;
    k.Ctot = Ctot;
    k.Ktot = d->KH * d->KW * Ctot;
    k.nkt = (k.Ktot + BK - 1) / BK;
    k.mtiles = (int)((M + BM - 1) / BM);
    if (d->mode == YH_CONV_FWD) { k.sa = d->stride; k.sb = 1; k.sc = -d->pad; k.sdshift = 0; }
    else { k.sa = 1; k.sb = -1; k.sc = d->pad; k.sdshift = d->stride == 2 ? 1 : 0; }
    if (k.d.nsplit > k.d.N) k.d.nsplit = k.d.N + 8;   // everything goes to out0
    k.fast = 1;
    k.dbg = conv_dbg_mask();
    // the buffer-load kernel walks 32-channel blocks: a first segment must end on a block boundary, the last may be ragged
    if (d->nseg > 1 && d->seg[0].C % 32) k.fast = 0;
    const bool ragged = (d->seg[d->nseg - 1].C % 32) != 0;
    if ((long)d->B * d->Hi * d->Wi >= (1L << 31)) k.fast = 0;
    k.cls = 0; k.Hc = d->Ho; k.Wc = d->Wo;
    if (d->mode == YH_CONV_DGRAD && d->stride == 2 && d->Ho % 2 == 0 && d->Wo % 2 == 0 && d->KH >= 2 && d->KW >= 2 && !d->stats) {
        k.cls = 1; k.Hc = d->Ho / 2; k.Wc = d->Wo / 2;
        k.M = (int)(M / 4);
        k.mtiles = (k.M + BM - 1) / BM;
    }

    int gx, gy, bn;
    conv_grid(d, &gx, &gy, &bn);
    YH_CHECK_ARG(gy * bn <= d->Npad, "yh_conv_igemm: Npad too small for tile");
    HaloGeom hgeo;
    const bool halo160 = conv_halo160_ok(d, &hgeo);
    const bool halo = !halo160 && !stem_eligible(d) && conv_halo_ok(d, &hgeo);
    const int v3 = (halo || halo160) ? 0 : conv_v3_variant(d);
    const int zslots = k.cls ? d->KH * d->KW : 1;      // (parity class, slot) pairs: see cls_slot
    const int gx_all = gx < k.mtiles ? gx : k.mtiles;  // conv_v2_kernel walks the classes of a region inside the block: no z dimension
    if (k.cls && !v3) gx = cls_blocks_per_slot(gx, zslots, k.mtiles);
    dim3 grid(gx, gy, zslots), block(256);
    // ---- lean buffer-load kernel
    const int bkt = pick_bkt(d, bn);
    k.v2 = conv_v2_ok(d) ? 1 : 0;
    k.pointwise = (d->KH == 1 && d->KW == 1 && d->stride == 1 && d->pad == 0 && !k.cls) ? 1 : 0;
    for (int s2 = 0; s2 < 2; ++s2) {
        k.segbytes[s2] = 0; k.segbytes64[s2] = 0;
        if (s2 < d->nseg) {
            const yh_seg& g = d->seg[s2];
            if (g.ups) k.pointwise = 0;
            const unsigned long npix = (unsigned long)d->B * (d->Hi >> g.ups) * (d->Wi >> g.ups);
            const unsigned long bytes = ((npix - 1) * g.ld + g.C) * 2;
            if (bytes >= (1ul << 31)) k.v2 = 0;
            k.segbytes[s2] = bytes >= (1ul << 31) ? 0x7fffffffu : (unsigned)bytes;          // only read where k.v2 / halo hold
            k.segbytes64[s2] = bytes;
        }
    }
    if (d->nseg == 1) k.segbytes64[1] = k.segbytes64[0];
    {
        const unsigned long wb = (unsigned long)d->Npad * k.Ktot * 2;
        if (wb >= (1ul << 31)) k.v2 = 0;
        k.wbytes = (unsigned)wb;
    }
    if (d->nseg == 1) { k.d.seg[1] = k.d.seg[0]; }
    if (stem_eligible(d)) {
        const int epi = d->scale ? 2 : (d->stats ? 1 : 0);
        const int ntl = (d->N + 31) / 32;
        if (name_out) { snprintf(name_out, name_len, "conv_stem_kernel<%d, %d>", epi, ntl); return YH_OK; }
        hipStream_t sst = (hipStream_t)stream;
        const dim3 sg(gx), sb(256);
        {   // XCD-aware strip order (see the kernel): whole bands per XCD and whole workgroup octets; YH_STEM_MAP=0: strips in order
            static const int smap = getenv("YH_STEM_MAP") ? atoi(getenv("YH_STEM_MAP")) : 1;
            const long rows = (long)d->B * d->Ho;
            k.xgx = (smap && gx % 8 == 0 && rows % (8 * STEM_BAND) == 0) ? 1 : 0;
        }
#define YH_LAUNCH_STEM(NTL_)                                                            \
        do {                                                                            \
            if (epi == 2)      conv_stem_kernel<2, NTL_><<<sg, sb, 0, sst>>>(k);        \
            else if (epi == 1) conv_stem_kernel<1, NTL_ <= 2 ? NTL_ : 2><<<sg, sb, 0, sst>>>(k); \
            else               conv_stem_kernel<0, NTL_><<<sg, sb, 0, sst>>>(k);        \
        } while (0)
        if (ntl == 1) YH_LAUNCH_STEM(1); else if (ntl == 2) YH_LAUNCH_STEM(2); else YH_LAUNCH_STEM(3);
#undef YH_LAUNCH_STEM
        YH_CHECK_LAUNCH("yh_conv_igemm(stem)");
        return YH_OK;
    }
    const bool generic = d->bias || d->scale || d->shift || d->act != YH_ACT_NONE || d->res || d->accumulate || k.d.nsplit < d->N;
    if (generic && d->stats) k.v2 = 0;            // statistics of an affine/activated output: generic kernel only
    if (ragged && !k.v2) k.fast = 0;              // the generic kernel's fast loader needs whole 32-channel blocks
    if (d->bnr_part) {
        const bool generic_na = d->bias || d->scale || d->shift || d->act != YH_ACT_NONE || d->res || k.d.nsplit < d->N;
        YH_CHECK_ARG((k.v2 || v3) && !generic_na && !d->stats && d->mode == YH_CONV_DGRAD && !stem_eligible(d),
                     "yh_conv_igemm: the fused BatchNorm-backward reduction needs the plain buffer-load data-gradient path");
        YH_CHECK_ARG(d->bnr_z && yh_aligned16(d->bnr_z) && d->bnr_ldz % 8 == 0 && d->bnr_ws && d->bnr_C >= d->N && d->N % 8 == 0,
                     "yh_conv_igemm: bad fused-reduction operands");
    }
    if (halo160) {
        YH_CHECK_ARG(k.v2 && !k.cls, "yh_conv_igemm: the halo kernel needs the buffer-load path");
        const int epi = generic ? 2 : 0;
        const bool tl = (k.Ctot % 64) != 0;
        if (name_out) { snprintf(name_out, name_len, tl ? "conv_halo160_kernel<%d, true>" : "conv_halo160_kernel<%d, false>", epi); return YH_OK; }
        hipStream_t sth = (hipStream_t)stream;
        static const int rowmajor = [] { const char* e = getenv("YH_HALO_MAP"); return (e && atoi(e) == 0) ? 1 : 0; }();
        hgeo.gx = gx; hgeo.gy = gy; hgeo.rowmajor = (rowmajor || gy == 1) ? 1 : 0;
        hgeo.stamps = g_halo_stamps;
        const dim3 gridh(gx * gy), blkh(512);              // one row of workgroups: halo_block_map
        const size_t sm = conv_halo160_smem_bytes();
        static YhDevOnce attr_set;      
        if (attr_set.need()) {
            attr_set.set((const void*)conv_halo160_kernel<0, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sm);
            attr_set.set((const void*)conv_halo160_kernel<0, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sm);
            attr_set.set((const void*)conv_halo160_kernel<2, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sm);
            attr_set.set((const void*)conv_halo160_kernel<2, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sm);
            attr_set.done(); 
        }
        if (epi == 2) { if (tl) conv_halo160_kernel<2, true><<<gridh, blkh, sm, sth>>>(k, hgeo); else conv_halo160_kernel<2, false><<<gridh, blkh, sm, sth>>>(k, hgeo); }
        else          { if (tl) conv_halo160_kernel<0, true><<<gridh, blkh, sm, sth>>>(k, hgeo); else conv_halo160_kernel<0, false><<<gridh, blkh, sm, sth>>>(k, hgeo); }
        YH_CHECK_LAUNCH("yh_conv_igemm(halo160)");
        return YH_OK;
    }
    if (halo) {
        YH_CHECK_ARG(k.v2 && !k.cls, "yh_conv_igemm: the halo kernel needs the buffer-load path");
        const int epi = d->bnr_part ? 3 : (generic ? 2 : (d->stats ? 1 : 0));
        const bool tl = (k.Ctot % 64) != 0;
        if (name_out) { snprintf(name_out, name_len, tl ? "conv_halo_kernel<%d, %d, true>" : "conv_halo_kernel<%d, %d, false>", bn, epi); return YH_OK; }
        hipStream_t sth = (hipStream_t)stream;
        // measured (profiles/r04_step_experiments.txt j): +0.5 % on YOLOv5x inference, -1 % on the YOLOv5l train step (forward with
        // statistics / data gradients) -> the XCD-major order only under the inference epilogue
        static const int rowmajor = [] { const char* e = getenv("YH_HALO_MAP"); return (e && atoi(e) == 0) ? 1 : 0; }();
        hgeo.gx = gx; hgeo.gy = gy; hgeo.rowmajor = (rowmajor || gy == 1 || epi != 2) ? 1 : 0;
        hgeo.stamps = nullptr;
        const dim3 gridh(gx * gy), blkh(512);              // one row of workgroups: halo_block_map
#define YH_LAUNCH_HALO(BN_, TL_)                                                                                     \
        do {                                                                                                         \
            const size_t sm = conv_halo_smem_bytes<BN_>();                                                           \
            static YhDevOnce attr_set;                                                                                  \
            if (attr_set.need()) {                                                                                         \
                attr_set.set((const void*)conv_halo_kernel<BN_, 0, TL_>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sm); \
                attr_set.set((const void*)conv_halo_kernel<BN_, 1, TL_>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sm); \
                attr_set.set((const void*)conv_halo_kernel<BN_, 2, TL_>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sm); \
                attr_set.set((const void*)conv_halo_kernel<BN_, 3, TL_>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sm); \
                attr_set.done();                                                                                      \
            }                                                                                                        \
            if (epi == 3)      conv_halo_kernel<BN_, 3, TL_><<<gridh, blkh, sm, sth>>>(k, hgeo);                     \
            else if (epi == 2) conv_halo_kernel<BN_, 2, TL_><<<gridh, blkh, sm, sth>>>(k, hgeo);                     \
            else if (epi == 1) conv_halo_kernel<BN_, 1, TL_><<<gridh, blkh, sm, sth>>>(k, hgeo);                     \
            else               conv_halo_kernel<BN_, 0, TL_><<<gridh, blkh, sm, sth>>>(k, hgeo);                     \
        } while (0)
        if (tl) { if (bn == 64) YH_LAUNCH_HALO(64, true); else YH_LAUNCH_HALO(128, true); }
        else    { if (bn == 64) YH_LAUNCH_HALO(64, false); else YH_LAUNCH_HALO(128, false); }
#undef YH_LAUNCH_HALO
        YH_CHECK_LAUNCH("yh_conv_igemm(halo)");
        return YH_OK;
    }
    if (v3) {
        const int bkt3 = conv_v3_bkt(d);          // conv_v3_variant already checked the addressing (conv_buf_ok, re-based)
        const bool tl3 = bkt3 == 64 && (k.Ctot % 64) != 0;
        const int epi = d->bnr_part ? 3 : (generic ? 2 : (d->stats ? 1 : 0));
        const int bmt = (v3 == 1 || v3 == 4) ? 256 : 128;
        const int stg = v3 == 4 ? 2 : (v3 == 1 ? (bkt3 == 64 ? 3 : 4) : (v3 == 2 ? (bkt3 == 64 ? 2 : 4) : (bkt3 == 64 ? 3 : 4)));
        if (name_out) {
            snprintf(name_out, name_len, "conv_v3_kernel<%d, %d, %d, %d, %d, %d, %d%s>", bmt, bn, v3 == 1 ? 4 : 2, v3 == 4 ? 4 : 2, bkt3, stg, epi,
                     tl3 ? ", true" : ", false");
            return YH_OK;
        }
        hipStream_t st3 = (hipStream_t)stream;
        {
            // measured: no gain on YOLOv5x inference (707 / 712 -> 691 / 717 img/s), -2 % on the YOLOv5l train step: off (YH_HALO_MAP=2 enables it)
            static const int rowmajor3 = [] { const char* e = getenv("YH_HALO_MAP"); return (e && atoi(e) == 2) ? 0 : 1; }();
            k.xgx = k.xgy = 0;
            if (!k.cls && grid.y > 1 && grid.z == 1 && !rowmajor3) { k.xgx = (int)grid.x; k.xgy = (int)grid.y; grid = dim3(grid.x * grid.y, 1, 1); }
        }
#define YH_LAUNCH_V3W(BMT_, BN_, WM_, WN_, BKT_, STG_, TL_)                                                               \
        do {                                                                                                         \
            const size_t sm = conv3_smem_bytes<BMT_, BN_, WM_, BKT_, STG_>();                                        \
            const dim3 blk(WM_ * WN_ * 64);                                                                           \
            static YhDevOnce attr_set;                                                                                  \
            if (attr_set.need()) {                                                                                         \
                attr_set.set((const void*)conv_v3_kernel<BMT_, BN_, WM_, WN_, BKT_, STG_, 0, TL_>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sm); \
                attr_set.set((const void*)conv_v3_kernel<BMT_, BN_, WM_, WN_, BKT_, STG_, 1, TL_>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sm); \
                attr_set.set((const void*)conv_v3_kernel<BMT_, BN_, WM_, WN_, BKT_, STG_, 2, TL_>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sm); \
                attr_set.set((const void*)conv_v3_kernel<BMT_, BN_, WM_, WN_, BKT_, STG_, 3, TL_>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sm); \
                attr_set.done();                                                                                      \
            }                                                                                                        \
            if (epi == 3)      conv_v3_kernel<BMT_, BN_, WM_, WN_, BKT_, STG_, 3, TL_><<<grid, blk, sm, st3>>>(k);     \
            else if (epi == 2) conv_v3_kernel<BMT_, BN_, WM_, WN_, BKT_, STG_, 2, TL_><<<grid, blk, sm, st3>>>(k);     \
            else if (epi == 1) conv_v3_kernel<BMT_, BN_, WM_, WN_, BKT_, STG_, 1, TL_><<<grid, blk, sm, st3>>>(k);     \
            else               conv_v3_kernel<BMT_, BN_, WM_, WN_, BKT_, STG_, 0, TL_><<<grid, blk, sm, st3>>>(k);     \
        } while (0)
#define YH_LAUNCH_V3(BMT_, BN_, WM_, BKT_, STG_, TL_) YH_LAUNCH_V3W(BMT_, BN_, WM_, 2, BKT_, STG_, TL_)
        if (v3 == 4) YH_LAUNCH_V3W(256, 256, 2, 4, 64, 2, false);
        else if (v3 == 1) { if (tl3) YH_LAUNCH_V3(256, 128, 4, 64, 3, true); else if (bkt3 == 64) YH_LAUNCH_V3(256, 128, 4, 64, 3, false); else YH_LAUNCH_V3(256, 128, 4, 32, 4, false); }
        else if (v3 == 2) { if (tl3) YH_LAUNCH_V3(128, 128, 2, 64, 2, true); else if (bkt3 == 64) YH_LAUNCH_V3(128, 128, 2, 64, 2, false); else YH_LAUNCH_V3(128, 128, 2, 32, 4, false); }
        else { if (tl3) YH_LAUNCH_V3(128, 64, 2, 64, 3, true); else if (bkt3 == 64) YH_LAUNCH_V3(128, 64, 2, 64, 3, false); else YH_LAUNCH_V3(128, 64, 2, 32, 4, false); }
#undef YH_LAUNCH_V3W
#undef YH_LAUNCH_V3
        YH_CHECK_LAUNCH("yh_conv_igemm(v3)");
        return YH_OK;
    }
    if (name_out) {
        const int wm = bn == 128 ? 2 : 4, wn = bn == 128 ? 2 : 1, minw = bn == 32 ? 4 : (bn == 64 ? 3 : 2);
        if (k.v2) {
            const bool e8 = bn == 128;
            snprintf(name_out, name_len, "conv_v2_kernel<%d, %d, %d, %d, %d, %d>", bn, e8 ? 4 : wm, e8 ? 2 : wn, e8 ? 4 : minw,
                     d->bnr_part ? 3 : (generic ? 2 : (d->stats ? 1 : 0)), bkt);
        }
        else snprintf(name_out, name_len, "conv_igemm_kernel<%d, %d, %d, %s, %d>", bn, wm, wn, k.fast ? "true" : "false", minw);
        return YH_OK;
    }
    if (k.v2) {
        hipStream_t st2 = (hipStream_t)stream;
        const int epi = d->bnr_part ? 3 : (generic ? 2 : (d->stats ? 1 : 0));
        if (k.cls && k.mtiles >= CLS_INNER_MIN_TILES) grid = dim3(cls_inner_blocks(gx_all), gy, 1);
#define YH_LAUNCH_V2(BN_, WM_, WN_, MINW_, BKT_)                                                               \
        do {                                                                                                   \
            const size_t sm = conv_smem_bytes<BN_, WM_, WN_, BKT_>();                                          \
            const dim3 blk(WM_ * WN_ * 64);                                                                    \
            if (epi == 3)      conv_v2_kernel<BN_, WM_, WN_, MINW_, 3, BKT_><<<grid, blk, sm, st2>>>(k);       \
            else if (epi == 2) conv_v2_kernel<BN_, WM_, WN_, MINW_, 2, BKT_><<<grid, blk, sm, st2>>>(k);       \
            else if (epi == 1) conv_v2_kernel<BN_, WM_, WN_, MINW_, 1, BKT_><<<grid, blk, sm, st2>>>(k);       \
            else               conv_v2_kernel<BN_, WM_, WN_, MINW_, 0, BKT_><<<grid, blk, sm, st2>>>(k);       \
        } while (0)
        if (bn == 128) {                     // 8 waves: 4 per SIMD with two resident blocks, wave tile 32 x 64
            if (bkt == 64) YH_LAUNCH_V2(128, 4, 2, 4, 64);
            else           YH_LAUNCH_V2(128, 4, 2, 4, 32);
        } else if (bn == 64) {
            YH_LAUNCH_V2(64, 4, 1, 3, 32);   // (8 waves of 32 x 32 measured equal: LDS reads per MFMA double)
        } else {
            YH_LAUNCH_V2(32, 4, 1, 4, 32);
        }
#undef YH_LAUNCH_V2
        YH_CHECK_LAUNCH("yh_conv_igemm(v2)");
        return YH_OK;
    }
    hipStream_t st = (hipStream_t)stream;
#define YH_LAUNCH_CONV(BN_, WM_, WN_, MINW_)                                                                  \
    do {                                                                                                      \
        const size_t sm = conv_smem_bytes<BN_, WM_, WN_>();                                                   \
        if (k.fast) conv_igemm_kernel<BN_, WM_, WN_, true, MINW_><<<grid, block, sm, st>>>(k);                \
        else        conv_igemm_kernel<BN_, WM_, WN_, false, MINW_><<<grid, block, sm, st>>>(k);               \
    } while (0)
    if (bn == 32) YH_LAUNCH_CONV(32, 4, 1, 4);
    else if (bn == 64) YH_LAUNCH_CONV(64, 4, 1, 3);
    else YH_LAUNCH_CONV(128, 2, 2, 2);
#undef YH_LAUNCH_CONV
    YH_CHECK_LAUNCH("yh_conv_igemm");
    return YH_OK;
}
}  // namespace

extern "C" int yh_conv_igemm(const yh_conv_desc* d, yh_stream stream) { return conv_run(d, stream, nullptr, 0); }

/* rows of the partial-sum slab a data-gradient launch with the fused BatchNorm-backward reduction (bnr_*) writes:
 * [rows][2][N] floats (sum dz | sum dz*z), consumed by yh_bn_bwd_finalize(part, rows, ...).  0: this descriptor
 * cannot take the fused path (the caller keeps the separate yh_bn_silu_bwd_reduce pass). */
extern "C" int yh_conv_bnr_rows(const yh_conv_desc* d)
{
    if (!conv_desc_plannable(d) || d->mode != YH_CONV_DGRAD || d->nseg != 1 || d->seg[0].C % 8 || d->seg[0].ups || d->N % 8) return 0;
    if (d->bias || d->scale || d->shift || d->act != YH_ACT_NONE || d->res || d->nsplit < d->N || d->stats) return 0;
    if (conv_dbg_mask() & 16) return 0;
    if (d->algo == 7 || d->algo == 8 || d->algo == 13) {
        const int r7 = d->algo == 7 ? yh_dg2_rows(d) : (d->algo == 8 ? yh_p3_rows(d) : yh_pt_rows(d));
        if (r7 > 0) return r7;
        yh_conv_desc d0 = *d;
        d0.algo = 0;
        return yh_conv_bnr_rows(&d0);
    }
    const unsigned long M = (unsigned long)d->B * d->Ho * d->Wo;
    if (M >= (1ul << 31) - BM) return 0;
    const unsigned long npix = (unsigned long)d->B * d->Hi * d->Wi;
    if (((npix - 1) * d->seg[0].ld + d->seg[0].C) * 2 >= (1ul << 31)) return 0;
    if ((unsigned long)d->Npad * d->KH * d->KW * d->seg[0].C * 2 >= (1ul << 31)) return 0;
    int gx, gy, bn;
    conv_grid(d, &gx, &gy, &bn);
    const bool cls = d->stride == 2 && d->Ho % 2 == 0 && d->Wo % 2 == 0 && d->KH >= 2 && d->KW >= 2;
    if (!stem_eligible(d) && conv_halo_ok(d, nullptr)) return gx;
    const int zslots = d->KH * d->KW;
    if (conv_v3_variant(d)) return cls ? gx * zslots : gx;
    if (cls) {
        const long mt = ((long)(M / 4) + BM - 1) / BM;          // conv_v2_kernel (the only non-v3 kernel with this epilogue): one row per block
        if (mt >= CLS_INNER_MIN_TILES) return cls_inner_blocks(gx < mt ? gx : (int)mt);
        return cls_blocks_per_slot(gx, zslots, mt) * zslots;
    }
    return gx;
}

/* name of the kernel instantiation yh_conv_igemm launches for this descriptor, as profilers print it */
extern "C" int yh_conv_kernel_name(const yh_conv_desc* d, char* buf, int buflen)
{
    YH_CHECK_ARG(buf && buflen >= 64, "yh_conv_kernel_name: buffer too small");
    return conv_run(d, nullptr, buf, buflen);
}

/* diagnostics: a device buffer of grid x 8 x 8 uint64 that receives cycle sums of every workgroup's 2nd tile in conv_halo160_kernel (NULL: off) */
extern "C" void yh_halo_set_stamps(void* p) { g_halo_stamps = (unsigned long long*)p; }

#ifdef YH_CONV_STAMPS
extern "C" int yh_debug_read_stamps(long long* host_out32)
{
    return hipMemcpyFromSymbol(host_out32, HIP_SYMBOL(g_stamps), sizeof(long long) * 32) == hipSuccess ? 0 : -1;
}
#endif

// diagnostics: resident blocks per CU the runtime predicts for each instantiation (bn = 32/64/128)
extern "C" int yh_debug_conv_occupancy(int bn)
{
    int nb = -1;
    hipError_t e;
    if (bn == 32) e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, conv_igemm_kernel<32, 4, 1, true, 4>, 256, conv_smem_bytes<32, 4, 1>());
    else if (bn == 64) e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, conv_igemm_kernel<64, 4, 1, true, 3>, 256, conv_smem_bytes<64, 4, 1>());
    else e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, conv_igemm_kernel<128, 2, 2, true, 2>, 256, conv_smem_bytes<128, 2, 2>());
    return e == hipSuccess ? nb : -(int)e;
}
