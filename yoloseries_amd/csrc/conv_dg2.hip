// Data gradient of a 3x3 / stride-2 / pad-1 convolution (the six down-sampling layers of YOLOv5 / YOLOX) as ONE pass over
// the gradient map gz of the conv output:
//
//   ga[b][2i+ph][2j+pw][c] = sum over the taps (kh, kw) of parity class (ph, pw), n:
//                            gz[b][i+dy(kh)][j+dx(kw)][n] * W[n][c][kh][kw]          dy(0) = 1, dy(1) = dy(2) = 0
//
// A block owns a region of <= 128 gz pixels of one image.  It stages the (TH+1) x (TW+1) patch of gz ONCE per channel block
// in LDS and forms all four parity classes of the 2TH x 2TW output pixels from it: the nine taps read only FOUR shifted views
// of the patch (a view feeds 4 / 2 / 2 / 1 taps), so gz leaves HBM once (the im2col form re-reads every gz row for each class,
// on different XCDs: PMC 2.2x the algorithmic bytes) and a class's pixels — every other pixel of a row — are interleaved in LDS
// before they are stored, so whole 128-byte lines are written instead of 64-byte halves.  MFMA 32x32x16 bf16 with swapped
// operands (D = W X^T): a lane ends up with 4 consecutive channels of one pixel (8-byte LDS stores into the output staging).
// Epilogues: plain / accumulating store (EPI 0) and the fused BatchNorm+SiLU backward reduction of the producer layer (EPI 3,
// yh_conv_desc.bnr_*), identical in arithmetic to conv_v2_kernel's.
// Replaces autograd's conv_transpose for `ConvBnAct(k=3, s=2)` (utils/layer_tools.py:82-94, models/normal/yolov5s.py:18-40).
#include "common.h"
#include <stdlib.h>

namespace {

typedef __attribute__((ext_vector_type(4))) unsigned int u32x4_t;
typedef unsigned int u32x4_nt __attribute__((ext_vector_type(4)));

__device__ __forceinline__ uint4 dg_ld_nt16(const uint16_t* p) {
    u32x4_nt v = __builtin_nontemporal_load(reinterpret_cast<const u32x4_nt*>(p));
    return make_uint4(v.x, v.y, v.z, v.w);
}
__device__ __forceinline__ void dg_st_nt16(uint16_t* p, uint4 v) {
    u32x4_nt w = {v.x, v.y, v.z, v.w};
    __builtin_nontemporal_store(w, reinterpret_cast<u32x4_nt*>(p));
}

constexpr int DG_RPX1 = 128;        // region pixels per block and pixel tile (PT tiles of 32 pixels per wave: 128 * PT pixels per block)
constexpr int dg_pmax(int pt) { return pt == 1 ? 168 : 304; }     // most patch pixels (TH+1)*(TW+1)

struct Dg2K {
    const uint16_t* gz; int ldg, Nk;
    const uint16_t* w; int Ktot;
    uint16_t* out; int ld0, C;
    int B, Hg, Wg, Ho, Wo;
    int TH, TW, tx, ty, ntiles, nchunk, npatch;
    int accumulate;
    const uint16_t* z; int ldz; const float* ws; int wsC; float* part;
    unsigned gzbytes, wbytes;
};

template <int CT, int KC, int PT>
constexpr int dg_smem_bytes() {
    // patch [PMAX][KC+8] | weights [9][32*CT][KC+8] | scale,shift [2][32*CT] floats;  the output staging (one row parity at a
    // time: [2*128*PT][32*CT+8]) aliases the patch (+ the weights when it is larger than the patch)
    const int main_b = (dg_pmax(PT) + 9 * 32 * CT) * (KC + 8) * 2;
    const int stg_b = 2 * DG_RPX1 * PT * (32 * CT + 8) * 2;
    return (main_b > stg_b ? main_b : stg_b) + 2 * 32 * CT * 4;
}

// block = 4*CT waves: wave (ct, q) multiplies PT tiles of 32 region pixels (q*PT .. q*PT+PT-1) with the ct-th 32 output channels,
// all four parity classes (4 x PT x 16 accumulator registers); CB = 32*CT channels per block; KC gz channels per step
template <int CT, int KC, int EPI, int PT>
__global__ __launch_bounds__(256 * CT, 1) void conv_dg2_kernel(const Dg2K p)
{
    constexpr int NT = 256 * CT;
    constexpr int DG_RPX = DG_RPX1 * PT;
    constexpr int DG_PMAX = dg_pmax(PT);
    constexpr int PITCH = KC + 8;                 // LDS row pitch in elements (80 / 144 bytes: conflict-free ds_read_b128)
    constexpr int CHR = KC / 8;                   // 16-byte chunks per patch pixel / weight row
    constexpr int CB = 32 * CT;                   // output channels of the block
    constexpr int PPP = NT / CHR;                 // patch pixels / weight rows covered by one pass of the block's threads
    constexpr int NPI = (DG_PMAX + PPP - 1) / PPP;               // patch chunks per thread
    constexpr int NWI = (9 * CB + PPP - 1) / PPP;                // weight chunks per thread
    constexpr int SP = CB + 8;                    // staging row pitch (elements)
    constexpr int CPR = CB / 8;                   // 16-byte chunks per staging row
    constexpr int NOI = 2 * DG_RPX * CPR / NT;    // read-out items per thread and row parity (== 4 * PT)
    constexpr unsigned OOB = 0x80000000u;
    constexpr int PATCH_BYTES = DG_PMAX * PITCH * 2;
    constexpr int STG_BYTES = 2 * DG_RPX * SP * 2;
    constexpr bool STG_IN_PATCH = STG_BYTES <= PATCH_BYTES;        // then the weights survive the epilogue
    static_assert(NT % CPR == 0 && NOI == 4 * PT, "read-out mapping");

    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    uint16_t* sP = reinterpret_cast<uint16_t*>(smem);
    uint16_t* sW = sP + DG_PMAX * PITCH;
    float* sStat = reinterpret_cast<float*>(smem + dg_smem_bytes<CT, KC, PT>() - 2 * CB * 4);
    uint16_t* sStg = reinterpret_cast<uint16_t*>(smem);

    const int t = threadIdx.x;
    const int lane = t & 63;
    const int wave = t >> 6;
    const int wq = wave & 3, wct = wave >> 2;
    const int c0 = blockIdx.y * CB;
    const int TWp = p.TW + 1;

    const __amdgpu_buffer_rsrc_t rsg = __builtin_amdgcn_make_buffer_rsrc((void*)p.gz, 0, p.gzbytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsw = __builtin_amdgcn_make_buffer_rsrc((void*)p.w, 0, p.wbytes, 0x00020000);

    // ---- per-thread constants.  A thread keeps its 16-byte chunk column (t % CHR) and walks rows t / CHR + j * PPP.
    const int ch = t % CHR, row0 = t / CHR;
    // weights: row = tap * CB + c; PPP is a multiple or a divisor-multiple of CB, so pass j starts at tap j * (PPP / CB) (PPP >= CB)
    // or at (tap, c) = (j / (CB/PPP), ...): handled generically with one division per pass, done once here
    unsigned woff[NWI];
#pragma unroll
    for (int j = 0; j < NWI; ++j) {
        const int row = row0 + j * PPP;
        const int tap = row / CB, c = row - tap * CB;
        woff[j] = row < 9 * CB ? (unsigned)((((size_t)(c0 + c) * p.Ktot) + tap * p.Nk + ch * 8) * 2) : OOB;
    }
    int ppij[NPI];                                  // patch pixel (pi << 8 | pj) of item j; -1: past the patch
#pragma unroll
    for (int j = 0; j < NPI; ++j) {
        const int pp = row0 + j * PPP;
        const int pi = pp / TWp;
        ppij[j] = pp < p.npatch ? (pi << 8) | (pp - pi * TWp) : -1;
    }
    // MFMA fragments: region pixel of this lane (B operand column) and its patch position
    const int koff = (lane >> 5) * 8;
    int rpx[PT];
    const uint16_t* xbase[PT];
#pragma unroll
    for (int pt = 0; pt < PT; ++pt) {
        rpx[pt] = (wq * PT + pt) * 32 + (lane & 31);
        const bool rvalid = rpx[pt] < p.TH * p.TW;
        const int ri = rvalid ? rpx[pt] / p.TW : 0;
        const int rj = rvalid ? rpx[pt] - ri * p.TW : 0;
        xbase[pt] = sP + (ri * TWp + rj) * PITCH + koff;
    }
    const uint16_t* const wbase = sW + (wct * 32 + (lane & 31)) * PITCH + koff;

    if (EPI == 3) {
        for (int i = t; i < 2 * CB; i += NT) {
            const int which = i / CB, c = i - which * CB;
            sStat[i] = (c0 + c < p.C) ? p.ws[(size_t)which * p.wsC + c0 + c] : 0.f;
        }
    }
    float bs_[8], bq_[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) { bs_[e] = 0.f; bq_[e] = 0.f; }

    u32x4_t rp[NPI], rw[NWI];
    const bool w_resident = STG_IN_PATCH && p.nchunk == 1;       // one channel block: the weights are loaded once per block

    auto tile_origin = [&](int tile, int& b, int& i0, int& j0) {
        const int per = p.tx * p.ty;
        b = tile / per;
        const int r = tile - b * per;
        const int tyi = r / p.tx;
        i0 = tyi * p.TH; j0 = (r - tyi * p.tx) * p.TW;
    };
    auto load_regs = [&](int tile, int kc, bool with_w) {
        int b, i0, j0;
        tile_origin(tile, b, i0, j0);
        const int so = kc * KC * 2;
#pragma unroll
        for (int j = 0; j < NPI; ++j) {
            const int gi = i0 + (ppij[j] >> 8), gj = j0 + (ppij[j] & 0xff);
            const bool ok = ppij[j] >= 0 && gi < p.Hg && gj < p.Wg;
            const unsigned off = ok ? (unsigned)(((b * p.Hg + gi) * p.Wg + gj) * (p.ldg * 2) + ch * 16) : OOB;
            rp[j] = __builtin_amdgcn_raw_buffer_load_b128(rsg, off, so, 0);
        }
        if (with_w) {
#pragma unroll
            for (int j = 0; j < NWI; ++j) rw[j] = __builtin_amdgcn_raw_buffer_load_b128(rsw, woff[j], so, 0);
        }
    };
    auto store_regs = [&](bool with_w) {
        uint16_t* dp = sP + row0 * PITCH + ch * 8;
#pragma unroll
        for (int j = 0; j < NPI; ++j)
            if (row0 + j * PPP < DG_PMAX) *reinterpret_cast<u32x4_t*>(dp + j * PPP * PITCH) = rp[j];
        if (with_w) {
            uint16_t* dw = sW + row0 * PITCH + ch * 8;
#pragma unroll
            for (int j = 0; j < NWI; ++j)
                if (row0 + j * PPP < 9 * CB) *reinterpret_cast<u32x4_t*>(dw + j * PPP * PITCH) = rw[j];
        }
    };

    f32x16_t acc[4][PT];
    int tile = blockIdx.x;
    if (tile >= p.ntiles) return;
    load_regs(tile, 0, true);
    bool first = true;
    for (; tile < p.ntiles; tile += gridDim.x) {
#pragma unroll
        for (int q = 0; q < 4; ++q)
#pragma unroll
            for (int pt = 0; pt < PT; ++pt)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[q][pt][r] = 0.f;
        for (int kc = 0; kc < p.nchunk; ++kc) {
            const bool with_w = !w_resident || first;
            __syncthreads();                           // the previous step's fragment reads / read-out are done
            store_regs(with_w);
            __syncthreads();
            first = false;
            {   // request the next step's operands: they arrive while this step is multiplied
                int nt = tile, nk = kc + 1;
                if (nk == p.nchunk) { nk = 0; nt = tile + gridDim.x; }
                if (nt < p.ntiles) load_regs(nt, nk, !w_resident);
            }
#pragma unroll
            for (int sh = 0; sh < 4; ++sh) {
                const int dy = sh >> 1, dx = sh & 1;
                bf16x8_t xf[PT][KC / 16];
#pragma unroll
                for (int pt = 0; pt < PT; ++pt)
#pragma unroll
                    for (int ks = 0; ks < KC / 16; ++ks)
                        xf[pt][ks] = __builtin_bit_cast(bf16x8_t, *reinterpret_cast<const uint4*>(xbase[pt] + (dy * TWp + dx) * PITCH + ks * 16));
#pragma unroll
                for (int a = 0; a < (dy ? 1 : 2); ++a) {
                    const int kh = dy ? 0 : 1 + a;
#pragma unroll
                    for (int b2 = 0; b2 < (dx ? 1 : 2); ++b2) {
                        const int kw = dx ? 0 : 1 + b2;
                        const int tap = kh * 3 + kw;
                        const int cls = (kh != 1 ? 2 : 0) + (kw != 1 ? 1 : 0);
#pragma unroll
                        for (int ks = 0; ks < KC / 16; ++ks) {
                            const bf16x8_t wf = __builtin_bit_cast(bf16x8_t, *reinterpret_cast<const uint4*>(wbase + tap * CB * PITCH + ks * 16));
#pragma unroll
                            for (int pt = 0; pt < PT; ++pt)
                                acc[cls][pt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf, xf[pt][ks], acc[cls][pt], 0, 0, 0);
                        }
                    }
                }
            }
        }

        // ---- epilogue: one row parity at a time through the staging buffer, stored as whole output rows
        int b, i0, j0;
        tile_origin(tile, b, i0, j0);
        const int cch = t % CPR;                   // this thread's chunk column of the staging rows
        const int n = c0 + cch * 8;
#pragma unroll
        for (int ph = 0; ph < 2; ++ph) {
            // pixel index of this thread's read-out items (and, EPI 3, the producer's z chunks: requested before the barriers)
            int oidx[NOI];
            constexpr bool ZPRE = EPI == 3 && PT == 1;     // PT 2: the z chunks are requested in the read-out loop (register budget)
            uint4 zpre[ZPRE ? NOI : 1];
#pragma unroll
            for (int it = 0; it < NOI; ++it) {
                const int row = t / CPR + it * (NT / CPR);
                const int r = row >> 1, pw = row & 1;
                const int i = r / p.TW, j = r - i * p.TW;
                const int gi = i0 + i, gj = j0 + j;
                const bool ok = r < p.TH * p.TW && gi < p.Hg && gj < p.Wg && n < p.C;
                oidx[it] = ok ? (b * p.Ho + 2 * gi + ph) * p.Wo + 2 * gj + pw : -1;
                if (ZPRE) {
                    zpre[it] = make_uint4(0, 0, 0, 0);
                    if (ok) zpre[it] = dg_ld_nt16(p.z + (size_t)oidx[it] * p.ldz + n);
                }
            }
            __syncthreads();                           // fragment reads of the last channel block (ph 0) / read-out of ph 0 (ph 1) done
#pragma unroll
            for (int pw = 0; pw < 2; ++pw)
#pragma unroll
                for (int pt = 0; pt < PT; ++pt) {
                    uint16_t* dst = sStg + (rpx[pt] * 2 + pw) * SP + wct * 32 + 4 * (lane >> 5);
                    const f32x16_t& a = acc[ph * 2 + pw][pt];
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        uint2 v;
                        v.x = pack2(a[4 * q + 0], a[4 * q + 1]);
                        v.y = pack2(a[4 * q + 2], a[4 * q + 3]);
                        *reinterpret_cast<uint2*>(dst + 8 * q) = v;
                    }
                }
            __syncthreads();
#pragma unroll
            for (int it = 0; it < NOI; ++it) {
                if (oidx[it] < 0) continue;
                const int row = t / CPR + it * (NT / CPR);
                uint4 v = *reinterpret_cast<const uint4*>(sStg + row * SP + cch * 8);
                uint16_t* dst = p.out + (size_t)oidx[it] * p.ld0 + n;
                if (p.accumulate) {
                    const uint4 ov = *reinterpret_cast<const uint4*>(dst);
                    float f[8], g0[8];
                    unpack8(v, f);
                    unpack8(ov, g0);
#pragma unroll
                    for (int e = 0; e < 8; ++e) f[e] += g0[e];
                    v = pack8(f);
                }
                if (EPI == 3) dg_st_nt16(dst, v);
                else *reinterpret_cast<uint4*>(dst) = v;
                if (EPI == 3) {
                    float g[8], z[8];
                    unpack8(v, g);
                    const uint4 zv = ZPRE ? zpre[it] : dg_ld_nt16(p.z + (size_t)oidx[it] * p.ldz + n);
                    unpack8(zv, z);
                    const float4 s0 = *reinterpret_cast<const float4*>(sStat + cch * 8);
                    const float4 s1 = *reinterpret_cast<const float4*>(sStat + cch * 8 + 4);
                    const float4 h0 = *reinterpret_cast<const float4*>(sStat + CB + cch * 8);
                    const float4 h1 = *reinterpret_cast<const float4*>(sStat + CB + cch * 8 + 4);
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

    if (EPI == 3) {
        float* sRed = reinterpret_cast<float*>(smem);          // [NT][16]: aliases the idle patch / weight buffers
        __syncthreads();
#pragma unroll
        for (int e = 0; e < 8; ++e) { sRed[t * 16 + e] = bs_[e]; sRed[t * 16 + 8 + e] = bq_[e]; }
        __syncthreads();
        for (int i = t; i < 2 * CB; i += NT) {
            const int which = i / CB, c = i - which * CB;
            float v = 0.f;
            for (int j = c / 8; j < NT; j += CPR) v += sRed[j * 16 + which * 8 + (c & 7)];     // a thread keeps chunk column t % CPR
            if (c0 + c < p.C) p.part[((size_t)blockIdx.x * 2 + which) * p.C + c0 + c] = v;
        }
    }
}

// region geometry for a gz map of Hg x Wg: TH x TW <= 128 pixels with a patch of <= DG_PMAX pixels, most useful MFMA rows first
bool dg2_geom(int Hg, int Wg, int pt, int* TH, int* TW, int* tx, int* ty)
{
    double best = -1.0;
    const int rpx = DG_RPX1 * pt;
    for (int tw = 4; tw <= 64; ++tw)
        for (int th = 1; th * tw <= rpx; ++th) {
            if ((th + 1) * (tw + 1) > dg_pmax(pt)) continue;
            const int nx = (Wg + tw - 1) / tw, ny = (Hg + th - 1) / th;
            double eff = (double)Hg * Wg / ((double)nx * ny * rpx);
            eff += 1e-4 * tw;                                  // ties: longer rows (whole lines per store)
            if (eff > best) { best = eff; *TH = th; *TW = tw; *tx = nx; *ty = ny; }
        }
    return best > 0.0;
}

struct Dg2Plan { int ct, kc, pt, gx, gy; Dg2K k; };

bool dg2_plan(const yh_conv_desc* d, Dg2Plan* pl)
{
    if (d->mode != YH_CONV_DGRAD || d->nseg != 1 || d->seg[0].ups) return false;
    if (d->KH != 3 || d->KW != 3 || d->stride != 2 || d->pad != 1) return false;
    if (d->Ho != 2 * d->Hi || d->Wo != 2 * d->Wi) return false;
    if (d->bias || d->scale || d->shift || d->act != YH_ACT_NONE || d->res || d->nsplit < d->N || d->stats) return false;
    const int Nk = d->seg[0].C;
    if (Nk % 32 || d->N % 8) return false;
    const unsigned long gzb = ((unsigned long)d->B * d->Hi * d->Wi - 1) * d->seg[0].ld * 2 + (unsigned long)Nk * 2;
    const unsigned long wb = (unsigned long)d->Npad * 9 * Nk * 2;
    if (gzb >= (1ul << 31) || wb >= (1ul << 31)) return false;
    if ((unsigned long)d->B * d->Ho * d->Wo >= (1ul << 31)) return false;
    Dg2K& k = pl->k;
    pl->ct = d->N <= 32 ? 1 : 2;
    pl->kc = (Nk % 64 == 0 && d->tile_k != 32) ? 64 : 32;
    // PT = 2 (two pixel tiles per wave, 256-pixel regions: twice the MFMAs per weight fragment and per barrier) is implemented and
    // correct, and measured at HALF the speed of PT = 1 on the YOLOv5s / YOLOv5l layers: 128 accumulator registers plus the register-
    // staged operands of the next step need > 256 VGPRs (spills with 8 waves, one 4-wave block per CU with 4) — not instantiated
    pl->pt = 1;
    if (!dg2_geom(d->Hi, d->Wi, pl->pt, &k.TH, &k.TW, &k.tx, &k.ty)) return false;
    const int cb = 32 * pl->ct;
    pl->gy = (d->N + cb - 1) / cb;
    if (pl->gy * cb > d->Npad) return false;
    k.gz = d->seg[0].ptr; k.ldg = d->seg[0].ld; k.Nk = Nk;
    k.w = d->w; k.Ktot = 9 * Nk;
    k.out = d->out0; k.ld0 = d->ld0; k.C = d->N;
    k.B = d->B; k.Hg = d->Hi; k.Wg = d->Wi; k.Ho = d->Ho; k.Wo = d->Wo;
    k.ntiles = d->B * k.tx * k.ty;
    k.nchunk = Nk / pl->kc;
    k.npatch = (k.TH + 1) * (k.TW + 1);
    k.accumulate = d->accumulate;
    k.z = d->bnr_z; k.ldz = d->bnr_ldz; k.ws = d->bnr_ws; k.wsC = d->bnr_C; k.part = d->bnr_part;
    k.gzbytes = (unsigned)gzb; k.wbytes = (unsigned)wb;
    int cap = (256 * (pl->ct == 1 ? 2 : 1)) / pl->gy;
    if (cap < 1) cap = 1;
    if (d->grid_cap > 0) cap = d->grid_cap;
    pl->gx = k.ntiles < cap ? k.ntiles : cap;
    return true;
}

}  // namespace

// entry points used by conv_igemm.hip (yh_conv_igemm with algo 7): eligibility + grid rows, kernel name, launch
int yh_dg2_rows(const yh_conv_desc* d)
{
    Dg2Plan pl;
    return dg2_plan(d, &pl) ? pl.gx : 0;
}

int yh_dg2_run(const yh_conv_desc* d, yh_stream stream, char* name_out, int name_len)
{
    Dg2Plan pl;
    YH_CHECK_ARG(dg2_plan(d, &pl), "yh_conv_igemm: algo 7 (stride-2 data-gradient kernel) is not eligible for this descriptor");
    const int epi = d->bnr_part ? 3 : 0;
    if (d->bnr_part)
        YH_CHECK_ARG(d->bnr_z && yh_aligned16(d->bnr_z) && d->bnr_ldz % 8 == 0 && d->bnr_ws && d->bnr_C >= d->N, "yh_conv_igemm: bad fused-reduction operands");
    if (name_out) { snprintf(name_out, name_len, "conv_dg2_kernel<%d, %d, %d, %d>", pl.ct, pl.kc, epi, pl.pt); return YH_OK; }
    hipStream_t st = (hipStream_t)stream;
    const dim3 grid(pl.gx, pl.gy), blk(256 * pl.ct);
#define YH_LAUNCH_DG2(CT_, KC_, PT_)                                                                                   \
    do {                                                                                                               \
        const int sm = dg_smem_bytes<CT_, KC_, PT_>();                                                                 \
        static YhDevOnce attr_set;                                                                                        \
        if (attr_set.need()) {                                                                                               \
            attr_set.set((const void*)conv_dg2_kernel<CT_, KC_, 0, PT_>, hipFuncAttributeMaxDynamicSharedMemorySize, sm); \
            attr_set.set((const void*)conv_dg2_kernel<CT_, KC_, 3, PT_>, hipFuncAttributeMaxDynamicSharedMemorySize, sm); \
            attr_set.done();                                                                                            \
        }                                                                                                              \
        if (epi == 3) conv_dg2_kernel<CT_, KC_, 3, PT_><<<grid, blk, sm, st>>>(pl.k);                                  \
        else          conv_dg2_kernel<CT_, KC_, 0, PT_><<<grid, blk, sm, st>>>(pl.k);                                  \
    } while (0)
    if (pl.ct == 1)      { if (pl.kc == 64) YH_LAUNCH_DG2(1, 64, 1); else YH_LAUNCH_DG2(1, 32, 1); }
    else                 { if (pl.kc == 64) YH_LAUNCH_DG2(2, 64, 1); else YH_LAUNCH_DG2(2, 32, 1); }
#undef YH_LAUNCH_DG2
    YH_CHECK_LAUNCH("yh_conv_igemm(dg2)");
    return YH_OK;
}
