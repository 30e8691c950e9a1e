// 3x3 / stride-1 / pad-1 convolution (forward and data gradient) for SMALL channel counts (32 / 64 channels on the large maps
// of YOLOv5 stage 1 / 2: the bottlenecks' conv_bn_act_2, utils/layer_tools.py:97-114), built like conv_dg2.hip:
//
//   out[b][i][j][n] = sum over taps (kh, kw), c:  x[b][i + sy(kh)][j + sx(kw)][c] * W[n][tap*C + c]
//                     forward: (sy, sx) = (kh - 1, kw - 1);  data gradient (x = gz, flipped image packed by the caller): (1 - kh, 1 - kw)
//
// A block owns a region of <= 128 * PT output pixels of one image, stages the (TH+2) x (TW+2) input patch ONCE per channel block
// in LDS and reads the nine taps as nine shifted views of it — the register-staged im2col kernel these layers ran on fetches
// every input row nine times through L2 (YOLOv5s stage 1: 1.6-2.0 TB/s of a 5.5 TB/s roof).  The weights of a layer with one
// channel block stay resident in LDS.  MFMA 32x32x16 bf16 with swapped operands (D = W X^T), output staged through LDS and
// stored as whole rows.  Epilogues: 0 plain / accumulating store, 1 + BatchNorm partial sums of the stored (bf16-rounded) values
// (forward, one slab row per block), 3 fused BatchNorm+SiLU backward reduction of the producer (data gradient, yh_conv_desc.bnr_*).
// Chosen per layer by the engine's timing (yh_conv_desc.algo 8).
#include "common.h"
#include <stdlib.h>

namespace {

typedef __attribute__((ext_vector_type(4))) unsigned int u32x4_t;
typedef unsigned int u32x4_nt __attribute__((ext_vector_type(4)));

__device__ __forceinline__ uint4 p3_ld_nt16(const uint16_t* p) {
    u32x4_nt v = __builtin_nontemporal_load(reinterpret_cast<const u32x4_nt*>(p));
    return make_uint4(v.x, v.y, v.z, v.w);
}
__device__ __forceinline__ void p3_st_nt16(uint16_t* p, uint4 v) {
    u32x4_nt w = {v.x, v.y, v.z, v.w};
    __builtin_nontemporal_store(w, reinterpret_cast<u32x4_nt*>(p));
}

// most patch pixels for 128 * PT region pixels: stride 1 (TH+2)*(TW+2), stride 2 (2 TH+1)*(2 TW+1)
constexpr int p3_pmax(int pt, int s = 1) { return s == 2 ? (pt == 1 ? 600 : 1120) : (pt == 1 ? 200 : 340); }

struct P3K {
    const uint16_t* x; int ldx, Cin;
    const uint16_t* w; int Ktot;
    uint16_t* out; int ld0, N, Npad;
    int B, H, W;                           // output grid
    int Hi, Wi;                            // input grid (== output grid at stride 1)
    int TH, TW, tx, ty, ntiles, nchunk, npatch;
    int PW;                                // patch row length in pixels: TW + 2 (stride 1), 2 TW + 1 (stride 2)
    int accumulate, flip;                  // flip: data gradient (tap (kh, kw) reads the patch at (2 - kh, 2 - kw))
    float* stats;                          // EPI 1: [gridDim.x][2][Npad]
    const uint16_t* z; int ldz; const float* ws; int wsC; float* part;      // EPI 3
    unsigned xbytes, wbytes;
};

template <int CT, int KC, int PT, int S = 1>
constexpr int p3_smem_bytes() {
    const int main_b = (p3_pmax(PT, S) + 9 * 32 * CT) * (KC + 8) * 2;
    const int stg_b = 128 * PT * (32 * CT + 8) * 2;
    return (main_b > stg_b ? main_b : stg_b) + 2 * 32 * CT * 4;
}

// block = 4*CT waves: wave (ct, q) multiplies PT tiles of 32 region pixels with the ct-th 32 output channels.
// S = 2 (forward only): the stride-2 downsampling layers (utils/layer_tools.py:82-94 with models/normal/yolov5s.py:18-40's
// ConvBnAct(.., 3, 2, 1)).  The patch of a TH x TW output tile is (2 TH + 1) x (2 TW + 1) input pixels; its columns are stored
// DE-INTERLEAVED by parity (even input columns first, then the odd ones), so the 32 consecutive output pixels of a fragment read
// consecutive patch slots for every tap — the same conflict-free LDS stride as at stride 1 (an interleaved patch would put the
// lanes 2 slots apart: a two-way bank conflict on every fragment read).  The im2col ring kernel these layers ran on stages every
// input pixel 2.25 times through the CU's LDS-DMA path (512 B per MFMA at 64 output channels); here each is staged once.
template <int CT, int KC, int EPI, int PT, int S = 1>
__global__ __launch_bounds__(256 * CT, 1) void conv_p3_kernel(const P3K p)
{
    static_assert(S == 1 || (S == 2 && EPI != 3), "stride 2: forward only");
    constexpr int NT = 256 * CT;
    constexpr int RPX = 128 * PT;
    constexpr int PMAX = p3_pmax(PT, S);
    constexpr int PITCH = KC + 8;
    constexpr int CHR = KC / 8;
    constexpr int CB = 32 * CT;
    constexpr int PPP = NT / CHR;
    constexpr int NPI = (PMAX + PPP - 1) / PPP;
    constexpr int NWI = (9 * CB + PPP - 1) / PPP;
    constexpr int SP = CB + 8;
    constexpr int CPR = CB / 8;
    constexpr int NOI = RPX * CPR / NT;            // read-out items per thread (== 2 * PT)
    constexpr unsigned OOB = 0x80000000u;
    constexpr bool STG_IN_PATCH = RPX * SP * 2 <= PMAX * PITCH * 2;
    static_assert(NT % CPR == 0 && NOI == 2 * PT, "read-out mapping");

    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    uint16_t* sP = reinterpret_cast<uint16_t*>(smem);
    uint16_t* sW = sP + PMAX * PITCH;
    float* sStat = reinterpret_cast<float*>(smem + p3_smem_bytes<CT, KC, PT, S>() - 2 * CB * 4);
    uint16_t* sStg = reinterpret_cast<uint16_t*>(smem);

    const int t = threadIdx.x;
    const int lane = t & 63;
    const int wave = t >> 6;
    const int wq = wave & 3, wct = wave >> 2;
    const int c0 = blockIdx.y * CB;
    const int TWp = p.PW;
    const int PWe = p.TW + 1;              // stride 2: even-column slots of a patch row (the odd columns follow them)

    const __amdgpu_buffer_rsrc_t rsx = __builtin_amdgcn_make_buffer_rsrc((void*)p.x, 0, p.xbytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsw = __builtin_amdgcn_make_buffer_rsrc((void*)p.w, 0, p.wbytes, 0x00020000);

    const int ch = t % CHR, row0 = t / CHR;
    unsigned woff[NWI];
#pragma unroll
    for (int j = 0; j < NWI; ++j) {
        const int row = row0 + j * PPP;
        const int tap = row / CB, c = row - tap * CB;
        woff[j] = row < 9 * CB ? (unsigned)((((size_t)(c0 + c) * p.Ktot) + tap * p.Cin + ch * 8) * 2) : OOB;
    }
    int ppij[NPI];
#pragma unroll
    for (int j = 0; j < NPI; ++j) {
        const int pp = row0 + j * PPP;
        const int pi = pp / TWp;
        int pj = pp - pi * TWp;
        if (S == 2) pj = pj < PWe ? 2 * pj : 2 * (pj - PWe) + 1;          // slot -> input column of the patch
        ppij[j] = pp < p.npatch ? (pi << 8) | pj : -1;
    }
    const int koff = (lane >> 5) * 8;
    int rpx[PT];
    const uint16_t* xbase[PT];
#pragma unroll
    for (int pt = 0; pt < PT; ++pt) {
        rpx[pt] = (wq * PT + pt) * 32 + (lane & 31);
        const bool rvalid = rpx[pt] < p.TH * p.TW;
        const int ri = rvalid ? rpx[pt] / p.TW : 0;
        const int rj = rvalid ? rpx[pt] - ri * p.TW : 0;
        xbase[pt] = sP + (S * ri * TWp + rj) * PITCH + koff;
    }
    const uint16_t* const wbase = sW + (wct * 32 + (lane & 31)) * PITCH + koff;

    if (EPI == 3) {
        for (int i = t; i < 2 * CB; i += NT) {
            const int which = i / CB, c = i - which * CB;
            sStat[i] = (c0 + c < p.N) ? p.ws[(size_t)which * p.wsC + c0 + c] : 0.f;
        }
    }
    float bs_[8], bq_[8];                  // EPI 1: sum, sum of squares;  EPI 3: sum dz, sum dz*z  (this thread's chunk column)
#pragma unroll
    for (int e = 0; e < 8; ++e) { bs_[e] = 0.f; bq_[e] = 0.f; }

    u32x4_t rp[NPI], rw[NWI];
    const bool w_resident = STG_IN_PATCH && p.nchunk == 1;

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
            const int gi = S * i0 - 1 + (ppij[j] >> 8), gj = S * j0 - 1 + (ppij[j] & 0xff);
            const bool ok = ppij[j] >= 0 && gi >= 0 && gj >= 0 && gi < p.Hi && gj < p.Wi;
            const unsigned off = ok ? (unsigned)(((b * p.Hi + gi) * p.Wi + gj) * (p.ldx * 2) + ch * 16) : OOB;
            rp[j] = __builtin_amdgcn_raw_buffer_load_b128(rsx, off, so, 0);
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
            if (row0 + j * PPP < PMAX) *reinterpret_cast<u32x4_t*>(dp + j * PPP * PITCH) = rp[j];
        if (with_w) {
            uint16_t* dw = sW + row0 * PITCH + ch * 8;
#pragma unroll
            for (int j = 0; j < NWI; ++j)
                if (row0 + j * PPP < 9 * CB) *reinterpret_cast<u32x4_t*>(dw + j * PPP * PITCH) = rw[j];
        }
    };

    f32x16_t acc[PT];
    int tile = blockIdx.x;
    if (tile < p.ntiles) load_regs(tile, 0, true);
    bool first = true;
    for (; tile < p.ntiles; tile += gridDim.x) {
#pragma unroll
        for (int pt = 0; pt < PT; ++pt)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[pt][r] = 0.f;
        for (int kc = 0; kc < p.nchunk; ++kc) {
            const bool with_w = !w_resident || first;
            __syncthreads();
            store_regs(with_w);
            __syncthreads();
            first = false;
            {
                int nt = tile, nk = kc + 1;
                if (nk == p.nchunk) { nk = 0; nt = tile + gridDim.x; }
                if (nt < p.ntiles) load_regs(nt, nk, !w_resident);
            }
#pragma unroll
            for (int sy = 0; sy < 3; ++sy)
#pragma unroll
                for (int sx = 0; sx < 3; ++sx) {
                    // the tap that reads the patch at (sy, sx): forward (kh, kw) = (sy, sx); data gradient (2 - sy, 2 - sx)
                    const int tap_f = sy * 3 + sx, tap_d = (2 - sy) * 3 + (2 - sx);
                    const uint16_t* wt = wbase + (p.flip ? tap_d : tap_f) * CB * PITCH;
                    // patch slot of this tap relative to the pixel's: stride 2 reads input column 2 rj + sx = slot (sx & 1) * PWe + rj + (sx >> 1)
                    const int tslot = S == 2 ? sy * TWp + (sx & 1) * PWe + (sx >> 1) : sy * TWp + sx;
#pragma unroll
                    for (int ks = 0; ks < KC / 16; ++ks) {
                        const bf16x8_t wf = __builtin_bit_cast(bf16x8_t, *reinterpret_cast<const uint4*>(wt + ks * 16));
#pragma unroll
                        for (int pt = 0; pt < PT; ++pt) {
                            const bf16x8_t xf = __builtin_bit_cast(bf16x8_t, *reinterpret_cast<const uint4*>(xbase[pt] + tslot * PITCH + ks * 16));
                            acc[pt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf, xf, acc[pt], 0, 0, 0);
                        }
                    }
                }
        }

        // ---- epilogue: the tile through the staging buffer, stored as whole rows
        int b, i0, j0;
        tile_origin(tile, b, i0, j0);
        const int cch = t % CPR;
        const int n = c0 + cch * 8;
        int oidx[NOI];
        uint4 zpre[EPI == 3 ? NOI : 1];                // EPI 3: the producer's z chunks are requested before the barriers of the staging
#pragma unroll
        for (int it = 0; it < NOI; ++it) {
            const int r = t / CPR + it * (NT / CPR);
            const int i = r / p.TW, j = r - i * p.TW;
            const int gi = i0 + i, gj = j0 + j;
            const bool ok = r < p.TH * p.TW && gi < p.H && gj < p.W && n < p.N;
            oidx[it] = ok ? (b * p.H + gi) * p.W + gj : -1;
            if (EPI == 3) {
                zpre[it] = make_uint4(0, 0, 0, 0);
                if (ok) zpre[it] = p3_ld_nt16(p.z + (size_t)oidx[it] * p.ldz + n);
            }
        }
        __syncthreads();                               // fragment reads of the last channel block are done
#pragma unroll
        for (int pt = 0; pt < PT; ++pt) {
            uint16_t* dst = sStg + rpx[pt] * SP + wct * 32 + 4 * (lane >> 5);
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                uint2 v;
                v.x = pack2(acc[pt][4 * q + 0], acc[pt][4 * q + 1]);
                v.y = pack2(acc[pt][4 * q + 2], acc[pt][4 * q + 3]);
                *reinterpret_cast<uint2*>(dst + 8 * q) = v;
            }
        }
        __syncthreads();
#pragma unroll
        for (int it = 0; it < NOI; ++it) {
            if (oidx[it] < 0) continue;
            const int r = t / CPR + it * (NT / CPR);
            uint4 v = *reinterpret_cast<const uint4*>(sStg + r * SP + cch * 8);
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
            if (EPI == 3) p3_st_nt16(dst, v);
            else *reinterpret_cast<uint4*>(dst) = v;
            if (EPI == 1) {
                float f[8];
                unpack8(v, f);
#pragma unroll
                for (int e = 0; e < 8; ++e) { bs_[e] += f[e]; bq_[e] += f[e] * f[e]; }
            }
            if (EPI == 3) {
                float g[8], z[8];
                unpack8(v, g);
                unpack8(zpre[it], z);
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

    if (EPI == 1 || EPI == 3) {
        float* sRed = reinterpret_cast<float*>(smem);          // [NT][16]: aliases the idle patch / weight buffers
        __syncthreads();
#pragma unroll
        for (int e = 0; e < 8; ++e) { sRed[t * 16 + e] = bs_[e]; sRed[t * 16 + 8 + e] = bq_[e]; }
        __syncthreads();
        for (int i = t; i < 2 * CB; i += NT) {
            const int which = i / CB, c = i - which * CB;
            float v = 0.f;
            for (int j = c / 8; j < NT; j += CPR) v += sRed[j * 16 + which * 8 + (c & 7)];
            if (EPI == 1) { if (c0 + c < p.Npad) p.stats[((size_t)blockIdx.x * 2 + which) * p.Npad + c0 + c] = (c0 + c < p.N) ? v : 0.f; }
            else if (c0 + c < p.N) p.part[((size_t)blockIdx.x * 2 + which) * p.N + c0 + c] = v;
        }
    }
}

bool p3_geom(int H, int W, int pt, int* TH, int* TW, int* tx, int* ty, int s = 1)
{
    double best = -1.0;
    const int rpx = 128 * pt;
    for (int tw = 4; tw <= 64; ++tw)
        for (int th = 1; th * tw <= rpx; ++th) {
            const int ph = s == 2 ? 2 * th + 1 : th + 2, pw = s == 2 ? 2 * tw + 1 : tw + 2;
            if (ph * pw > p3_pmax(pt, s)) continue;
            const int nx = (W + tw - 1) / tw, ny = (H + th - 1) / th;
            double eff = (double)H * W / ((double)nx * ny * rpx);
            eff *= (double)(s * s * th * tw) / (ph * pw);           // halo overhead of the patch
            eff += 1e-4 * tw;
            if (eff > best) { best = eff; *TH = th; *TW = tw; *tx = nx; *ty = ny; }
        }
    return best > 0.0;
}

struct P3Plan { int ct, kc, pt, gx, gy, s; P3K k; };

bool p3_plan(const yh_conv_desc* d, P3Plan* pl)
{
    if (d->nseg != 1 || d->seg[0].ups) return false;
    if (d->KH != 3 || d->KW != 3 || d->pad != 1) return false;
    const int S = d->stride;
    if (S == 1) { if (d->Ho != d->Hi || d->Wo != d->Wi) return false; }
    else if (S == 2) {      // the downsampling layers, forward only (their data gradient is conv_dg2_kernel's)
        if (d->mode != YH_CONV_FWD || d->Ho != (d->Hi - 1) / 2 + 1 || d->Wo != (d->Wi - 1) / 2 + 1) return false;
    } else return false;
    pl->s = S;
    if (d->bias || d->scale || d->shift || d->act != YH_ACT_NONE || d->res || d->nsplit < d->N) return false;
    if (d->mode == YH_CONV_FWD && (d->bnr_part || d->accumulate)) return false;
    if (d->mode == YH_CONV_DGRAD && d->stats) return false;
    const int Cin = d->seg[0].C;
    if (Cin % 32 || Cin > 128 || d->N % 8 || d->N > 128) return false;         // the small-channel layers this kernel is for
    const unsigned long xb = ((unsigned long)d->B * d->Hi * d->Wi - 1) * d->seg[0].ld * 2 + (unsigned long)Cin * 2;
    const unsigned long wb = (unsigned long)d->Npad * 9 * Cin * 2;
    if (xb >= (1ul << 31) || wb >= (1ul << 31) || (unsigned long)d->B * d->Ho * d->Wo >= (1ul << 31)) return false;
    P3K& k = pl->k;
    pl->ct = d->N <= 32 ? 1 : 2;
    pl->kc = (Cin % 64 == 0 && d->tile_k != 32 && S == 1) ? 64 : 32;      // stride 2: the patch of 64-channel blocks does not fit beside the weights
    pl->pt = (d->tile_n == 32) ? 1 : 2;
    if (!p3_geom(d->Ho, d->Wo, pl->pt, &k.TH, &k.TW, &k.tx, &k.ty, S)) return false;
    const int cb = 32 * pl->ct;
    pl->gy = (d->N + cb - 1) / cb;
    if (pl->gy * cb > d->Npad) return false;
    k.x = d->seg[0].ptr; k.ldx = d->seg[0].ld; k.Cin = Cin;
    k.w = d->w; k.Ktot = 9 * Cin;
    k.out = d->out0; k.ld0 = d->ld0; k.N = d->N; k.Npad = d->Npad;
    k.B = d->B; k.H = d->Ho; k.W = d->Wo; k.Hi = d->Hi; k.Wi = d->Wi;
    k.ntiles = d->B * k.tx * k.ty;
    k.nchunk = Cin / pl->kc;
    k.PW = S == 2 ? 2 * k.TW + 1 : k.TW + 2;
    k.npatch = (S == 2 ? 2 * k.TH + 1 : k.TH + 2) * k.PW;
    k.accumulate = d->accumulate;
    k.flip = d->mode == YH_CONV_DGRAD ? 1 : 0;
    k.stats = d->stats;
    k.z = d->bnr_z; k.ldz = d->bnr_ldz; k.ws = d->bnr_ws; k.wsC = d->bnr_C; k.part = d->bnr_part;
    k.xbytes = (unsigned)xb; k.wbytes = (unsigned)wb;
    // resident blocks per CU by LDS: the grid is one resident wave of blocks (bounds the partial-sum rows too)
    const int smem = S == 2 ? ((pl->ct == 1) ? (pl->pt == 2 ? p3_smem_bytes<1, 32, 2, 2>() : p3_smem_bytes<1, 32, 1, 2>())
                                             : (pl->pt == 2 ? p3_smem_bytes<2, 32, 2, 2>() : p3_smem_bytes<2, 32, 1, 2>()))
                   : (pl->ct == 1) ? (pl->kc == 64 ? (pl->pt == 2 ? p3_smem_bytes<1, 64, 2>() : p3_smem_bytes<1, 64, 1>())
                                                   : (pl->pt == 2 ? p3_smem_bytes<1, 32, 2>() : p3_smem_bytes<1, 32, 1>()))
                                   : (pl->kc == 64 ? (pl->pt == 2 ? p3_smem_bytes<2, 64, 2>() : p3_smem_bytes<2, 64, 1>())
                                                   : (pl->pt == 2 ? p3_smem_bytes<2, 32, 2>() : p3_smem_bytes<2, 32, 1>()));
    int occ = (160 * 1024) / smem;
    if (occ > 4) occ = 4;
    if (occ < 1) occ = 1;
    int cap = (256 * occ) / pl->gy;
    if (cap < 1) cap = 1;
    if (d->grid_cap > 0) cap = d->grid_cap;
    pl->gx = k.ntiles < cap ? k.ntiles : cap;
    return true;
}

}  // namespace

int yh_p3_rows(const yh_conv_desc* d)
{
    P3Plan pl;
    return p3_plan(d, &pl) ? pl.gx : 0;
}

int yh_p3_run(const yh_conv_desc* d, yh_stream stream, char* name_out, int name_len)
{
    P3Plan pl;
    YH_CHECK_ARG(p3_plan(d, &pl), "yh_conv_igemm: algo 8 (3x3 patch kernel) is not eligible for this descriptor");
    const int epi = d->bnr_part ? 3 : (d->stats ? 1 : 0);
    if (d->bnr_part)
        YH_CHECK_ARG(d->bnr_z && yh_aligned16(d->bnr_z) && d->bnr_ldz % 8 == 0 && d->bnr_ws && d->bnr_C >= d->N, "yh_conv_igemm: bad fused-reduction operands");
    if (name_out) {
        snprintf(name_out, name_len, "conv_p3_kernel<%d, %d, %d, %d, %d>", pl.ct, pl.kc, epi, pl.pt, pl.s);     // as profilers print it
        return YH_OK;
    }
    hipStream_t st = (hipStream_t)stream;
    const dim3 grid(pl.gx, pl.gy), blk(256 * pl.ct);
#define YH_LAUNCH_P3(CT_, KC_, PT_)                                                                                    \
    do {                                                                                                               \
        const int sm = p3_smem_bytes<CT_, KC_, PT_>();                                                                 \
        static YhDevOnce attr_set;                                                                                        \
        if (attr_set.need()) {                                                                                               \
            attr_set.set((const void*)conv_p3_kernel<CT_, KC_, 0, PT_>, hipFuncAttributeMaxDynamicSharedMemorySize, sm); \
            attr_set.set((const void*)conv_p3_kernel<CT_, KC_, 1, PT_>, hipFuncAttributeMaxDynamicSharedMemorySize, sm); \
            attr_set.set((const void*)conv_p3_kernel<CT_, KC_, 3, PT_>, hipFuncAttributeMaxDynamicSharedMemorySize, sm); \
            attr_set.done();                                                                                            \
        }                                                                                                              \
        if (epi == 3)      conv_p3_kernel<CT_, KC_, 3, PT_><<<grid, blk, sm, st>>>(pl.k);                              \
        else if (epi == 1) conv_p3_kernel<CT_, KC_, 1, PT_><<<grid, blk, sm, st>>>(pl.k);                              \
        else               conv_p3_kernel<CT_, KC_, 0, PT_><<<grid, blk, sm, st>>>(pl.k);                              \
    } while (0)
#define YH_LAUNCH_P3S2(CT_, PT_)                                                                                        \
    do {                                                                                                               \
        const int sm = p3_smem_bytes<CT_, 32, PT_, 2>();                                                               \
        static YhDevOnce attr_set;                                                                                     \
        if (attr_set.need()) {                                                                                         \
            attr_set.set((const void*)conv_p3_kernel<CT_, 32, 0, PT_, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, sm); \
            attr_set.set((const void*)conv_p3_kernel<CT_, 32, 1, PT_, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, sm); \
            attr_set.done();                                                                                           \
        }                                                                                                              \
        if (epi == 1) conv_p3_kernel<CT_, 32, 1, PT_, 2><<<grid, blk, sm, st>>>(pl.k);                                 \
        else          conv_p3_kernel<CT_, 32, 0, PT_, 2><<<grid, blk, sm, st>>>(pl.k);                                 \
    } while (0)
#define YH_P3_PT(CT_, KC_) do { if (pl.pt == 2) YH_LAUNCH_P3(CT_, KC_, 2); else YH_LAUNCH_P3(CT_, KC_, 1); } while (0)
    if (pl.s == 2) {
        if (pl.ct == 1) { if (pl.pt == 2) YH_LAUNCH_P3S2(1, 2); else YH_LAUNCH_P3S2(1, 1); }
        else            { if (pl.pt == 2) YH_LAUNCH_P3S2(2, 2); else YH_LAUNCH_P3S2(2, 1); }
    }
    else if (pl.ct == 1) { if (pl.kc == 64) YH_P3_PT(1, 64); else YH_P3_PT(1, 32); }
    else                 { if (pl.kc == 64) YH_P3_PT(2, 64); else YH_P3_PT(2, 32); }
#undef YH_P3_PT
#undef YH_LAUNCH_P3S2
#undef YH_LAUNCH_P3
    YH_CHECK_LAUNCH("yh_conv_igemm(p3)");
    return YH_OK;
}
