// Weight gradient of the 3x3 layers with FEW channels on LARGE maps (YOLOv5 stem, stage-1 / stage-2 convs and bottlenecks) in
// "patch form":
//
//   dW[n][tap*Ctot + coff_k + c] += sum over the region's pixels m:  gy[m][n] * x[src(m, tap)][c]
//
// The im2col form (conv_wgrad.hip) gathers the nine taps of every pixel from global memory: 9x the loads of x, 18 separate
// 32-byte pieces per pixel on the stem.  Here a block owns a region of TH x 16 output pixels of one image, stages the gy tile and
// the input PATCH ((TH s + 2) x (16 s + 2) pixels, s = stride) ONCE in LDS in their natural [pixel][channel] layout, and forms
// the MFMA operands with transposing LDS reads (ds_read_b64_tr_b16: a 16-lane group reads 4 pixel rows x 16 channels, every
// lane supplying its own row address) — so the nine taps are nine ADDRESS OFFSETS into the same patch.  The (n-tile, tap,
// channel-tile) accumulator tiles of 32 x 32 are dealt over the block's four waves; blocks are persistent and keep their
// accumulators across all their regions, so the fp32 atomics of the epilogue happen once per block (N x 9C x 4 bytes), not once
// per split of 64 KB.  Chosen per layer by the engine's timing (yh_wgrad_desc.tile_k == 40).
// Replaces autograd's conv weight gradient (train_yolov5.py:337) for these layers.
#include "common.h"
#include <stdlib.h>

namespace {

typedef __attribute__((ext_vector_type(4))) short v4s;
typedef __attribute__((ext_vector_type(8))) short v8s;
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4_t;

__device__ __forceinline__ v4s wgp_tr(const uint16_t* p) {
    return __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) v4s*)(p));
}

#ifndef WGP_MINB
#define WGP_MINB 3
#endif
constexpr int WGP_TW = 16;            // region width: one 16-pixel MFMA k-group per region row

struct WgpK {
    const uint16_t* gy; int ldg, N;
    const uint16_t* x; int ldx, C;
    float* dw; int Ktot, Ctot, coff_k;
    int B, Ho, Wo, Hi, Wi, stride;
    int TH, tx, ty, ntiles, PW, npatch;
    int ntile_n, ntile_c, ntiles_acc, tpw;      // accumulator tiles: n-tiles x channel-tiles (over the taps), tiles per wave
    int ntaps, kw_, pad;                        // 9 taps (3x3, pad 1) or 1 (1x1, pad 0: the "patch" is the region itself)
    unsigned gybytes, xbytes;
    // fused BatchNorm+SiLU backward apply (conv_wgpf_kernel): gy holds ga, gz = A*dz + (Bc*z + D) is formed while the tile is staged
    const uint16_t* z; int ldz; const float* ws; const float* gamma; const float* coef; unsigned zbytes;
};

// PG / PX: LDS pitches (elements) of the gy tile and of the patch: 64 or 192 bytes mod 256 (conflict-free transposing reads)
constexpr int wgp_pitch(int cols) { const int r = (cols + 31) / 32 * 32; return (r % 64 == 32) ? r : r + 32; }

// NW waves per block (chosen so that the accumulator tiles divide evenly over them), TPW accumulator tiles per wave, NGI / NXI
// 16-byte chunks per thread of the gy tile / of the patch
template <int NW, int TPW, int NGI, int NXI, bool FBN>
__device__ __forceinline__ void wgp_body(const WgpK& p, const int PG, const int PX)
{
    constexpr int NT = 64 * NW;
    constexpr unsigned OOB = 0x80000000u;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    uint16_t* sG = reinterpret_cast<uint16_t*>(smem);             // [TH*16][PG]
    uint16_t* sX = sG + p.TH * WGP_TW * PG;                       // [npatch][PX]

    const int t = threadIdx.x;
    const int lane = t & 63;
    const int wave = t >> 6;
    const int s = p.stride;
    const __amdgpu_buffer_rsrc_t rsg = __builtin_amdgcn_make_buffer_rsrc((void*)p.gy, 0, p.gybytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsx = __builtin_amdgcn_make_buffer_rsrc((void*)p.x, 0, p.xbytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsz = __builtin_amdgcn_make_buffer_rsrc((void*)(FBN ? p.z : p.gy), 0, FBN ? p.zbytes : p.gybytes, 0x00020000);
    // FBN: gz = A*dz + (Bc*z + D), dz = ga*silu'(z*sc + sh) — bn_silu_bwd_apply_kernel's arithmetic (elementwise.hip), the five
    // per-channel constants in LDS behind the patch: sCst[5][64] = sc | sh | A | Bc | D
    float* const sCst = reinterpret_cast<float*>(sX + p.npatch * PX);
    if (FBN) {
        for (int c = t; c < 64; c += NT) {
            const int cc = c < p.N ? c : 0;
            const float mu = p.ws[2 * p.N + cc], is = p.ws[3 * p.N + cc];
            const float gi = p.gamma[cc] * is;
            const float c1 = p.coef[cc], c2 = p.coef[p.N + cc];
            sCst[c] = p.ws[cc]; sCst[64 + c] = p.ws[p.N + cc];
            sCst[128 + c] = gi;
            sCst[192 + c] = -gi * is * c2;
            sCst[256 + c] = gi * (mu * is * c2 - c1);
        }
    }

    // ---- staging maps: 16-byte chunks of the gy tile / the patch, dealt over the threads
    const int gch = (p.N + 7) / 8, xch = p.C / 8;               // chunks per pixel
    const int ng_items = p.TH * WGP_TW * gch, nx_items = p.npatch * xch;

    // ---- fragment addressing (see tr_read in conv_wgrad.hip): lane 4q+pp of a 16-lane group supplies row q, columns 4pp..4pp+3;
    // the group g16 covers rows 8 (g16 >> 1) + {0..3} (+4 for the second read) and columns 16 (g16 & 1) .. +16 of the 32-wide tile
    const int g16 = lane >> 4, i16 = lane & 15;
    const int q = i16 >> 2, pp = i16 & 3;
    const int prow = 8 * (g16 >> 1) + q;                        // pixel of the k-group this lane addresses (second read: + 4)
    const int chalf = 16 * (g16 & 1) + 4 * pp;                  // column inside the 32-wide tile

    // this wave's accumulator tiles, dealt round-robin: tile id = k * NW + wave  ->  (n-tile, channel-tile over the taps)
    int a_off[TPW], b_off[TPW];                                 // LDS element offsets of the lane's A / B row 0 for pixel 0 of a group
    bool live[TPW];
#pragma unroll
    for (int k = 0; k < TPW; ++k) {
        const int tile = k * NW + wave;
        live[k] = tile < p.ntiles_acc;                          // wave-uniform
        const int nt = live[k] ? tile / p.ntile_c : 0;
        const int tc = live[k] ? tile - nt * p.ntile_c : 0;
        a_off[k] = prow * PG + nt * 32 + chalf;
        // B: column (tc * 32 + chalf) of the (tap, c) axis: tap = col / C, c = col % C (a 16-column half never straddles a tap: C >= 16)
        const int col = tc * 32 + chalf;
        const int tap = col / p.C, c = col - tap * p.C;
        // a 16-column half past the last tap (C = 16: 9 taps fill 4.5 tiles) reads tap 8 again: its columns are dropped in the epilogue
        const int tp = tap < p.ntaps ? tap : p.ntaps - 1;
        const int kh = tp / p.kw_, kw = tp - kh * p.kw_;
        b_off[k] = (kh * p.PW + kw + prow * s) * PX + c;
    }

    f32x16_t acc[TPW];
#pragma unroll
    for (int k = 0; k < TPW; ++k)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[k][r] = 0.f;

    u32x4_t rg[NGI], rx[NXI], rz[FBN ? NGI : 1];
    unsigned gok = 0;                                           // FBN: which of this thread's gy chunks are pixels of the map
    auto tile_origin = [&](int tile, int& b, int& i0, int& j0) {
        const int per = p.tx * p.ty;
        b = tile / per;
        const int r = tile - b * per;
        const int tyi = r / p.tx;
        i0 = tyi * p.TH; j0 = (r - tyi * p.tx) * WGP_TW;
    };
    auto load_regs = [&](int tile) {
        int b, i0, j0;
        tile_origin(tile, b, i0, j0);
#pragma unroll
        for (int j = 0; j < NGI; ++j) {
            const int id = t + j * NT;
            const int px = id / gch, chn = id - px * gch;
            const int i = px / WGP_TW, jj = px - i * WGP_TW;
            const int gi = i0 + i, gj = j0 + jj;
            const bool ok = id < ng_items && gi < p.Ho && gj < p.Wo;
            rg[j] = __builtin_amdgcn_raw_buffer_load_b128(rsg, ok ? (unsigned)(((b * p.Ho + gi) * p.Wo + gj) * (p.ldg * 2) + chn * 16) : OOB, 0, 0);
            if (FBN) {
                rz[j] = __builtin_amdgcn_raw_buffer_load_b128(rsz, ok ? (unsigned)(((b * p.Ho + gi) * p.Wo + gj) * (p.ldz * 2) + chn * 16) : OOB, 0, 0);
                gok = (gok & ~(1u << j)) | ((ok ? 1u : 0u) << j);
            }
        }
        const int pi0 = i0 * s - p.pad, pj0 = j0 * s - p.pad;
#pragma unroll
        for (int j = 0; j < NXI; ++j) {
            const int id = t + j * NT;
            const int px = id / xch, chn = id - px * xch;
            const int pi = px / p.PW, pj = px - pi * p.PW;
            const int gi = pi0 + pi, gj = pj0 + pj;
            const bool ok = id < nx_items && gi >= 0 && gj >= 0 && gi < p.Hi && gj < p.Wi;
            rx[j] = __builtin_amdgcn_raw_buffer_load_b128(rsx, ok ? (unsigned)(((b * p.Hi + gi) * p.Wi + gj) * (p.ldx * 2) + chn * 16) : OOB, 0, 2);
        }
    };
    auto store_regs = [&]() {
#pragma unroll
        for (int j = 0; j < NGI; ++j) {
            const int id = t + j * NT;
            if (id < ng_items) {
                const int px = id / gch, chn = id - px * gch;
                u32x4_t v = rg[j];
                if (FBN) {
                    float g[8], z[8], o[8];
                    unpack8(make_uint4(rg[j].x, rg[j].y, rg[j].z, rg[j].w), g);
                    unpack8(make_uint4(rz[j].x, rz[j].y, rz[j].z, rz[j].w), z);
#pragma unroll
                    for (int e = 0; e < 8; ++e) {
                        const int c = chn * 8 + e;
                        const float a = z[e] * sCst[c] + sCst[64 + c];
                        const float sg = sigmoid_fast(a);
                        const float dz = g[e] * (sg * (1.f + a * (1.f - sg)));
                        o[e] = sCst[128 + c] * dz + (sCst[192 + c] * z[e] + sCst[256 + c]);
                    }
                    // pixels past the map / padded channels were loaded as zeros: dz = 0, but Bc*0 + D is not — they stay zero
                    const uint4 w = ((gok >> j) & 1u) ? pack8(o) : make_uint4(0, 0, 0, 0);
                    v = u32x4_t{w.x, w.y, w.z, w.w};
                    if (chn * 8 + 8 > p.N) {                    // ragged last chunk: channels >= N stay zero
                        uint16_t* q = reinterpret_cast<uint16_t*>(&v);
                        for (int e = 0; e < 8; ++e) if (chn * 8 + e >= p.N) q[e] = 0;
                    }
                }
                *reinterpret_cast<u32x4_t*>(sG + px * PG + chn * 8) = v;
            }
        }
#pragma unroll
        for (int j = 0; j < NXI; ++j) {
            const int id = t + j * NT;
            if (id < nx_items) { const int px = id / xch, chn = id - px * xch; *reinterpret_cast<u32x4_t*>(sX + px * PX + chn * 8) = rx[j]; }
        }
    };

    int tile = blockIdx.x;
    if (tile < p.ntiles) load_regs(tile);
    for (; tile < p.ntiles; tile += gridDim.x) {
        __syncthreads();                               // the previous region's fragment reads are done
        store_regs();
        __syncthreads();
        if (tile + (int)gridDim.x < p.ntiles) load_regs(tile + gridDim.x);
        for (int g = 0; g < p.TH; ++g) {               // one k-group = the 16 pixels of region row g
            const uint16_t* ga = sG + g * WGP_TW * PG;
            const uint16_t* xa = sX + g * s * p.PW * PX;
            int last_nt = -1;
            bf16x8_t af = {};
#pragma unroll
            for (int k = 0; k < TPW; ++k) {
                if (!live[k]) continue;
                const int nt = (k * NW + wave) / p.ntile_c;
                if (nt != last_nt) {                   // consecutive tiles of a wave mostly share their n-tile
                    const v4s lo = wgp_tr(ga + a_off[k]);
                    const v4s hi = wgp_tr(ga + a_off[k] + 4 * PG);
                    const v8s v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
                    af = __builtin_bit_cast(bf16x8_t, v);
                    last_nt = nt;
                }
                const v4s lo = wgp_tr(xa + b_off[k]);
                const v4s hi = wgp_tr(xa + b_off[k] + 4 * s * PX);
                const v8s v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
                acc[k] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af, __builtin_bit_cast(bf16x8_t, v), acc[k], 0, 0, 0);
            }
        }
    }

    // ---- epilogue, once per block: C[n][col]: lane holds column (lane & 31), rows (r & 3) + 8 (r >> 2) + 4 (lane >> 5)
#pragma unroll
    for (int k = 0; k < TPW; ++k) {
        if (!live[k]) continue;
        const int tile_id = k * NW + wave;
        const int nt = tile_id / p.ntile_c, tc = tile_id - nt * p.ntile_c;
        const int col = tc * 32 + (lane & 31);
        const int tap = col / p.C, c = col - tap * p.C;
        if (tap >= p.ntaps) continue;
        float* base = p.dw + (size_t)tap * p.Ctot + p.coff_k + c;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int n = nt * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
            if (n < p.N) atomicAdd(base + (size_t)n * p.Ktot, acc[k][r]);
        }
    }
}

template <int NW, int TPW, int NGI, int NXI>
__global__ __launch_bounds__(64 * NW, (NW >= 8 ? 1 : (TPW <= 2 ? WGP_MINB : 2))) void conv_wgp_kernel(const WgpK p, const int PG, const int PX)
{
    wgp_body<NW, TPW, NGI, NXI, false>(p, PG, PX);
}
// the stem's form: BatchNorm+SiLU backward apply inside the staging of the gy tile (yh_wgrad_desc.bn_z)
template <int NW, int TPW, int NGI, int NXI>
__global__ __launch_bounds__(64 * NW, (NW >= 8 ? 1 : (TPW <= 2 ? WGP_MINB : 2))) void conv_wgpf_kernel(const WgpK p, const int PG, const int PX)
{
    wgp_body<NW, TPW, NGI, NXI, true>(p, PG, PX);
}

struct WgpPlan { WgpK k; int nw, tpw, ngi, nxi, PG, PX, gx, smem; };

bool wgp_plan(const yh_wgrad_desc* d, WgpPlan* pl)
{
    const bool k3 = d->KH == 3 && d->KW == 3 && d->pad == 1 && (d->stride == 1 || d->stride == 2);
    const bool k1 = d->KH == 1 && d->KW == 1 && d->pad == 0 && d->stride == 1;
    if (!(k3 || k1) || d->seg.ups) return false;
    if (d->partial) return false;
    const int C = d->seg.C, N = d->N;
    if (!(C == 16 || C == 32 || C == 64) || N < 8 || N > 64) return false;      // staging registers: <= 8 gy chunks per pixel
    if (d->stride == 2 && C > 32) return false;                                    // ... and <= 11 patch chunks per thread
    if (d->stride == 2 && (d->Hi != 2 * d->Ho || d->Wi != 2 * d->Wo)) return false;
    if (d->stride == 1 && (d->Hi != d->Ho || d->Wi != d->Wo)) return false;
    const unsigned long gyb = ((unsigned long)d->B * d->Ho * d->Wo - 1) * d->ldg * 2 + (unsigned long)((N + 7) / 8 * 8) * 2;
    const unsigned long xb = ((unsigned long)d->B * d->Hi * d->Wi - 1) * d->seg.ld * 2 + (unsigned long)C * 2;
    if (gyb >= (1ul << 31) || xb >= (1ul << 31)) return false;
    WgpK& k = pl->k;
    const int s = d->stride;
    k.TH = s == 1 ? 16 : 8;                             // 256 / 128 region pixels; patch 18 x 18 / 17 x 33 + 1
    if (k.TH > d->Ho) k.TH = d->Ho;
    k.ntaps = k3 ? 9 : 1; k.kw_ = k3 ? 3 : 1; k.pad = d->pad;
    k.PW = WGP_TW * s + 2 * d->pad;
    k.npatch = (k.TH * s + 2 * d->pad) * k.PW;
    k.tx = (d->Wo + WGP_TW - 1) / WGP_TW; k.ty = (d->Ho + k.TH - 1) / k.TH;
    k.ntiles = d->B * k.tx * k.ty;
    k.ntile_n = (N + 31) / 32;
    k.ntile_c = (k.ntaps * C + 31) / 32;
    k.ntiles_acc = k.ntile_n * k.ntile_c;
    // waves per block: five where the accumulator tiles divide evenly over them (the stems: 5 tiles = 5 x 1, 10 = 5 x 2: 172 ->
    // 151 us on YOLOv5s), else four (measured: 3 waves x 3 tiles for the 9-tile layers equal, 6 x 3 for the 18-tile layer 20 % slower)
    const int na = k.ntiles_acc;
    const int nw = (na == 5 || na == 10) ? 5 : (na > 20 ? 8 : 4);      // > 20 tiles (64 -> 64, 3x3): eight waves x 5
    if (d->bn_z && nw != 5) return false;               // fused BatchNorm backward: the stem shapes only
    pl->nw = nw;
    k.tpw = (na + nw - 1) / nw;
    if (k.tpw > 5) return false;                        // <= 5 x 16 accumulator registers per lane
    pl->tpw = k.tpw;
    const int nt = 64 * nw;
    pl->ngi = (k.TH * WGP_TW * ((N + 7) / 8) + nt - 1) / nt;     // 16-byte chunks per thread: gy tile, patch
    pl->nxi = (k.npatch * (C / 8) + nt - 1) / nt;
    k.gy = d->gy; k.ldg = d->ldg; k.N = N;
    k.x = d->seg.ptr; k.ldx = d->seg.ld; k.C = C;
    k.dw = d->dw; k.Ktot = k.ntaps * d->Ctot; k.Ctot = d->Ctot; k.coff_k = d->coff_k;
    k.B = d->B; k.Ho = d->Ho; k.Wo = d->Wo; k.Hi = d->Hi; k.Wi = d->Wi; k.stride = s;
    k.gybytes = (unsigned)gyb; k.xbytes = (unsigned)xb;
    k.z = d->bn_z; k.ldz = d->bn_ldz; k.ws = d->bn_ws; k.gamma = d->bn_gamma; k.coef = d->bn_coef; k.zbytes = 0;
    if (d->bn_z) {
        if (!d->bn_ws || !d->bn_gamma || !d->bn_coef || d->bn_ldz % 8 || (((uintptr_t)d->bn_z) & 15)) return false;
        const unsigned long zb = ((unsigned long)d->B * d->Ho * d->Wo - 1) * d->bn_ldz * 2 + (unsigned long)((N + 7) / 8 * 8) * 2;
        if (zb >= (1ul << 31)) return false;
        k.zbytes = (unsigned)zb;
    }
    pl->PG = wgp_pitch(k.ntile_n * 32);
    pl->PX = wgp_pitch(C < 32 ? 32 : C);
    pl->smem = (k.TH * WGP_TW * pl->PG + k.npatch * pl->PX) * 2 + (d->bn_z ? 5 * 64 * 4 : 0);
    int occ = (150 * 1024) / pl->smem;
    const int occ_max = nw >= 8 ? 1 : (k.tpw <= 2 ? WGP_MINB : 2);      // __launch_bounds__
    if (occ > occ_max) occ = occ_max;
    if (occ < 1) return false;
    int gx = 256 * occ;
    if (d->splits > 0 && d->splits < gx) gx = d->splits;   // `splits` caps the persistent blocks (each adds one set of atomics)
    pl->gx = k.ntiles < gx ? k.ntiles : gx;
    return true;
}

}  // namespace

// instantiation table shared by the launcher and the name query: (accumulator tiles per wave, gy chunks, patch chunks per thread)
// (waves, accumulator tiles per wave, gy chunks, patch chunks per thread)
static const int kWgpInst[7][4] = {{5, 1, 4, 3}, {5, 2, 7, 3}, {4, 1, 8, 8}, {4, 3, 4, 6}, {4, 5, 4, 10}, {4, 5, 8, 11}, {8, 5, 4, 6}};
static int wgp_pick(const WgpPlan& pl)
{
    for (int i = 0; i < 7; ++i)
        if (pl.nw == kWgpInst[i][0] && pl.tpw <= kWgpInst[i][1] && pl.ngi <= kWgpInst[i][2] && pl.nxi <= kWgpInst[i][3]) return i;
    return -1;
}
// eligibility == a plan AND an instantiation for it: the query, the name query and the launcher agree (C = 16 / stride 2 has a plan
// but no instance: yh_conv_wgrad then takes the im2col form instead of failing)
int yh_wgp_ok(const yh_wgrad_desc* d) { WgpPlan pl; return (wgp_plan(d, &pl) && wgp_pick(pl) >= 0) ? 1 : 0; }

/* profiler spelling of the instantiation the patch form launches for this descriptor ("" when it does not apply) */
extern "C" int yh_conv_wgrad_patch_name(const yh_wgrad_desc* d, char* buf, int buflen)
{
    WgpPlan pl;
    if (!buf || buflen < 40) return YH_EINVAL;
    buf[0] = 0;
    if (!d || !wgp_plan(d, &pl)) return YH_OK;
    const int i = wgp_pick(pl);
    if (i >= 0) snprintf(buf, buflen, "conv_wgp%s_kernel<%d, %d, %d, %d>", d->bn_z ? "f" : "", kWgpInst[i][0], kWgpInst[i][1], kWgpInst[i][2], kWgpInst[i][3]);
    return YH_OK;
}

int yh_wgp_run(const yh_wgrad_desc* d, yh_stream stream)
{
    WgpPlan pl;
    YH_CHECK_ARG(wgp_plan(d, &pl), "yh_conv_wgrad: the patch form (tile_k 40) is not eligible for this layer");
    hipStream_t st = (hipStream_t)stream;
    const int inst = wgp_pick(pl);
    YH_CHECK_ARG(inst >= 0, "yh_conv_wgrad: no patch-form instantiation for this layer");
#define YH_TRY_WGP(I_, NW_, TPW_, NGI_, NXI_)                                                                     \
    if (inst == I_) {                                                                                             \
        static YhDevOnce attr_set;                                                                                   \
        if (attr_set.need()) { attr_set.set((const void*)conv_wgp_kernel<NW_, TPW_, NGI_, NXI_>, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024); attr_set.done();  } \
        conv_wgp_kernel<NW_, TPW_, NGI_, NXI_><<<dim3(pl.gx), dim3(64 * NW_), pl.smem, st>>>(pl.k, pl.PG, pl.PX);  \
    }
#define YH_TRY_WGPF(I_, NW_, TPW_, NGI_, NXI_)                                                                    \
    if (inst == I_ && d->bn_z) {                                                                                  \
        static YhDevOnce attr_set;                                                                                   \
        if (attr_set.need()) { attr_set.set((const void*)conv_wgpf_kernel<NW_, TPW_, NGI_, NXI_>, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024); attr_set.done();  } \
        conv_wgpf_kernel<NW_, TPW_, NGI_, NXI_><<<dim3(pl.gx), dim3(64 * NW_), pl.smem, st>>>(pl.k, pl.PG, pl.PX); \
        YH_CHECK_LAUNCH("yh_conv_wgrad(patch, fused BatchNorm backward)");                                        \
        return YH_OK;                                                                                             \
    }
    YH_TRY_WGPF(0, 5, 1, 4, 3)
    YH_TRY_WGPF(1, 5, 2, 7, 3)
#undef YH_TRY_WGPF
    YH_CHECK_ARG(!d->bn_z, "yh_conv_wgrad: the patch form fuses the BatchNorm backward for the stem shapes only");
    YH_TRY_WGP(0, 5, 1, 4, 3)        // stem, <= 32 outputs (C = 16: 5 tiles, one per wave)
    YH_TRY_WGP(1, 5, 2, 7, 3)        // stem, 64 outputs (10 tiles)
    YH_TRY_WGP(2, 4, 1, 8, 8)        // 1x1 layers: <= 64 channels on both sides (<= 4 tiles)
    YH_TRY_WGP(3, 4, 3, 4, 6)        // 32 -> 32, 3x3 stride 1 (9 tiles)
    YH_TRY_WGP(4, 4, 5, 4, 10)       // 32 -> 64, 3x3 stride 2 (18 tiles)
    YH_TRY_WGP(5, 4, 5, 8, 11)       // the rest on four waves (32 -> 64 stride 1, 64 -> 32)
    YH_TRY_WGP(6, 8, 5, 4, 6)        // 64 -> 64, 3x3 stride 1 (36 tiles: eight waves)
#undef YH_TRY_WGP
    YH_CHECK_LAUNCH("yh_conv_wgrad(patch)");
    return YH_OK;
}
