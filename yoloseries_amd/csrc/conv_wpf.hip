// Forward convolution / data gradient (stride 1: 3x3 pad 1 or 1x1) for the K-heavy layers, built like conv_wgs.hip (round 4):
//
//   out[m][n] = sum over taps, c:  x[pixel(m) + shift(tap)][c] * W[n][tap*C + c]
//
//   * ONE workgroup of four waves per CU, one wave per SIMD: every wave owns 128 consecutive output pixels x 128 output channels
//     (256 accumulator registers, MFMA 32x32x16 with swapped operands D = W X^T: a lane holds a pixel, its registers four
//     consecutive channels at a time);
//   * a k-stage is 32 input channels of one tap: the weight tile [128 n][64 B] is SHARED by the four waves (each loads a quarter),
//     the im2col rows [128 px][64 B] are PRIVATE to a wave — 320 bytes of LDS-DMA per MFMA instead of the 384 of a 256 x 128 tile
//     shared by eight waves, one ds_read_b128 per TWO MFMAs (8 fragments feed 16 MFMAs) instead of one per MFMA;
//   * ring of three stages, every wave waits for its own transfers by a counted vmcnt (one stage stays in flight), ONE raw
//     s_barrier per stage (32 MFMAs per wave) publishes the weight quarters and frees the slot consumed a stage earlier;
//     stages past the end are dummies of out-of-range offsets: every step issues the same 10 transfers;
//   * LDS rows of 64 B are unpadded (DMA writes lane * 16 B); the chunk index is XORed with (row >> 2) & 3 on the source side of
//     the DMA and on the fragment reads (conflict-free ds_read_b128);
//   * epilogue through LDS: [pixel][channel] rows of the wave's tile, stored as whole 256-byte rows.  Epilogues: 0 plain /
//     accumulating store, 1 + BatchNorm partial sums of the stored values (forward), 3 fused BatchNorm+SiLU backward reduction
//     (data gradient; yh_conv_desc.bnr_*).
// Chosen per layer by the engine's timing (yh_conv_desc.algo 11).  Replaces nn.Conv2d forward / backward-data of
// utils/layer_tools.py:82-94 for the 3x3 / 1x1 layers with >= 128 channels.
#include "common.h"
#include <stdlib.h>

namespace {

typedef unsigned int u32x4_nt __attribute__((ext_vector_type(4)));
__device__ __forceinline__ uint4 wpf_ld_nt16(const uint16_t* p) {
    u32x4_nt v = __builtin_nontemporal_load(reinterpret_cast<const u32x4_nt*>(p));
    return make_uint4(v.x, v.y, v.z, v.w);
}

constexpr int WPF_STG = 3;
constexpr int WPF_STAGE = 40960;                     // W [128][64 B] | X of wave 0..3 [128][64 B] each
static_assert(WPF_STG * WPF_STAGE <= 4 * 128 * 132 * 2, "the ring lies inside the staging buffer");
constexpr int WPF_SP = 132;                          // staging pitch in bf16 elements (264 B: two-way conflicts at most)
constexpr int WPF_STAGING = 4 * 128 * WPF_SP * 2;    // 135168 B: the four waves' [128 px][128 n] tiles
constexpr int WPF_STAT_OFF = WPF_STAGING;            // [4 waves][2][128] floats behind it
constexpr int WPF_LDS = WPF_STAGING + 4 * 2 * 128 * 4 + 2 * 128 * 4;
constexpr unsigned WPF_OOB = 0x80000000u;

struct WpfK {
    const uint16_t* x; int ldx, Cin;
    const uint16_t* w; int Ktot;
    uint16_t* out; int ld0, N, Npad;
    int B, H, W, M;
    int taps, pad, flip;                   // taps = 9 | 1; flip: data gradient (tap t reads the shift of tap taps-1-t)
    int ncb, nkt, mtiles, gx;              // 32-channel blocks per tap, stages, 512-pixel tiles, persistent workgroups per n-tile
    int accumulate;
    float* stats;                          // EPI 1: [gx][2][Npad]
    const uint16_t* z; int ldz; const float* ws; int wsC; float* part;      // EPI 3: [gx][2][N]
    unsigned xbytes, wbytes;
};

__device__ __forceinline__ void wpf_dma(unsigned lds, unsigned voff, const __amdgpu_buffer_rsrc_t rs, int soff) {
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %3 offen lds" :: "s"(lds), "v"(voff), "s"(rs), "s"(soff) : "memory");
}

template <int EPI>
__global__ __launch_bounds__(256, 1) void conv_wpf_kernel(const WpfK p)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int n0 = blockIdx.y * 128;
    const unsigned lbase = (unsigned)(size_t)((__attribute__((address_space(3))) unsigned char*)smem);

    const __amdgpu_buffer_rsrc_t rsx = __builtin_amdgcn_make_buffer_rsrc((void*)p.x, 0, p.xbytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsw = __builtin_amdgcn_make_buffer_rsrc((void*)p.w, 0, p.wbytes, 0x00020000);

    // loader: a transfer fills 16 rows x 64 B; this lane's row inside it and the source chunk its LDS position holds
    const int lr = lane >> 2;
    const int srcch = (lane & 3) ^ ((lane >> 4) & 3);          // (row >> 2) & 3 with row = 16 i + lr
    // weights: this wave's two transfers = rows 32 wave .. 32 wave + 31 of the tile
    unsigned voffW[2];
#pragma unroll
    for (int j = 0; j < 2; ++j)
        voffW[j] = (unsigned)(((n0 + 32 * wave + 16 * j + lr) * p.Ktot + srcch * 8) * 2);
    // fragment reads: lane (r = lane & 31, h = lane >> 5) reads row r of a 32-row block, chunk (2 ks + h) ^ ((r >> 2) & 3)
    const int r31 = lane & 31, h = lane >> 5;
    int rk[2];
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) rk[ks] = r31 * 64 + (((2 * ks + h) ^ ((r31 >> 2) & 3)) << 4);
    const int xreg = 8192 + wave * 8192;                       // this wave's X region inside a stage

    float bs_[8], bq_[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) { bs_[e] = 0.f; bq_[e] = 0.f; }
    float* const sWs = reinterpret_cast<float*>(smem + WPF_STAT_OFF + 4 * 2 * 128 * 4);      // EPI 3: scale | shift of the tile's channels
    if (EPI == 3) {
        for (int i = threadIdx.x; i < 2 * 128; i += 256) {
            const int which = i >> 7, c = i & 127;
            sWs[i] = (n0 + c < p.N) ? p.ws[(size_t)which * p.wsC + n0 + c] : 0.f;
        }
    }

    for (int mt = blockIdx.x; mt < p.mtiles; mt += p.gx) {
        const int m0 = mt * 512 + wave * 128;
        // pixel of this lane's row in each of the wave's 8 transfers: packed (h + 1) << 16 | (w + 1), image index; rows past M: -1
        int phw[8], pim[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int m = m0 + 16 * i + lr;
            phw[i] = -1; pim[i] = 0;
            if (m < p.M) {
                const int im = m / (p.H * p.W);
                const int rem = m - im * (p.H * p.W);
                const int hh = rem / p.W;
                phw[i] = ((hh + 1) << 16) | (rem - hh * p.W + 1);
                pim[i] = im;
            }
        }
        unsigned voffX[8];
        auto tap_offsets = [&](int tap) {                       // im2col row offsets of the wave's rows for one tap
            const int tp = p.flip ? p.taps - 1 - tap : tap;
            const int dy = p.taps == 9 ? tp / 3 - 1 : 0, dx = p.taps == 9 ? tp - (tp / 3) * 3 - 1 : 0;
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const int hh = (phw[i] >> 16) - 1 + dy, ww = (phw[i] & 0xffff) - 1 + dx;
                const bool ok = phw[i] >= 0 && (unsigned)hh < (unsigned)p.H && (unsigned)ww < (unsigned)p.W;
                voffX[i] = ok ? (unsigned)(((pim[i] * p.H + hh) * p.W + ww) * (p.ldx * 2) + srcch * 16) : WPF_OOB;
            }
        };

        f32x16_t acc[4][4];                 // [n block][pixel block]
#pragma unroll
        for (int x = 0; x < 4; ++x)
#pragma unroll
            for (int y = 0; y < 4; ++y)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[x][y][r] = 0.f;

        // stage s = (tap, channel block): the next one to issue is (itap, icb)
        int itap = 0, icb = 0;
        tap_offsets(0);
        auto issue = [&](int slot, unsigned oob) {
            const unsigned la = lbase + slot * WPF_STAGE;
            const int sw = (itap * p.Cin + icb * 32) * 2, sx = icb * 64;
            wpf_dma(la + (2 * wave) * 1024, voffW[0] | oob, rsw, sw);
            wpf_dma(la + (2 * wave + 1) * 1024, voffW[1] | oob, rsw, sw);
#pragma unroll
            for (int i = 0; i < 8; ++i) wpf_dma(la + xreg + i * 1024, voffX[i] | oob, rsx, sx);
            if (++icb == p.ncb) { icb = 0; ++itap; if (itap < p.taps) tap_offsets(itap); }
        };
        issue(0, 0u);
        issue(1, 1 < p.nkt ? 0u : WPF_OOB);
        int slot = 0;
        for (int k = 0; k < p.nkt; ++k) {
            YH_VMCNT(10);                                  // this wave's transfers of stage k have landed (stage k+1 stays in flight)
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();                  // stage k complete for every wave; the slot of stage k-1 is free
            __builtin_amdgcn_sched_barrier(0);
            const unsigned char* const sb = smem + slot * WPF_STAGE;
            bf16x8_t wf0[4], xf0[4], wf1[4], xf1[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                wf0[i] = __builtin_bit_cast(bf16x8_t, *reinterpret_cast<const uint4*>(sb + i * 2048 + rk[0]));
                xf0[i] = __builtin_bit_cast(bf16x8_t, *reinterpret_cast<const uint4*>(sb + xreg + i * 2048 + rk[0]));
            }
            {
                const int s2 = slot + 2 >= WPF_STG ? slot + 2 - WPF_STG : slot + 2;
                issue(s2, k + 2 < p.nkt ? 0u : WPF_OOB);
            }
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                wf1[i] = __builtin_bit_cast(bf16x8_t, *reinterpret_cast<const uint4*>(sb + i * 2048 + rk[1]));
                xf1[i] = __builtin_bit_cast(bf16x8_t, *reinterpret_cast<const uint4*>(sb + xreg + i * 2048 + rk[1]));
            }
#pragma unroll
            for (int x = 0; x < 4; ++x)
#pragma unroll
                for (int y = 0; y < 4; ++y) acc[x][y] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf0[x], xf0[y], acc[x][y], 0, 0, 0);
#pragma unroll
            for (int x = 0; x < 4; ++x)
#pragma unroll
                for (int y = 0; y < 4; ++y) acc[x][y] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf1[x], xf1[y], acc[x][y], 0, 0, 0);
            slot = slot + 1 == WPF_STG ? 0 : slot + 1;
        }
        YH_VMCNT(0);                                       // the dummies behind the last stage
        __syncthreads();                                   // every wave has left the ring: it becomes the staging buffer

        // ---- epilogue: C[n][pixel] (lane: pixel lane & 31 of block y; register r: channel (r & 3) + 8 (r >> 2) + 4 (lane >> 5) of
        // block x) -> this wave's [128 px][WPF_SP] staging rows -> whole 256-byte rows of out
        uint16_t* const stg = reinterpret_cast<uint16_t*>(smem) + wave * 128 * WPF_SP;
#pragma unroll
        for (int y = 0; y < 4; ++y) {
            uint16_t* const dst = stg + (32 * y + (lane & 31)) * WPF_SP + 4 * (lane >> 5);
#pragma unroll
            for (int x = 0; x < 4; ++x)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    uint2 v;
                    v.x = pack2(acc[x][y][4 * q + 0], acc[x][y][4 * q + 1]);
                    v.y = pack2(acc[x][y][4 * q + 2], acc[x][y][4 * q + 3]);
                    *reinterpret_cast<uint2*>(dst + 32 * x + 8 * q) = v;
                }
        }
        // read-out: lane = (row group lane >> 4, chunk lane & 15): rows (lane >> 4) + 4 it, it = 0 .. 31
        const int cch = lane & 15;
        const int n = n0 + cch * 8;
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // the wave reads back only its own rows: no barrier
#pragma unroll 4
        for (int it = 0; it < 32; ++it) {
            const int row = (lane >> 4) + 4 * it;
            const int m = m0 + row;
            if (m >= p.M || n >= p.N) continue;
            const uint2 va = *reinterpret_cast<const uint2*>(stg + row * WPF_SP + cch * 8);        // 264-byte pitch: 8-byte aligned
            const uint2 vb = *reinterpret_cast<const uint2*>(stg + row * WPF_SP + cch * 8 + 4);
            uint4 v = make_uint4(va.x, va.y, vb.x, vb.y);
            uint16_t* const dst = p.out + (size_t)m * p.ld0 + n;
            if (p.accumulate) {
                const uint4 ov = *reinterpret_cast<const uint4*>(dst);
                float f[8], g0[8];
                unpack8(v, f);
                unpack8(ov, g0);
#pragma unroll
                for (int e = 0; e < 8; ++e) f[e] += g0[e];
                v = pack8(f);
            }
            *reinterpret_cast<uint4*>(dst) = v;
            if (EPI == 1) {
                float f[8];
                unpack8(v, f);
#pragma unroll
                for (int e = 0; e < 8; ++e) { bs_[e] += f[e]; bq_[e] += f[e] * f[e]; }
            }
            if (EPI == 3) {
                float g[8], z[8];
                unpack8(v, g);
                unpack8(wpf_ld_nt16(p.z + (size_t)m * p.ldz + n), z);
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const float a = z[e] * sWs[cch * 8 + e] + sWs[128 + cch * 8 + e];
                    const float sg = sigmoid_fast(a);
                    const float dz = g[e] * (sg * (1.f + a * (1.f - sg)));
                    bs_[e] += dz; bq_[e] += dz * z[e];
                }
            }
        }
        __syncthreads();                                   // staging rows read: the ring may be refilled
    }

    if (EPI == 1 || EPI == 3) {
        // this thread's sums belong to channel chunk lane & 15: the four row groups of a wave meet by shuffles, the waves in LDS
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            bs_[e] += __shfl_xor(bs_[e], 16, 64); bs_[e] += __shfl_xor(bs_[e], 32, 64);
            bq_[e] += __shfl_xor(bq_[e], 16, 64); bq_[e] += __shfl_xor(bq_[e], 32, 64);
        }
        float* const sSt = reinterpret_cast<float*>(smem + WPF_STAT_OFF);          // [wave][2][128]
        if (lane < 16) {
#pragma unroll
            for (int e = 0; e < 8; ++e) { sSt[(wave * 2 + 0) * 128 + lane * 8 + e] = bs_[e]; sSt[(wave * 2 + 1) * 128 + lane * 8 + e] = bq_[e]; }
        }
        __syncthreads();
        {
            const int which = threadIdx.x >> 7, c = threadIdx.x & 127;
            const float v = sSt[(0 * 2 + which) * 128 + c] + sSt[(1 * 2 + which) * 128 + c] + sSt[(2 * 2 + which) * 128 + c] + sSt[(3 * 2 + which) * 128 + c];
            if (EPI == 1) { if (n0 + c < p.Npad) p.stats[((size_t)blockIdx.x * 2 + which) * p.Npad + n0 + c] = (n0 + c < p.N) ? v : 0.f; }
            else if (n0 + c < p.N) p.part[((size_t)blockIdx.x * 2 + which) * p.N + n0 + c] = v;
        }
    }
}

struct WpfPlan { int gx, gy; WpfK k; };

bool wpf_plan(const yh_conv_desc* d, WpfPlan* pl)
{
    if (!d || d->nseg != 1 || d->seg[0].ups || d->stride != 1 || d->Ho != d->Hi || d->Wo != d->Wi) return false;
    const bool k3 = d->KH == 3 && d->KW == 3 && d->pad == 1, k1 = d->KH == 1 && d->KW == 1 && d->pad == 0;
    if (!k3 && !k1) return false;
    if (d->bias || d->scale || d->shift || d->act != YH_ACT_NONE || d->res || d->nsplit < d->N || d->acc_rows) return false;
    if (d->mode == YH_CONV_FWD && (d->bnr_part || d->accumulate)) return false;
    if (d->mode == YH_CONV_DGRAD && d->stats) return false;
    const int Cin = d->seg[0].C;
    if (Cin % 32 || Cin < 64 || d->N % 8 || d->N < 64 || d->seg[0].ld % 8 || d->ld0 % 8) return false;
    const long M = (long)d->B * d->Ho * d->Wo;
    const unsigned long xb = ((unsigned long)M - 1) * d->seg[0].ld * 2 + (unsigned long)Cin * 2;
    const unsigned long wb = (unsigned long)d->Npad * d->KH * d->KW * Cin * 2;
    if (xb >= (1ul << 31) - 4096 || wb >= (1ul << 31) || M >= (1L << 31) - 1024 || d->Ho >= 32768 || d->Wo >= 32768) return false;
    WpfK& k = pl->k;
    pl->gy = (d->N + 127) / 128;
    if (pl->gy * 128 > d->Npad) return false;
    k.x = d->seg[0].ptr; k.ldx = d->seg[0].ld; k.Cin = Cin;
    k.w = d->w; k.Ktot = d->KH * d->KW * Cin;
    k.out = d->out0; k.ld0 = d->ld0; k.N = d->N; k.Npad = d->Npad;
    k.B = d->B; k.H = d->Ho; k.W = d->Wo; k.M = (int)M;
    k.taps = d->KH * d->KW; k.pad = d->pad; k.flip = d->mode == YH_CONV_DGRAD ? 1 : 0;
    k.ncb = Cin / 32; k.nkt = k.taps * k.ncb; k.mtiles = (int)((M + 511) / 512);
    int cap = 256 / pl->gy;
    if (cap < 1) cap = 1;
    if (d->grid_cap > 0) cap = d->grid_cap;
    pl->gx = k.mtiles < cap ? k.mtiles : cap;
    k.gx = pl->gx;
    k.accumulate = d->accumulate;
    k.stats = d->stats;
    k.z = d->bnr_z; k.ldz = d->bnr_ldz; k.ws = d->bnr_ws; k.wsC = d->bnr_C; k.part = d->bnr_part;
    k.xbytes = (unsigned)xb; k.wbytes = (unsigned)wb;
    return true;
}

}  // namespace

int yh_wpf_rows(const yh_conv_desc* d)
{
    WpfPlan pl;
    return wpf_plan(d, &pl) ? pl.gx : 0;
}

int yh_wpf_run(const yh_conv_desc* d, yh_stream stream, char* name_out, int name_len)
{
    WpfPlan pl;
    YH_CHECK_ARG(wpf_plan(d, &pl), "yh_conv_igemm: algo 11 (wave-private tiles) is not eligible for this descriptor");
    const int epi = d->bnr_part ? 3 : (d->stats ? 1 : 0);
    if (d->bnr_part)
        YH_CHECK_ARG(d->bnr_z && yh_aligned16(d->bnr_z) && d->bnr_ldz % 8 == 0 && d->bnr_ws && d->bnr_C >= d->N, "yh_conv_igemm: bad fused-reduction operands");
    if (name_out) { snprintf(name_out, name_len, "conv_wpf_kernel<%d>", epi); return YH_OK; }
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute((const void*)conv_wpf_kernel<0>, hipFuncAttributeMaxDynamicSharedMemorySize, WPF_LDS);
        (void)hipFuncSetAttribute((const void*)conv_wpf_kernel<1>, hipFuncAttributeMaxDynamicSharedMemorySize, WPF_LDS);
        (void)hipFuncSetAttribute((const void*)conv_wpf_kernel<3>, hipFuncAttributeMaxDynamicSharedMemorySize, WPF_LDS);
        attr_set = true;
    }
    hipStream_t st = (hipStream_t)stream;
    const dim3 grid(pl.gx, pl.gy), blk(256);
    if (epi == 3)      conv_wpf_kernel<3><<<grid, blk, WPF_LDS, st>>>(pl.k);
    else if (epi == 1) conv_wpf_kernel<1><<<grid, blk, WPF_LDS, st>>>(pl.k);
    else               conv_wpf_kernel<0><<<grid, blk, WPF_LDS, st>>>(pl.k);
    YH_CHECK_LAUNCH("yh_conv_igemm(wpf)");
    return YH_OK;
}
