// YOLOX loss with SimOTA label assignment on gfx950 (loss/yolox_loss.py:11-458 of the reference).
//
//   assign   one workgroup per image and stage, no host round trips (the reference runs a Python loop
//            over images and ground-truth boxes with ~4 syncs per image, :114,:330-350):
//            in-box / in-centre cell masks -> ordered candidate list -> IoU and cost matrices ->
//            per-gt dynamic-k (wave arg-max top-k) -> k smallest costs -> conflict resolution ->
//            ordered foreground list.  Orderings follow the reference: candidates and foreground
//            cells in cell order, ties resolved towards the lower index.
//   fg_fwd   per foreground cell (16-lane group): L1, IoU/GIoU/CIoU loss (:378-415), class BCE
//            against one-hot * matched IoU.
//   obj_fwd  objectness BCE over every cell;  finalize: /max(num_fg,1), balances EMA, weighted sum.
//   backward forward-mode dual numbers carry d/d(x,y,w,h) through the IoU-family formulas, including
//            the reference's gradient of the class TARGET w.r.t. the predicted box (matched IoU is
//            not detached there, :92,:150).
// Reference quirk kept: the class/objectness part of the assignment cost is a constant
// (`cls_cost_const`, computed by the host with the reference's expression on zero logits, :111-147).
#include "common.h"

namespace {

constexpr int MAXS = 4;
constexpr int MAXG = 128;          // ground-truth boxes per image held in LDS
constexpr int PARTS = 1024;

struct XLayout {
    size_t fg_count, nfg_stage, bal_used, fg_cell, fg_gt, fg_iou, cellmap, total;   // saved
    size_t w_flags, w_cand, w_iou, w_cost, w_cnt, w_mgt, w_part, w_total;           // workspace
    int ncell[MAXS];
    size_t cell_off[MAXS];         // element offset of stage s inside per-cell arrays (all images)
    size_t mat_off[MAXS];          // element offset of stage s inside the G x Y matrices
    int G;
};

XLayout make_xlayout(const yh_yolox_desc& d) {
    XLayout L;
    size_t cells = 0, mats = 0;
    L.G = d.maxbox < MAXG ? d.maxbox : MAXG;
    for (int s = 0; s < MAXS; ++s) {
        L.cell_off[s] = cells; L.mat_off[s] = mats; L.ncell[s] = 0;
        if (s < d.num_stage) {
            L.ncell[s] = d.H[s] * d.W[s];
            cells += (size_t)d.B * L.ncell[s];
            mats += (size_t)d.B * L.ncell[s] * L.G;
        }
    }
    size_t o = 0;
    auto take = [&](size_t bytes) { size_t r = o; o += (bytes + 255) & ~(size_t)255; return r; };
    L.fg_count = take(sizeof(int32_t) * MAXS * d.B);
    L.nfg_stage = take(64);
    L.bal_used = take(64);
    L.fg_cell = take(sizeof(int32_t) * cells);
    L.fg_gt = take(sizeof(int32_t) * cells);
    L.fg_iou = take(sizeof(float) * cells);
    L.cellmap = take(sizeof(int32_t) * cells);
    L.total = o;
    o = 0;
    L.w_flags = take(cells);
    L.w_cand = take(sizeof(int32_t) * cells);
    L.w_iou = take(sizeof(float) * mats);
    L.w_cost = take(sizeof(float) * mats);
    L.w_cnt = take(sizeof(int32_t) * cells);
    L.w_mgt = take(sizeof(int32_t) * cells);
    L.w_part = take(sizeof(double) * MAXS * 4 * PARTS);
    L.w_total = o;
    return L;
}

struct XK {
    yh_yolox_desc d;
    XLayout L;
};

template <typename T> __device__ __forceinline__ float ldx(const T* p);
template <> __device__ __forceinline__ float ldx<float>(const float* p) { return *p; }
template <> __device__ __forceinline__ float ldx<uint16_t>(const uint16_t* p) { return bf2f(*p); }
template <typename T> __device__ __forceinline__ void stx(T* p, float v);
template <> __device__ __forceinline__ void stx<float>(float* p, float v) { *p = v; }
template <> __device__ __forceinline__ void stx<uint16_t>(uint16_t* p, float v) { *p = f2bf(v); }

__device__ __forceinline__ float sigm(float x) { return 1.0f / (1.0f + expf(-x)); }

// gpu_iou of two xyxy boxes (utils/bbox_tools.py:164-190)
__device__ __forceinline__ float iou_xyxy(float ax0, float ay0, float ax1, float ay1, float bx0, float by0, float bx1, float by1) {
    const float a1 = (ax1 - ax0) * (ay1 - ay0);
    const float a2 = (bx1 - bx0) * (by1 - by0);
    const float w = fmaxf(fminf(ax1, bx1) - fmaxf(ax0, bx0), 0.f);
    const float h = fmaxf(fminf(ay1, by1) - fmaxf(ay0, by0), 0.f);
    const float inter = w * h;
    return inter / fmaxf(a1 + a2 - inter, 1e-9f);
}

// Block-free arg-max of a (value, index) pair over a wavefront: the pair is packed into one 64-bit key whose unsigned
// order is (value desc, index asc) for want_max and (value asc, index asc) otherwise, reduced with six DPP steps (row shifts
// 1/2/4/8, then row_bcast 15 and 31) and read back from lane 63.  Values must not be NaN or -0.0; index 0x7fffffff is the
// "nothing" sentinel and loses every tie.
__device__ __forceinline__ void wave_pick(float& v, int& idx, bool want_max)
{
    unsigned b = __float_as_uint(v);
    b ^= (b >> 31) ? 0xffffffffu : 0x80000000u;          // monotone float -> unsigned
    unsigned hi = want_max ? b : ~b, lo = ~(unsigned)idx;
#define YH_DPP_MAX(ctrl, rows) { \
        const unsigned olo = (unsigned)__builtin_amdgcn_update_dpp((int)lo, (int)lo, ctrl, rows, 0xf, false); \
        const unsigned ohi = (unsigned)__builtin_amdgcn_update_dpp((int)hi, (int)hi, ctrl, rows, 0xf, false); \
        if (ohi > hi || (ohi == hi && olo > lo)) { lo = olo; hi = ohi; } }
    YH_DPP_MAX(0x111, 0xf) YH_DPP_MAX(0x112, 0xf) YH_DPP_MAX(0x114, 0xf) YH_DPP_MAX(0x118, 0xf)
    YH_DPP_MAX(0x142, 0xa) YH_DPP_MAX(0x143, 0xc)
#undef YH_DPP_MAX
    hi = (unsigned)__builtin_amdgcn_readlane((int)hi, 63);
    lo = (unsigned)__builtin_amdgcn_readlane((int)lo, 63);
    b = want_max ? hi : ~hi;
    b = (b & 0x80000000u) ? (b ^ 0x80000000u) : ~b;
    v = __uint_as_float(b);
    idx = (int)~lo;
}

struct StageX { const void* pred; void* gpred; int s, H, W, ld; float stride; };

// ------------------------------------------------------------------------------------------------
struct StagesX { StageX s[MAXS]; };

// grid (B, stages): the stages are matched independently (loss/yolox_loss.py:60-69) and share one launch
template <typename T>
__global__ __launch_bounds__(1024) void yolox_assign_kernel(const XK p, const StagesX sts, const float* __restrict__ targets,
                                                            unsigned char* __restrict__ wsb, unsigned char* __restrict__ svb)
{
    const StageX st = sts.s[blockIdx.y];
    __shared__ float gt[MAXG][4];          // x, y, w, h (pixels)
    __shared__ int gt_row[MAXG];
    __shared__ int s_near[MAXG];
    __shared__ int s_cnt[16];
    __shared__ int s_misc[8];
    __shared__ int s_vali[16];
    const yh_yolox_desc& d = p.d;
    const int b = blockIdx.x, s = st.s;
    const int t = threadIdx.x, lane = t & 63, wv = t >> 6;
    const int n = st.H * st.W;
    const size_t cbase = p.L.cell_off[s] + (size_t)b * n;
    unsigned char* flags = wsb + p.L.w_flags + cbase;
    int32_t* cand = reinterpret_cast<int32_t*>(wsb + p.L.w_cand) + cbase;
    float* iou_m = reinterpret_cast<float*>(wsb + p.L.w_iou) + p.L.mat_off[s] + (size_t)b * n * p.L.G;
    float* cost_m = reinterpret_cast<float*>(wsb + p.L.w_cost) + p.L.mat_off[s] + (size_t)b * n * p.L.G;
    int32_t* cnt = reinterpret_cast<int32_t*>(wsb + p.L.w_cnt) + cbase;
    int32_t* mgt = reinterpret_cast<int32_t*>(wsb + p.L.w_mgt) + cbase;
    int32_t* fg_count = reinterpret_cast<int32_t*>(svb + p.L.fg_count) + s * d.B + b;
    int32_t* fg_cell = reinterpret_cast<int32_t*>(svb + p.L.fg_cell) + cbase;
    int32_t* fg_gt = reinterpret_cast<int32_t*>(svb + p.L.fg_gt) + cbase;
    float* fg_iou = reinterpret_cast<float*>(svb + p.L.fg_iou) + cbase;
    int32_t* cellmap = reinterpret_cast<int32_t*>(svb + p.L.cellmap) + cbase;
    const T* pred = reinterpret_cast<const T*>(st.pred) + (size_t)b * n * st.ld;
    const float stride = st.stride;

    // ---- valid ground truths (class >= 0), in row order (:116-118)
    if (t == 0) {
        int g = 0;
        for (int j = 0; j < d.maxbox && g < MAXG; ++j) {
            const float* r = targets + ((size_t)b * d.maxbox + j) * 6;
            if (r[4] >= 0.f) { gt[g][0] = r[0]; gt[g][1] = r[1]; gt[g][2] = r[2]; gt[g][3] = r[3]; gt_row[g] = j; ++g; }
        }
        s_misc[0] = g;
    }
    for (int i = t; i < n; i += 1024) cellmap[i] = -1;
    __syncthreads();
    const int G = s_misc[0];
    if (G == 0) { if (t == 0) *fg_count = 0; return; }

    // ---- cell masks (select_grid :235-303)
    const float eps = 1e-9f;
    int nbox = 0, nctr = 0;
    for (int i = t; i < n; i += 1024) {
        const float cx = ((float)(i % st.W) + 0.5f) * stride;
        const float cy = ((float)(i / st.W) + 0.5f) * stride;
        bool anyb = false, anyc = false;
        for (int g = 0; g < G; ++g) {
            const float x = gt[g][0], y = gt[g][1], w = gt[g][2], h = gt[g][3];
            const float xmin = x + w * -0.5f, ymin = y + h * -0.5f, xmax = x + w * 0.5f, ymax = y + h * 0.5f;
            const float m1 = fminf(fminf(-xmin + cx, -ymin + cy), fminf(xmax + -cx, ymax + -cy));
            anyb |= m1 > eps;
            const float r = d.center_radius;
            const float m2 = fminf(fminf(cx + -(x + -r), cy + -(y + -r)), fminf(-cx + (x + r), -cy + (y + r)));
            anyc |= m2 > eps;
        }
        flags[i] = (anyb ? 1 : 0) | (anyc ? 2 : 0);
        nbox += anyb; nctr += anyc;
    }
    for (int o = 32; o > 0; o >>= 1) { nbox += __shfl_xor(nbox, o, 64); nctr += __shfl_xor(nctr, o, 64); }
    if (lane == 0) { s_cnt[wv] = nbox; s_vali[wv] = nctr; }
    __syncthreads();
    if (t == 0) {
        int a = 0, c = 0;
        for (int w = 0; w < 16; ++w) { a += s_cnt[w]; c += s_vali[w]; }
        s_misc[1] = a; s_misc[2] = c;
    }
    __syncthreads();
    if (s_misc[1] == 0) {
        // reference: random subset of the cells nearest to each gt centre (torch.randperm, :270-278).  Deterministic
        // here: the nearest cell of the first `choose_num` ground truths, in gt order.
        if (t == 0) {
            int uniq = 0;
            for (int g = 0; g < G; ++g) {
                int best = 0; float bd = INFINITY;
                for (int i = 0; i < n; ++i) {
                    const float dx = gt[g][0] - ((float)(i % st.W) + 0.5f) * stride, dy = gt[g][1] - ((float)(i / st.W) + 0.5f) * stride;
                    const float dd = sqrtf(dx * dx + dy * dy);
                    if (dd < bd) { bd = dd; best = i; }
                }
                if (!(flags[best] & 4)) { flags[best] |= 4; ++uniq; }
                s_near[g] = best;
            }
            const int choose = (uniq * 0.2f > 2.f) ? (int)(uniq * 0.2f) : 1;
            for (int g = 0; g < choose && g < G; ++g) flags[s_near[g]] |= 1;
            for (int i = 0; i < n; ++i) flags[i] &= 3;
        }
        __syncthreads();
    }
    const bool ctr_is_box = (s_misc[2] == 0);      // :295-296

    // ---- ordered candidate list (cells in the union mask)
    if (t == 0) s_misc[3] = 0;
    __syncthreads();
    for (int i0 = 0; i0 < n; i0 += 1024) {
        const int i = i0 + t;
        bool f = false;
        if (i < n) { const int fl = flags[i]; f = (fl & 1) || (ctr_is_box ? (fl & 1) : (fl & 2)); }
        const unsigned long long bal = __ballot(f);
        if (lane == 0) s_cnt[wv] = __popcll(bal);
        __syncthreads();
        int before = 0, total = 0;
        for (int w = 0; w < 16; ++w) { if (w < wv) before += s_cnt[w]; total += s_cnt[w]; }
        const int base = s_misc[3];
        if (f) cand[base + before + __popcll(bal & ((1ull << lane) - 1ull))] = i;
        __syncthreads();
        if (t == 0) s_misc[3] = base + total;
        __syncthreads();
    }
    const int Y = s_misc[3];
    if (Y == 0) { if (t == 0) *fg_count = 0; return; }

    // ---- register-resident path (Y <= 8192 candidates and topk <= 16: every 640x640 case): thread t owns candidates
    // t, t+1024, ...; their decoded boxes stay in registers and the IoU / cost of a (gt, candidate) pair are computed
    // on the fly with the same fp32 expressions as the matrix path below.
    constexpr int MAXJ = 8;
    if (Y <= MAXJ * 1024 && d.topk <= 16) {
        float qx0[MAXJ], qy0[MAXJ], qx1[MAXJ], qy1[MAXJ];
        int qxy[MAXJ], cnt_l[MAXJ], mgt_l[MAXJ];         // qxy: cell column | row << 16
        unsigned invalid = 0;
#pragma unroll
        for (int j = 0; j < MAXJ; ++j) {
            const int y = t + 1024 * j;
            cnt_l[j] = 0; mgt_l[j] = -1; qxy[j] = 0;
            qx0[j] = qy0[j] = qx1[j] = qy1[j] = 0.f;
            if (y < Y) {
                const int cell = cand[y];
                const T* row = pred + (size_t)cell * st.ld;
                const float gxc = (float)(cell % st.W), gyc = (float)(cell / st.W);
                const float px = (ldx<T>(row) + gxc) * stride, py = (ldx<T>(row + 1) + gyc) * stride;
                const float pw = expf(ldx<T>(row + 2)) * stride, ph = expf(ldx<T>(row + 3)) * stride;
                qx0[j] = px - pw / 2.f; qy0[j] = py - ph / 2.f; qx1[j] = px + pw / 2.f; qy1[j] = py + ph / 2.f;
                qxy[j] = (cell % st.W) | ((cell / st.W) << 16);
            } else {
                invalid |= 1u << j;
            }
        }
        auto pair = [&](int g, int j, float& iou, float& cost) __attribute__((always_inline)) {
            const float x = gt[g][0], yy = gt[g][1], w = gt[g][2], h = gt[g][3];
            iou = iou_xyxy(x - w / 2.f, yy - h / 2.f, x + w / 2.f, yy + h / 2.f, qx0[j], qy0[j], qx1[j], qy1[j]);
            const float cx = ((float)(qxy[j] & 0xffff) + 0.5f) * stride, cy = ((float)(qxy[j] >> 16) + 0.5f) * stride;
            const float xmin = x + w * -0.5f, ymin = yy + h * -0.5f, xmax = x + w * 0.5f, ymax = yy + h * 0.5f;
            const bool inb = fminf(fminf(-xmin + cx, -ymin + cy), fminf(xmax + -cx, ymax + -cy)) > eps;
            const float r = d.center_radius;
            const bool inc = fminf(fminf(cx + -(x + -r), cy + -(yy + -r)), fminf(-cx + (x + r), -cy + (yy + r))) > eps;
            cost = (d.cls_cost_const + 3.f * (-logf(iou + 1e-9f))) + 100000.f * ((inb && inc) ? 0.f : 1.f);
        };
        // The ground truths are matched independently of each other (only the per-candidate claim counts meet), so
        // a round takes GG of them at once: every wave lists its own K best candidates per ground truth with wave
        // shuffles alone, one wave per ground truth merges the 16 lists (the global top-K is contained in their union
        // because (value, index) is a strict total order), and the whole round costs five block barriers instead of
        // one per pick.
        constexpr int GG = 16, KMAX = 16;
        __shared__ float s_lv[GG][16][KMAX];
        __shared__ int s_li[GG][16][KMAX];
        __shared__ int s_dk[GG];
        __shared__ int s_pick[GG][KMAX];
        const int K = d.topk < Y ? d.topk : Y;          // topk <= KMAX on this path
        // the wave's `count` best of its own candidates for ground truth slot gi -> s_lv/s_li[gi][wv][*]
        auto wave_list = [&](int gi, const float (&val)[MAXJ], int count, bool want_max) __attribute__((always_inline)) {
            unsigned taken = invalid;
            for (int k = 0; k < count; ++k) {
                float bv = want_max ? -INFINITY : INFINITY; int bi = 0x7fffffff;
#pragma unroll
                for (int j = 0; j < MAXJ; ++j) {
                    const bool better = want_max ? (val[j] > bv) : (val[j] < bv);
                    if (!(taken & (1u << j)) && (better || (val[j] == bv && t + 1024 * j < bi))) { bv = val[j]; bi = t + 1024 * j; }
                }
                wave_pick(bv, bi, want_max);
                if (want_max && !(bv > 0.f)) {
                    // IoUs are >= 0: everything this wave has left is exactly 0 (or nothing is left, -inf) and adds
                    // nothing to the top-K sum, whichever of those entries the merge would pick
                    for (int kk = k + lane; kk < count; kk += 64) { s_lv[gi][wv][kk] = bv; s_li[gi][wv][kk] = 0x7fffffff; }
                    break;
                }
                if (lane == 0) { s_lv[gi][wv][k] = bv; s_li[gi][wv][k] = bi; }
                if (bi != 0x7fffffff && (bi & 1023) == t) taken |= 1u << (bi >> 10);
            }
        };
        for (int g0 = 0; g0 < G; g0 += GG) {
            const int ng = (G - g0) < GG ? (G - g0) : GG;
            // ---- K largest IoUs per ground truth, (value desc, index asc)
            for (int gi = 0; gi < ng; ++gi) {
                float vio[MAXJ];
#pragma unroll
                for (int j = 0; j < MAXJ; ++j) {
                    float co;
                    vio[j] = -INFINITY;
                    if (!(invalid & (1u << j))) { pair(g0 + gi, j, vio[j], co); vio[j] += 0.f; }          // -0.0 -> +0.0
                }
                wave_list(gi, vio, K, true);
            }
            __syncthreads();
            if (wv < ng) {
                // 16 lists of K entries: a lane holds entries lane, lane + 64, ... of the 16 x K table
                constexpr int EPL = 16 * KMAX / 64;
                float ev[EPL]; int ei[EPL];
#pragma unroll
                for (int q = 0; q < EPL; ++q) {
                    const int e = lane + 64 * q, w = e / KMAX, k = e % KMAX;
                    const bool ok = k < K;
                    ev[q] = ok ? s_lv[wv][w][k] : -INFINITY;
                    ei[q] = ok ? s_li[wv][w][k] : 0x7fffffff;
                }
                unsigned tk = 0;
                float ksum = 0.f;
                for (int k = 0; k < K; ++k) {
                    float bv = -INFINITY; int bi = 0x7fffffff; int bq = -1;
#pragma unroll
                    for (int q = 0; q < EPL; ++q)
                        if (!(tk & (1u << q)) && (ev[q] > bv || (ev[q] == bv && ei[q] < bi))) { bv = ev[q]; bi = ei[q]; bq = q; }
                    const int mine = bi;
                    wave_pick(bv, bi, true);
                    ksum += bv;
                    if (bq >= 0 && mine == bi && bi != 0x7fffffff) tk |= 1u << bq;
                }
                int dk = (int)ksum;
                dk = dk < 1 ? 1 : (dk > Y ? Y : dk);
                if (lane == 0) s_dk[wv] = dk < KMAX ? dk : KMAX;          // dk <= K: every IoU is <= 1
            }
            __syncthreads();
            // ---- dk smallest costs per ground truth, (value asc, index asc)
            for (int gi = 0; gi < ng; ++gi) {
                float vco[MAXJ];
#pragma unroll
                for (int j = 0; j < MAXJ; ++j) {
                    float io;
                    vco[j] = INFINITY;
                    if (!(invalid & (1u << j))) pair(g0 + gi, j, io, vco[j]);
                }
                wave_list(gi, vco, s_dk[gi], false);
            }
            __syncthreads();
            if (wv < ng) {
                constexpr int EPL = 16 * KMAX / 64;
                const int dk = s_dk[wv];
                float ev[EPL]; int ei[EPL];
#pragma unroll
                for (int q = 0; q < EPL; ++q) {
                    const int e = lane + 64 * q, w = e / KMAX, k = e % KMAX;
                    const bool ok = k < dk;
                    ev[q] = ok ? s_lv[wv][w][k] : INFINITY;
                    ei[q] = ok ? s_li[wv][w][k] : 0x7fffffff;
                }
                unsigned tk = 0;
                for (int k = 0; k < dk; ++k) {
                    float bv = INFINITY; int bi = 0x7fffffff; int bq = -1;
#pragma unroll
                    for (int q = 0; q < EPL; ++q)
                        if (!(tk & (1u << q)) && (ev[q] < bv || (ev[q] == bv && ei[q] < bi))) { bv = ev[q]; bi = ei[q]; bq = q; }
                    const int mine = bi;
                    wave_pick(bv, bi, false);
                    if (lane == 0) s_pick[wv][k] = bi == 0x7fffffff ? -1 : bi;
                    if (bq >= 0 && mine == bi && bi != 0x7fffffff) tk |= 1u << bq;
                }
            }
            __syncthreads();
            // ---- claims
            for (int gi = 0; gi < ng; ++gi) {
                const int dk = s_dk[gi];
                for (int k = 0; k < dk; ++k) {
                    const int bi = s_pick[gi][k];
                    if (bi >= 0 && (bi & 1023) == t) {
                        const int jj = bi >> 10;
#pragma unroll
                        for (int j = 0; j < MAXJ; ++j)
                            if (j == jj) { cnt_l[j] += 1; mgt_l[j] = g0 + gi; }
                    }
                }
            }
        }
        // conflicts: a candidate claimed by several gts goes to the one with the smallest cost over ALL gts (:341-346)
#pragma unroll
        for (int j = 0; j < MAXJ; ++j) {
            if (cnt_l[j] > 1) {
                float bv = INFINITY; int bg = 0;
                for (int g = 0; g < G; ++g) { float io, co; pair(g, j, io, co); if (co < bv) { bv = co; bg = g; } }
                mgt_l[j] = bg;
            }
        }
        // ordered foreground list
        if (t == 0) s_misc[4] = 0;
        __syncthreads();
#pragma unroll
        for (int j = 0; j < MAXJ; ++j) {
            if (1024 * j >= Y) break;                 // block-uniform
            const bool f = cnt_l[j] > 0;
            const unsigned long long bal = __ballot(f);
            if (lane == 0) s_cnt[wv] = __popcll(bal);
            __syncthreads();
            int before = 0, total = 0;
            for (int w = 0; w < 16; ++w) { if (w < wv) before += s_cnt[w]; total += s_cnt[w]; }
            const int base = s_misc[4];
            if (f) {
                const int pos = base + before + __popcll(bal & ((1ull << lane) - 1ull));
                float io, co;
                pair(mgt_l[j], j, io, co);
                const int cell = (qxy[j] >> 16) * st.W + (qxy[j] & 0xffff);
                fg_cell[pos] = cell;
                fg_gt[pos] = gt_row[mgt_l[j]];
                fg_iou[pos] = io;
                cellmap[cell] = pos;
            }
            __syncthreads();
            if (t == 0) s_misc[4] = base + total;
            __syncthreads();
        }
        if (t == 0) *fg_count = s_misc[4];
        return;
    }

    // ---- IoU and cost matrices (label_assign :131-149)
    for (int e = t; e < G * Y; e += 1024) {
        const int g = e / Y, y = e - g * Y;
        const int cell = cand[y];
        const T* row = pred + (size_t)cell * st.ld;
        const float gxc = (float)(cell % st.W), gyc = (float)(cell / st.W);
        const float px = (ldx<T>(row) + gxc) * stride, py = (ldx<T>(row + 1) + gyc) * stride;
        const float pw = expf(ldx<T>(row + 2)) * stride, ph = expf(ldx<T>(row + 3)) * stride;
        const float x = gt[g][0], yy = gt[g][1], w = gt[g][2], h = gt[g][3];
        const float iou = iou_xyxy(x - w / 2.f, yy - h / 2.f, x + w / 2.f, yy + h / 2.f, px - pw / 2.f, py - ph / 2.f, px + pw / 2.f, py + ph / 2.f);
        const float cx = (gxc + 0.5f) * stride, cy = (gyc + 0.5f) * stride;
        const float xmin = x + w * -0.5f, ymin = yy + h * -0.5f, xmax = x + w * 0.5f, ymax = yy + h * 0.5f;
        const bool inb = fminf(fminf(-xmin + cx, -ymin + cy), fminf(xmax + -cx, ymax + -cy)) > eps;
        const float r = d.center_radius;
        const bool inc = fminf(fminf(cx + -(x + -r), cy + -(yy + -r)), fminf(-cx + (x + r), -cy + (yy + r))) > eps;
        iou_m[e] = iou;
        cost_m[e] = (d.cls_cost_const + 3.f * (-logf(iou + 1e-9f))) + 100000.f * ((inb && inc) ? 0.f : 1.f);
    }
    for (int y = t; y < Y; y += 1024) { cnt[y] = 0; mgt[y] = -1; }
    __syncthreads();

    // ---- dynamic-k matching (simple_ota :305-359): one wave per ground truth
    const int K = d.topk < Y ? d.topk : Y;
    for (int g = wv; g < G; g += 16) {
        const float* io = iou_m + (size_t)g * Y;
        const float* co = cost_m + (size_t)g * Y;
        // sum of the K largest IoUs, selected in (value desc, index asc) order
        float lastv = INFINITY; int lasti = -1; float ksum = 0.f;
        for (int k = 0; k < K; ++k) {
            float bv = -INFINITY; int bi = 0x7fffffff;
            for (int y = lane; y < Y; y += 64) {
                const float v = io[y];
                const bool elig = (v < lastv) || (v == lastv && y > lasti);
                if (elig && (v > bv || (v == bv && y < bi))) { bv = v; bi = y; }
            }
            for (int o = 32; o > 0; o >>= 1) {
                const float ov = __shfl_xor(bv, o, 64); const int oi = __shfl_xor(bi, o, 64);
                if (ov > bv || (ov == bv && oi < bi)) { bv = ov; bi = oi; }
            }
            ksum += bv; lastv = bv; lasti = bi;
        }
        int dk = (int)ksum;
        dk = dk < 1 ? 1 : (dk > Y ? Y : dk);
        lastv = -INFINITY; lasti = -1;
        for (int k = 0; k < dk; ++k) {          // dk smallest costs, (value asc, index asc)
            float bv = INFINITY; int bi = 0x7fffffff;
            for (int y = lane; y < Y; y += 64) {
                const float v = co[y];
                const bool elig = (v > lastv) || (v == lastv && y > lasti);
                if (elig && (v < bv || (v == bv && y < bi))) { bv = v; bi = y; }
            }
            for (int o = 32; o > 0; o >>= 1) {
                const float ov = __shfl_xor(bv, o, 64); const int oi = __shfl_xor(bi, o, 64);
                if (ov < bv || (ov == bv && oi < bi)) { bv = ov; bi = oi; }
            }
            if (bi == 0x7fffffff) break;
            if (lane == 0) { atomicAdd(&cnt[bi], 1); mgt[bi] = g; }
            lastv = bv; lasti = bi;
        }
    }
    __threadfence_block();
    __syncthreads();

    // ---- conflicts: a candidate claimed by several gts goes to the one with the smallest cost over ALL gts (:341-346)
    for (int y = t; y < Y; y += 1024) {
        if (cnt[y] > 1) {
            float bv = INFINITY; int bg = 0;
            for (int g = 0; g < G; ++g) { const float v = cost_m[(size_t)g * Y + y]; if (v < bv) { bv = v; bg = g; } }
            mgt[y] = bg;
        }
    }
    __syncthreads();

    // ---- ordered foreground list
    if (t == 0) s_misc[4] = 0;
    __syncthreads();
    for (int y0 = 0; y0 < Y; y0 += 1024) {
        const int y = y0 + t;
        const bool f = y < Y && cnt[y] > 0;
        const unsigned long long bal = __ballot(f);
        if (lane == 0) s_cnt[wv] = __popcll(bal);
        __syncthreads();
        int before = 0, total = 0;
        for (int w = 0; w < 16; ++w) { if (w < wv) before += s_cnt[w]; total += s_cnt[w]; }
        const int base = s_misc[4];
        if (f) {
            const int pos = base + before + __popcll(bal & ((1ull << lane) - 1ull));
            const int g = mgt[y];
            fg_cell[pos] = cand[y];
            fg_gt[pos] = gt_row[g];
            fg_iou[pos] = iou_m[(size_t)g * Y + y];
            cellmap[cand[y]] = pos;
        }
        __syncthreads();
        if (t == 0) s_misc[4] = base + total;
        __syncthreads();
    }
    if (t == 0) *fg_count = s_misc[4];
}

// ------------------------------------------------------------------------------------------------
// forward-mode dual number over the 4 box unknowns (x, y, w, h)
struct Dual {
    float v, g[4];
    __device__ Dual() {}
    __device__ Dual(float c) : v(c) { g[0] = g[1] = g[2] = g[3] = 0.f; }
    __device__ static Dual var(float c, int i) { Dual d(c); d.g[i] = 1.f; return d; }
};
__device__ __forceinline__ Dual operator+(Dual a, Dual b) { Dual r; r.v = a.v + b.v; for (int i = 0; i < 4; ++i) r.g[i] = a.g[i] + b.g[i]; return r; }
__device__ __forceinline__ Dual operator-(Dual a, Dual b) { Dual r; r.v = a.v - b.v; for (int i = 0; i < 4; ++i) r.g[i] = a.g[i] - b.g[i]; return r; }
__device__ __forceinline__ Dual operator*(Dual a, Dual b) { Dual r; r.v = a.v * b.v; for (int i = 0; i < 4; ++i) r.g[i] = a.g[i] * b.v + a.v * b.g[i]; return r; }
__device__ __forceinline__ Dual operator/(Dual a, Dual b) { Dual r; r.v = a.v / b.v; for (int i = 0; i < 4; ++i) r.g[i] = (a.g[i] * b.v - a.v * b.g[i]) / (b.v * b.v); return r; }
// torch minimum/maximum split the gradient on ties; clamp(min) passes the gradient at x >= min
__device__ __forceinline__ Dual dmin(Dual a, Dual b) { const float wa = a.v < b.v ? 1.f : (a.v == b.v ? 0.5f : 0.f); Dual r; r.v = fminf(a.v, b.v); for (int i = 0; i < 4; ++i) r.g[i] = wa * a.g[i] + (1.f - wa) * b.g[i]; return r; }
__device__ __forceinline__ Dual dmax(Dual a, Dual b) { const float wa = a.v > b.v ? 1.f : (a.v == b.v ? 0.5f : 0.f); Dual r; r.v = fmaxf(a.v, b.v); for (int i = 0; i < 4; ++i) r.g[i] = wa * a.g[i] + (1.f - wa) * b.g[i]; return r; }
__device__ __forceinline__ Dual dclamp0(Dual a) { Dual r; const float k = a.v >= 0.f ? 1.f : 0.f; r.v = fmaxf(a.v, 0.f); for (int i = 0; i < 4; ++i) r.g[i] = k * a.g[i]; return r; }
__device__ __forceinline__ Dual dclampmin(Dual a, float m) { Dual r; const float k = a.v >= m ? 1.f : 0.f; r.v = fmaxf(a.v, m); for (int i = 0; i < 4; ++i) r.g[i] = k * a.g[i]; return r; }
__device__ __forceinline__ Dual dclamp(Dual a, float lo, float hi) { Dual r; const float k = (a.v >= lo && a.v <= hi) ? 1.f : 0.f; r.v = fminf(fmaxf(a.v, lo), hi); for (int i = 0; i < 4; ++i) r.g[i] = k * a.g[i]; return r; }
__device__ __forceinline__ Dual datan(Dual a) { Dual r; r.v = atanf(a.v); const float k = 1.f / (1.f + a.v * a.v); for (int i = 0; i < 4; ++i) r.g[i] = k * a.g[i]; return r; }
__device__ __forceinline__ Dual dabs(Dual a) { Dual r; r.v = fabsf(a.v); const float k = a.v > 0.f ? 1.f : (a.v < 0.f ? -1.f : 0.f); for (int i = 0; i < 4; ++i) r.g[i] = k * a.g[i]; return r; }
__device__ __forceinline__ Dual dconst(Dual a) { return Dual(a.v); }

// YOLOXLoss.iou_loss on xywh boxes (:378-415); box1 = prediction (differentiated), box2 = target
__device__ Dual iou_loss_dual(int type, const float* pb, const float* tb)
{
    const float eps = 1e-9f;
    const Dual x1 = Dual::var(pb[0], 0), y1 = Dual::var(pb[1], 1), w1 = Dual::var(pb[2], 2), h1 = Dual::var(pb[3], 3);
    const Dual x2(tb[0]), y2(tb[1]), w2(tb[2]), h2(tb[3]);
    const Dual two(2.f);
    const Dual ax0 = x1 - w1 / two, ay0 = y1 - h1 / two, ax1 = x1 + w1 / two, ay1 = y1 + h1 / two;
    const Dual bx0 = x2 - w2 / two, by0 = y2 - h2 / two, bx1 = x2 + w2 / two, by1 = y2 + h2 / two;
    const Dual uni = dclamp0(w1 * h1) + dclamp0(w2 * h2);
    const Dual inter = dclamp0(dmin(ax1, bx1) - dmax(ax0, bx0)) * dclamp0(dmin(ay1, by1) - dmax(ay0, by0));
    const Dual iou = inter / (uni - inter + Dual(eps));
    if (type == 0) return Dual(1.f) - iou * iou;
    if (type == 1) {
        const Dual convex = dclamp0(dmax(ax1, bx1) - dmin(ax0, bx0)) * dclamp0(dmax(ay1, by1) - dmin(ay0, by0));
        const Dual giou = iou - dabs(convex - uni) / (convex + Dual(eps));
        return Dual(1.f) - dclamp(giou, -1.f, 1.f);
    }
    const Dual c_hs = dclamp0(dmax(ay1, by1) - dmin(ay0, by0));
    const Dual c_ws = dclamp0(dmax(ax1, bx1) - dmin(ax0, bx0));
    const Dual c_d = c_ws * c_ws + c_hs * c_hs + Dual(eps);
    const Dual ctr = (x1 - x2) * (x1 - x2) + (y1 - y2) * (y1 - y2);
    const Dual da = datan(w1 / h1) - datan(w2 / h2);
    const Dual v = Dual((float)(4.0 / (3.14159265358979323846 * 3.14159265358979323846))) * da * da;
    const Dual alpha = dconst(v / dclampmin(Dual(1.f) - iou + v, eps));          // torch.no_grad (:409-410)
    return Dual(1.f) - (iou - ctr / c_d - v * alpha);
}

// gpu_iou(target xyxy, prediction xyxy) as a dual over the prediction's (x,y,w,h): the matched IoU
__device__ Dual matched_iou_dual(const float* pb, const float* tb)
{
    const Dual x1 = Dual::var(pb[0], 0), y1 = Dual::var(pb[1], 1), w1 = Dual::var(pb[2], 2), h1 = Dual::var(pb[3], 3);
    const Dual two(2.f);
    const Dual px0 = x1 - w1 / two, py0 = y1 - h1 / two, px1 = x1 + w1 / two, py1 = y1 + h1 / two;
    const Dual gx0(tb[0] - tb[2] / 2.f), gy0(tb[1] - tb[3] / 2.f), gx1(tb[0] + tb[2] / 2.f), gy1(tb[1] + tb[3] / 2.f);
    const Dual a1 = (gx1 - gx0) * (gy1 - gy0);
    const Dual a2 = (px1 - px0) * (py1 - py0);
    const Dual w = dclamp0(dmin(gx1, px1) - dmax(gx0, px0));
    const Dual h = dclamp0(dmin(gy1, py1) - dmax(gy0, py0));
    const Dual inter = w * h;
    return inter / dclampmin(a1 + a2 - inter, 1e-9f);
}

__device__ __forceinline__ float bce_logits(float x, float t, float pw, float* dx, float* dt) {
    const float lw = 1.0f + (pw - 1.0f) * t;
    const float sp = log1pf(expf(-fabsf(x))) + fmaxf(-x, 0.0f);      // softplus(-x) = -log(sigmoid(x))
    const float sg = sigm(x);
    if (dx) *dx = (1.0f - t) - lw * (1.0f - sg);
    if (dt) *dt = -x + (pw - 1.0f) * sp;                              // d/dt [(1-t)x + (1+(pw-1)t) sp]
    return (1.0f - t) * x + lw * sp;
}
__device__ __forceinline__ float focal_factor(float x, float t, float gamma, float alpha, float* dx, float* dt) {
    const float pr = sigm(x);
    const float acc = t * pr + (1.0f - t) * (1.0f - pr);
    const float om = 1.0f - acc;
    const float gf = powf(om, gamma);
    const float af = t * alpha + (1.0f - t) * (1.0f - alpha);
    const float dgf_dom = om > 0.f ? gamma * powf(om, gamma - 1.0f) : 0.f;
    if (dx) *dx = dgf_dom * (-(2.0f * t - 1.0f) * pr * (1.0f - pr)) * af;
    if (dt) *dt = dgf_dom * (-(2.0f * pr - 1.0f)) * af + gf * (2.0f * alpha - 1.0f);
    return gf * af;
}

// ------------------------------------------------------------------------------------------------
template <typename T, bool BWD>
__global__ __launch_bounds__(256) void yolox_fg_kernel(const XK p, const StagesX sts, const float* __restrict__ targets,
                                                       const unsigned char* __restrict__ svb, double* __restrict__ part,
                                                       const float* __restrict__ gout)
{
    const StageX st = sts.s[blockIdx.z];          // one launch for all stages
    __shared__ double sred[3][4];
    const yh_yolox_desc& d = p.d;
    const int s = st.s, n = st.H * st.W, nc = d.num_class, E = 5 + nc;
    const int t = threadIdx.x, sub = t & 15, grp = t >> 4;
    const int32_t* fg_count = reinterpret_cast<const int32_t*>(svb + p.L.fg_count) + s * d.B;
    const int32_t* fg_cell = reinterpret_cast<const int32_t*>(svb + p.L.fg_cell) + p.L.cell_off[s];
    const int32_t* fg_gt = reinterpret_cast<const int32_t*>(svb + p.L.fg_gt) + p.L.cell_off[s];
    const float* fg_iou = reinterpret_cast<const float*>(svb + p.L.fg_iou) + p.L.cell_off[s];
    const int32_t* nfg_stage = reinterpret_cast<const int32_t*>(svb + p.L.nfg_stage);
    const double* bal_used = reinterpret_cast<const double*>(svb + p.L.bal_used);
    (void)bal_used;
    const T* pred = reinterpret_cast<const T*>(st.pred);
    T* gp = reinterpret_cast<T*>(st.gpred);
    double a_iou = 0.0, a_l1 = 0.0, a_cls = 0.0;
    float kiou = 0.f, kl1 = 0.f, kcls = 0.f;
    if (BWD) {
        const float nf = (float)(nfg_stage[s] > 1 ? nfg_stage[s] : 1);
        const float go = *gout;
        kiou = go * d.iou_scale / nf; kl1 = go * d.l1_scale / nf; kcls = go * d.cls_scale / nf;
    }
    // work items: (image b, foreground index j); gridDim.y walks the images in parallel (a sequential image loop was a
    // chain of B dependent load latencies: 0.25-0.35 ms per launch at B=64), every 16-lane group a strided share of j
    for (int b = blockIdx.y; b < d.B; b += gridDim.y) {
        const int nfgb = fg_count[b];
        for (int j = blockIdx.x * 16 + grp; j < nfgb; j += gridDim.x * 16) {
            const int cell = fg_cell[(size_t)b * n + j];
            const int grow = fg_gt[(size_t)b * n + j];
            const float miou = fg_iou[(size_t)b * n + j];
            const float* tg = targets + ((size_t)b * d.maxbox + grow) * 6;
            const float tb[4] = {tg[0], tg[1], tg[2], tg[3]};
            const int tcls = (int)tg[4];
            const size_t roff = ((size_t)b * n + cell) * st.ld;
            const T* row = pred + roff;
            float lg[4];
            {
                float mine = (sub < 4) ? ldx<T>(row + sub) : 0.f;
#pragma unroll
                for (int i = 0; i < 4; ++i) lg[i] = __shfl(mine, i, 16);
            }
            const float gxc = (float)(cell % st.W), gyc = (float)(cell / st.W);
            const float pb[4] = {(lg[0] + gxc) * st.stride, (lg[1] + gyc) * st.stride, expf(lg[2]) * st.stride, expf(lg[3]) * st.stride};
            const float tl1[4] = {tb[0] / st.stride - gxc, tb[1] / st.stride - gyc, logf(tb[2] / st.stride + 1e-16f), logf(tb[3] / st.stride + 1e-16f)};
            const Dual il = iou_loss_dual(d.iou_type, pb, tb);
            float l1 = 0.f;
            if (d.use_l1) l1 = (fabsf(lg[0] - tl1[0]) + fabsf(lg[1] - tl1[1]) + fabsf(lg[2] - tl1[2]) + fabsf(lg[3] - tl1[3])) / 4.f;
            // class loss over this lane's classes; target = onehot * smooth * matched_iou
            float csum = 0.f, dmiou = 0.f;
            for (int e = 5 + sub; e < E; e += 16) {
                const float x = ldx<T>(row + e);
                const bool hot = (e - 5 == tcls);
                const float tt = hot ? d.cls_smooth * miou : 0.f;
                float dx, dtb, fdx = 0.f, fdt = 0.f, f = 1.f;
                const float l = bce_logits(x, tt, d.cls_pos_weight, &dx, &dtb);
                if (d.use_focal) f = focal_factor(x, tt, d.focal_gamma, d.focal_alpha, &fdx, &fdt);
                csum += l * f;
                if (BWD) {
                    stx<T>(gp + roff + e, kcls * (dx * f + l * fdx) / (float)nc);
                    if (hot) dmiou = kcls * (dtb * f + l * fdt) * d.cls_smooth / (float)nc;
                }
            }
#pragma unroll
            for (int o = 8; o > 0; o >>= 1) { csum += __shfl_xor(csum, o, 16); dmiou += __shfl_xor(dmiou, o, 16); }
            if (sub == 0) {
                a_iou += (double)il.v; a_l1 += (double)l1; a_cls += (double)(csum / (float)nc);
                if (BWD) {
                    const Dual mi = matched_iou_dual(pb, tb);       // class target carries gradient into the box (:150)
                    float gb[4];
#pragma unroll
                    for (int i = 0; i < 4; ++i) gb[i] = kiou * il.g[i] + dmiou * mi.g[i];
                    float gl[4] = {gb[0] * st.stride, gb[1] * st.stride, gb[2] * pb[2], gb[3] * pb[3]};
                    if (d.use_l1) {
#pragma unroll
                        for (int i = 0; i < 4; ++i) {
                            const float df = lg[i] - tl1[i];
                            gl[i] += kl1 * (df > 0.f ? 1.f : (df < 0.f ? -1.f : 0.f)) / 4.f;
                        }
                    }
#pragma unroll
                    for (int i = 0; i < 4; ++i) stx<T>(gp + roff + i, gl[i]);
                }
            }
        }
    }
    if (!BWD) {
        a_iou = wave_sum_d(a_iou); a_l1 = wave_sum_d(a_l1); a_cls = wave_sum_d(a_cls);
        if ((t & 63) == 0) { sred[0][t >> 6] = a_iou; sred[1][t >> 6] = a_l1; sred[2][t >> 6] = a_cls; }
        __syncthreads();
        if (t == 0) {
            for (int k = 0; k < 3; ++k)
                part[((size_t)s * 4 + k) * PARTS + blockIdx.y * gridDim.x + blockIdx.x] = sred[k][0] + sred[k][1] + sred[k][2] + sred[k][3];
        }
    }
}

template <typename T>
__global__ __launch_bounds__(256) void yolox_obj_fwd_kernel(const XK p, const StagesX sts, const unsigned char* __restrict__ svb,
                                                            double* __restrict__ part)
{
    const StageX st = sts.s[blockIdx.y];          // one launch for all stages
    __shared__ double sred[4];
    const yh_yolox_desc& d = p.d;
    const int s = st.s, n = st.H * st.W;
    const long tot = (long)d.B * n;
    const int32_t* cellmap = reinterpret_cast<const int32_t*>(svb + p.L.cellmap) + p.L.cell_off[s];
    const T* pred = reinterpret_cast<const T*>(st.pred);
    double acc = 0.0;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < tot; i += (long)gridDim.x * blockDim.x) {
        const float x = ldx<T>(pred + (size_t)i * st.ld + 4);
        const float tt = cellmap[i] >= 0 ? 1.f : 0.f;
        float l = bce_logits(x, tt, d.cof_pos_weight, nullptr, nullptr);
        if (d.use_focal) l *= focal_factor(x, tt, d.focal_gamma, d.focal_alpha, nullptr, nullptr);
        acc += (double)l;
    }
    acc = wave_sum_d(acc);
    if ((threadIdx.x & 63) == 0) sred[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) part[((size_t)s * 4 + 3) * PARTS + blockIdx.x] = sred[0] + sred[1] + sred[2] + sred[3];
}

__global__ __launch_bounds__(1024) void yolox_finalize_kernel(const XK p, const double* __restrict__ part, int nb_fg, int nb_obj,
                                                              const float* __restrict__ targets, unsigned char* __restrict__ svb,
                                                              double* balances, float* result)
{
    __shared__ double ssum[MAXS * 4];
    __shared__ int sfg[MAXS];
    __shared__ int sgt;
    const yh_yolox_desc& d = p.d;
    const int S = d.num_stage;
    const int wv = threadIdx.x >> 6, lane = threadIdx.x & 63;
    if (wv < S * 4) {
        const int kind = wv & 3;
        const int nbb = kind == 3 ? nb_obj : nb_fg;
        double a = 0.0;
        for (int i = lane; i < nbb; i += 64) a += part[(size_t)wv * PARTS + i];
        a = wave_sum_d(a);
        if (lane == 0) ssum[wv] = a;
    }
    if (threadIdx.x < S) {
        const int32_t* fc = reinterpret_cast<const int32_t*>(svb + p.L.fg_count) + threadIdx.x * d.B;
        int a = 0;
        for (int b = 0; b < d.B; ++b) a += fc[b];
        sfg[threadIdx.x] = a;
    }
    if (threadIdx.x == 0) sgt = 0;
    __syncthreads();
    {
        int g = 0;
        for (int i = threadIdx.x; i < d.B * d.maxbox; i += 1024) g += targets[(size_t)i * 6 + 4] >= 0.f;
        for (int o = 32; o > 0; o >>= 1) g += __shfl_xor(g, o, 64);
        if (lane == 0 && g) atomicAdd(&sgt, g);
    }
    __syncthreads();
    if (threadIdx.x != 0) return;
    int32_t* nfg_stage = reinterpret_cast<int32_t*>(svb + p.L.nfg_stage);
    double* bal_used = reinterpret_cast<double*>(svb + p.L.bal_used);
    double ti = 0, tl = 0, tc = 0, to = 0;
    long nfg_tot = 0;
    for (int s = 0; s < S; ++s) {
        const int nf = sfg[s] > 1 ? sfg[s] : 1;          // tot_fg_num = max(tot_fg_num, 1) (:205)
        nfg_stage[s] = nf;
        nfg_tot += nf;
        ti += (double)(float)(ssum[s * 4 + 0] / nf);
        tl += (double)(float)(ssum[s * 4 + 1] / nf);
        tc += (double)(float)(ssum[s * 4 + 2] / nf);
        const double cof = (double)(float)(ssum[s * 4 + 3] / nf);
        const double bal = balances[s];
        bal_used[s] = bal;
        const double tmp = (double)(float)(cof * bal);
        balances[s] = bal * 0.9999 + 0.0001 / tmp;       // :64
        to += tmp;
    }
    const double b1 = balances[1];
    for (int s = 0; s < S; ++s) balances[s] /= b1;       // :74
    ti *= d.iou_scale; tc *= d.cls_scale; to *= d.cof_scale; tl *= d.l1_scale;
    result[0] = (float)(ti + tc + to + tl);
    result[1] = (float)ti; result[2] = (float)tl; result[3] = (float)tc; result[4] = (float)to;
    result[5] = (float)nfg_tot;
    result[6] = (float)(sgt * S);                        // tot_num_gt accumulates over the stages (:68)
    result[7] = 0.f;
}

// whole gradient tensor: zeros + objectness gradient (channel 4); fg_bwd then fills the other channels of fg cells.
// Tiles of 256 cells: one thread per cell evaluates the objectness gradient (dense lanes) into LDS, then the block
// streams the tile's 16-byte chunks.
template <typename T>
__global__ __launch_bounds__(256) void yolox_obj_bwd_kernel(const XK p, const StagesX sts, const unsigned char* __restrict__ svb,
                                                            const float* __restrict__ gout)
{
    const StageX st = sts.s[blockIdx.y];          // one launch for all stages
    constexpr int TP = 256;
    __shared__ float sG[TP];
    const yh_yolox_desc& d = p.d;
    const int s = st.s, n = st.H * st.W;
    const int cpr = st.ld / 8;
    const long ncell = (long)d.B * n;
    const int t = threadIdx.x;
    const int32_t* cellmap = reinterpret_cast<const int32_t*>(svb + p.L.cellmap) + p.L.cell_off[s];
    const int32_t* nfg_stage = reinterpret_cast<const int32_t*>(svb + p.L.nfg_stage);
    const double* bal_used = reinterpret_cast<const double*>(svb + p.L.bal_used);
    const T* pred = reinterpret_cast<const T*>(st.pred);
    T* gp = reinterpret_cast<T*>(st.gpred);
    const float coef = (float)((double)(*gout) * d.cof_scale * bal_used[s] / (double)(nfg_stage[s] > 1 ? nfg_stage[s] : 1));
    const long ntile = (ncell + TP - 1) / TP;
    for (long tile = blockIdx.x; tile < ntile; tile += gridDim.x) {
        const long cell0 = tile * TP;
        {
            const long cell = cell0 + t;
            float gv = 0.f;
            if (cell < ncell) {
                const float x = ldx<T>(pred + (size_t)cell * st.ld + 4);
                const float tt = cellmap[cell] >= 0 ? 1.f : 0.f;
                float dx, fdx = 0.f, f = 1.f;
                const float l = bce_logits(x, tt, d.cof_pos_weight, &dx, nullptr);
                if (d.use_focal) f = focal_factor(x, tt, d.focal_gamma, d.focal_alpha, &fdx, nullptr);
                gv = coef * (dx * f + l * fdx);
            }
            sG[t] = gv;
        }
        __syncthreads();
        for (int i = t; i < TP * cpr; i += 256) {
            const int pl = i / cpr;
            const int c0 = (i - pl * cpr) * 8;
            const long cell = cell0 + pl;
            if (cell >= ncell) break;
            float g[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) g[e] = 0.f;
            if (c0 == 0) g[4] = sG[pl];
            T* dst = gp + (size_t)cell * st.ld + c0;
            if (sizeof(T) == 2) *reinterpret_cast<uint4*>(dst) = pack8(g);
            else {
                float4* d4 = reinterpret_cast<float4*>(dst);
                d4[0] = make_float4(g[0], g[1], g[2], g[3]);
                d4[1] = make_float4(g[4], g[5], g[6], g[7]);
            }
        }
        __syncthreads();
    }
}

int check_xdesc(const yh_yolox_desc* d, const char* who) {
    YH_CHECK_ARG(d != nullptr, "%s: null desc", who);
    YH_CHECK_ARG(d->B > 0 && d->maxbox > 0 && d->num_class >= 1 && 5 + d->num_class <= 128, "%s: bad B/maxbox/classes", who);
    YH_CHECK_ARG(d->num_stage >= 1 && d->num_stage <= MAXS, "%s: bad num_stage", who);
    YH_CHECK_ARG(d->iou_type >= 0 && d->iou_type <= 2 && d->topk >= 1, "%s: bad iou_type/topk", who);
    for (int s = 0; s < d->num_stage; ++s)
        YH_CHECK_ARG(d->H[s] > 0 && d->W[s] > 0 && d->ldp[s] % 8 == 0 && d->ldp[s] >= 5 + d->num_class && (long)d->H[s] * d->W[s] <= (1 << 20),
                     "%s: stage %d dims / ld invalid", who, s);
    return YH_OK;
}

}  // namespace

extern "C" size_t yh_yolox_saved_bytes(const yh_yolox_desc* d) { return d ? make_xlayout(*d).total : 0; }
extern "C" size_t yh_yolox_ws_bytes(const yh_yolox_desc* d) { return d ? make_xlayout(*d).w_total : 0; }
/* byte offsets inside `saved`: out[0]=fg_count [S][B] i32, out[1]=fg_cell, out[2]=fg_gt, out[3]=fg_iou (per-cell arrays),
 * out[4..7] = element offset of each stage inside the per-cell arrays */
extern "C" int yh_yolox_layout(const yh_yolox_desc* d, int64_t* out) {
    YH_CHECK_ARG(d && out, "yh_yolox_layout: null");
    XLayout L = make_xlayout(*d);
    out[0] = (int64_t)L.fg_count; out[1] = (int64_t)L.fg_cell; out[2] = (int64_t)L.fg_gt; out[3] = (int64_t)L.fg_iou;
    for (int s = 0; s < MAXS; ++s) out[4 + s] = (int64_t)L.cell_off[s];
    return YH_OK;
}

extern "C" int yh_yolox_loss_fwd(const yh_yolox_desc* d, const void* const* preds, const float* targets_xywh,
                                 double* balances, float* result, void* saved, void* ws, yh_stream stream)
{
    int rc = check_xdesc(d, "yh_yolox_loss_fwd");
    if (rc) return rc;
    YH_CHECK_ARG(preds && targets_xywh && balances && result && saved && ws, "yh_yolox_loss_fwd: null pointer");
    hipStream_t st = (hipStream_t)stream;
    XK k; k.d = *d; k.L = make_xlayout(*d);
    unsigned char* sv = (unsigned char*)saved;
    unsigned char* wsb = (unsigned char*)ws;
    double* part = reinterpret_cast<double*>(wsb + k.L.w_part);
    const int nb_fg = 256, nb_obj = 512;
    const dim3 fg_grid(4, nb_fg / 4);              // x: shares of an image's foreground list, y: images
    StagesX all;
    for (int s = 0; s < d->num_stage; ++s) {
        YH_CHECK_ARG(preds[s] && yh_aligned16(preds[s]), "yh_yolox_loss_fwd: preds[%d] null/unaligned", s);
        StageX& sg = all.s[s];
        sg.pred = preds[s]; sg.gpred = nullptr; sg.s = s; sg.H = d->H[s]; sg.W = d->W[s]; sg.ld = d->ldp[s];
        sg.stride = d->img_size0 / (float)d->H[s];
    }
    if (d->pred_is_f32) hipLaunchKernelGGL((yolox_assign_kernel<float>), dim3(d->B, d->num_stage), dim3(1024), 0, st, k, all, targets_xywh, wsb, sv);
    else                hipLaunchKernelGGL((yolox_assign_kernel<uint16_t>), dim3(d->B, d->num_stage), dim3(1024), 0, st, k, all, targets_xywh, wsb, sv);
    for (int s = d->num_stage; s < MAXS; ++s) all.s[s] = all.s[0];
    const dim3 fg_grid3(fg_grid.x, fg_grid.y, d->num_stage);
    if (d->pred_is_f32) {
        hipLaunchKernelGGL((yolox_fg_kernel<float, false>), fg_grid3, dim3(256), 0, st, k, all, targets_xywh, sv, part, (const float*)nullptr);
        hipLaunchKernelGGL((yolox_obj_fwd_kernel<float>), dim3(nb_obj, d->num_stage), dim3(256), 0, st, k, all, sv, part);
    } else {
        hipLaunchKernelGGL((yolox_fg_kernel<uint16_t, false>), fg_grid3, dim3(256), 0, st, k, all, targets_xywh, sv, part, (const float*)nullptr);
        hipLaunchKernelGGL((yolox_obj_fwd_kernel<uint16_t>), dim3(nb_obj, d->num_stage), dim3(256), 0, st, k, all, sv, part);
    }
    hipLaunchKernelGGL(yolox_finalize_kernel, dim3(1), dim3(1024), 0, st, k, part, nb_fg, nb_obj, targets_xywh, sv, balances, result);
    YH_CHECK_LAUNCH("yh_yolox_loss_fwd");
    return YH_OK;
}

extern "C" int yh_yolox_loss_bwd(const yh_yolox_desc* d, const void* const* preds, const float* targets_xywh, const float* gout,
                                 const void* saved, void* const* gpreds, yh_stream stream)
{
    int rc = check_xdesc(d, "yh_yolox_loss_bwd");
    if (rc) return rc;
    YH_CHECK_ARG(preds && targets_xywh && gout && saved && gpreds, "yh_yolox_loss_bwd: null pointer");
    hipStream_t st = (hipStream_t)stream;
    XK k; k.d = *d; k.L = make_xlayout(*d);
    const unsigned char* sv = (const unsigned char*)saved;
    StagesX all;
    long ntile_max = 1;
    for (int s = 0; s < d->num_stage; ++s) YH_CHECK_ARG(preds[s] && gpreds[s] && yh_aligned16(gpreds[s]), "yh_yolox_loss_bwd: stage %d pointers null/unaligned", s);
    for (int s = 0; s < MAXS; ++s) {
        const int q = s < d->num_stage ? s : 0;
        StageX& sg = all.s[s];
        sg.pred = preds[q]; sg.gpred = gpreds[q]; sg.s = q; sg.H = d->H[q]; sg.W = d->W[q]; sg.ld = d->ldp[q];
        sg.stride = d->img_size0 / (float)d->H[q];
        const long ntile = ((long)d->B * sg.H * sg.W + 255) / 256;
        if (s < d->num_stage && ntile > ntile_max) ntile_max = ntile;
    }
    const int gb = (int)(ntile_max > 4096 ? 4096 : ntile_max);
    // objectness pass (writes every stage's whole gradient tensor), then the foreground cells fill their channels: two launches
    if (d->pred_is_f32) {
        hipLaunchKernelGGL((yolox_obj_bwd_kernel<float>), dim3(gb, d->num_stage), dim3(256), 0, st, k, all, sv, gout);
        hipLaunchKernelGGL((yolox_fg_kernel<float, true>), dim3(4, 64, d->num_stage), dim3(256), 0, st, k, all, targets_xywh, sv, (double*)nullptr, gout);
    } else {
        hipLaunchKernelGGL((yolox_obj_bwd_kernel<uint16_t>), dim3(gb, d->num_stage), dim3(256), 0, st, k, all, sv, gout);
        hipLaunchKernelGGL((yolox_fg_kernel<uint16_t, true>), dim3(4, 64, d->num_stage), dim3(256), 0, st, k, all, targets_xywh, sv, (double*)nullptr, gout);
    }
    YH_CHECK_LAUNCH("yh_yolox_loss_bwd");
    return YH_OK;
}
