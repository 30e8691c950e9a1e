// YOLOv5 loss on gfx950 (loss/yolov5_loss.py:30-235 of the reference):
//   assign   — anchor-ratio filter + 5-neighbour grid expansion, order-preserving
//              compaction with wave ballots (YOLOV5Loss.match, :142-214); indices are
//              bit-exact with the reference (same fp32 operation order, no FMA).
//   pos_fwd  — per positive (16-lane group): gather 5+nc logits, CIoU, focal-BCE class loss;
//              positives sharing a cell are chained in a per-cell list (t_cof scatter with
//              last-writer-wins == largest row index, :114).
//   obj_fwd  — focal-BCE objectness over every cell (:116-123).
//   finalize — deterministic reduction, loss scaling and the stateful `balances` EMA (:124-131).
//   backward — obj_bwd streams the whole gradient tensor once (zeros + objectness grads),
//              pos_bwd overwrites the (5+nc) channels of cells that own positives.
// Predictions are NHWC (cell-major) [B][H][W][ld] with channel a*(5+nc)+e, bf16 or fp32.
#include "common.h"

namespace {

constexpr int MAXS = 4;

struct Layout {                // byte offsets inside `saved`
    size_t count, bal_used, tbox, tidx, ciou, next, head, total;
    size_t head_off[MAXS];     // element offsets of each stage inside head
    int cap;
    int ncell[MAXS];
};

Layout make_layout(const yh_v5loss_desc& d) {
    Layout L;
    L.cap = 5 * d.num_anchor * d.B * d.maxbox;
    size_t o = 0;
    L.count = o; o += 64;
    L.bal_used = o; o += 64;
    L.tbox = o; o += (size_t)d.num_stage * L.cap * 4 * sizeof(float);
    L.tidx = o; o += (size_t)d.num_stage * L.cap * 5 * sizeof(int32_t);
    L.ciou = o; o += (size_t)d.num_stage * L.cap * sizeof(float);
    L.next = o; o += (size_t)d.num_stage * L.cap * sizeof(int32_t);
    o = (o + 255) & ~(size_t)255;
    L.head = o;
    size_t e = 0;
    for (int s = 0; s < d.num_stage; ++s) {
        L.head_off[s] = e;
        L.ncell[s] = d.B * d.num_anchor * d.H[s] * d.W[s];
        e += L.ncell[s];
    }
    o += e * sizeof(int32_t);
    L.total = (o + 255) & ~(size_t)255;
    return L;
}

constexpr int PART_BLOCKS = 1024;   // partial-sum slots per (stage, kind)
// ws layout: double part[MAXS][3][PART_BLOCKS]
size_t ws_bytes() { return sizeof(double) * MAXS * 3 * PART_BLOCKS; }

struct LossK {
    yh_v5loss_desc d;
    Layout L;
};

__device__ __forceinline__ float remainder1(float a) {     // torch.remainder(a, 1.0)
    float m = fmodf(a, 1.0f);
    if (m != 0.f && m < 0.f) m += 1.0f;
    return m;
}

// ---------------------------------------------------------------- assignment
__global__ __launch_bounds__(1024) void v5_assign_kernel(const LossK p, const float* __restrict__ targets,
                                                         int32_t* __restrict__ count, float* __restrict__ tbox,
                                                         int32_t* __restrict__ tidx)
{
    __shared__ int wave_cnt[16];
    __shared__ int base_s;
    const yh_v5loss_desc& d = p.d;
    const int s = blockIdx.x;
    const int t = threadIdx.x, lane = t & 63, wv = t >> 6;
    const int A = d.num_anchor, B = d.B, MB = d.maxbox;
    const int per_k = A * B * MB;
    const int T = 5 * per_k;
    const float fw = (float)d.W[s], fh = (float)d.H[s];
    const float ds = d.img_size1 / fw;                       // :66 ds_scale
    if (t == 0) base_s = 0;
    __syncthreads();
    float* ob = tbox + (size_t)s * p.L.cap * 4;
    int32_t* oi = tidx + (size_t)s * p.L.cap * 5;

    for (int q0 = 0; q0 < T; q0 += 1024) {
        const int q = q0 + t;
        bool flag = false;
        float gxf = 0, gyf = 0, gw = 0, gh = 0, offx = 0, offy = 0;
        int a = 0, cls = 0, img = 0;
        if (q < T) {
            const int k = q / per_k;
            int r = q - k * per_k;
            a = r / (B * MB);
            r -= a * (B * MB);
            const float* tg = targets + (size_t)r * 6;      // r = b*MB + j
            const float x1 = tg[0], y1 = tg[1], x2 = tg[2], y2 = tg[3];
            // xyxy2xywhn (utils/bbox_tools.py:103-119), then * fm size (:151-153)
            float cxn, cyn, wn, hn;
            if (d.targets_xywhn) { cxn = x1; cyn = y1; wn = x2; hn = y2; }
            else {
                cxn = ((x1 + x2) / 2.0f) / d.img_size0;
                cyn = ((y1 + y2) / 2.0f) / d.img_size1;
                wn = (x2 - x1) / d.img_size0;
                hn = (y2 - y1) / d.img_size1;
            }
            gxf = cxn * fw; gyf = cyn * fh; gw = wn * fw; gh = hn * fh;
            const float aw = d.anchors[s][a][0] / ds, ah = d.anchors[s][a][1] / ds;
            const float rw = gw / aw + 1e-16f, rh = gh / ah + 1e-16f;
            const float m = fmaxf(fmaxf(rw, 1.0f / rw), fmaxf(rh, 1.0f / rh));
            bool ok = m < d.anchor_thr;                     // :170
            if (ok) {
                const float ox = fw - gxf, oy = fh - gyf;   // :175
                switch (k) {                                // :180-186
                    case 0: break;
                    case 1: ok = (remainder1(gxf) < 0.5f) && (gxf > 1.0f); offx = 0.5f; break;
                    case 2: ok = (remainder1(gyf) < 0.5f) && (gyf > 1.0f); offy = 0.5f; break;
                    case 3: ok = (remainder1(ox) < 0.5f) && (ox > 1.0f); offx = -0.5f; break;
                    default: ok = (remainder1(oy) < 0.5f) && (oy > 1.0f); offy = -0.5f; break;
                }
            }
            flag = ok;
            cls = (int)tg[4];
            img = (int)tg[5];
        }
        const unsigned long long bal = __ballot(flag);
        const int within = __popcll(bal & ((1ull << lane) - 1ull));
        if (lane == 0) wave_cnt[wv] = __popcll(bal);
        __syncthreads();
        int before = 0, total = 0;
#pragma unroll
        for (int w = 0; w < 16; ++w) {
            int c = wave_cnt[w];
            if (w < wv) before += c;
            total += c;
        }
        const int base = base_s;
        if (flag) {
            const int pos = base + before + within;
            const long cxc = (long)(gxf - offx);            // .long() truncates toward zero (:196)
            const long cyc = (long)(gyf - offy);
            float* b = ob + (size_t)pos * 4;
            b[0] = gxf - (float)cxc; b[1] = gyf - (float)cyc; b[2] = gw; b[3] = gh;   // offsets before the clamp (:198)
            int gx = (int)(cxc < 0 ? 0 : (cxc > d.W[s] - 1 ? d.W[s] - 1 : cxc));
            int gy = (int)(cyc < 0 ? 0 : (cyc > d.H[s] - 1 ? d.H[s] - 1 : cyc));
            int32_t* ii = oi + (size_t)pos * 5;
            ii[0] = cls; ii[1] = img; ii[2] = a; ii[3] = gy; ii[4] = gx;
        }
        __syncthreads();
        if (t == 0) base_s = base + total;
        __syncthreads();
    }
    if (t == 0) count[s] = base_s;
}

// ---------------------------------------------------------------- math helpers
__device__ __forceinline__ float sigm(float x) { return 1.0f / (1.0f + expf(-x)); }

// BCEWithLogits(x, t, pos_weight) as torch computes it, and d/dx
__device__ __forceinline__ float bce_logits(float x, float t, float pw, float* dx) {
    const float lw = 1.0f + (pw - 1.0f) * t;
    const float sp = log1pf(expf(-fabsf(x))) + fmaxf(-x, 0.0f);     // softplus(-x)
    const float sg = sigm(x);
    if (dx) *dx = (1.0f - t) - lw * (1.0f - sg);
    return (1.0f - t) * x + lw * sp;
}
// focal factor (:216-235) and d/dx
__device__ __forceinline__ float focal_factor(float x, float t, float gamma, float alpha, float* dx) {
    const float pr = sigm(x);
    const float acc = t * pr + (1.0f - t) * (1.0f - pr);
    const float one_m = 1.0f - acc;
    const float gf = powf(one_m, gamma);
    const float af = t * alpha + (1.0f - t) * (1.0f - alpha);
    if (dx) {
        // d(1-acc)/dx = -(2t-1) p(1-p)
        const float dgf = (one_m > 0.f) ? gamma * powf(one_m, gamma - 1.0f) * (-(2.0f * t - 1.0f) * pr * (1.0f - pr)) : 0.f;
        *dx = dgf * af;
    }
    return gf * af;
}

// CIoU of xyxy boxes (utils/bbox_tools.py:286-339) + gradient w.r.t. b1, alpha constant
__device__ float ciou_fwd_bwd(const float* b1, const float* b2, float* g /*4 or null*/)
{
    const float eps = 1e-9f;
    const float x1 = b1[0], y1 = b1[1], x2 = b1[2], y2 = b1[3];
    const float X1 = b2[0], Y1 = b2[1], X2 = b2[2], Y2 = b2[3];
    const float w1 = x2 - x1, h1 = y2 - y1, w2 = X2 - X1, h2 = Y2 - Y1;
    const float ixmax = fminf(x2, X2), ixmin = fmaxf(x1, X1);
    const float iymax = fminf(y2, Y2), iymin = fmaxf(y1, Y1);
    const float iwr = ixmax - ixmin, ihr = iymax - iymin;
    const float iw = fmaxf(iwr, 0.f), ih = fmaxf(ihr, 0.f);
    const float inter = iw * ih;
    const float union_raw = w1 * h1 + w2 * h2 - inter;
    const float uni = fmaxf(union_raw, eps);
    const float iou = inter / uni;
    const float cxmin = fminf(x1, X1), cxmax = fmaxf(x2, X2);
    const float cymin = fminf(y1, Y1), cymax = fmaxf(y2, Y2);
    const float c_hs = cymax - cymin, c_ws = cxmax - cxmin;
    const float cd_raw = c_ws * c_ws + c_hs * c_hs;
    const float c1x = (x1 + x2) / 2.f, c1y = (y1 + y2) / 2.f;
    const float c2x = (X1 + X2) / 2.f, c2y = (Y1 + Y2) / 2.f;
    const float ctr_ws = c1x - c2x, ctr_hs = c1y - c2y;
    const float ctr = ctr_hs * ctr_hs + ctr_ws * ctr_ws;
    const float kk = (float)(4.0 / (3.14159265358979323846 * 3.14159265358979323846));
    const float h1c = fmaxf(h1, eps), h2c = fmaxf(h2, eps);
    const float u1 = w1 / h1c;
    const float dA = atanf(u1) - atanf(w2 / h2c);
    const float v = kk * dA * dA;
    const float alpha = v / fmaxf(1.f - iou + v, eps);
    const float cd = fmaxf(cd_raw, eps);
    const float ciou = iou - (ctr / cd + v * alpha);
    if (g) {
        // order of unknowns: x1, y1, x2, y2
        // torch.min/max split the gradient evenly on ties (minimum/maximum backward); clamp(min=0) passes at >= 0
        auto lt = [](float a, float b) { return a < b ? 1.f : (a == b ? 0.5f : 0.f); };
        const float pw_ = (iwr >= 0.f) ? 1.f : 0.f, ph_ = (ihr >= 0.f) ? 1.f : 0.f;
        const float diw[4] = {-pw_ * lt(X1, x1), 0.f, pw_ * lt(x2, X2), 0.f};
        const float dih[4] = {0.f, -ph_ * lt(Y1, y1), 0.f, ph_ * lt(y2, Y2)};
        const float dw1[4] = {-1.f, 0.f, 1.f, 0.f};
        const float dh1[4] = {0.f, -1.f, 0.f, 1.f};
        const float dcw[4] = {-lt(x1, X1), 0.f, lt(X2, x2), 0.f};
        const float dch[4] = {0.f, -lt(y1, Y1), 0.f, lt(Y2, y2)};
        const float dcx[4] = {0.5f, 0.f, 0.5f, 0.f};
        const float dcy[4] = {0.f, 0.5f, 0.f, 0.5f};
        const float du_dw = 1.f / h1c;
        const float du_dh = (h1 >= eps) ? -w1 / (h1c * h1c) : 0.f;
        const float datan = 1.f / (1.f + u1 * u1);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const float dinter = ih * diw[i] + iw * dih[i];
            const float dunion = (union_raw >= eps) ? (dw1[i] * h1 + w1 * dh1[i] - dinter) : 0.f;
            const float diou = (dinter * uni - inter * dunion) / (uni * uni);
            const float dcd = (cd_raw >= eps) ? (2.f * c_ws * dcw[i] + 2.f * c_hs * dch[i]) : 0.f;
            const float dctr = 2.f * ctr_ws * dcx[i] + 2.f * ctr_hs * dcy[i];
            const float dterm = (dctr * cd - ctr * dcd) / (cd * cd);
            const float dv = 2.f * kk * dA * datan * (du_dw * dw1[i] + du_dh * dh1[i]);
            g[i] = diou - dterm - alpha * dv;
        }
    }
    return ciou;
}

template <typename T> __device__ __forceinline__ float ldp(const T* p);
template <> __device__ __forceinline__ float ldp<float>(const float* p) { return *p; }
template <> __device__ __forceinline__ float ldp<uint16_t>(const uint16_t* p) { return bf2f(*p); }
template <typename T> __device__ __forceinline__ void stp(T* p, float v);
template <> __device__ __forceinline__ void stp<float>(float* p, float v) { *p = v; }
template <> __device__ __forceinline__ void stp<uint16_t>(uint16_t* p, float v) { *p = f2bf(v); }

struct Stage {
    const void* pred;
    void* gpred;
    int s, H, W, ld;
};
struct Stages { Stage s[4]; };

// Decode the predicted box of one positive and evaluate CIoU (+ grads w.r.t. the 4 logits)
__device__ __forceinline__ float pos_box(const yh_v5loss_desc& d, int s, int anc, const float* lg, const float* tb,
                                         float* dlogit /*4 or null*/)
{
    const float ds = d.img_size1 / (float)d.W[s];
    const float aw = d.anchors[s][anc][0] / ds, ah = d.anchors[s][anc][1] / ds;
    const float s0 = sigm(lg[0]), s1 = sigm(lg[1]), s2 = sigm(lg[2]), s3 = sigm(lg[3]);
    const float px = s0 * 2.f - 0.5f, py = s1 * 2.f - 0.5f;
    const float t2 = s2 * 2.f, t3 = s3 * 2.f;
    const float pw = t2 * t2 * aw, ph = t3 * t3 * ah;
    const float b1[4] = {px - pw / 2.f, py - ph / 2.f, px + pw / 2.f, py + ph / 2.f};
    const float b2[4] = {tb[0] - tb[2] / 2.f, tb[1] - tb[3] / 2.f, tb[0] + tb[2] / 2.f, tb[1] + tb[3] / 2.f};
    float g[4];
    const float c = ciou_fwd_bwd(b1, b2, dlogit ? g : nullptr);
    if (dlogit) {
        const float dpx = g[0] + g[2], dpy = g[1] + g[3];
        const float dpw = (g[2] - g[0]) / 2.f, dph = (g[3] - g[1]) / 2.f;
        dlogit[0] = dpx * 2.f * s0 * (1.f - s0);
        dlogit[1] = dpy * 2.f * s1 * (1.f - s1);
        dlogit[2] = dpw * (2.f * t2 * aw) * (2.f * s2 * (1.f - s2));
        dlogit[3] = dph * (2.f * t3 * ah) * (2.f * s3 * (1.f - s3));
    }
    return c;
}

// ---------------------------------------------------------------- positives, forward
template <typename T>
__global__ __launch_bounds__(256) void v5_pos_fwd_kernel(const LossK p, const Stages sts, const int32_t* __restrict__ count,
                                                         const float* __restrict__ tbox, const int32_t* __restrict__ tidx,
                                                         float* __restrict__ ciou_out, int32_t* __restrict__ next,
                                                         int32_t* __restrict__ head, double* __restrict__ part)
{
    const Stage st = sts.s[blockIdx.y];          // one launch for all stages: they are independent of each other
    __shared__ double sred[2][4];
    const yh_v5loss_desc& d = p.d;
    const int s = st.s;
    const int N = count[s];
    const int E = 5 + d.num_class;
    const int t = threadIdx.x, sub = t & 15, grp = t >> 4;
    const T* pred = reinterpret_cast<const T*>(st.pred);
    double acc_iou = 0.0, acc_cls = 0.0;
    for (int n = blockIdx.x * 16 + grp; n < N; n += gridDim.x * 16) {
        const int32_t* ii = tidx + ((size_t)s * p.L.cap + n) * 5;
        const float* tb = tbox + ((size_t)s * p.L.cap + n) * 4;
        const int cls = ii[0], img = ii[1], anc = ii[2], gy = ii[3], gx = ii[4];
        if (img < 0 || img >= d.B) continue;        // malformed img_id: the reference would index out of range
        const T* row = pred + (((size_t)img * st.H + gy) * st.W + gx) * st.ld + anc * E;
        float lg[4];
        {
            float mine = (sub < 4) ? ldp<T>(row + sub) : 0.f;
#pragma unroll
            for (int i = 0; i < 4; ++i) lg[i] = __shfl(mine, i, 16);
        }
        const float c = pos_box(d, s, anc, lg, tb, nullptr);
        float cls_sum = 0.f;
        if (d.num_class > 1) {
            for (int e = 5 + sub; e < E; e += 16) {
                const float x = ldp<T>(row + e);
                const float tt = (e - 5 == cls) ? d.cls_smooth : 0.f;
                float l = bce_logits(x, tt, d.cls_pos_weight, nullptr);
                if (d.use_focal) l *= focal_factor(x, tt, d.focal_gamma, d.focal_alpha, nullptr);
                cls_sum += l;
            }
#pragma unroll
            for (int o = 8; o > 0; o >>= 1) cls_sum += __shfl_xor(cls_sum, o, 16);
        }
        if (sub == 0) {
            ciou_out[(size_t)s * p.L.cap + n] = c;
            acc_iou += (double)(1.0f - c);
            acc_cls += (double)cls_sum;
            const int cell = ((img * d.num_anchor + anc) * st.H + gy) * st.W + gx;
            const int prev = atomicExch(head + p.L.head_off[s] + cell, n);
            next[(size_t)s * p.L.cap + n] = prev;
        }
    }
    acc_iou = wave_sum_d(acc_iou);
    acc_cls = wave_sum_d(acc_cls);
    if ((t & 63) == 0) { sred[0][t >> 6] = acc_iou; sred[1][t >> 6] = acc_cls; }
    __syncthreads();
    if (t == 0) {
        part[((size_t)s * 3 + 0) * PART_BLOCKS + blockIdx.x] = sred[0][0] + sred[0][1] + sred[0][2] + sred[0][3];
        part[((size_t)s * 3 + 1) * PART_BLOCKS + blockIdx.x] = sred[1][0] + sred[1][1] + sred[1][2] + sred[1][3];
    }
}

// largest row index in a cell's list == the last writer of t_cof[...] = iou (:114)
__device__ __forceinline__ int list_max(const int32_t* next, int h) {
    int m = h;
    for (int it = 0; h >= 0 && it < (1 << 20); ++it) { m = h > m ? h : m; h = next[h]; }
    return m;
}

// ---------------------------------------------------------------- objectness, forward
template <typename T>
__global__ __launch_bounds__(256) void v5_obj_fwd_kernel(const LossK p, const Stages sts, const float* __restrict__ ciou,
                                                         const int32_t* __restrict__ next, const int32_t* __restrict__ head,
                                                         double* __restrict__ part)
{
    const Stage st = sts.s[blockIdx.y];          // one launch for all stages: they are independent of each other
    __shared__ double sred[4];
    const yh_v5loss_desc& d = p.d;
    const int s = st.s, A = d.num_anchor, E = 5 + d.num_class;
    const int ncell = p.L.ncell[s];
    const T* pred = reinterpret_cast<const T*>(st.pred);
    const int32_t* hd = head + p.L.head_off[s];
    const int32_t* nx = next + (size_t)s * p.L.cap;
    const float* ci = ciou + (size_t)s * p.L.cap;
    double acc = 0.0;
    // thread index enumerates (img, y, x, a) with a fastest so that neighbouring lanes share cache lines
    for (int id = blockIdx.x * blockDim.x + threadIdx.x; id < ncell; id += gridDim.x * blockDim.x) {
        const int a = id % A;
        const int pix = id / A;                               // (img*H + y)*W + x
        const int x = pix % st.W;
        const int r2 = pix / st.W;
        const int y = r2 % st.H;
        const int img = r2 / st.H;
        const float lg = ldp<T>(pred + (size_t)pix * st.ld + a * E + 4);
        const int h = hd[((img * A + a) * st.H + y) * st.W + x];
        float tt = 0.f;
        if (h >= 0) tt = fmaxf(ci[list_max(nx, h)], 0.f);
        float l = bce_logits(lg, tt, d.cof_pos_weight, nullptr);
        if (d.use_focal) l *= focal_factor(lg, tt, d.focal_gamma, d.focal_alpha, nullptr);
        acc += (double)l;
    }
    acc = wave_sum_d(acc);
    if ((threadIdx.x & 63) == 0) sred[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) part[((size_t)s * 3 + 2) * PART_BLOCKS + blockIdx.x] = sred[0] + sred[1] + sred[2] + sred[3];
}

// ---------------------------------------------------------------- finalize
__global__ __launch_bounds__(1024) void v5_finalize_kernel(const LossK p, const int32_t* __restrict__ count, const double* __restrict__ part,
                                   int nb_pos, int nb_obj, double* balances, double* bal_used, float* result)
{
    // deterministic tree: thread t sums the strided partials of (stage, kind) = t / 64; lanes combine in fixed order
    __shared__ double ssum[MAXS * 3];
    const yh_v5loss_desc& d = p.d;
    const int S = d.num_stage;
    {
        const int wv = threadIdx.x >> 6, lane = threadIdx.x & 63;
        if (wv < S * 3) {
            const int kind = wv % 3;
            const int n = kind == 2 ? nb_obj : nb_pos;
            double a = 0.0;
            for (int i = lane; i < n; i += 64) a += part[(size_t)wv * PART_BLOCKS + i];
            a = wave_sum_d(a);
            if (lane == 0) ssum[wv] = a;
        }
    }
    __syncthreads();
    if (threadIdx.x != 0) return;
    const double s3 = 3.0 / S;
    double iou_loss = 0.0, cof_loss = 0.0, cls_loss = 0.0;
    long tar = 0;
    for (int s = 0; s < S; ++s) {
        const double a = ssum[s * 3 + 0], b = ssum[s * 3 + 1], c = ssum[s * 3 + 2];
        const int N = count[s];
        tar += N;
        if (d.num_class > 1) cls_loss += (double)(float)(b / ((double)N * d.num_class));   // mean of an empty tensor is NaN, as in the reference
        if (N > 0) iou_loss += (double)(float)(a / N);
        const double cof_mean = (double)(float)(c / (double)p.L.ncell[s]);
        const double bal = balances[s];
        bal_used[s] = bal;
        const double cof_tmp = (double)(float)(cof_mean * bal);
        balances[s] = bal * 0.9999 + 0.0001 / cof_tmp;        // :124
        cof_loss += cof_tmp;
    }
    const double b1 = balances[1];
    for (int s = 0; s < S; ++s) balances[s] /= b1;            // :127
    iou_loss *= d.iou_scale * s3;
    cof_loss *= d.cof_scale * s3 * (S == 3 ? 1.0 : 1.4);
    cls_loss *= d.cls_scale * s3;
    result[0] = (float)((iou_loss + cof_loss + cls_loss) * d.B);
    result[1] = (float)(iou_loss * d.B);
    result[2] = (float)(cof_loss * d.B);
    result[3] = (float)(cls_loss * d.B);
    result[4] = (float)tar;
    result[5] = result[6] = result[7] = 0.f;
}

// ---------------------------------------------------------------- backward
// Whole gradient tensor = zeros + objectness gradients.  A block walks tiles of 64 pixels: first ONE THREAD PER
// (pixel, anchor) evaluates the objectness gradient (dense lanes: BCE / focal terms, positive-list lookup) into LDS,
// then all threads stream the tile's 16-byte chunks.  (One thread per chunk with the objectness math inlined left
// 6 of 64 lanes busy in that branch while every wave paid for it: 0.5 ms for the stride-8 head.)
template <typename T>
__global__ __launch_bounds__(256) void v5_obj_bwd_kernel(const LossK p, const Stages sts, const float* __restrict__ gout,
                                                         const double* __restrict__ bal_used, const float* __restrict__ ciou,
                                                         const int32_t* __restrict__ next, const int32_t* __restrict__ head)
{
    const Stage st = sts.s[blockIdx.y];          // one launch for all stages: they are independent of each other
    constexpr int TP = 64;                 // pixels per tile
    __shared__ float sG[TP * 4];
    const yh_v5loss_desc& d = p.d;
    const int s = st.s, A = d.num_anchor, E = 5 + d.num_class;
    const int cpr = st.ld / 8;
    const int npix = d.B * st.H * st.W;
    const int t = threadIdx.x;
    const T* pred = reinterpret_cast<const T*>(st.pred);
    T* gp = reinterpret_cast<T*>(st.gpred);
    const int32_t* hd = head + p.L.head_off[s];
    const int32_t* nx = next + (size_t)s * p.L.cap;
    const float* ci = ciou + (size_t)s * p.L.cap;
    const double s3 = 3.0 / d.num_stage;
    const float coef = (float)((double)(*gout) * d.B * d.cof_scale * s3 * (d.num_stage == 3 ? 1.0 : 1.4) * bal_used[s] / (double)p.L.ncell[s]);
    const int ntile = (npix + TP - 1) / TP;
    for (int tile = blockIdx.x; tile < ntile; tile += gridDim.x) {
        const int pix0 = tile * TP;
        if (t < TP * A) {
            const int pl = t / A, a = t - pl * A;
            const int pix = pix0 + pl;
            float gv = 0.f;
            if (pix < npix) {
                const int x = pix % st.W;
                const int r2 = pix / st.W;
                const int y = r2 % st.H;
                const int img = r2 / st.H;
                const float lg = ldp<T>(pred + (size_t)pix * st.ld + a * E + 4);
                const int h = hd[((img * A + a) * st.H + y) * st.W + x];
                float tt = 0.f;
                if (h >= 0) tt = fmaxf(ci[list_max(nx, h)], 0.f);
                float dl, df = 0.f, f = 1.f;
                const float l = bce_logits(lg, tt, d.cof_pos_weight, &dl);
                if (d.use_focal) f = focal_factor(lg, tt, d.focal_gamma, d.focal_alpha, &df);
                gv = coef * (dl * f + l * df);
            }
            sG[pl * 4 + a] = gv;
        }
        __syncthreads();
        for (int i = t; i < TP * cpr; i += 256) {
            const int pl = i / cpr;
            const int c0 = (i - pl * cpr) * 8;
            const int pix = pix0 + pl;
            if (pix >= npix) break;
            float g[8];
#pragma unroll
            for (int e2 = 0; e2 < 8; ++e2) g[e2] = 0.f;
            for (int a = 0; a < A; ++a) {                       // objectness channels are a*E+4
                const int ch = a * E + 4;
                if (ch >= c0 && ch < c0 + 8) g[ch - c0] = sG[pl * 4 + a];
            }
            T* dst = gp + (size_t)pix * st.ld + c0;
            if (sizeof(T) == 2) {
                *reinterpret_cast<uint4*>(dst) = pack8(g);
            } else {
                float4* d4 = reinterpret_cast<float4*>(dst);
                d4[0] = make_float4(g[0], g[1], g[2], g[3]);
                d4[1] = make_float4(g[4], g[5], g[6], g[7]);
            }
        }
        __syncthreads();
    }
}

// The member with the largest row index of each cell list sums the gradients of all
// members (ascending row order) and writes the cell's (5+nc)-1 non-objectness channels.
template <typename T>
__global__ __launch_bounds__(256) void v5_pos_bwd_kernel(const LossK p, const Stages sts, const float* __restrict__ gout,
                                                         const int32_t* __restrict__ count, const float* __restrict__ tbox,
                                                         const int32_t* __restrict__ tidx, const int32_t* __restrict__ next,
                                                         const int32_t* __restrict__ head)
{
    const Stage st = sts.s[blockIdx.y];          // one launch for all stages: they are independent of each other
    const yh_v5loss_desc& d = p.d;
    const int s = st.s;
    const int N = count[s];
    const int E = 5 + d.num_class;
    const int t = threadIdx.x, sub = t & 15, grp = t >> 4;
    const T* pred = reinterpret_cast<const T*>(st.pred);
    T* gp = reinterpret_cast<T*>(st.gpred);
    const int32_t* nx = next + (size_t)s * p.L.cap;
    const double s3 = 3.0 / d.num_stage;
    const float go = *gout;
    const float kbox = (float)(-(double)go * d.B * d.iou_scale * s3 / (double)N);
    const float kcls = (float)((double)go * d.B * d.cls_scale * s3 / ((double)N * d.num_class));
    for (int n = blockIdx.x * 16 + grp; n < N; n += gridDim.x * 16) {
        const int32_t* ii = tidx + ((size_t)s * p.L.cap + n) * 5;
        const int img = ii[1], anc = ii[2], gy = ii[3], gx = ii[4];
        if (img < 0 || img >= d.B) continue;
        const int cell = ((img * d.num_anchor + anc) * st.H + gy) * st.W + gx;
        const int h = head[p.L.head_off[s] + cell];
        if (list_max(nx, h) != n) continue;              // not this cell's writer
        const size_t roff = (((size_t)img * st.H + gy) * st.W + gx) * st.ld + anc * E;
        const T* row = pred + roff;
        float lg[4];
        {
            float mine = (sub < 4) ? ldp<T>(row + sub) : 0.f;
#pragma unroll
            for (int i = 0; i < 4; ++i) lg[i] = __shfl(mine, i, 16);
        }
        // per-lane accumulators for channels e = sub, sub+16, ...
        float gacc[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) gacc[i] = 0.f;
        // visit members in ascending row index: repeatedly pick the smallest index > last
        int last = -1;
        for (int guard = 0; guard < (1 << 16); ++guard) {
            int m = 0x7fffffff;
            for (int q = h; q >= 0; q = nx[q]) if (q > last && q < m) m = q;
            if (m == 0x7fffffff) break;
            last = m;
            const int32_t* mi = tidx + ((size_t)s * p.L.cap + m) * 5;
            const float* tb = tbox + ((size_t)s * p.L.cap + m) * 4;
            const int cls = mi[0];
            float dl4[4];
            pos_box(d, s, anc, lg, tb, dl4);
            int slot = 0;
            for (int e = sub; e < E; e += 16, ++slot) {
                float gv = 0.f;
                if (e < 4) gv = kbox * dl4[e];
                else if (e >= 5 && d.num_class > 1) {
                    const float x = ldp<T>(row + e);
                    const float tt = (e - 5 == cls) ? d.cls_smooth : 0.f;
                    float dl, df = 0.f, f = 1.f;
                    const float l = bce_logits(x, tt, d.cls_pos_weight, &dl);
                    if (d.use_focal) f = focal_factor(x, tt, d.focal_gamma, d.focal_alpha, &df);
                    gv = kcls * (dl * f + l * df);
                }
                if (slot < 8) gacc[slot] += gv;
            }
        }
        int slot = 0;
        for (int e = sub; e < E; e += 16, ++slot)
            if (e != 4 && slot < 8) stp<T>(gp + roff + e, gacc[slot]);
    }
}

int check_desc(const yh_v5loss_desc* d, const char* who) {
    YH_CHECK_ARG(d != nullptr, "%s: null desc", who);
    YH_CHECK_ARG(d->B > 0 && d->maxbox > 0 && d->num_class >= 1 && d->num_anchor >= 1 && d->num_anchor <= 3, "%s: bad B/maxbox/classes/anchors", who);
    YH_CHECK_ARG(d->num_stage >= 1 && d->num_stage <= MAXS, "%s: bad num_stage", who);
    YH_CHECK_ARG(5 + d->num_class <= 128, "%s: at most 123 classes supported", who);
    for (int s = 0; s < d->num_stage; ++s) {
        YH_CHECK_ARG(d->H[s] > 0 && d->W[s] > 0, "%s: bad stage dims", who);
        YH_CHECK_ARG(d->ldp[s] % 8 == 0 && d->ldp[s] >= d->num_anchor * (5 + d->num_class), "%s: ldp[%d]=%d must be a multiple of 8 and >= A*(5+nc)", who, s, d->ldp[s]);
        YH_CHECK_ARG((long)d->B * d->num_anchor * d->H[s] * d->W[s] < (1L << 31), "%s: too many cells", who);
    }
    YH_CHECK_ARG((long)5 * d->num_anchor * d->B * d->maxbox < (1L << 28), "%s: too many targets", who);
    return YH_OK;
}

}  // namespace

extern "C" size_t yh_v5loss_ws_bytes(const yh_v5loss_desc* d) { (void)d; return ws_bytes(); }
extern "C" size_t yh_v5loss_saved_bytes(const yh_v5loss_desc* d) {
    if (!d) return 0;
    return make_layout(*d).total;
}

extern "C" int yh_v5_assign(const yh_v5loss_desc* d, const float* targets, int32_t* count, float* tbox,
                            int32_t* tidx, void* ws, yh_stream stream)
{
    (void)ws;
    int rc = check_desc(d, "yh_v5_assign");
    if (rc) return rc;
    YH_CHECK_ARG(targets && count && tbox && tidx, "yh_v5_assign: null pointer");
    LossK k; k.d = *d; k.L = make_layout(*d);
    hipLaunchKernelGGL(v5_assign_kernel, dim3(d->num_stage), dim3(1024), 0, (hipStream_t)stream, k, targets, count, tbox, tidx);
    YH_CHECK_LAUNCH("yh_v5_assign");
    return YH_OK;
}

/* the assignment step of yh_v5_loss_fwd (it depends on the targets only), into the `saved` state of the call */
static int v5_loss_assign(const yh_v5loss_desc* d, const float* targets, void* saved, yh_stream stream)
{
    int rc;
    LossK k; k.d = *d; k.L = make_layout(*d);
    char* sv = (char*)saved;
    int32_t* head = (int32_t*)(sv + k.L.head);
    rc = yh_fill_u32(head, 0xffffffffu, (int64_t)((k.L.total - k.L.head) / 4), stream);
    if (rc) return rc;
    hipLaunchKernelGGL(v5_assign_kernel, dim3(d->num_stage), dim3(1024), 0, (hipStream_t)stream, k, targets,
                       (int32_t*)(sv + k.L.count), (float*)(sv + k.L.tbox), (int32_t*)(sv + k.L.tidx));
    YH_CHECK_LAUNCH("yh_v5_loss_fwd(assign)");
    return YH_OK;
}

extern "C" int yh_v5_loss_fwd(const yh_v5loss_desc* d, const void* const* preds, const float* targets,
                              double* balances, float* result, void* saved, void* ws, yh_stream stream)
{
    int rc = check_desc(d, "yh_v5_loss_fwd");
    if (rc) return rc;
    YH_CHECK_ARG(preds && targets && balances && result && saved && ws, "yh_v5_loss_fwd: null pointer");
    for (int s = 0; s < d->num_stage; ++s) YH_CHECK_ARG(preds[s] && yh_aligned16(preds[s]), "yh_v5_loss_fwd: preds[%d] null/unaligned", s);
    hipStream_t st = (hipStream_t)stream;
    LossK k; k.d = *d; k.L = make_layout(*d);
    char* sv = (char*)saved;
    int32_t* count = (int32_t*)(sv + k.L.count);
    double* bal_used = (double*)(sv + k.L.bal_used);
    float* tbox = (float*)(sv + k.L.tbox);
    int32_t* tidx = (int32_t*)(sv + k.L.tidx);
    float* ciou = (float*)(sv + k.L.ciou);
    int32_t* next = (int32_t*)(sv + k.L.next);
    int32_t* head = (int32_t*)(sv + k.L.head);
    double* part = (double*)ws;
    rc = v5_loss_assign(d, targets, saved, stream);
    if (rc) return rc;
    int nb_pos = (k.L.cap + 15) / 16;
    if (nb_pos > PART_BLOCKS) nb_pos = PART_BLOCKS;
    int nb_obj = PART_BLOCKS;
    Stages all;
    for (int s = 0; s < 4; ++s) {
        const int q = s < d->num_stage ? s : 0;
        Stage& sg = all.s[s];
        sg.pred = preds[q]; sg.gpred = nullptr; sg.s = q; sg.H = d->H[q]; sg.W = d->W[q]; sg.ld = d->ldp[q];
    }
    // positives of every stage, then the objectness pass of every stage (it reads the stage's CIoU values): two launches
    if (d->pred_is_f32) {
        hipLaunchKernelGGL((v5_pos_fwd_kernel<float>), dim3(nb_pos, d->num_stage), dim3(256), 0, st, k, all, count, tbox, tidx, ciou, next, head, part);
        hipLaunchKernelGGL((v5_obj_fwd_kernel<float>), dim3(nb_obj, d->num_stage), dim3(256), 0, st, k, all, ciou, next, head, part);
    } else {
        hipLaunchKernelGGL((v5_pos_fwd_kernel<uint16_t>), dim3(nb_pos, d->num_stage), dim3(256), 0, st, k, all, count, tbox, tidx, ciou, next, head, part);
        hipLaunchKernelGGL((v5_obj_fwd_kernel<uint16_t>), dim3(nb_obj, d->num_stage), dim3(256), 0, st, k, all, ciou, next, head, part);
    }
    hipLaunchKernelGGL(v5_finalize_kernel, dim3(1), dim3(1024), 0, st, k, count, part, nb_pos, nb_obj, balances, bal_used, result);
    YH_CHECK_LAUNCH("yh_v5_loss_fwd");
    return YH_OK;
}

extern "C" int yh_v5_loss_bwd(const yh_v5loss_desc* d, const void* const* preds, const float* gout,
                              const void* saved, void* const* gpreds, void* ws, yh_stream stream)
{
    (void)ws;
    int rc = check_desc(d, "yh_v5_loss_bwd");
    if (rc) return rc;
    YH_CHECK_ARG(preds && gout && saved && gpreds, "yh_v5_loss_bwd: null pointer");
    hipStream_t st = (hipStream_t)stream;
    LossK k; k.d = *d; k.L = make_layout(*d);
    const char* sv = (const char*)saved;
    const int32_t* count = (const int32_t*)(sv + k.L.count);
    const double* bal_used = (const double*)(sv + k.L.bal_used);
    const float* tbox = (const float*)(sv + k.L.tbox);
    const int32_t* tidx = (const int32_t*)(sv + k.L.tidx);
    const float* ciou = (const float*)(sv + k.L.ciou);
    const int32_t* next = (const int32_t*)(sv + k.L.next);
    const int32_t* head = (const int32_t*)(sv + k.L.head);
    int nb_pos = (k.L.cap + 15) / 16;
    if (nb_pos > PART_BLOCKS) nb_pos = PART_BLOCKS;
    Stages all;
    long ntile_max = 1;
    for (int s = 0; s < d->num_stage; ++s) YH_CHECK_ARG(preds[s] && gpreds[s] && yh_aligned16(gpreds[s]), "yh_v5_loss_bwd: stage %d pointers null/unaligned", s);
    for (int s = 0; s < 4; ++s) {
        const int q = s < d->num_stage ? s : 0;
        Stage& sg = all.s[s];
        sg.pred = preds[q]; sg.gpred = gpreds[q]; sg.s = q; sg.H = d->H[q]; sg.W = d->W[q]; sg.ld = d->ldp[q];
        const long ntile = ((long)d->B * sg.H * sg.W + 63) / 64;
        if (s < d->num_stage && ntile > ntile_max) ntile_max = ntile;
    }
    const int gb = (int)(ntile_max > 4096 ? 4096 : ntile_max);
    // the objectness pass writes every stage's whole gradient tensor, then the positives fill their channels: two launches
    if (d->pred_is_f32) {
        hipLaunchKernelGGL((v5_obj_bwd_kernel<float>), dim3(gb, d->num_stage), dim3(256), 0, st, k, all, gout, bal_used, ciou, next, head);
        hipLaunchKernelGGL((v5_pos_bwd_kernel<float>), dim3(nb_pos, d->num_stage), dim3(256), 0, st, k, all, gout, count, tbox, tidx, next, head);
    } else {
        hipLaunchKernelGGL((v5_obj_bwd_kernel<uint16_t>), dim3(gb, d->num_stage), dim3(256), 0, st, k, all, gout, bal_used, ciou, next, head);
        hipLaunchKernelGGL((v5_pos_bwd_kernel<uint16_t>), dim3(nb_pos, d->num_stage), dim3(256), 0, st, k, all, gout, count, tbox, tidx, next, head);
    }
    YH_CHECK_LAUNCH("yh_v5_loss_bwd");
    return YH_OK;
}

// ---------------------------------------------------------------- box utilities
namespace {
__global__ void iou_matrix_kernel(const float* __restrict__ b1, int n1, const float* __restrict__ b2, int n2,
                                  float eps_clamp, float* __restrict__ out)
{
    long tot = (long)n1 * n2;
    for (long id = (long)blockIdx.x * blockDim.x + threadIdx.x; id < tot; id += (long)gridDim.x * blockDim.x) {
        int i = (int)(id / n2), j = (int)(id - (long)i * n2);
        const float* a = b1 + (size_t)i * 4;
        const float* b = b2 + (size_t)j * 4;
        const float a1 = (a[2] - a[0]) * (a[3] - a[1]);
        const float a2 = (b[2] - b[0]) * (b[3] - b[1]);
        float w = fminf(a[2], b[2]) - fmaxf(a[0], b[0]);
        float h = fminf(a[3], b[3]) - fmaxf(a[1], b[1]);
        if (eps_clamp >= 0.f) { w = fmaxf(w, 0.f); h = fmaxf(h, 0.f); }      // < 0: the evaluators' bbox_iou (trainer/eval_yolov5.py:237-258) clamps nothing
        const float inter = w * h;
        float den = a1 + a2 - inter;
        if (eps_clamp > 0.f) den = fmaxf(den, eps_clamp);      // gpu_iou; 0 -> numba_iou (no clamp, 0/0 = NaN)
        out[id] = inter / den;
    }
}

__global__ void iou_pairwise_kernel(int kind, const float* __restrict__ b1, const float* __restrict__ b2, int n,
                                    float* __restrict__ out, float* __restrict__ grad)
{
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float* a = b1 + (size_t)i * 4;
    const float* b = b2 + (size_t)i * 4;
    if (kind == 2) {
        float g[4];
        out[i] = ciou_fwd_bwd(a, b, grad ? g : nullptr);
        if (grad) { grad[i * 4 + 0] = g[0]; grad[i * 4 + 1] = g[1]; grad[i * 4 + 2] = g[2]; grad[i * 4 + 3] = g[3]; }
        return;
    }
    const float eps = 1e-6f;
    const float a1 = (a[2] - a[0]) * (a[3] - a[1]);
    const float a2 = (b[2] - b[0]) * (b[3] - b[1]);
    const float w = fmaxf(fminf(a[2], b[2]) - fmaxf(a[0], b[0]), 0.f);
    const float h = fmaxf(fminf(a[3], b[3]) - fmaxf(a[1], b[1]), 0.f);
    const float inter = w * h;
    const float uni = a1 + a2 - inter;
    const float iou = inter / fmaxf(uni, eps);
    const float cw = fmaxf(a[2], b[2]) - fminf(a[0], b[0]);
    const float chh = fmaxf(a[3], b[3]) - fminf(a[1], b[1]);
    if (kind == 0) {            // GIoU utils/bbox_tools.py:193-230
        const float ca = cw * chh;
        out[i] = iou - fabsf(ca - uni) / fabsf(fmaxf(ca, eps));
    } else {                    // DIoU :233-283
        const float cd = cw * cw + chh * chh;
        const float dx = (a[2] + a[0]) / 2.f - (b[2] + b[0]) / 2.f;
        const float dy = (a[3] + a[1]) / 2.f - (b[3] + b[1]) / 2.f;
        float v = iou - (dx * dx + dy * dy) / fmaxf(cd, eps);
        out[i] = fminf(fmaxf(v, -1.f), 1.f);
    }
}
}  // namespace

extern "C" int yh_iou_matrix(const float* b1, int n1, const float* b2, int n2, float eps_clamp, float* out, yh_stream stream)
{
    YH_CHECK_ARG(b1 && b2 && out && n1 >= 0 && n2 >= 0, "yh_iou_matrix: bad args");
    long tot = (long)n1 * n2;
    if (tot == 0) return YH_OK;
    int g = (int)((tot + 255) / 256 > 4096 ? 4096 : (tot + 255) / 256);
    hipLaunchKernelGGL(iou_matrix_kernel, dim3(g), dim3(256), 0, (hipStream_t)stream, b1, n1, b2, n2, eps_clamp, out);
    YH_CHECK_LAUNCH("yh_iou_matrix");
    return YH_OK;
}

extern "C" int yh_iou_pairwise(int kind, const float* b1, const float* b2, int n, float* out, float* grad_b1, yh_stream stream)
{
    YH_CHECK_ARG(b1 && b2 && out && n >= 0 && kind >= 0 && kind <= 2, "yh_iou_pairwise: bad args");
    YH_CHECK_ARG(grad_b1 == nullptr || kind == 2, "yh_iou_pairwise: gradient only for CIoU");
    if (n == 0) return YH_OK;
    hipLaunchKernelGGL(iou_pairwise_kernel, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream, kind, b1, b2, n, out, grad_b1);
    YH_CHECK_LAUNCH("yh_iou_pairwise");
    return YH_OK;
}
