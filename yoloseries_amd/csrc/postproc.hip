// Inference post-processing on gfx950 (trainer/eval_yolov5.py:182-317, trainer/eval_yolox.py:123-259,
// utils/nms.py:10-65 of the reference):
//   decode_full   — sigmoid / grid / anchor decode to the (B, N, 5+nc) tensor do_inference returns.
//   decode_filter — fused decode + confidence filter + best-class selection; candidates are
//                   compacted per image IN PREDICTION ORDER (wave ballots + block scan), so the
//                   whole decoded tensor never leaves the GPU (the reference copies it to the host).
//   nms_batched   — one workgroup per image: greedy arg-max NMS exactly as numba_nms does it
//                   (first maximum on ties, class offset added in fp32 before the IoU, inclusive
//                   threshold, NaN IoU never suppresses), early exit after max_keep picks, then
//                   the "merge" filter of eval_yolov5.py:306-315.  Wave-level arg-max via shuffles.
// All comparisons are in fp32 with the reference's operation order (compiled with -ffp-contract=off).
#include "common.h"

namespace {

constexpr int MAXS = 4;

template <typename T> __device__ __forceinline__ float ldv(const T* p);
template <> __device__ __forceinline__ float ldv<float>(const float* p) { return *p; }
template <> __device__ __forceinline__ float ldv<uint16_t>(const uint16_t* p) { return bf2f(*p); }

__device__ __forceinline__ float sigm(float x) { return 1.0f / (1.0f + expf(-x)); }

struct DecK {
    yh_decode_desc d;
    const void* pred[MAXS];
    int start[MAXS + 1];      // first prediction index of each stage (per image)
    int ntot;
    int cstart[MAXS + 1];     // two-pass filter: first 64-pixel chunk of each stage; chunk keys in prediction order are
    int kstart[MAXS + 1];     //   kstart[s] + a * nchunk(s) + c  (stage, anchor, chunk)
};

// prediction index (within an image) -> stage, anchor, y, x
__device__ __forceinline__ void locate(const DecK& k, int pi, int& s, int& a, int& y, int& x) {
    s = 0;
#pragma unroll
    for (int i = 1; i < MAXS; ++i) if (i < k.d.num_stage && pi >= k.start[i]) s = i;
    int r = pi - k.start[s];
    const int hw = k.d.H[s] * k.d.W[s];
    a = r / hw; r -= a * hw;
    y = r / k.d.W[s];
    x = r - y * k.d.W[s];
}

template <typename T>
__device__ __forceinline__ const T* cell_ptr(const DecK& k, int b, int s, int a, int y, int x) {
    const T* base = reinterpret_cast<const T*>(k.pred[s]);
    return base + (((size_t)b * k.d.H[s] + y) * k.d.W[s] + x) * k.d.ldp[s] + a * (5 + k.d.num_class);
}

// box decode: returns center-format box (cx, cy, w, h) in pixels
template <typename T>
__device__ __forceinline__ void decode_box(const DecK& k, const T* row, int s, int a, int y, int x, float* box) {
    const float st = k.d.stride[s];
    if (k.d.yolox) {          // eval_yolox.py:140-143
        box[0] = (ldv<T>(row + 0) + (float)x) * st;
        box[1] = (ldv<T>(row + 1) + (float)y) * st;
        box[2] = expf(ldv<T>(row + 2)) * st;
        box[3] = expf(ldv<T>(row + 3)) * st;
    } else {                  // eval_yolov5.py:203-205
        const float aw = k.d.anchors[s][a][0] / st, ah = k.d.anchors[s][a][1] / st;
        box[0] = (sigm(ldv<T>(row + 0)) * 2.f - 0.5f + (float)x) * st;
        box[1] = (sigm(ldv<T>(row + 1)) * 2.f - 0.5f + (float)y) * st;
        const float tw = sigm(ldv<T>(row + 2)) * 2.f, th = sigm(ldv<T>(row + 3)) * 2.f;
        box[2] = tw * tw * aw * st;
        box[3] = th * th * ah * st;
    }
}

template <typename T>
__global__ void decode_full_kernel(const DecK k, float* __restrict__ out)
{
    const int E = 5 + k.d.num_class;
    const long tot = (long)k.d.B * k.ntot * E;
    for (long id = (long)blockIdx.x * blockDim.x + threadIdx.x; id < tot; id += (long)gridDim.x * blockDim.x) {
        const int e = (int)(id % E);
        const long r = id / E;
        const int pi = (int)(r % k.ntot);
        const int b = (int)(r / k.ntot);
        int s, a, y, x;
        locate(k, pi, s, a, y, x);
        const T* row = cell_ptr<T>(k, b, s, a, y, x);
        float v;
        if (e < 4) {
            float box[4];
            decode_box<T>(k, row, s, a, y, x, box);
            v = box[e];
        } else {
            v = sigm(ldv<T>(row + e));
        }
        out[id] = v;
    }
}

// confidence filter + best class + box of ONE prediction row (eval_yolov5.py:266-285, eval_yolox.py:206-227); true = candidate
template <typename T>
__device__ __forceinline__ bool eval_pred(const DecK& k, const T* row, int s, int a, int y, int x, float conf_thr, float cls_thr,
                                          float* box, float& conf, int& cls)
{
    const int nc = k.d.num_class;
    const float obj = sigm(ldv<T>(row + 4));
    bool pass = k.d.yolox ? true : (obj >= conf_thr);                 // eval_yolov5.py:266
    if (!pass) return false;
    float best = -INFINITY, best_raw = -INFINITY;
    for (int c = 0; c < nc; ++c) {
        const float pc = sigm(ldv<T>(row + 5 + c));
        const float sc = pc * obj;                                // x[:, 5:] *= x[:, 4:5]
        if (sc > best) { best = sc; cls = c; }                    // first maximum
        if (pc > best_raw) best_raw = pc;
    }
    if (k.d.yolox) pass = (obj * best_raw) >= conf_thr && best >= cls_thr;   // eval_yolox.py:206-207,227
    else           pass = best > cls_thr;                                      // eval_yolov5.py:285
    if (!pass) return false;
    float cb[4];
    decode_box<T>(k, row, s, a, y, x, cb);
    box[0] = cb[0] - cb[2] / 2.f; box[1] = cb[1] - cb[3] / 2.f;           // numba_xywh2xyxy
    box[2] = cb[0] + cb[2] / 2.f; box[3] = cb[1] + cb[3] / 2.f;
    conf = best;
    return true;
}

// One workgroup (1024 threads) per image walks the predictions in order (kept for heads too small to fill the chip otherwise
// and as the cross-check of the two-pass path in the tests: yh_decode_filter with ws == NULL).
template <typename T>
__global__ __launch_bounds__(1024) void decode_filter_kernel(const DecK k, float conf_thr, float cls_thr,
                                                             float* __restrict__ cand, int32_t* __restrict__ ncand, int cap)
{
    __shared__ int wave_cnt[16];
    __shared__ int base_s;
    const int b = blockIdx.x;
    const int t = threadIdx.x, lane = t & 63, wv = t >> 6;
    if (t == 0) base_s = 0;
    __syncthreads();
    float* out = cand + (size_t)b * cap * 6;
    for (int p0 = 0; p0 < k.ntot; p0 += 1024) {
        const int pi = p0 + t;
        bool flag = false;
        float box[4] = {0, 0, 0, 0}, conf = 0.f;
        int cls = 0;
        if (pi < k.ntot) {
            int s, a, y, x;
            locate(k, pi, s, a, y, x);
            const T* row = cell_ptr<T>(k, b, s, a, y, x);
            flag = eval_pred<T>(k, row, s, a, y, x, conf_thr, cls_thr, box, conf, cls);
        }
        const unsigned long long bal = __ballot(flag);
        const int within = __popcll(bal & ((1ull << lane) - 1ull));
        if (lane == 0) wave_cnt[wv] = __popcll(bal);
        __syncthreads();
        int before = 0, total = 0;
#pragma unroll
        for (int w = 0; w < 16; ++w) { int c = wave_cnt[w]; if (w < wv) before += c; total += c; }
        const int base = base_s;
        if (flag) {
            const int pos = base + before + within;
            if (pos < cap) {
                float* o = out + (size_t)pos * 6;
                o[0] = box[0]; o[1] = box[1]; o[2] = box[2]; o[3] = box[3]; o[4] = conf; o[5] = (float)cls;
            }
        }
        __syncthreads();
        if (t == 0) base_s = base + total;
        __syncthreads();
    }
    if (t == 0) ncand[b] = base_s;
}



// Two-pass form of the same filter for heads that matter (17.1 MB per 1280^2 image): pass 1 spreads the image over
// blocks — a block owns 64 consecutive pixels of one stage, copies their prediction rows to LDS with coalesced 16-byte
// loads and evaluates one (pixel, anchor) per thread, wave a = anchor a, so a wave ballot orders the candidates of the key
// (stage, anchor, chunk) — and leaves them in a staging area [B][nkeys][64][6] with their counts; pass 2 (one block per
// image) turns the counts into offsets in prediction order and moves the candidates to their final rows.  Same arithmetic,
// same order as decode_filter_kernel.
constexpr int DPIX = 64;
template <typename T>
__global__ __launch_bounds__(256) void decode_scan_kernel(const DecK k, float conf_thr, float cls_thr,
                                                          float* __restrict__ stage, int32_t* __restrict__ counts, int nkeys)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char dsm[];
    T* tile = reinterpret_cast<T*>(dsm);
    const int b = blockIdx.y;
    int s = 0;
#pragma unroll
    for (int i = 1; i < MAXS; ++i) if (i < k.d.num_stage && (int)blockIdx.x >= k.cstart[i]) s = i;
    const int c = blockIdx.x - k.cstart[s];
    const int hw = k.d.H[s] * k.d.W[s];
    const int nchunk = (hw + DPIX - 1) / DPIX;
    const int p0 = c * DPIX;
    const int npix = min(DPIX, hw - p0);
    const int ldp = k.d.ldp[s];
    // LDS rows are 16 bytes longer than the prediction rows: with the rows back to back (256 floats / 256 bf16 = a multiple of the
    // 64 banks) the 64 lanes of a wave — one pixel each — read the same bank for every element: a 32-way conflict on each of the
    // 85 reads per lane, which is what the pass took most of its time for (2.7 of 8 TB/s, round 5)
    constexpr int LPAD = 16 / (int)sizeof(T);
    const int lds_ld = ldp + LPAD;
    const T* src = reinterpret_cast<const T*>(k.pred[s]) + ((size_t)b * hw + p0) * ldp;
    const int nelem = npix * ldp;
    const int t = threadIdx.x;
    if (((size_t)src & 15) == 0 && (ldp * (int)sizeof(T)) % 16 == 0) {
        const int cpr = ldp * (int)sizeof(T) / 16;            // 16-byte chunks per row
        const int nv = npix * cpr;
        // every load of the tile is requested before the first LDS store (a load -> store loop kept one or two in flight per thread:
        // with two 65 KB blocks per CU the pass was bound by the round trips, 2.7 TB/s)
        constexpr int NLD = 16;
        for (int i0 = 0; i0 < nv; i0 += 256 * NLD) {
            uint4 v[NLD];
#pragma unroll
            for (int u = 0; u < NLD; ++u) {
                const int i = i0 + u * 256 + t;
                v[u] = make_uint4(0, 0, 0, 0);
                if (i < nv) v[u] = reinterpret_cast<const uint4*>(src)[i];
            }
#pragma unroll
            for (int u = 0; u < NLD; ++u) {
                const int i = i0 + u * 256 + t;
                if (i < nv) {
                    const int r = i / cpr, c16 = i - r * cpr;
                    *reinterpret_cast<uint4*>(tile + r * lds_ld + c16 * LPAD) = v[u];
                }
            }
        }
    } else {
        for (int i = t; i < nelem; i += 256) { const int r = i / ldp; tile[r * lds_ld + (i - r * ldp)] = src[i]; }
    }
    __syncthreads();
    const int a = t >> 6, lane = t & 63;
    if (a >= k.d.num_anchor) return;
    bool flag = false;
    float box[4] = {0, 0, 0, 0}, conf = 0.f;
    int cls = 0;
    if (lane < npix) {
        const int pidx = p0 + lane;
        const int y = pidx / k.d.W[s], x = pidx - y * k.d.W[s];
        const T* row = tile + lane * lds_ld + a * (5 + k.d.num_class);
        flag = eval_pred<T>(k, row, s, a, y, x, conf_thr, cls_thr, box, conf, cls);
    }
    const unsigned long long bal = __ballot(flag);
    const int key = k.kstart[s] + a * nchunk + c;
    if (lane == 0) counts[(size_t)b * nkeys + key] = __popcll(bal);
    if (flag) {
        float* o = stage + (((size_t)b * nkeys + key) * DPIX + __popcll(bal & ((1ull << lane) - 1ull))) * 6;
        o[0] = box[0]; o[1] = box[1]; o[2] = box[2]; o[3] = box[3]; o[4] = conf; o[5] = (float)cls;
    }
}

__global__ __launch_bounds__(1024) void decode_gather_kernel(const float* __restrict__ stage, const int32_t* __restrict__ counts, int nkeys,
                                                              float* __restrict__ cand, int32_t* __restrict__ ncand, int cap)
{
    __shared__ int wsum[16];
    __shared__ int base_s;
    const int b = blockIdx.x;
    const int t = threadIdx.x, lane = t & 63, wv = t >> 6;
    if (t == 0) base_s = 0;
    __syncthreads();
    float* out = cand + (size_t)b * cap * 6;
    for (int k0 = 0; k0 < nkeys; k0 += 1024) {
        const int key = k0 + t;
        const int cnt = key < nkeys ? counts[(size_t)b * nkeys + key] : 0;
        int incl = cnt;                                  // inclusive scan inside the wave
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) { const int v = __shfl_up(incl, o, 64); if (lane >= o) incl += v; }
        if (lane == 63) wsum[wv] = incl;
        __syncthreads();
        int before = 0, total = 0;
#pragma unroll
        for (int w = 0; w < 16; ++w) { const int v = wsum[w]; if (w < wv) before += v; total += v; }
        const int base = base_s;
        int pos = base + before + incl - cnt;
        const float* src = stage + ((size_t)b * nkeys + key) * DPIX * 6;
        for (int i = 0; i < cnt && pos < cap; ++i, ++pos) {
#pragma unroll
            for (int e = 0; e < 6; ++e) out[(size_t)pos * 6 + e] = src[i * 6 + e];
        }
        __syncthreads();
        if (t == 0) base_s = base + total;
        __syncthreads();
    }
    if (t == 0) ncand[b] = base_s;
}

// Multi-label candidate filter (hyp['mutil_label'], eval_yolov5.py:276-279 / eval_yolox.py:218-221): every (prediction, class) with
// cls*obj >= cls_thr among the pre-filtered predictions — YOLOv5: obj >= conf_thr; YOLOX (yolox != 0): obj * max(cls) >= conf_thr —
// is a candidate of its own; rows in (prediction, class) order like np.nonzero.
// A thread counts its prediction's classes, a wave / block prefix sum places them.
__global__ __launch_bounds__(1024) void filter_decoded_multi_kernel(const float* __restrict__ dec, int N, int nc, float conf_thr,
                                                                    float cls_thr, int yolox, float* __restrict__ cand, int32_t* __restrict__ ncand, int cap)
{
    __shared__ int wave_cnt[16];
    __shared__ int base_s;
    const int b = blockIdx.x;
    const int t = threadIdx.x, lane = t & 63, wv = t >> 6;
    const int E = 5 + nc;
    if (t == 0) base_s = 0;
    __syncthreads();
    float* out = cand + (size_t)b * cap * 6;
    for (int p0 = 0; p0 < N; p0 += 1024) {
        const int pi = p0 + t;
        const float* row = dec + ((size_t)b * N + (pi < N ? pi : 0)) * E;
        int cnt = 0;
        float obj = 0.f;
        if (pi < N) {
            obj = row[4];
            bool pre = obj >= conf_thr;
            if (yolox) {                                  // np.max over the classes, then one fp32 product (eval_yolox.py:206-207)
                float mx = row[5];
                for (int c = 1; c < nc; ++c) mx = fmaxf(mx, row[5 + c]);
                pre = obj * mx >= conf_thr;
            }
            if (pre)
                for (int c = 0; c < nc; ++c) cnt += (row[5 + c] * obj >= cls_thr) ? 1 : 0;
        }
        int incl = cnt;                                   // inclusive prefix sum over the wave
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const int v = __shfl_up(incl, o, 64);
            if (lane >= o) incl += v;
        }
        if (lane == 63) wave_cnt[wv] = incl;
        __syncthreads();
        int before = 0, total = 0;
#pragma unroll
        for (int w = 0; w < 16; ++w) { int c = wave_cnt[w]; if (w < wv) before += c; total += c; }
        int pos = base_s + before + incl - cnt;
        if (cnt > 0) {
            const float x0 = row[0] - row[2] / 2.f, y0 = row[1] - row[3] / 2.f, x1 = row[0] + row[2] / 2.f, y1 = row[1] + row[3] / 2.f;
            for (int c = 0; c < nc; ++c) {
                const float sc = row[5 + c] * obj;
                if (sc >= cls_thr) {
                    if (pos < cap) {
                        float* o = out + (size_t)pos * 6;
                        o[0] = x0; o[1] = y0; o[2] = x1; o[3] = y1; o[4] = sc; o[5] = (float)c;
                    }
                    ++pos;
                }
            }
        }
        __syncthreads();
        if (t == 0) base_s += total;
        __syncthreads();
    }
    if (t == 0) ncand[b] = base_s;
}

// Candidate filter on an already decoded (B, N, 5+nc) fp32 tensor (the argument of
// YOLOV5Evaluator.numba_nms, eval_yolov5.py:261-286), order preserving.
__global__ __launch_bounds__(1024) void filter_decoded_kernel(const float* __restrict__ dec, int N, int nc, float conf_thr,
                                                              float cls_thr, int yolox, float* __restrict__ cand,
                                                              int32_t* __restrict__ ncand, int cap)
{
    __shared__ int wave_cnt[16];
    __shared__ int base_s;
    const int b = blockIdx.x;
    const int t = threadIdx.x, lane = t & 63, wv = t >> 6;
    const int E = 5 + nc;
    if (t == 0) base_s = 0;
    __syncthreads();
    float* out = cand + (size_t)b * cap * 6;
    for (int p0 = 0; p0 < N; p0 += 1024) {
        const int pi = p0 + t;
        bool flag = false;
        float box[4] = {0, 0, 0, 0}, conf = 0.f;
        int cls = 0;
        if (pi < N) {
            const float* row = dec + ((size_t)b * N + pi) * E;
            const float obj = row[4];
            bool pass = yolox ? true : (obj >= conf_thr);
            if (pass) {
                float best = -INFINITY, best_raw = -INFINITY;
                for (int c = 0; c < nc; ++c) {
                    const float pc = row[5 + c];
                    const float sc = pc * obj;
                    if (sc > best) { best = sc; cls = c; }
                    if (pc > best_raw) best_raw = pc;
                }
                if (yolox) pass = (obj * best_raw) >= conf_thr && best >= cls_thr;
                else       pass = best > cls_thr;
                if (pass) {
                    box[0] = row[0] - row[2] / 2.f; box[1] = row[1] - row[3] / 2.f;
                    box[2] = row[0] + row[2] / 2.f; box[3] = row[1] + row[3] / 2.f;
                    conf = best;
                    flag = true;
                }
            }
        }
        const unsigned long long bal = __ballot(flag);
        const int within = __popcll(bal & ((1ull << lane) - 1ull));
        if (lane == 0) wave_cnt[wv] = __popcll(bal);
        __syncthreads();
        int before = 0, total = 0;
#pragma unroll
        for (int w = 0; w < 16; ++w) { int c = wave_cnt[w]; if (w < wv) before += c; total += c; }
        const int base = base_s;
        if (flag) {
            const int pos = base + before + within;
            if (pos < cap) {
                float* o = out + (size_t)pos * 6;
                o[0] = box[0]; o[1] = box[1]; o[2] = box[2]; o[3] = box[3]; o[4] = conf; o[5] = (float)cls;
            }
        }
        __syncthreads();
        if (t == 0) base_s = base + total;
        __syncthreads();
    }
    if (t == 0) ncand[b] = base_s;
}

// ---------------------------------------------------------------- NMS
// numba_iou of one pair (utils/bbox_tools.py:12-35): no eps, 0/0 -> NaN
__device__ __forceinline__ float pair_iou(const float4 a, const float4 b, float union_clamp) {
    const float a1 = (a.z - a.x) * (a.w - a.y);
    const float a2 = (b.z - b.x) * (b.w - b.y);
    const float w = fmaxf(0.f, fminf(a.z, b.z) - fmaxf(a.x, b.x));
    const float h = fmaxf(0.f, fminf(a.w, b.w) - fmaxf(a.y, b.y));
    const float inter = w * h;
    float den = a1 + a2 - inter;
    if (union_clamp > 0.f) den = fmaxf(den, union_clamp);
    return inter / den;
}

// Greedy NMS == "walk the candidates in (score descending, index ascending) order; keep one iff no earlier KEPT box suppresses
// it" (utils/nms.py:10-27 picks arg-max with first index on ties and only ever zeroes scores, so the pick order is exactly
// that order restricted to the kept boxes; non-positive scores are never picked).  One workgroup (16 waves) per image:
//   1. 64-bit keys (score bits << 32 | ~index) are bitonic-sorted once — in LDS up to 8192 candidates, in the L2-resident
//      workspace above — and the class-offset boxes are permuted into that order;
//   2. the sorted list is consumed in chunks of 64: a chunk none of whose members is still alive is skipped without a barrier
//      (every wave reads the same 64 flags); otherwise wave 0 builds, with wave shuffles, the 64 x 64 suppression bit matrix
//      of the chunk (lane j: mask of earlier members that suppress j), resolves it in score order, appends the survivors to
//      the keep list (stopping at max_keep), and all 16 waves strike out the later candidates those survivors suppress.
//   At most max_keep chunks are ever resolved (each resolved chunk keeps >= 1 box), two barriers per resolved chunk instead
//   of three per kept box, and no arg-max scans.  Selection order, tie rule, NaN-IoU and threshold conventions are unchanged.
constexpr int NMS_LDS_KEYS = 8192;

__device__ __forceinline__ void bitonic_sort_desc(unsigned long long* keys, int P2, int t)
{
    for (int k = 2; k <= P2; k <<= 1) {
        for (int j = k >> 1; j > 0; j >>= 1) {
            for (int i = t; i < P2; i += 1024) {
                const int ixj = i ^ j;
                if (ixj > i) {
                    const unsigned long long a = keys[i], c = keys[ixj];
                    const bool desc = (i & k) == 0;
                    if (desc ? (a < c) : (a > c)) { keys[i] = c; keys[ixj] = a; }
                }
            }
            __syncthreads();
        }
    }
}

__global__ __launch_bounds__(1024) void nms_kernel(const float* __restrict__ cand, const int32_t* __restrict__ ncand, int cap, int P2cap,
                                                   float iou_thr, int class_aware, int inclusive, int max_keep, int merge_filter,
                                                   float* __restrict__ out, int32_t* __restrict__ nkeep, int32_t* __restrict__ keep_idx,
                                                   unsigned char* __restrict__ ws)
{
    __shared__ unsigned long long s_keys[NMS_LDS_KEYS];
    __shared__ float4 s_kbox[64];
    __shared__ int s_nk, s_k, s_pick, s_nvalid;
    __shared__ int s_flag[512];
    const int b = blockIdx.x;
    const int t = threadIdx.x, lane = t & 63, wv = t >> 6;
    int n = ncand[b];
    if (n > cap) n = cap;
    const float* cb = cand + (size_t)b * cap * 6;
    // workspace of this image: keys [P2cap] u64 | boxes in candidate order [cap] | boxes in sorted order [cap] | removed flags [cap]
    unsigned char* wb = ws + (size_t)b * ((size_t)P2cap * 8 + (size_t)cap * 36);
    unsigned long long* gkeys = reinterpret_cast<unsigned long long*>(wb);
    float4* bx = reinterpret_cast<float4*>(wb + (size_t)P2cap * 8);
    float4* sbx = bx + cap;
    unsigned char* rem = reinterpret_cast<unsigned char*>(sbx + cap);
    int32_t* kp = keep_idx + (size_t)b * max_keep;
    const float uclamp = inclusive ? 0.f : 1e-9f;          // gpu_nms uses gpu_iou (union clamp 1e-9)

    int P2 = 1024;
    while (P2 < n) P2 <<= 1;
    unsigned long long* keys = P2 <= NMS_LDS_KEYS ? s_keys : gkeys;
    if (t == 0) s_nvalid = 0;
    __syncthreads();
    int myvalid = 0;
    for (int i = t; i < P2; i += 1024) {
        unsigned long long key = 0ull;
        if (i < n) {
            const float* r = cb + (size_t)i * 6;
            const float off = class_aware ? r[5] * 4096.f : r[5] * 0.f;       // eval_yolov5.py:293-298
            bx[i] = make_float4(r[0] + off, r[1] + off, r[2] + off, r[3] + off);
            rem[i] = 0;
            const float sc = r[4];
            if (sc > 0.f) { key = ((unsigned long long)__float_as_uint(sc) << 32) | (unsigned long long)(0xFFFFFFFFu - (unsigned)i); ++myvalid; }
        }
        keys[i] = key;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) myvalid += __shfl_xor(myvalid, o, 64);
    if (lane == 0 && myvalid) atomicAdd(&s_nvalid, myvalid);
    __syncthreads();
    const int nvalid = s_nvalid;                           // candidates with a positive score: the only ones ever picked
    bitonic_sort_desc(keys, P2, t);
    for (int q = t; q < nvalid; q += 1024) sbx[q] = bx[0xFFFFFFFFu - (unsigned)(keys[q] & 0xFFFFFFFFull)];
    __syncthreads();

    int k = 0;
    for (int base = 0; base < nvalid && k < max_keep; base += 64) {
        const int q = base + lane;
        const bool alive = q < nvalid && rem[q] == 0;
        const unsigned long long alive_mask = __ballot(alive);            // identical in every wave: no barrier needed to skip
        if (alive_mask == 0ull) continue;
        if (wv == 0) {
            const float4 mine = alive ? sbx[q] : make_float4(0.f, 0.f, 0.f, 0.f);
            unsigned long long supp_by = 0ull;                            // earlier alive members of the chunk that suppress this one
            for (unsigned long long m = alive_mask; m; m &= m - 1) {
                const int i = __builtin_ctzll(m);
                const float4 bi = make_float4(__shfl(mine.x, i, 64), __shfl(mine.y, i, 64), __shfl(mine.z, i, 64), __shfl(mine.w, i, 64));
                if (alive && lane > i) {
                    const float iou = pair_iou(bi, mine, uclamp);
                    if (inclusive ? (iou >= iou_thr) : (iou > iou_thr)) supp_by |= 1ull << i;
                }
            }
            unsigned long long kept = 0ull;
            int kk = k;
            for (unsigned long long m = alive_mask; m && kk < max_keep; m &= m - 1) {
                const int i = __builtin_ctzll(m);
                const unsigned lo = __shfl((unsigned)supp_by, i, 64), hi = __shfl((unsigned)(supp_by >> 32), i, 64);
                const unsigned long long si = ((unsigned long long)hi << 32) | lo;
                if ((si & kept) == 0ull) { kept |= 1ull << i; ++kk; }
            }
            if ((kept >> lane) & 1ull) {
                const int r = __popcll(kept & ((1ull << lane) - 1ull));
                kp[k + r] = (int)(0xFFFFFFFFu - (unsigned)(keys[q] & 0xFFFFFFFFull));
                s_kbox[r] = mine;
            }
            if (lane == 0) { s_nk = kk - k; s_k = kk; }
        }
        __syncthreads();
        const int nk = s_nk;
        k = s_k;
        if (k < max_keep) {
            for (int q2 = base + 64 + t; q2 < nvalid; q2 += 1024) {
                if (rem[q2]) continue;
                const float4 bq = sbx[q2];
                for (int r = 0; r < nk; ++r) {
                    const float iou = pair_iou(s_kbox[r], bq, uclamp);
                    if (inclusive ? (iou >= iou_thr) : (iou > iou_thr)) { rem[q2] = 1; break; }
                }
            }
        }
        __syncthreads();
    }

    // merge filter: keep i iff more than one candidate overlaps it with iou > thr (eval_yolov5.py:306-315)
    const bool do_merge = merge_filter && n > 1 && n < 3000;
    if (do_merge) {
        for (int r0 = 0; r0 < k; r0 += 16) {
            const int r = r0 + wv;
            int cnt = 0;
            if (r < k) {
                const float4 pb = bx[kp[r]];
                for (int i = lane; i < n; i += 64) cnt += (pair_iou(pb, bx[i], 0.f) > iou_thr) ? 1 : 0;
            }
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) cnt += __shfl_xor(cnt, o, 64);
            if (lane == 0 && r < k) s_flag[r] = cnt > 1;
        }
        __syncthreads();
        if (t == 0) {
            int m = 0;
            for (int r = 0; r < k; ++r) if (s_flag[r]) kp[m++] = kp[r];
            s_pick = m;
        }
        __syncthreads();
        k = s_pick;
    }
    if (t == 0) nkeep[b] = k;
    __syncthreads();
    for (int i = t; i < k * 6; i += 1024) {
        const int r = i / 6, c = i - r * 6;
        out[((size_t)b * max_keep + r) * 6 + c] = cb[(size_t)kp[r] * 6 + c];
    }
}

int fill_deck(const yh_decode_desc* d, const void* const* preds, DecK* k, const char* who) {
    YH_CHECK_ARG(d && preds, "%s: null desc", who);
    YH_CHECK_ARG(d->B > 0 && d->num_class >= 1 && d->num_anchor >= 1 && d->num_anchor <= 3 && d->num_stage >= 1 && d->num_stage <= MAXS, "%s: bad dims", who);
    k->d = *d;
    int acc = 0;
    for (int s = 0; s < MAXS; ++s) {
        k->start[s] = acc;
        k->pred[s] = nullptr;
        if (s < d->num_stage) {
            YH_CHECK_ARG(preds[s] != nullptr && d->H[s] > 0 && d->W[s] > 0 && d->ldp[s] >= d->num_anchor * (5 + d->num_class), "%s: stage %d invalid", who, s);
            k->pred[s] = preds[s];
            acc += d->num_anchor * d->H[s] * d->W[s];
        }
    }
    k->start[MAXS] = acc;
    k->ntot = acc;
    int cacc = 0, kacc = 0;
    for (int s = 0; s <= MAXS; ++s) {
        k->cstart[s] = cacc;
        k->kstart[s] = kacc;
        if (s < d->num_stage) {
            const int nchunk = (d->H[s] * d->W[s] + DPIX - 1) / DPIX;
            cacc += nchunk;
            kacc += nchunk * d->num_anchor;
        }
    }
    return YH_OK;
}

}  // namespace

extern "C" int yh_decode_full(const yh_decode_desc* d, const void* const* preds, float* out, yh_stream stream)
{
    DecK k;
    int rc = fill_deck(d, preds, &k, "yh_decode_full");
    if (rc) return rc;
    YH_CHECK_ARG(out != nullptr, "yh_decode_full: null output");
    long tot = (long)d->B * k.ntot * (5 + d->num_class);
    int g = (int)((tot + 255) / 256 > 8192 ? 8192 : (tot + 255) / 256);
    if (d->pred_is_f32) hipLaunchKernelGGL((decode_full_kernel<float>), dim3(g), dim3(256), 0, (hipStream_t)stream, k, out);
    else                hipLaunchKernelGGL((decode_full_kernel<uint16_t>), dim3(g), dim3(256), 0, (hipStream_t)stream, k, out);
    YH_CHECK_LAUNCH("yh_decode_full");
    return YH_OK;
}

/* staging area [B][nkeys][64][6] fp32 + counts [B][nkeys] int32 of the two-pass filter */
extern "C" size_t yh_decode_filter_ws_bytes(const yh_decode_desc* d)
{
    if (!d || d->B <= 0 || d->num_stage < 1 || d->num_stage > MAXS || d->num_anchor < 1) return 0;
    size_t nkeys = 0;
    for (int s = 0; s < d->num_stage; ++s) nkeys += (size_t)((d->H[s] * d->W[s] + DPIX - 1) / DPIX) * d->num_anchor;
    return (size_t)d->B * nkeys * (DPIX * 6 * 4 + 4);
}

extern "C" int yh_decode_filter(const yh_decode_desc* d, const void* const* preds, float conf_thr, float cls_thr,
                                float* cand, int32_t* ncand, int cap, void* ws, yh_stream stream)
{
    DecK k;
    int rc = fill_deck(d, preds, &k, "yh_decode_filter");
    if (rc) return rc;
    YH_CHECK_ARG(cand && ncand && cap > 0 && cap % 4 == 0, "yh_decode_filter: cand/ncand null or cap not a multiple of 4");
    if (ws) {
        YH_CHECK_ARG(yh_aligned16(ws) && d->num_anchor <= 4, "yh_decode_filter: workspace unaligned");
        const int nkeys = k.kstart[MAXS], nchunks = k.cstart[MAXS];
        float* stage = reinterpret_cast<float*>(ws);
        int32_t* counts = reinterpret_cast<int32_t*>(stage + (size_t)d->B * nkeys * DPIX * 6);
        int ldmax = 0;
        for (int s = 0; s < d->num_stage; ++s) ldmax = d->ldp[s] > ldmax ? d->ldp[s] : ldmax;
        const size_t sm = (size_t)DPIX * ((size_t)ldmax * (d->pred_is_f32 ? 4 : 2) + 16);      // rows padded by 16 bytes (bank conflicts)
        YH_CHECK_ARG(sm <= 160 * 1024, "yh_decode_filter: prediction rows of %d elements do not fit the LDS tile", ldmax);
        const dim3 grid(nchunks, d->B);
        if (d->pred_is_f32) {
            static YhDevOnce attr;
            if (attr.need()) { attr.set((const void*)decode_scan_kernel<float>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); attr.done(); }
            hipLaunchKernelGGL((decode_scan_kernel<float>), grid, dim3(256), sm, (hipStream_t)stream, k, conf_thr, cls_thr, stage, counts, nkeys);
        } else {
            static YhDevOnce attr;
            if (attr.need()) { attr.set((const void*)decode_scan_kernel<uint16_t>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); attr.done(); }
            hipLaunchKernelGGL((decode_scan_kernel<uint16_t>), grid, dim3(256), sm, (hipStream_t)stream, k, conf_thr, cls_thr, stage, counts, nkeys);
        }
        YH_CHECK_LAUNCH("yh_decode_filter(scan)");
        hipLaunchKernelGGL(decode_gather_kernel, dim3(d->B), dim3(1024), 0, (hipStream_t)stream, stage, counts, nkeys, cand, ncand, cap);
        YH_CHECK_LAUNCH("yh_decode_filter(gather)");
        return YH_OK;
    }
    if (d->pred_is_f32) hipLaunchKernelGGL((decode_filter_kernel<float>), dim3(d->B), dim3(1024), 0, (hipStream_t)stream, k, conf_thr, cls_thr, cand, ncand, cap);
    else                hipLaunchKernelGGL((decode_filter_kernel<uint16_t>), dim3(d->B), dim3(1024), 0, (hipStream_t)stream, k, conf_thr, cls_thr, cand, ncand, cap);
    YH_CHECK_LAUNCH("yh_decode_filter");
    return YH_OK;
}

extern "C" int yh_filter_decoded(const float* dec, int B, int N, int num_class, float conf_thr, float cls_thr, int yolox,
                                 float* cand, int32_t* ncand, int cap, yh_stream stream)
{
    YH_CHECK_ARG(dec && cand && ncand && B > 0 && N > 0 && num_class >= 1 && cap > 0 && cap % 4 == 0, "yh_filter_decoded: bad args");
    YH_CHECK_ARG(yolox >= 0 && yolox <= 3, "yh_filter_decoded: mode must be 0 (YOLOv5), 1 (YOLOX), 2 (YOLOv5 multi-label) or 3 (YOLOX multi-label)");
    if (yolox >= 2)
        hipLaunchKernelGGL(filter_decoded_multi_kernel, dim3(B), dim3(1024), 0, (hipStream_t)stream, dec, N, num_class, conf_thr, cls_thr, yolox == 3 ? 1 : 0, cand, ncand, cap);
    else
        hipLaunchKernelGGL(filter_decoded_kernel, dim3(B), dim3(1024), 0, (hipStream_t)stream, dec, N, num_class, conf_thr, cls_thr, yolox, cand, ncand, cap);
    YH_CHECK_LAUNCH("yh_filter_decoded");
    return YH_OK;
}

static int nms_pow2(int cap) { int p = 1024; while (p < cap) p <<= 1; return p; }
/* per image: sort keys [pow2(cap)] u64 | class-offset boxes in candidate order and in sorted order [2][cap] float4 | flags [cap] (+pad) */
extern "C" size_t yh_nms_ws_bytes(int B, int cap) { return (size_t)B * ((size_t)nms_pow2(cap) * 8 + (size_t)cap * 36); }

extern "C" int yh_nms_batched(const float* cand, const int32_t* ncand, int B, int cap,
                              float iou_thr, int class_aware, int thr_inclusive, int max_keep, int merge_filter,
                              float* out, int32_t* nkeep, int32_t* keep_idx, void* ws, yh_stream stream)
{
    YH_CHECK_ARG(cand && ncand && out && nkeep && keep_idx && ws, "yh_nms_batched: null pointer");
    YH_CHECK_ARG(B > 0 && cap > 0 && cap % 4 == 0, "yh_nms_batched: cap must be a positive multiple of 4");
    YH_CHECK_ARG(max_keep > 0 && (max_keep <= 512 || !merge_filter), "yh_nms_batched: max_keep must be in [1,512] when merge_filter is set");
    YH_CHECK_ARG(yh_aligned16(ws), "yh_nms_batched: workspace unaligned");
    hipLaunchKernelGGL(nms_kernel, dim3(B), dim3(1024), 0, (hipStream_t)stream, cand, ncand, cap, nms_pow2(cap), iou_thr, class_aware,
                       thr_inclusive, max_keep, merge_filter, out, nkeep, keep_idx, (unsigned char*)ws);
    YH_CHECK_LAUNCH("yh_nms_batched");
    return YH_OK;
}
