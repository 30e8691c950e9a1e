// Flat parameter-arena kernels: index-map gather between the fp32 master parameters
// (PyTorch OIHW layout, state_dict compatible) and the packed bf16 weight images read
// by conv_igemm.hip, plus the optimizer-side streaming kernels (SGD-nesterov with
// per-element parameter groups, grad-norm, EMA) of train_yolov5.py:258-280,342-350 and
// trainer/ema_model.py:20-28.  All HBM-bound, one launch per step each.
#include "common.h"
#include <stdarg.h>

namespace {
constexpr int TH = 256;
inline int grid_for(long n) {
    long g = (n + TH - 1) / TH;
    if (g > 4096) g = 4096;
    if (g < 1) g = 1;
    return (int)g;
}

// eight elements per thread: two 16-byte index loads, eight gathered floats in flight, ONE 16-byte store (a 2-byte store per lane
// wrote 128 bytes per wave instruction: 77 us per step for the 18 MB of YOLOv5s' packed weights); the tail and unaligned
// destinations take the scalar path
__global__ void pack_bf16_kernel(const float* __restrict__ src, const int32_t* __restrict__ idx, long n, uint16_t* __restrict__ dst)
{
    const long n8 = ((reinterpret_cast<uintptr_t>(dst) | reinterpret_cast<uintptr_t>(idx)) & 15) ? 0 : n >> 3;
    for (long c = (long)blockIdx.x * blockDim.x + threadIdx.x; c < n8; c += (long)gridDim.x * blockDim.x) {
        const int4 j0 = *reinterpret_cast<const int4*>(idx + c * 8), j1 = *reinterpret_cast<const int4*>(idx + c * 8 + 4);
        const int j[8] = {j0.x, j0.y, j0.z, j0.w, j1.x, j1.y, j1.z, j1.w};
        float f[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) f[e] = j[e] >= 0 ? src[j[e]] : 0.f;
        uint4 v;
        v.x = (uint32_t)f2bf(f[0]) | ((uint32_t)f2bf(f[1]) << 16);
        v.y = (uint32_t)f2bf(f[2]) | ((uint32_t)f2bf(f[3]) << 16);
        v.z = (uint32_t)f2bf(f[4]) | ((uint32_t)f2bf(f[5]) << 16);
        v.w = (uint32_t)f2bf(f[6]) | ((uint32_t)f2bf(f[7]) << 16);
        *reinterpret_cast<uint4*>(dst + c * 8) = v;
    }
    for (long i = n8 * 8 + (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        int32_t j = idx[i];
        dst[i] = j >= 0 ? f2bf(src[j]) : (uint16_t)0;
    }
}
__global__ void gather_f32_kernel(const float* __restrict__ src, const int32_t* __restrict__ idx, long n, float* __restrict__ dst)
{
    const long n4 = ((reinterpret_cast<uintptr_t>(dst) | reinterpret_cast<uintptr_t>(idx)) & 15) ? 0 : n >> 2;      // four per thread: 16-byte stores
    for (long c = (long)blockIdx.x * blockDim.x + threadIdx.x; c < n4; c += (long)gridDim.x * blockDim.x) {
        const int4 j = *reinterpret_cast<const int4*>(idx + c * 4);
        float4 v;
        v.x = j.x >= 0 ? src[j.x] : 0.f; v.y = j.y >= 0 ? src[j.y] : 0.f; v.z = j.z >= 0 ? src[j.z] : 0.f; v.w = j.w >= 0 ? src[j.w] : 0.f;
        *reinterpret_cast<float4*>(dst + c * 4) = v;
    }
    for (long i = n4 * 4 + (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        int32_t j = idx[i];
        dst[i] = j >= 0 ? src[j] : 0.f;
    }
}
__global__ void sgd_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ buf,
                           const uint8_t* __restrict__ group, long n, const float* __restrict__ lr,
                           const float* __restrict__ wd, float momentum, int nesterov, int first,
                           const float* __restrict__ grad_scale)
{
    const float gs = grad_scale ? *grad_scale : 1.f;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        int gi = group ? group[i] : 0;
        float pv = p[i];
        float gv = g[i] * gs + wd[gi] * pv;
        float b = first ? gv : momentum * buf[i] + gv;
        buf[i] = b;
        float upd = nesterov ? gv + momentum * b : b;
        p[i] = pv - lr[gi] * upd;
    }
}
// same update with the per-step scalars read from device memory (scal = lr[0..2] | wd[0..2] | momentum | first-step flag):
// nothing about the step is baked into the launch arguments, so a captured hipGraph replays with the values the host wrote
// into `scal` before the replay (warm-up of lr / momentum, train_yolov5.py:437-456)
__global__ void sgd_dev_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ buf,
                               const uint8_t* __restrict__ group, long n, const float* __restrict__ scal, int nesterov,
                               const float* __restrict__ grad_scale)
{
    const float gs = grad_scale ? *grad_scale : 1.f;
    const float lr[3] = {scal[0], scal[1], scal[2]}, wd[3] = {scal[3], scal[4], scal[5]};
    const float momentum = scal[6];
    const bool first = scal[7] != 0.f;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        int gi = group ? group[i] : 0;
        gi = gi > 2 ? 2 : gi;
        float pv = p[i];
        float gv = g[i] * gs + (gi == 0 ? wd[0] : (gi == 1 ? wd[1] : wd[2])) * pv;
        float b = first ? gv : momentum * buf[i] + gv;
        buf[i] = b;
        float upd = nesterov ? gv + momentum * b : b;
        p[i] = pv - (gi == 0 ? lr[0] : (gi == 1 ? lr[1] : lr[2])) * upd;
    }
}
__global__ void ema_dev_kernel(float* __restrict__ e, const float* __restrict__ p, long n, const float* __restrict__ decay_dev)
{
    const float decay = *decay_dev;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x)
        e[i] = decay * e[i] + (1.f - decay) * p[i];
}
// EMA decay schedule kept on the device (trainer/ema_model.py:12: decay = ratio * (1 - exp(-n / tau)), n = update count):
// one thread advances the counter and writes the decay of THIS update, evaluated in double like the reference's Python float
__global__ void ema_advance_kernel(long long* __restrict__ counter, float* __restrict__ decay, double ratio, double tau)
{
    if (threadIdx.x == 0 && blockIdx.x == 0) {
        const long long n = counter[0] + 1;
        counter[0] = n;
        decay[0] = (float)(ratio * (1.0 - exp(-(double)n / tau)));
    }
}
__global__ __launch_bounds__(TH) void sumsq_part_kernel(const float* __restrict__ x, long n, float* __restrict__ part)
{
    __shared__ double sw[TH / 64];
    double s = 0.0;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        float v = x[i];
        s += (double)v * (double)v;
    }
    s = wave_sum_d(s);
    if ((threadIdx.x & 63) == 0) sw[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) {
        double tsum = 0.0;
        for (int w = 0; w < TH / 64; ++w) tsum += sw[w];
        part[blockIdx.x] = (float)tsum;
    }
}
__global__ __launch_bounds__(256) void sum_final_kernel(const float* __restrict__ part, int nb, float* out)
{
    __shared__ double sw[4];
    double s = 0.0;
    for (int i = threadIdx.x; i < nb; i += 256) s += (double)part[i];
    s = wave_sum_d(s);
    if ((threadIdx.x & 63) == 0) sw[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) *out = (float)(sw[0] + sw[1] + sw[2] + sw[3]);
}
__global__ void clip_scale_kernel(const float* sumsq, float max_norm, float* out)
{
    // torch.nn.utils.clip_grad_norm_: coef = max_norm / (total_norm + 1e-6), clamped to 1
    if (threadIdx.x == 0 && blockIdx.x == 0) {
        float c = max_norm / (sqrtf(*sumsq) + 1e-6f);
        *out = c < 1.f ? c : 1.f;
    }
}
__global__ void ema_kernel(float* __restrict__ e, const float* __restrict__ p, long n, float decay)
{
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x)
        e[i] = decay * e[i] + (1.f - decay) * p[i];
}
}  // namespace

// ---- error plumbing shared by every translation unit ------------------------
static thread_local char g_err[512] = "";
void yh_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}
extern "C" const char* yh_last_error(void) { return g_err; }
extern "C" int yh_version(void) { return 100; }
extern "C" int yh_device_cus(void) {
    int dev = 0;
    hipDeviceProp_t prop;
    if (hipGetDevice(&dev) != hipSuccess) return -1;
    if (hipGetDeviceProperties(&prop, dev) != hipSuccess) return -1;
    return prop.multiProcessorCount;
}

extern "C" int yh_pack_bf16(const float* src, const int32_t* idx, int64_t n, yh_bf16* dst, yh_stream stream)
{
    YH_CHECK_ARG(src && idx && dst && n >= 0, "yh_pack_bf16: bad args");
    if (n == 0) return YH_OK;
    hipLaunchKernelGGL(pack_bf16_kernel, dim3(grid_for(n)), dim3(TH), 0, (hipStream_t)stream, src, idx, (long)n, dst);
    YH_CHECK_LAUNCH("yh_pack_bf16");
    return YH_OK;
}
extern "C" int yh_gather_f32(const float* src, const int32_t* idx, int64_t n, float* dst, yh_stream stream)
{
    YH_CHECK_ARG(src && idx && dst && n >= 0, "yh_gather_f32: bad args");
    if (n == 0) return YH_OK;
    hipLaunchKernelGGL(gather_f32_kernel, dim3(grid_for(n)), dim3(TH), 0, (hipStream_t)stream, src, idx, (long)n, dst);
    YH_CHECK_LAUNCH("yh_gather_f32");
    return YH_OK;
}
extern "C" int yh_sgd_step(float* p, const float* g, float* buf, const uint8_t* group, int64_t n,
                           const float* lr, const float* wd, int ngroups, float momentum, int nesterov,
                           int first_step, const float* grad_scale, yh_stream stream)
{
    YH_CHECK_ARG(p && g && buf && lr && wd && n >= 0 && ngroups >= 1 && ngroups <= 256, "yh_sgd_step: bad args");
    if (n == 0) return YH_OK;
    hipLaunchKernelGGL(sgd_kernel, dim3(grid_for(n)), dim3(TH), 0, (hipStream_t)stream,
                       p, g, buf, group, (long)n, lr, wd, momentum, nesterov, first_step, grad_scale);
    YH_CHECK_LAUNCH("yh_sgd_step");
    return YH_OK;
}
extern "C" int yh_sgd_step_dev(float* p, const float* g, float* buf, const uint8_t* group, int64_t n,
                               const float* scal, int nesterov, const float* grad_scale, yh_stream stream)
{
    YH_CHECK_ARG(p && g && buf && scal && n >= 0, "yh_sgd_step_dev: bad args");
    if (n == 0) return YH_OK;
    hipLaunchKernelGGL(sgd_dev_kernel, dim3(grid_for(n)), dim3(TH), 0, (hipStream_t)stream, p, g, buf, group, (long)n, scal, nesterov, grad_scale);
    YH_CHECK_LAUNCH("yh_sgd_step_dev");
    return YH_OK;
}
extern "C" int yh_ema_update_dev(float* ema, const float* p, int64_t n, const float* decay_dev, yh_stream stream)
{
    YH_CHECK_ARG(ema && p && decay_dev && n >= 0, "yh_ema_update_dev: bad args");
    if (n == 0) return YH_OK;
    hipLaunchKernelGGL(ema_dev_kernel, dim3(grid_for(n)), dim3(TH), 0, (hipStream_t)stream, ema, p, (long)n, decay_dev);
    YH_CHECK_LAUNCH("yh_ema_update_dev");
    return YH_OK;
}
extern "C" int yh_ema_advance(int64_t* counter, float* decay, double ratio, double tau, yh_stream stream)
{
    YH_CHECK_ARG(counter && decay && tau > 0.0, "yh_ema_advance: bad args");
    hipLaunchKernelGGL(ema_advance_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, (long long*)counter, decay, ratio, tau);
    YH_CHECK_LAUNCH("yh_ema_advance");
    return YH_OK;
}
extern "C" int yh_sumsq(const float* x, int64_t n, float* part, float* out, yh_stream stream)
{
    YH_CHECK_ARG(x && part && out && n > 0, "yh_sumsq: bad args (part needs 4096 floats)");
    int nb = grid_for(n);
    hipLaunchKernelGGL(sumsq_part_kernel, dim3(nb), dim3(TH), 0, (hipStream_t)stream, x, (long)n, part);
    hipLaunchKernelGGL(sum_final_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, part, nb, out);
    YH_CHECK_LAUNCH("yh_sumsq");
    return YH_OK;
}
extern "C" int yh_ema_update(float* ema, const float* p, int64_t n, float decay, yh_stream stream)
{
    YH_CHECK_ARG(ema && p && n >= 0, "yh_ema_update: bad args");
    if (n == 0) return YH_OK;
    hipLaunchKernelGGL(ema_kernel, dim3(grid_for(n)), dim3(TH), 0, (hipStream_t)stream, ema, p, (long)n, decay);
    YH_CHECK_LAUNCH("yh_ema_update");
    return YH_OK;
}

extern "C" int yh_clip_scale(const float* sumsq, float max_norm, float* out, yh_stream stream)
{
    YH_CHECK_ARG(sumsq && out && max_norm > 0.f, "yh_clip_scale: bad args");
    hipLaunchKernelGGL(clip_scale_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, sumsq, max_norm, out);
    YH_CHECK_LAUNCH("yh_clip_scale");
    return YH_OK;
}
