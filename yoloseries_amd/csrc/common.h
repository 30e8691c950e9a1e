// Shared device helpers for the gfx950 kernels (wave64, bf16 storage as uint16).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>
#include "../../include/yolohip.h"

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_t;
typedef __attribute__((ext_vector_type(16))) float f32x16_t;
typedef __attribute__((ext_vector_type(4))) float f32x4_t;

#define YH_WAVE 64

__device__ __forceinline__ float bf2f(uint16_t u) { return __uint_as_float(((uint32_t)u) << 16); }
// round-to-nearest-even, NaN preserving (hipcc lowers the cast to v_cvt_pk_bf16_f32)
__device__ __forceinline__ uint16_t f2bf(float f) {
    __bf16 b = (__bf16)f;
    return __builtin_bit_cast(uint16_t, b);
}
__device__ __forceinline__ float bf_round(float f) { return bf2f(f2bf(f)); }

__device__ __forceinline__ void unpack8(const uint4& v, float* f) {
    f[0] = __uint_as_float(v.x << 16); f[1] = __uint_as_float(v.x & 0xffff0000u);
    f[2] = __uint_as_float(v.y << 16); f[3] = __uint_as_float(v.y & 0xffff0000u);
    f[4] = __uint_as_float(v.z << 16); f[5] = __uint_as_float(v.z & 0xffff0000u);
    f[6] = __uint_as_float(v.w << 16); f[7] = __uint_as_float(v.w & 0xffff0000u);
}
// two floats -> packed bf16 pair (lo in bits 0..15), one v_cvt_pk_bf16_f32
__device__ __forceinline__ uint32_t pack2(float lo, float hi) {
    typedef __attribute__((ext_vector_type(2))) float f32x2_t;
    typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2_t;
    f32x2_t v = {lo, hi};
    bf16x2_t b = __builtin_convertvector(v, bf16x2_t);
    return __builtin_bit_cast(uint32_t, b);
}
__device__ __forceinline__ uint4 pack8(const float* f) {
    uint4 v;
    v.x = pack2(f[0], f[1]);
    v.y = pack2(f[2], f[3]);
    v.z = pack2(f[4], f[5]);
    v.w = pack2(f[6], f[7]);
    return v;
}

__device__ __forceinline__ float sigmoidf_(float x) { return 1.0f / (1.0f + __expf(-x)); }
__device__ __forceinline__ float siluf_(float x) { return x * sigmoidf_(x); }
// approximate-reciprocal variants (v_exp_f32 + v_rcp_f32, ~1 ulp each) for the bf16 activation path;
// the loss / assignment kernels keep the IEEE-division forms above
__device__ __forceinline__ float sigmoid_fast(float x) { return __builtin_amdgcn_rcpf(1.0f + __expf(-x)); }
__device__ __forceinline__ float silu_fast(float x) { return x * sigmoid_fast(x); }

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ double wave_sum_d(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

// ---- LDS-DMA (global -> LDS without VGPR staging) ---------------------------
#define YH_VMCNT(n) asm volatile("s_waitcnt vmcnt(%0)" :: "n"(n) : "memory")
// counted wait with a wave-uniform run-time count (the instruction takes an immediate)
#define YH_VMCNT_SW(n) do { switch (n) { case 0: YH_VMCNT(0); break; case 1: YH_VMCNT(1); break; case 2: YH_VMCNT(2); break; case 3: YH_VMCNT(3); break; \
    case 4: YH_VMCNT(4); break; case 5: YH_VMCNT(5); break; case 6: YH_VMCNT(6); break; case 7: YH_VMCNT(7); break; case 8: YH_VMCNT(8); break; \
    case 9: YH_VMCNT(9); break; default: YH_VMCNT(10); break; } } while (0)
// workgroup barrier that orders LDS accesses only: unlike __syncthreads() it does not drain LDS-DMA transfers in flight
#define YH_LDS_BARRIER() do { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); __builtin_amdgcn_s_barrier(); } while (0)

// one LDS-DMA wave instruction: 64 lanes x 16 bytes from (rsrc, per-lane voff + scalar soff) to lds .. lds + 1024
__device__ __forceinline__ void lds_dma16(const __amdgpu_buffer_rsrc_t rs, unsigned char* lds, unsigned voff, int soff)
{
#if defined(__HIP_DEVICE_COMPILE__)
    typedef __attribute__((address_space(3))) void lds_void;
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_void*)lds, 16, voff, soff, 0, 0);
#endif
}

// ---- host side helpers -----------------------------------------------------
void yh_set_error(const char* fmt, ...);

// hipFuncAttributeMaxDynamicSharedMemorySize is a PER-DEVICE attribute of a kernel: a process that launches on a second GPU must set
// it there too.  One flag per device (bit mask over the first 64 devices, others set the attribute at every launch), written only
// after every hipFuncSetAttribute of the block has succeeded; two threads may both set the attributes (idempotent), none launches
// before they are set.  A failed set leaves the flag clear and the launch behind it reports hipErrorInvalidValue (YH_CHECK_LAUNCH).
#include <atomic>
struct YhDevOnce {
    std::atomic<uint64_t> mask{0};
    bool failed = false;
    static int dev_() { int d = -1; return hipGetDevice(&d) == hipSuccess ? d : -1; }
    bool need() { const int d = dev_(); failed = false; return d < 0 || d >= 64 || !((mask.load(std::memory_order_acquire) >> d) & 1ull); }
    void set(const void* fn, hipFuncAttribute attr, int value) {
        const hipError_t e = hipFuncSetAttribute(fn, attr, value);
        if (e != hipSuccess) { failed = true; yh_set_error("hipFuncSetAttribute(%d bytes of LDS): %s", value, hipGetErrorString(e)); }
    }
    void done() { const int d = dev_(); if (!failed && d >= 0 && d < 64) mask.fetch_or(1ull << d, std::memory_order_release); }
};
#define YH_CHECK_ARG(cond, ...) do { if (!(cond)) { yh_set_error(__VA_ARGS__); return YH_EINVAL; } } while (0)
#define YH_CHECK_LAUNCH(name) do { hipError_t e_ = hipGetLastError(); \
    if (e_ != hipSuccess) { yh_set_error("%s: launch failed: %s", name, hipGetErrorString(e_)); return YH_ELAUNCH; } } while (0)

// conv_dg2.hip: the stride-2 data-gradient kernel behind yh_conv_igemm (algo 7)
int yh_dg2_rows(const yh_conv_desc* d);                 // grid rows (== slab rows of the fused reduction); 0 = not eligible
int yh_dg2_run(const yh_conv_desc* d, yh_stream stream, char* name_out, int name_len);

// conv_p3.hip: the 3x3 / stride-1 patch kernel for small channel counts behind yh_conv_igemm (algo 8)
int yh_p3_rows(const yh_conv_desc* d);                  // grid rows (== statistics / fused-reduction slab rows); 0 = not eligible
int yh_p3_run(const yh_conv_desc* d, yh_stream stream, char* name_out, int name_len);

// conv_h80.hip: the 80-channel 3x3 halo kernel (YOLOv5x stage 1, inference epilogues) behind yh_conv_igemm (algo 9)
int yh_h80_rows(const yh_conv_desc* d);                 // grid rows; 0 = not eligible
int yh_h80_run(const yh_conv_desc* d, yh_stream stream, char* name_out, int name_len);

// conv_pw.hip: the pointwise (1x1) kernel for the 80- / 160-channel layers of YOLOv5x (inference epilogues) behind yh_conv_igemm (algo 10)
int yh_c80_rows(const yh_conv_desc* d);                 // grid size; 0 = not eligible (conv_c80.hip)
int yh_c80_run(const yh_conv_desc* d, yh_stream stream, char* name_out, int name_len);
int yh_pw_rows(const yh_conv_desc* d);                  // grid rows; 0 = not eligible
int yh_pw_run(const yh_conv_desc* d, yh_stream stream, char* name_out, int name_len);

// conv_pt.hip: the pointwise (1x1) kernel of the TRAINING step (128 / 256 input channels; all four epilogues) behind yh_conv_igemm (algo 13)
int yh_pt_rows(const yh_conv_desc* d);                  // pixel slots of the grid (== statistics / fused-reduction slab rows); 0 = not eligible
int yh_pt_run(const yh_conv_desc* d, yh_stream stream, char* name_out, int name_len);

// conv_wgp.hip: the patch form of the weight gradient behind yh_conv_wgrad (tile_k 40)
int yh_wgp_ok(const yh_wgrad_desc* d);
int yh_wgp_run(const yh_wgrad_desc* d, yh_stream stream);

// conv_wgs.hip: wave-private 128 x 128 tiles + stream-K for the K-heavy layers behind yh_conv_wgrad (tile_k 129)
int yh_wgs_ok(const yh_wgrad_desc* d);
int yh_wgs_tiles(const yh_wgrad_desc* d);
const char* yh_wgs_name(const yh_wgrad_desc* d);
int yh_wgs_run(const yh_wgrad_desc* d, yh_stream stream);

static inline bool yh_aligned16(const void* p) { return (((uintptr_t)p) & 15) == 0; }
