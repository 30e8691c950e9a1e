// Packed-fp32 instruction forms beside MFMA on MI355X (DESIGN §5 (l); driver: tools/pk_victim_probe.py).
//   pk_victim_kernel<FORM>: every thread repeats ONE packed operation (inline asm: the exact VOP3P form; FORM >= 100 = every form the
//     -O3 build of v5_pos_bwd_kernel contains) and the same two scalar operations, and counts bit differences per half.
//   pk_trigger_kernel<KIND>: what the kernel on the other stream executes (plain VALU, v_accvgpr moves, MFMA 32x32x16 / 16x16x32 with
//     VGPR / AGPR accumulators, transposing LDS reads).
//   pk_poison_kernel: every vector register of a wave := a pattern (leftover register contents are NOT what the fault depends on).
// Result on the boxes of this pool: `v_pk_{add,mul}_f32 D, X, Y op_sel:[0,1] ...` — the LOW lane reading the HIGH dword of the second
// source — is wrong in ~5 % of the threads (2000 operations each) while v_mfma_f32_16x16x32_bf16 runs on the other stream or the
// engine's weight-gradient kernels do; 0 wrong of 10^13 for every other form, and for this one with nothing / plain VALU beside it.
// Build: hipcc -O3 -fno-slp-vectorize -fno-vectorize --offload-arch=gfx950 -shared -fPIC -o libpk_victim.so pk_victim.hip
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef float f2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ float smul(float a, float b) { float r; asm volatile("v_mul_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }
__device__ __forceinline__ float sadd(float a, float b) { float r; asm volatile("v_add_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }
__device__ __forceinline__ float sfma(float a, float b, float c) { float r; asm volatile("v_fma_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c)); return r; }

template <int FORM>
__global__ __launch_bounds__(256) void pk_victim_kernel(int iters, const float* __restrict__ in, unsigned long long* __restrict__ mism,
                                                        unsigned* __restrict__ first)
{
    const int tid = blockIdx.x * blockDim.x + threadIdx.x;
    float a = in[(tid * 4 + 0) & 0xffff], b = in[(tid * 4 + 1) & 0xffff], c = in[(tid * 4 + 2) & 0xffff], d = in[(tid * 4 + 3) & 0xffff];
    unsigned bad = 0;
    const float su0 = in[blockIdx.x & 0xff], su1 = in[(blockIdx.x + 77) & 0xff];
    for (int it = 0; it < iters; ++it) {
        f2 x = {a, b}, y = {c, d}, r;
        const f2 ysf = {su0 + 0.001f * (it & 7), su1};
        const f2 ys = {__builtin_amdgcn_readfirstlane(__float_as_int(ysf.x)) == 0 ? 0.f : __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(ysf.x))), __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(ysf.y)))};
        float lo, hi;
        if (FORM == 0) {
            asm volatile("v_pk_mul_f32 %0, %1, %2" : "=v"(r) : "v"(x), "v"(y));
            lo = smul(a, c); hi = smul(b, d);
        } else if (FORM == 1) {
            asm volatile("v_pk_mul_f32 %0, %1, %2 neg_lo:[0,1] neg_hi:[0,1]" : "=v"(r) : "v"(x), "v"(y));
            lo = smul(a, -c); hi = smul(b, -d);
        } else if (FORM == 2) {
            asm volatile("v_pk_mul_f32 %0, %1, %2 op_sel_hi:[1,0]" : "=v"(r) : "v"(x), "v"(y));      // hi uses the LOW half of y
            lo = smul(a, c); hi = smul(b, c);
        } else if (FORM == 3) {
            asm volatile("v_pk_add_f32 %0, %1, %2 neg_lo:[0,1] neg_hi:[0,1]" : "=v"(r) : "v"(x), "v"(y));
            lo = sadd(a, -c); hi = sadd(b, -d);
        } else if (FORM == 4) {
            f2 z = {d, a};
            asm volatile("v_pk_fma_f32 %0, %1, %2, %3" : "=v"(r) : "v"(x), "v"(y), "v"(z));
            lo = sfma(a, c, d); hi = sfma(b, d, a);
        } else if (FORM == 5) {
            asm volatile("v_pk_mov_b32 %0, %1, %2 op_sel:[1,0]" : "=v"(r) : "v"(x), "v"(y));        // lo = x.hi, hi = y.lo
            lo = b; hi = c;
        } else if (FORM == 6) {
            // the compiler's own sequence around the CIoU comparisons: a half of the pair written by v_cndmask right before the packed read
            float m0, m1;
            asm volatile("v_cmp_lt_f32 vcc, %2, %3\n\ts_nop 1\n\tv_cndmask_b32 %0, 0, 1.0, vcc\n\t"
                         "v_cmp_lt_f32 vcc, %4, %5\n\ts_nop 1\n\tv_cndmask_b32 %1, 0, 1.0, vcc"
                         : "=&v"(m0), "=&v"(m1) : "v"(a), "v"(c), "v"(b), "v"(d) : "vcc");
            f2 m = {m0, m1};
            asm volatile("v_pk_mul_f32 %0, %1, %2 neg_lo:[0,1] neg_hi:[0,1]" : "=v"(r) : "v"(m), "v"(y));
            lo = smul(a < c ? 1.f : 0.f, -c); hi = smul(b < d ? 1.f : 0.f, -d);
        } else if (FORM == 100) {
            asm volatile("v_pk_add_f32 %0, %1, %2 neg_lo:[0,1] neg_hi:[0,1]" : "=v"(r) : "v"(x), "v"(y));
            lo = sadd(x.x, (-y.x)); hi = sadd(x.y, (-y.y));
        } else if (FORM == 101) {
            asm volatile("v_pk_add_f32 %0, %1, %2" : "=v"(r) : "v"(x), "v"(y));
            lo = sadd(x.x, y.x); hi = sadd(x.y, y.y);
        } else if (FORM == 102) {
            asm volatile("v_pk_mul_f32 %0, %1, %2" : "=v"(r) : "v"(x), "v"(y));
            lo = smul(x.x, y.x); hi = smul(x.y, y.y);
        } else if (FORM == 103) {
            asm volatile("v_pk_mul_f32 %0, %1, %2 op_sel_hi:[0,1] neg_lo:[0,1] neg_hi:[0,1]" : "=v"(r) : "v"(x), "v"(y));
            lo = smul(x.x, (-y.x)); hi = smul(x.x, (-y.y));
        } else if (FORM == 104) {
            asm volatile("v_pk_add_f32 %0, %1, %2 op_sel:[1,0] op_sel_hi:[0,1]" : "=v"(r) : "v"(x), "v"(y));
            lo = sadd(x.y, y.x); hi = sadd(x.x, y.y);
        } else if (FORM == 105) {
            asm volatile("v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0]" : "=v"(r) : "v"(x), "v"(y));
            lo = sadd(x.x, y.y); hi = sadd(x.y, y.x);
        } else if (FORM == 106) {
            asm volatile("v_pk_mul_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0]" : "=v"(r) : "v"(x), "v"(y));
            lo = smul(x.x, y.y); hi = smul(x.y, y.x);
        } else if (FORM == 107) {
            asm volatile("v_pk_mul_f32 %0, %1, %2 op_sel_hi:[0,1]" : "=v"(r) : "v"(x), "v"(y));
            lo = smul(x.x, y.x); hi = smul(x.x, y.y);
        } else if (FORM == 108) {
            asm volatile("v_pk_mul_f32 %0, %1, 0 op_sel_hi:[1,0]" : "=v"(r) : "v"(x));
            lo = smul(x.x, 0.f); hi = smul(x.y, 0.f);
        } else if (FORM == 109) {
            asm volatile("v_pk_add_f32 %0, %1, %2 op_sel:[1,0] op_sel_hi:[0,1] neg_lo:[0,1] neg_hi:[0,1]" : "=v"(r) : "v"(x), "v"(y));
            lo = sadd(x.y, (-y.x)); hi = sadd(x.x, (-y.y));
        } else if (FORM == 110) {
            asm volatile("v_pk_mul_f32 %0, %1, 0.5 op_sel_hi:[1,0]" : "=v"(r) : "v"(x));
            lo = smul(x.x, 0.5f); hi = smul(x.y, 0.5f);
        } else if (FORM == 111) {
            asm volatile("v_pk_add_f32 %0, %1, %2 op_sel_hi:[1,0] neg_lo:[0,1] neg_hi:[0,1]" : "=v"(r) : "v"(x), "v"(y));
            lo = sadd(x.x, (-y.x)); hi = sadd(x.y, (-y.x));
        } else if (FORM == 112) {
            asm volatile("v_pk_mul_f32 %0, %1, %2 op_sel_hi:[1,0]" : "=v"(r) : "v"(x), "v"(y));
            lo = smul(x.x, y.x); hi = smul(x.y, y.x);
        } else if (FORM == 113) {
            asm volatile("v_pk_mul_f32 %0, %1, %2 op_sel_hi:[0,1]" : "=v"(r) : "v"(x), "s"(ys));
            lo = smul(x.x, ysf.x); hi = smul(x.x, ysf.y);
        } else if (FORM == 114) {
            asm volatile("v_pk_add_f32 %0, %1, %2 neg_lo:[1,0] neg_hi:[1,0]" : "=v"(r) : "v"(x), "s"(ys));
            lo = sadd((-x.x), ysf.x); hi = sadd((-x.y), ysf.y);
        } else if (FORM == 115) {
            asm volatile("v_pk_add_f32 %0, %1, %2" : "=v"(r) : "v"(x), "s"(ys));
            lo = sadd(x.x, ysf.x); hi = sadd(x.y, ysf.y);
        } else if (FORM == 116) {
            asm volatile("v_pk_add_f32 %0, %1, 1.0 op_sel_hi:[1,0]" : "=v"(r) : "v"(x));
            lo = sadd(x.x, 1.0f); hi = sadd(x.y, 1.0f);
        } else if (FORM == 117) {
            asm volatile("v_pk_mul_f32 %0, %1, %2 neg_lo:[0,1] neg_hi:[0,1]" : "=v"(r) : "v"(x), "v"(y));
            lo = smul(x.x, (-y.x)); hi = smul(x.y, (-y.y));
        } else if (FORM == 118) {
            asm volatile("v_pk_add_f32 %0, %1, %2 op_sel_hi:[0,1]" : "=v"(r) : "v"(x), "v"(y));
            lo = sadd(x.x, y.x); hi = sadd(x.x, y.y);
        } else if (FORM == 119) {
            asm volatile("v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0] neg_lo:[0,1] neg_hi:[0,1]" : "=v"(r) : "v"(x), "v"(y));
            lo = sadd(x.x, (-y.y)); hi = sadd(x.y, (-y.x));
        } else if (FORM == 120) {
            asm volatile("v_pk_add_f32 %0, %1, 1.0 op_sel_hi:[1,0] neg_lo:[1,0] neg_hi:[1,0]" : "=v"(r) : "v"(x));
            lo = sadd((-x.x), 1.0f); hi = sadd((-x.y), 1.0f);
        } else if (FORM == 121) {
            asm volatile("v_pk_add_f32 %0, %1, -0.5 op_sel_hi:[1,0]" : "=v"(r) : "v"(x));
            lo = sadd(x.x, -0.5f); hi = sadd(x.y, -0.5f);
        } else if (FORM == 200) {
            asm volatile("v_pk_mul_f32 %0, %1, %2 op_sel:[0,1]" : "=v"(r) : "v"(x), "v"(y));                 // lo = x.lo * y.hi, hi = x.hi * y.hi
            lo = smul(x.x, y.y); hi = smul(x.y, y.y);
        } else if (FORM == 201) {
            asm volatile("v_pk_add_f32 %0, %1, %2 op_sel:[1,1] op_sel_hi:[0,0]" : "=v"(r) : "v"(x), "v"(y)); // both sources swapped
            lo = sadd(x.y, y.y); hi = sadd(x.x, y.x);
        } else if (FORM == 202) {
            f2 z = {d, a};
            asm volatile("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[0,1,0] op_sel_hi:[1,0,1]" : "=v"(r) : "v"(x), "v"(y), "v"(z));
            lo = sfma(x.x, y.y, z.x); hi = sfma(x.y, y.x, z.y);
        } else if (FORM == 203) {
            f2 z = {d, a};
            asm volatile("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[0,0,1] op_sel_hi:[1,1,0]" : "=v"(r) : "v"(x), "v"(y), "v"(z));
            lo = sfma(x.x, y.x, z.y); hi = sfma(x.y, y.y, z.x);
        } else if (FORM == 204) {
            asm volatile("v_pk_mov_b32 %0, %1, %2 op_sel:[0,1]" : "=v"(r) : "v"(x), "v"(y));                 // lo = x.lo, hi = y.hi
            lo = x.x; hi = y.y;
        } else if (FORM == 205) {
            asm volatile("v_pk_mov_b32 %0, %1, %2 op_sel:[1,1]" : "=v"(r) : "v"(x), "v"(y));                 // lo = x.hi, hi = y.hi
            lo = x.y; hi = y.y;
        } else {
            asm volatile("v_pk_mul_f32 %0, %1, 0 op_sel_hi:[1,0]" : "=v"(r) : "v"(x));
            lo = smul(a, 0.f); hi = smul(b, 0.f);
        }
        if (__float_as_uint(r.x) != __float_as_uint(lo)) bad |= 1;
        if (__float_as_uint(r.y) != __float_as_uint(hi)) bad |= 2;
        // next operands: bounded, data dependent
        a = a * 0.75f + 0.3f * d; b = b * 0.5f - 0.4f * c; c = c * 0.9f + 0.05f; d = 0.8f * d - 0.1f * a;
        if (!(fabsf(a) < 8.f)) a = 0.37f;
        if (!(fabsf(b) < 8.f)) b = -0.61f;
        if (!(fabsf(d) < 8.f)) d = 0.11f;
    }
    if (bad) { atomicAdd(mism + (bad & 1 ? 0 : 1), 1ull); if (bad & 2) atomicAdd(mism + 2, 1ull); atomicMax(first, (unsigned)tid + 1); }
}

extern "C" int pk_victim_launch(int form, int blocks, int iters, const float* in, unsigned long long* mism, unsigned* first, void* stream)
{
    hipStream_t st = (hipStream_t)stream;
    switch (form) {
    case 0: pk_victim_kernel<0><<<blocks, 256, 0, st>>>(iters, in, mism, first); break;
    case 1: pk_victim_kernel<1><<<blocks, 256, 0, st>>>(iters, in, mism, first); break;
    case 2: pk_victim_kernel<2><<<blocks, 256, 0, st>>>(iters, in, mism, first); break;
    case 3: pk_victim_kernel<3><<<blocks, 256, 0, st>>>(iters, in, mism, first); break;
    case 4: pk_victim_kernel<4><<<blocks, 256, 0, st>>>(iters, in, mism, first); break;
    case 5: pk_victim_kernel<5><<<blocks, 256, 0, st>>>(iters, in, mism, first); break;
    case 6: pk_victim_kernel<6><<<blocks, 256, 0, st>>>(iters, in, mism, first); break;
    case 100: pk_victim_kernel<100><<<blocks, 256, 0, st>>>(iters, in, mism, first); break;
    case 101: pk_victim_kernel<101><<<blocks, 256, 0, st>>>(iters, in, mism, first); break;
    case 102: pk_victim_kernel<102><<<blocks, 256, 0, st>>>(iters, in, mism, first); break;
    case 103: pk_victim_kernel<103><<<blocks, 256, 0, st>>>(iters, in, mism, first); break;
    case 104: pk_victim_kernel<104><<<blocks, 256, 0, st>>>(iters, in, mism, first); break;
    case 105: pk_victim_kernel<105><<<blocks, 256, 0, st>>>(iters, in, mism, first); break;
    case 106: pk_victim_kernel<106><<<blocks, 256, 0, st>>>(iters, in, mism, first); break;
    case 107: pk_victim_kernel<107><<<blocks, 256, 0, st>>>(iters, in, mism, first); break;
    case 108: pk_victim_kernel<108><<<blocks, 256, 0, st>>>(iters, in, mism, first); break;
    case 109: pk_victim_kernel<109><<<blocks, 256, 0, st>>>(iters, in, mism, first); break;
    case 110: pk_victim_kernel<110><<<blocks, 256, 0, st>>>(iters, in, mism, first); break;
    case 111: pk_victim_kernel<111><<<blocks, 256, 0, st>>>(iters, in, mism, first); break;
    case 112: pk_victim_kernel<112><<<blocks, 256, 0, st>>>(iters, in, mism, first); break;
    case 113: pk_victim_kernel<113><<<blocks, 256, 0, st>>>(iters, in, mism, first); break;
    case 114: pk_victim_kernel<114><<<blocks, 256, 0, st>>>(iters, in, mism, first); break;
    case 115: pk_victim_kernel<115><<<blocks, 256, 0, st>>>(iters, in, mism, first); break;
    case 116: pk_victim_kernel<116><<<blocks, 256, 0, st>>>(iters, in, mism, first); break;
    case 117: pk_victim_kernel<117><<<blocks, 256, 0, st>>>(iters, in, mism, first); break;
    case 118: pk_victim_kernel<118><<<blocks, 256, 0, st>>>(iters, in, mism, first); break;
    case 119: pk_victim_kernel<119><<<blocks, 256, 0, st>>>(iters, in, mism, first); break;
    case 120: pk_victim_kernel<120><<<blocks, 256, 0, st>>>(iters, in, mism, first); break;
    case 121: pk_victim_kernel<121><<<blocks, 256, 0, st>>>(iters, in, mism, first); break;
    case 200: pk_victim_kernel<200><<<blocks, 256, 0, st>>>(iters, in, mism, first); break;
    case 201: pk_victim_kernel<201><<<blocks, 256, 0, st>>>(iters, in, mism, first); break;
    case 202: pk_victim_kernel<202><<<blocks, 256, 0, st>>>(iters, in, mism, first); break;
    case 203: pk_victim_kernel<203><<<blocks, 256, 0, st>>>(iters, in, mism, first); break;
    case 204: pk_victim_kernel<204><<<blocks, 256, 0, st>>>(iters, in, mism, first); break;
    case 205: pk_victim_kernel<205><<<blocks, 256, 0, st>>>(iters, in, mism, first); break;
    default: pk_victim_kernel<7><<<blocks, 256, 0, st>>>(iters, in, mism, first); break;
    }
    return (int)hipGetLastError();
}

// every architectural vector register of the wave (255 VGPRs + 256 AGPRs) := pattern: what the NEXT wave on this SIMD finds in registers it
// never wrote (is the loss backward's packed build sensitive to that? tools/loss_race_diag.py POISON=1)
__global__ __launch_bounds__(256) void pk_poison_kernel(unsigned pattern, unsigned* sink)
{
    unsigned v = pattern;
    asm volatile("v_mov_b32 v1, %0\n\tv_mov_b32 v2, %0\n\tv_mov_b32 v3, %0\n\tv_mov_b32 v4, %0\n\tv_mov_b32 v5, %0\n\tv_mov_b32 v6, %0\n\tv_mov_b32 v7, %0\n\tv_mov_b32 v8, %0\n\tv_mov_b32 v9, %0\n\tv_mov_b32 v10, %0\n\tv_mov_b32 v11, %0\n\tv_mov_b32 v12, %0\n\tv_mov_b32 v13, %0\n\tv_mov_b32 v14, %0\n\tv_mov_b32 v15, %0\n\tv_mov_b32 v16, %0\n\tv_mov_b32 v17, %0\n\tv_mov_b32 v18, %0\n\tv_mov_b32 v19, %0\n\tv_mov_b32 v20, %0\n\tv_mov_b32 v21, %0\n\tv_mov_b32 v22, %0\n\tv_mov_b32 v23, %0\n\tv_mov_b32 v24, %0\n\tv_mov_b32 v25, %0\n\tv_mov_b32 v26, %0\n\tv_mov_b32 v27, %0\n\tv_mov_b32 v28, %0\n\tv_mov_b32 v29, %0\n\tv_mov_b32 v30, %0\n\tv_mov_b32 v31, %0\n\tv_mov_b32 v32, %0\n\tv_mov_b32 v33, %0\n\tv_mov_b32 v34, %0\n\tv_mov_b32 v35, %0\n\tv_mov_b32 v36, %0\n\tv_mov_b32 v37, %0\n\tv_mov_b32 v38, %0\n\tv_mov_b32 v39, %0\n\tv_mov_b32 v40, %0\n\tv_mov_b32 v41, %0\n\tv_mov_b32 v42, %0\n\tv_mov_b32 v43, %0\n\tv_mov_b32 v44, %0\n\tv_mov_b32 v45, %0\n\tv_mov_b32 v46, %0\n\tv_mov_b32 v47, %0\n\tv_mov_b32 v48, %0\n\tv_mov_b32 v49, %0\n\tv_mov_b32 v50, %0\n\tv_mov_b32 v51, %0\n\tv_mov_b32 v52, %0\n\tv_mov_b32 v53, %0\n\tv_mov_b32 v54, %0\n\tv_mov_b32 v55, %0\n\tv_mov_b32 v56, %0\n\tv_mov_b32 v57, %0\n\tv_mov_b32 v58, %0\n\tv_mov_b32 v59, %0\n\tv_mov_b32 v60, %0\n\tv_mov_b32 v61, %0\n\tv_mov_b32 v62, %0\n\tv_mov_b32 v63, %0\n\tv_mov_b32 v64, %0\n\tv_mov_b32 v65, %0\n\tv_mov_b32 v66, %0\n\tv_mov_b32 v67, %0\n\tv_mov_b32 v68, %0\n\tv_mov_b32 v69, %0\n\tv_mov_b32 v70, %0\n\tv_mov_b32 v71, %0\n\tv_mov_b32 v72, %0\n\tv_mov_b32 v73, %0\n\tv_mov_b32 v74, %0\n\tv_mov_b32 v75, %0\n\tv_mov_b32 v76, %0\n\tv_mov_b32 v77, %0\n\tv_mov_b32 v78, %0\n\tv_mov_b32 v79, %0\n\tv_mov_b32 v80, %0\n\tv_mov_b32 v81, %0\n\tv_mov_b32 v82, %0\n\tv_mov_b32 v83, %0\n\tv_mov_b32 v84, %0\n\tv_mov_b32 v85, %0\n\tv_mov_b32 v86, %0\n\tv_mov_b32 v87, %0\n\tv_mov_b32 v88, %0\n\tv_mov_b32 v89, %0\n\tv_mov_b32 v90, %0\n\tv_mov_b32 v91, %0\n\tv_mov_b32 v92, %0\n\tv_mov_b32 v93, %0\n\tv_mov_b32 v94, %0\n\tv_mov_b32 v95, %0\n\tv_mov_b32 v96, %0\n\tv_mov_b32 v97, %0\n\tv_mov_b32 v98, %0\n\tv_mov_b32 v99, %0\n\tv_mov_b32 v100, %0\n\tv_mov_b32 v101, %0\n\tv_mov_b32 v102, %0\n\tv_mov_b32 v103, %0\n\tv_mov_b32 v104, %0\n\tv_mov_b32 v105, %0\n\tv_mov_b32 v106, %0\n\tv_mov_b32 v107, %0\n\tv_mov_b32 v108, %0\n\tv_mov_b32 v109, %0\n\tv_mov_b32 v110, %0\n\tv_mov_b32 v111, %0\n\tv_mov_b32 v112, %0\n\tv_mov_b32 v113, %0\n\tv_mov_b32 v114, %0\n\tv_mov_b32 v115, %0\n\tv_mov_b32 v116, %0\n\tv_mov_b32 v117, %0\n\tv_mov_b32 v118, %0\n\tv_mov_b32 v119, %0\n\tv_mov_b32 v120, %0\n\tv_mov_b32 v121, %0\n\tv_mov_b32 v122, %0\n\tv_mov_b32 v123, %0\n\tv_mov_b32 v124, %0\n\tv_mov_b32 v125, %0\n\tv_mov_b32 v126, %0\n\tv_mov_b32 v127, %0\n\tv_mov_b32 v128, %0\n\tv_mov_b32 v129, %0\n\tv_mov_b32 v130, %0\n\tv_mov_b32 v131, %0\n\tv_mov_b32 v132, %0\n\tv_mov_b32 v133, %0\n\tv_mov_b32 v134, %0\n\tv_mov_b32 v135, %0\n\tv_mov_b32 v136, %0\n\tv_mov_b32 v137, %0\n\tv_mov_b32 v138, %0\n\tv_mov_b32 v139, %0\n\tv_mov_b32 v140, %0\n\tv_mov_b32 v141, %0\n\tv_mov_b32 v142, %0\n\tv_mov_b32 v143, %0\n\tv_mov_b32 v144, %0\n\tv_mov_b32 v145, %0\n\tv_mov_b32 v146, %0\n\tv_mov_b32 v147, %0\n\tv_mov_b32 v148, %0\n\tv_mov_b32 v149, %0\n\tv_mov_b32 v150, %0\n\tv_mov_b32 v151, %0\n\tv_mov_b32 v152, %0\n\tv_mov_b32 v153, %0\n\tv_mov_b32 v154, %0\n\tv_mov_b32 v155, %0\n\tv_mov_b32 v156, %0\n\tv_mov_b32 v157, %0\n\tv_mov_b32 v158, %0\n\tv_mov_b32 v159, %0\n\tv_mov_b32 v160, %0\n\tv_mov_b32 v161, %0\n\tv_mov_b32 v162, %0\n\tv_mov_b32 v163, %0\n\tv_mov_b32 v164, %0\n\tv_mov_b32 v165, %0\n\tv_mov_b32 v166, %0\n\tv_mov_b32 v167, %0\n\tv_mov_b32 v168, %0\n\tv_mov_b32 v169, %0\n\tv_mov_b32 v170, %0\n\tv_mov_b32 v171, %0\n\tv_mov_b32 v172, %0\n\tv_mov_b32 v173, %0\n\tv_mov_b32 v174, %0\n\tv_mov_b32 v175, %0\n\tv_mov_b32 v176, %0\n\tv_mov_b32 v177, %0\n\tv_mov_b32 v178, %0\n\tv_mov_b32 v179, %0\n\tv_mov_b32 v180, %0\n\tv_mov_b32 v181, %0\n\tv_mov_b32 v182, %0\n\tv_mov_b32 v183, %0\n\tv_mov_b32 v184, %0\n\tv_mov_b32 v185, %0\n\tv_mov_b32 v186, %0\n\tv_mov_b32 v187, %0\n\tv_mov_b32 v188, %0\n\tv_mov_b32 v189, %0\n\tv_mov_b32 v190, %0\n\tv_mov_b32 v191, %0\n\tv_mov_b32 v192, %0\n\tv_mov_b32 v193, %0\n\tv_mov_b32 v194, %0\n\tv_mov_b32 v195, %0\n\tv_mov_b32 v196, %0\n\tv_mov_b32 v197, %0\n\tv_mov_b32 v198, %0\n\tv_mov_b32 v199, %0\n\tv_mov_b32 v200, %0\n\tv_mov_b32 v201, %0\n\tv_mov_b32 v202, %0\n\tv_mov_b32 v203, %0\n\tv_mov_b32 v204, %0\n\tv_mov_b32 v205, %0\n\tv_mov_b32 v206, %0\n\tv_mov_b32 v207, %0\n\tv_mov_b32 v208, %0\n\tv_mov_b32 v209, %0\n\tv_mov_b32 v210, %0\n\tv_mov_b32 v211, %0\n\tv_mov_b32 v212, %0\n\tv_mov_b32 v213, %0\n\tv_mov_b32 v214, %0\n\tv_mov_b32 v215, %0\n\tv_mov_b32 v216, %0\n\tv_mov_b32 v217, %0\n\tv_mov_b32 v218, %0\n\tv_mov_b32 v219, %0\n\tv_mov_b32 v220, %0\n\tv_mov_b32 v221, %0\n\tv_mov_b32 v222, %0\n\tv_mov_b32 v223, %0\n\tv_mov_b32 v224, %0\n\tv_mov_b32 v225, %0\n\tv_mov_b32 v226, %0\n\tv_mov_b32 v227, %0\n\tv_mov_b32 v228, %0\n\tv_mov_b32 v229, %0\n\tv_mov_b32 v230, %0\n\tv_mov_b32 v231, %0\n\tv_mov_b32 v232, %0\n\tv_mov_b32 v233, %0\n\tv_mov_b32 v234, %0\n\tv_mov_b32 v235, %0\n\tv_mov_b32 v236, %0\n\tv_mov_b32 v237, %0\n\tv_mov_b32 v238, %0\n\tv_mov_b32 v239, %0\n\tv_mov_b32 v240, %0\n\tv_mov_b32 v241, %0\n\tv_mov_b32 v242, %0\n\tv_mov_b32 v243, %0\n\tv_mov_b32 v244, %0\n\tv_mov_b32 v245, %0\n\tv_mov_b32 v246, %0\n\tv_mov_b32 v247, %0\n\tv_mov_b32 v248, %0\n\tv_mov_b32 v249, %0\n\tv_mov_b32 v250, %0\n\tv_mov_b32 v251, %0\n\tv_mov_b32 v252, %0\n\tv_mov_b32 v253, %0\n\tv_mov_b32 v254, %0\n\tv_mov_b32 v255, %0\n\t"
                 "v_accvgpr_write_b32 a0, %0\n\tv_accvgpr_write_b32 a1, %0\n\tv_accvgpr_write_b32 a2, %0\n\tv_accvgpr_write_b32 a3, %0\n\tv_accvgpr_write_b32 a4, %0\n\tv_accvgpr_write_b32 a5, %0\n\tv_accvgpr_write_b32 a6, %0\n\tv_accvgpr_write_b32 a7, %0\n\tv_accvgpr_write_b32 a8, %0\n\tv_accvgpr_write_b32 a9, %0\n\tv_accvgpr_write_b32 a10, %0\n\tv_accvgpr_write_b32 a11, %0\n\tv_accvgpr_write_b32 a12, %0\n\tv_accvgpr_write_b32 a13, %0\n\tv_accvgpr_write_b32 a14, %0\n\tv_accvgpr_write_b32 a15, %0\n\tv_accvgpr_write_b32 a16, %0\n\tv_accvgpr_write_b32 a17, %0\n\tv_accvgpr_write_b32 a18, %0\n\tv_accvgpr_write_b32 a19, %0\n\tv_accvgpr_write_b32 a20, %0\n\tv_accvgpr_write_b32 a21, %0\n\tv_accvgpr_write_b32 a22, %0\n\tv_accvgpr_write_b32 a23, %0\n\tv_accvgpr_write_b32 a24, %0\n\tv_accvgpr_write_b32 a25, %0\n\tv_accvgpr_write_b32 a26, %0\n\tv_accvgpr_write_b32 a27, %0\n\tv_accvgpr_write_b32 a28, %0\n\tv_accvgpr_write_b32 a29, %0\n\tv_accvgpr_write_b32 a30, %0\n\tv_accvgpr_write_b32 a31, %0\n\tv_accvgpr_write_b32 a32, %0\n\tv_accvgpr_write_b32 a33, %0\n\tv_accvgpr_write_b32 a34, %0\n\tv_accvgpr_write_b32 a35, %0\n\tv_accvgpr_write_b32 a36, %0\n\tv_accvgpr_write_b32 a37, %0\n\tv_accvgpr_write_b32 a38, %0\n\tv_accvgpr_write_b32 a39, %0\n\tv_accvgpr_write_b32 a40, %0\n\tv_accvgpr_write_b32 a41, %0\n\tv_accvgpr_write_b32 a42, %0\n\tv_accvgpr_write_b32 a43, %0\n\tv_accvgpr_write_b32 a44, %0\n\tv_accvgpr_write_b32 a45, %0\n\tv_accvgpr_write_b32 a46, %0\n\tv_accvgpr_write_b32 a47, %0\n\tv_accvgpr_write_b32 a48, %0\n\tv_accvgpr_write_b32 a49, %0\n\tv_accvgpr_write_b32 a50, %0\n\tv_accvgpr_write_b32 a51, %0\n\tv_accvgpr_write_b32 a52, %0\n\tv_accvgpr_write_b32 a53, %0\n\tv_accvgpr_write_b32 a54, %0\n\tv_accvgpr_write_b32 a55, %0\n\tv_accvgpr_write_b32 a56, %0\n\tv_accvgpr_write_b32 a57, %0\n\tv_accvgpr_write_b32 a58, %0\n\tv_accvgpr_write_b32 a59, %0\n\tv_accvgpr_write_b32 a60, %0\n\tv_accvgpr_write_b32 a61, %0\n\tv_accvgpr_write_b32 a62, %0\n\tv_accvgpr_write_b32 a63, %0\n\tv_accvgpr_write_b32 a64, %0\n\tv_accvgpr_write_b32 a65, %0\n\tv_accvgpr_write_b32 a66, %0\n\tv_accvgpr_write_b32 a67, %0\n\tv_accvgpr_write_b32 a68, %0\n\tv_accvgpr_write_b32 a69, %0\n\tv_accvgpr_write_b32 a70, %0\n\tv_accvgpr_write_b32 a71, %0\n\tv_accvgpr_write_b32 a72, %0\n\tv_accvgpr_write_b32 a73, %0\n\tv_accvgpr_write_b32 a74, %0\n\tv_accvgpr_write_b32 a75, %0\n\tv_accvgpr_write_b32 a76, %0\n\tv_accvgpr_write_b32 a77, %0\n\tv_accvgpr_write_b32 a78, %0\n\tv_accvgpr_write_b32 a79, %0\n\tv_accvgpr_write_b32 a80, %0\n\tv_accvgpr_write_b32 a81, %0\n\tv_accvgpr_write_b32 a82, %0\n\tv_accvgpr_write_b32 a83, %0\n\tv_accvgpr_write_b32 a84, %0\n\tv_accvgpr_write_b32 a85, %0\n\tv_accvgpr_write_b32 a86, %0\n\tv_accvgpr_write_b32 a87, %0\n\tv_accvgpr_write_b32 a88, %0\n\tv_accvgpr_write_b32 a89, %0\n\tv_accvgpr_write_b32 a90, %0\n\tv_accvgpr_write_b32 a91, %0\n\tv_accvgpr_write_b32 a92, %0\n\tv_accvgpr_write_b32 a93, %0\n\tv_accvgpr_write_b32 a94, %0\n\tv_accvgpr_write_b32 a95, %0\n\tv_accvgpr_write_b32 a96, %0\n\tv_accvgpr_write_b32 a97, %0\n\tv_accvgpr_write_b32 a98, %0\n\tv_accvgpr_write_b32 a99, %0\n\tv_accvgpr_write_b32 a100, %0\n\tv_accvgpr_write_b32 a101, %0\n\tv_accvgpr_write_b32 a102, %0\n\tv_accvgpr_write_b32 a103, %0\n\tv_accvgpr_write_b32 a104, %0\n\tv_accvgpr_write_b32 a105, %0\n\tv_accvgpr_write_b32 a106, %0\n\tv_accvgpr_write_b32 a107, %0\n\tv_accvgpr_write_b32 a108, %0\n\tv_accvgpr_write_b32 a109, %0\n\tv_accvgpr_write_b32 a110, %0\n\tv_accvgpr_write_b32 a111, %0\n\tv_accvgpr_write_b32 a112, %0\n\tv_accvgpr_write_b32 a113, %0\n\tv_accvgpr_write_b32 a114, %0\n\tv_accvgpr_write_b32 a115, %0\n\tv_accvgpr_write_b32 a116, %0\n\tv_accvgpr_write_b32 a117, %0\n\tv_accvgpr_write_b32 a118, %0\n\tv_accvgpr_write_b32 a119, %0\n\tv_accvgpr_write_b32 a120, %0\n\tv_accvgpr_write_b32 a121, %0\n\tv_accvgpr_write_b32 a122, %0\n\tv_accvgpr_write_b32 a123, %0\n\tv_accvgpr_write_b32 a124, %0\n\tv_accvgpr_write_b32 a125, %0\n\tv_accvgpr_write_b32 a126, %0\n\tv_accvgpr_write_b32 a127, %0\n\tv_accvgpr_write_b32 a128, %0\n\tv_accvgpr_write_b32 a129, %0\n\tv_accvgpr_write_b32 a130, %0\n\tv_accvgpr_write_b32 a131, %0\n\tv_accvgpr_write_b32 a132, %0\n\tv_accvgpr_write_b32 a133, %0\n\tv_accvgpr_write_b32 a134, %0\n\tv_accvgpr_write_b32 a135, %0\n\tv_accvgpr_write_b32 a136, %0\n\tv_accvgpr_write_b32 a137, %0\n\tv_accvgpr_write_b32 a138, %0\n\tv_accvgpr_write_b32 a139, %0\n\tv_accvgpr_write_b32 a140, %0\n\tv_accvgpr_write_b32 a141, %0\n\tv_accvgpr_write_b32 a142, %0\n\tv_accvgpr_write_b32 a143, %0\n\tv_accvgpr_write_b32 a144, %0\n\tv_accvgpr_write_b32 a145, %0\n\tv_accvgpr_write_b32 a146, %0\n\tv_accvgpr_write_b32 a147, %0\n\tv_accvgpr_write_b32 a148, %0\n\tv_accvgpr_write_b32 a149, %0\n\tv_accvgpr_write_b32 a150, %0\n\tv_accvgpr_write_b32 a151, %0\n\tv_accvgpr_write_b32 a152, %0\n\tv_accvgpr_write_b32 a153, %0\n\tv_accvgpr_write_b32 a154, %0\n\tv_accvgpr_write_b32 a155, %0\n\tv_accvgpr_write_b32 a156, %0\n\tv_accvgpr_write_b32 a157, %0\n\tv_accvgpr_write_b32 a158, %0\n\tv_accvgpr_write_b32 a159, %0\n\tv_accvgpr_write_b32 a160, %0\n\tv_accvgpr_write_b32 a161, %0\n\tv_accvgpr_write_b32 a162, %0\n\tv_accvgpr_write_b32 a163, %0\n\tv_accvgpr_write_b32 a164, %0\n\tv_accvgpr_write_b32 a165, %0\n\tv_accvgpr_write_b32 a166, %0\n\tv_accvgpr_write_b32 a167, %0\n\tv_accvgpr_write_b32 a168, %0\n\tv_accvgpr_write_b32 a169, %0\n\tv_accvgpr_write_b32 a170, %0\n\tv_accvgpr_write_b32 a171, %0\n\tv_accvgpr_write_b32 a172, %0\n\tv_accvgpr_write_b32 a173, %0\n\tv_accvgpr_write_b32 a174, %0\n\tv_accvgpr_write_b32 a175, %0\n\tv_accvgpr_write_b32 a176, %0\n\tv_accvgpr_write_b32 a177, %0\n\tv_accvgpr_write_b32 a178, %0\n\tv_accvgpr_write_b32 a179, %0\n\tv_accvgpr_write_b32 a180, %0\n\tv_accvgpr_write_b32 a181, %0\n\tv_accvgpr_write_b32 a182, %0\n\tv_accvgpr_write_b32 a183, %0\n\tv_accvgpr_write_b32 a184, %0\n\tv_accvgpr_write_b32 a185, %0\n\tv_accvgpr_write_b32 a186, %0\n\tv_accvgpr_write_b32 a187, %0\n\tv_accvgpr_write_b32 a188, %0\n\tv_accvgpr_write_b32 a189, %0\n\tv_accvgpr_write_b32 a190, %0\n\tv_accvgpr_write_b32 a191, %0\n\tv_accvgpr_write_b32 a192, %0\n\tv_accvgpr_write_b32 a193, %0\n\tv_accvgpr_write_b32 a194, %0\n\tv_accvgpr_write_b32 a195, %0\n\tv_accvgpr_write_b32 a196, %0\n\tv_accvgpr_write_b32 a197, %0\n\tv_accvgpr_write_b32 a198, %0\n\tv_accvgpr_write_b32 a199, %0\n\tv_accvgpr_write_b32 a200, %0\n\tv_accvgpr_write_b32 a201, %0\n\tv_accvgpr_write_b32 a202, %0\n\tv_accvgpr_write_b32 a203, %0\n\tv_accvgpr_write_b32 a204, %0\n\tv_accvgpr_write_b32 a205, %0\n\tv_accvgpr_write_b32 a206, %0\n\tv_accvgpr_write_b32 a207, %0\n\tv_accvgpr_write_b32 a208, %0\n\tv_accvgpr_write_b32 a209, %0\n\tv_accvgpr_write_b32 a210, %0\n\tv_accvgpr_write_b32 a211, %0\n\tv_accvgpr_write_b32 a212, %0\n\tv_accvgpr_write_b32 a213, %0\n\tv_accvgpr_write_b32 a214, %0\n\tv_accvgpr_write_b32 a215, %0\n\tv_accvgpr_write_b32 a216, %0\n\tv_accvgpr_write_b32 a217, %0\n\tv_accvgpr_write_b32 a218, %0\n\tv_accvgpr_write_b32 a219, %0\n\tv_accvgpr_write_b32 a220, %0\n\tv_accvgpr_write_b32 a221, %0\n\tv_accvgpr_write_b32 a222, %0\n\tv_accvgpr_write_b32 a223, %0\n\tv_accvgpr_write_b32 a224, %0\n\tv_accvgpr_write_b32 a225, %0\n\tv_accvgpr_write_b32 a226, %0\n\tv_accvgpr_write_b32 a227, %0\n\tv_accvgpr_write_b32 a228, %0\n\tv_accvgpr_write_b32 a229, %0\n\tv_accvgpr_write_b32 a230, %0\n\tv_accvgpr_write_b32 a231, %0\n\tv_accvgpr_write_b32 a232, %0\n\tv_accvgpr_write_b32 a233, %0\n\tv_accvgpr_write_b32 a234, %0\n\tv_accvgpr_write_b32 a235, %0\n\tv_accvgpr_write_b32 a236, %0\n\tv_accvgpr_write_b32 a237, %0\n\tv_accvgpr_write_b32 a238, %0\n\tv_accvgpr_write_b32 a239, %0\n\tv_accvgpr_write_b32 a240, %0\n\tv_accvgpr_write_b32 a241, %0\n\tv_accvgpr_write_b32 a242, %0\n\tv_accvgpr_write_b32 a243, %0\n\tv_accvgpr_write_b32 a244, %0\n\tv_accvgpr_write_b32 a245, %0\n\tv_accvgpr_write_b32 a246, %0\n\tv_accvgpr_write_b32 a247, %0\n\tv_accvgpr_write_b32 a248, %0\n\tv_accvgpr_write_b32 a249, %0\n\tv_accvgpr_write_b32 a250, %0\n\tv_accvgpr_write_b32 a251, %0\n\tv_accvgpr_write_b32 a252, %0\n\tv_accvgpr_write_b32 a253, %0\n\tv_accvgpr_write_b32 a254, %0\n\tv_accvgpr_write_b32 a255, %0\n\t"
                 "s_nop 4"
                 :: "v"(v) : "v1", "v2", "v3", "v4", "v5", "v6", "v7", "v8", "v9", "v10", "v11", "v12", "v13", "v14", "v15", "v16", "v17", "v18", "v19", "v20", "v21", "v22", "v23", "v24", "v25", "v26", "v27", "v28", "v29", "v30", "v31", "v32", "v33", "v34", "v35", "v36", "v37", "v38", "v39", "v40", "v41", "v42", "v43", "v44", "v45", "v46", "v47", "v48", "v49", "v50", "v51", "v52", "v53", "v54", "v55", "v56", "v57", "v58", "v59", "v60", "v61", "v62", "v63", "v64", "v65", "v66", "v67", "v68", "v69", "v70", "v71", "v72", "v73", "v74", "v75", "v76", "v77", "v78", "v79", "v80", "v81", "v82", "v83", "v84", "v85", "v86", "v87", "v88", "v89", "v90", "v91", "v92", "v93", "v94", "v95", "v96", "v97", "v98", "v99", "v100", "v101", "v102", "v103", "v104", "v105", "v106", "v107", "v108", "v109", "v110", "v111", "v112", "v113", "v114", "v115", "v116", "v117", "v118", "v119", "v120", "v121", "v122", "v123", "v124", "v125", "v126", "v127", "v128", "v129", "v130", "v131", "v132", "v133", "v134", "v135", "v136", "v137", "v138", "v139", "v140", "v141", "v142", "v143", "v144", "v145", "v146", "v147", "v148", "v149", "v150", "v151", "v152", "v153", "v154", "v155", "v156", "v157", "v158", "v159", "v160", "v161", "v162", "v163", "v164", "v165", "v166", "v167", "v168", "v169", "v170", "v171", "v172", "v173", "v174", "v175", "v176", "v177", "v178", "v179", "v180", "v181", "v182", "v183", "v184", "v185", "v186", "v187", "v188", "v189", "v190", "v191", "v192", "v193", "v194", "v195", "v196", "v197", "v198", "v199", "v200", "v201", "v202", "v203", "v204", "v205", "v206", "v207", "v208", "v209", "v210", "v211", "v212", "v213", "v214", "v215", "v216", "v217", "v218", "v219", "v220", "v221", "v222", "v223", "v224", "v225", "v226", "v227", "v228", "v229", "v230", "v231", "v232", "v233", "v234", "v235", "v236", "v237", "v238", "v239", "v240", "v241", "v242", "v243", "v244", "v245", "v246", "v247", "v248", "v249", "v250", "v251", "v252", "v253", "v254", "v255", "a0", "a1", "a2", "a3", "a4", "a5", "a6", "a7", "a8", "a9", "a10", "a11", "a12", "a13", "a14", "a15", "a16", "a17", "a18", "a19", "a20", "a21", "a22", "a23", "a24", "a25", "a26", "a27", "a28", "a29", "a30", "a31", "a32", "a33", "a34", "a35", "a36", "a37", "a38", "a39", "a40", "a41", "a42", "a43", "a44", "a45", "a46", "a47", "a48", "a49", "a50", "a51", "a52", "a53", "a54", "a55", "a56", "a57", "a58", "a59", "a60", "a61", "a62", "a63", "a64", "a65", "a66", "a67", "a68", "a69", "a70", "a71", "a72", "a73", "a74", "a75", "a76", "a77", "a78", "a79", "a80", "a81", "a82", "a83", "a84", "a85", "a86", "a87", "a88", "a89", "a90", "a91", "a92", "a93", "a94", "a95", "a96", "a97", "a98", "a99", "a100", "a101", "a102", "a103", "a104", "a105", "a106", "a107", "a108", "a109", "a110", "a111", "a112", "a113", "a114", "a115", "a116", "a117", "a118", "a119", "a120", "a121", "a122", "a123", "a124", "a125", "a126", "a127", "a128", "a129", "a130", "a131", "a132", "a133", "a134", "a135", "a136", "a137", "a138", "a139", "a140", "a141", "a142", "a143", "a144", "a145", "a146", "a147", "a148", "a149", "a150", "a151", "a152", "a153", "a154", "a155", "a156", "a157", "a158", "a159", "a160", "a161", "a162", "a163", "a164", "a165", "a166", "a167", "a168", "a169", "a170", "a171", "a172", "a173", "a174", "a175", "a176", "a177", "a178", "a179", "a180", "a181", "a182", "a183", "a184", "a185", "a186", "a187", "a188", "a189", "a190", "a191", "a192", "a193", "a194", "a195", "a196", "a197", "a198", "a199", "a200", "a201", "a202", "a203", "a204", "a205", "a206", "a207", "a208", "a209", "a210", "a211", "a212", "a213", "a214", "a215", "a216", "a217", "a218", "a219", "a220", "a221", "a222", "a223", "a224", "a225", "a226", "a227", "a228", "a229", "a230", "a231", "a232", "a233", "a234", "a235", "a236", "a237", "a238", "a239", "a240", "a241", "a242", "a243", "a244", "a245", "a246", "a247", "a248", "a249", "a250", "a251", "a252", "a253", "a254", "a255");
    if (pattern == 0x12345u && sink) sink[threadIdx.x] = v;
}

extern "C" int pk_poison_launch(unsigned pattern, int blocks, void* stream)
{
    pk_poison_kernel<<<blocks, 256, 0, (hipStream_t)stream>>>(pattern, nullptr);
    return (int)hipGetLastError();
}

// ---- candidate TRIGGERS: what must the kernel on the other stream execute for the victim's op_sel:[0,1] forms to go wrong?
typedef float f16v __attribute__((ext_vector_type(16)));
typedef float f4v __attribute__((ext_vector_type(4)));
typedef short s8v __attribute__((ext_vector_type(8)));

template <int KIND>
__global__ __launch_bounds__(256) void pk_trigger_kernel(int iters, float* __restrict__ sink)
{
    __shared__ __attribute__((aligned(16))) unsigned short lds[4096];
    const int t = threadIdx.x;
    float x = 1.0f + t * 1e-3f, y = 0.5f;
    f16v acc = {0}, accb = {0}, accc = {0}, accd = {0};
    f4v acc4 = {0};
    s8v a = {1, 2, 3, 4, 5, 6, 7, 8}, b = {8, 7, 6, 5, 4, 3, 2, 1};
    for (int i = t; i < 4096; i += 256) lds[i] = (unsigned short)i;
    __syncthreads();
    for (int it = 0; it < iters; ++it) {
        if (KIND == 0) {
            asm volatile("v_fma_f32 %0, %0, %1, %1\n\tv_fma_f32 %0, %0, %1, %1\n\tv_fma_f32 %0, %0, %1, %1\n\tv_fma_f32 %0, %0, %1, %1" : "+v"(x) : "v"(y));
        } else if (KIND == 1) {
            float r;
            asm volatile("v_accvgpr_write_b32 a0, %1\n\tv_accvgpr_write_b32 a1, %1\n\tv_accvgpr_write_b32 a2, %1\n\tv_accvgpr_write_b32 a3, %1\n\t"
                         "s_nop 2\n\tv_accvgpr_read_b32 %0, a0\n\tv_accvgpr_read_b32 %0, a1\n\tv_accvgpr_read_b32 %0, a2\n\tv_accvgpr_read_b32 %0, a3"
                         : "=v"(r) : "v"(x) : "a0", "a1", "a2", "a3");
            x = r;
        } else if (KIND == 2) {
            asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0\n\tv_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(acc) : "v"(a), "v"(b));
        } else if (KIND == 3) {
            asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0\n\tv_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(acc) : "v"(a), "v"(b));
        } else if (KIND == 4) {
            asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0\n\tv_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(acc4) : "v"(a), "v"(b));
        } else if (KIND == 6) {
            asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0\n\tv_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+a"(acc4) : "v"(a), "v"(b));
        } else if (KIND == 7) {
            typedef short s4v_ __attribute__((ext_vector_type(4)));
            s4v_ a4 = {1, 2, 3, 4}, b4 = {4, 3, 2, 1};
            asm volatile("v_mfma_f32_32x32x8bf16_1k %0, %1, %2, %0\n\tv_mfma_f32_32x32x8bf16_1k %0, %1, %2, %0" : "+v"(acc) : "v"(a4), "v"(b4));
        } else if (KIND == 8) {
            typedef short s4v_ __attribute__((ext_vector_type(4)));
            s4v_ a4 = {1, 2, 3, 4}, b4 = {4, 3, 2, 1};
            asm volatile("v_mfma_f32_16x16x16bf16_1k %0, %1, %2, %0\n\tv_mfma_f32_16x16x16bf16_1k %0, %1, %2, %0" : "+v"(acc4) : "v"(a4), "v"(b4));
        } else if (KIND == 9) {           // four INDEPENDENT 32x32x16 accumulators: the matrix pipe never waits for a result
            asm volatile("v_mfma_f32_32x32x16_bf16 %0, %4, %5, %0\n\tv_mfma_f32_32x32x16_bf16 %1, %4, %5, %1\n\t"
                         "v_mfma_f32_32x32x16_bf16 %2, %4, %5, %2\n\tv_mfma_f32_32x32x16_bf16 %3, %4, %5, %3"
                         : "+v"(acc), "+v"(accb), "+v"(accc), "+v"(accd) : "v"(a), "v"(b));
        } else if (KIND == 5) {
            typedef short s4v __attribute__((ext_vector_type(4)));
            s4v r = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s4v*)(lds + ((t * 4 + it * 64) & 4092)));
            a[0] += r[0]; a[1] += r[1];
        }
    }
    if (sink && (x == 12345.678f || acc[0] == 3.f || acc4[0] == 5.f || a[0] == 77)) sink[t] = x + acc[1] + accb[1] + accc[1] + accd[1] + acc4[1] + a[1];
}

extern "C" int pk_trigger_launch(int kind, int blocks, int iters, float* sink, void* stream)
{
    hipStream_t st = (hipStream_t)stream;
    switch (kind) {
    case 0: pk_trigger_kernel<0><<<blocks, 256, 0, st>>>(iters, sink); break;
    case 1: pk_trigger_kernel<1><<<blocks, 256, 0, st>>>(iters, sink); break;
    case 2: pk_trigger_kernel<2><<<blocks, 256, 0, st>>>(iters, sink); break;
    case 3: pk_trigger_kernel<3><<<blocks, 256, 0, st>>>(iters, sink); break;
    case 4: pk_trigger_kernel<4><<<blocks, 256, 0, st>>>(iters, sink); break;
    case 6: pk_trigger_kernel<6><<<blocks, 256, 0, st>>>(iters, sink); break;
    case 7: pk_trigger_kernel<7><<<blocks, 256, 0, st>>>(iters, sink); break;
    case 8: pk_trigger_kernel<8><<<blocks, 256, 0, st>>>(iters, sink); break;
    case 9: pk_trigger_kernel<9><<<blocks, 256, 0, st>>>(iters, sink); break;
    default: pk_trigger_kernel<5><<<blocks, 256, 0, st>>>(iters, sink); break;
    }
    return (int)hipGetLastError();
}
