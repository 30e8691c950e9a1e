// Issue rate of the two ways to bring operand bytes into a CU on gfx950, per CU and per wave:
//   (a) LDS-DMA   buffer_load_dwordx4 ... offen lds   (1 KiB per wave instruction, M0 = LDS address), and
//   (b) ordinary  buffer_load_dwordx4 v[..]            (1 KiB per wave instruction into 4 VGPRs) followed by ds_write_b128,
// with W = 1, 2, 4, 8 waves per CU (one workgroup per CU, 256 CUs), N instructions per wave in a row, waited for in groups of 8.
// Sources: "zero" = a descriptor of 0 records (every lane out of range: no memory access, the issue / address path alone),
// "l2" = a 64 KiB buffer per workgroup (L2 resident after the first pass), "hbm" = a 1 GiB buffer streamed once.
// Build + run on the GPU box:  hipcc --offload-arch=gfx950 -O2 tools/micro/lds_dma_rate.hip -o /tmp/lds_dma_rate && /tmp/lds_dma_rate
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>

typedef __attribute__((ext_vector_type(4))) unsigned int u4;

template <int MODE>   // 0: LDS-DMA, 1: load to VGPR + ds_write_b128
__global__ __launch_bounds__(512, 1) void rate_kernel(const unsigned char* src, unsigned bytes, unsigned stride_wg, int n, unsigned long long* out)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int nw = blockDim.x >> 6;
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)(src + (size_t)blockIdx.x * stride_wg), 0, bytes, 0x00020000);
    const unsigned lbase = (unsigned)(size_t)((__attribute__((address_space(3))) unsigned char*)smem) + wave * 8192;
    unsigned voff = (unsigned)(wave * 1024 + lane * 16);
    const unsigned step = (unsigned)(nw * 1024);
    const unsigned wrap = bytes ? bytes : 1u << 30;
    __syncthreads();
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    u4 acc = {0, 0, 0, 0};
    for (int i = 0; i < n; i += 8) {
        if (MODE == 0) {
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, 0 offen lds" :: "s"(lbase + u * 1024), "v"(voff), "s"(rs) : "memory");
                voff += step; if (voff >= wrap) voff -= wrap;
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        } else {
            u4 v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                asm volatile("buffer_load_dwordx4 %0, %1, %2, 0 offen" : "=v"(v[u]) : "v"(voff), "s"(rs) : "memory");
                voff += step; if (voff >= wrap) voff -= wrap;
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                *reinterpret_cast<u4*>(smem + wave * 8192 + u * 1024 + lane * 16) = v[u];
                acc += v[u];
            }
        }
    }
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if (lane == 0) out[blockIdx.x * 8 + wave] = t1 - t0;
    if (acc[0] == 0x12345678u && acc[1] == 1u) out[0] = 0;      // keep the VGPR path alive
}

int main()
{
    const size_t big = 1ull << 30;
    unsigned char* d;
    hipMalloc(&d, big);
    hipMemset(d, 1, big);
    unsigned long long* out;
    hipMalloc(&out, 256 * 8 * 8);
    unsigned long long h[256 * 8];
    hipFuncSetAttribute((const void*)rate_kernel<0>, hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
    hipFuncSetAttribute((const void*)rate_kernel<1>, hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    const char* srcn[3] = {"zero", "l2", "hbm"};
    const int n = 2048;
    printf("%-8s %-5s %5s | %12s %14s %14s\n", "path", "src", "waves", "cycles/instr", "cycles/instr", "B/clk/CU");
    printf("%-8s %-5s %5s | %12s %14s %14s\n", "", "", "/CU", "per wave", "per CU", "");
    for (int mode = 0; mode < 2; ++mode)
        for (int s = 0; s < 3; ++s)
            for (int w = 1; w <= 8; w *= 2) {
                unsigned bytes = s == 0 ? 0u : (s == 1 ? 65536u : (unsigned)(big / 256));
                unsigned stride = s == 2 ? (unsigned)(big / 256) : 65536u;
                float ms = 0.f;
                for (int rep = 0; rep < 2; ++rep) {
                    hipEventRecord(e0, 0);
                    if (mode == 0) rate_kernel<0><<<256, w * 64, 65536>>>(d, bytes, stride, n, out);
                    else           rate_kernel<1><<<256, w * 64, 65536>>>(d, bytes, stride, n, out);
                    hipEventRecord(e1, 0);
                    hipDeviceSynchronize();
                    hipEventElapsedTime(&ms, e0, e1);
                }
                hipMemcpy(h, out, sizeof(h), hipMemcpyDeviceToHost);
                double sum = 0; int cnt = 0;
                for (int b = 0; b < 256; ++b) for (int i = 0; i < w; ++i) { sum += (double)h[b * 8 + i]; ++cnt; }
                const double cyc = sum / cnt;                       // s_memtime ticks of a wave for n instructions
                printf("%-8s %-5s %5d | %12.1f %14.1f %14.1f   kernel %.1f us (%.0f ticks/us), %.2f TB/s\n", mode == 0 ? "lds-dma" : "vgpr+ds", srcn[s], w, cyc / n, cyc / n / w, 1024.0 * w * n / cyc,
                       ms * 1e3, cyc / (ms * 1e3), 256.0 * w * n * 1024.0 / (ms * 1e-3) / 1e12);
            }
    printf("(s_memtime ticks; compare with the shader clock: rocm-smi / 2.4 GHz nominal)\n");
    return 0;
}
