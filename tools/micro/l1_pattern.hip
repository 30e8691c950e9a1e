// L1/TA throughput of 16-byte-per-lane buffer loads for different row shapes (timing experiment):
//   pattern R: a wave-load covers (64*16/R) rows of R contiguous bytes, rows `pitch` bytes apart.
// Each block walks its own small window (L1/L2 resident).  Build: hipcc -O3 --offload-arch=gfx950 l1_pattern.hip -o l1_pattern
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4_t;

template <int ROWB>
__global__ __launch_bounds__(256) void k(const unsigned char* base, unsigned bytes, int pitch, int iters, unsigned* sink, int win)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    constexpr int LPR = ROWB / 16;                    // lanes per row
    const int row = lane / LPR, sub = lane % LPR;
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)base, 0, bytes, 0x00020000);
    const unsigned blk_off = (unsigned)(blockIdx.x % 64) * (unsigned)win;
    unsigned voff = blk_off + (unsigned)((wave * (64 / LPR) + row) * pitch + sub * 16);
    u32x4_t acc = {0, 0, 0, 0};
    for (int i = 0; i < iters; ++i) {
        unsigned so = (unsigned)((i & 7) * ROWB);      // walk 8 column blocks (L1 resident window)
        u32x4_t a = __builtin_amdgcn_raw_buffer_load_b128(rs, voff, so, 0);
        u32x4_t b = __builtin_amdgcn_raw_buffer_load_b128(rs, voff + 64 * pitch, so, 0);
        u32x4_t c = __builtin_amdgcn_raw_buffer_load_b128(rs, voff + 128 * pitch, so, 0);
        u32x4_t d = __builtin_amdgcn_raw_buffer_load_b128(rs, voff + 192 * pitch, so, 0);
        acc += a + b + c + d;
    }
    if (acc[0] == 0x12345678u) sink[0] = acc[1];
}

int main(int argc, char** argv)
{
    const int iters = 2000;
    const unsigned bytes = 256u << 20;
    unsigned char* buf; unsigned* sink;
    hipMalloc(&buf, bytes); hipMemset(buf, 1, bytes); hipMalloc(&sink, 64);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int blocks_per_cu = 1; blocks_per_cu <= 4; blocks_per_cu *= 2) {
        for (int pat = 0; pat < 4; ++pat) {
            const int rowb = pat == 0 ? 64 : (pat == 1 ? 128 : (pat == 2 ? 256 : 64));
            const int pitch = pat == 3 ? 64 : 2304;        // pat 3: fully contiguous 64-B rows
            const int win = 256 * 2304 + 4096;
            dim3 grid(256 * blocks_per_cu), block(256);
            for (int rep = 0; rep < 2; ++rep) {
                hipEventRecord(e0);
                if (rowb == 64) k<64><<<grid, block>>>(buf, bytes, pitch, iters, sink, win);
                else if (rowb == 128) k<128><<<grid, block>>>(buf, bytes, pitch, iters, sink, win);
                else k<256><<<grid, block>>>(buf, bytes, pitch, iters, sink, win);
                hipEventRecord(e1); hipEventSynchronize(e1);
            }
            float ms; hipEventElapsedTime(&ms, e0, e1);
            double by = (double)grid.x * 4 * iters * 4 * 1024.0;
            printf("blocks/CU %d rowbytes %3d pitch %4d: %7.3f ms  %6.2f TB/s  = %5.1f B/clk/CU @2.4GHz\n", blocks_per_cu, rowb, pitch, ms, by / ms / 1e9,
                   by / ms / 1e-3 / 256 / 2.4e9);
        }
    }
    return 0;
}
