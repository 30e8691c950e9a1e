// Split-K epilogue experiment (gfx950): 512 blocks add 64 KB fp32 tiles into T tile buffers with global atomics.
//   mode 0: agent-scope atomicAdd, the splits of a tile spread over all XCDs (what conv_wgrad does)
//   mode 1: agent-scope atomicAdd, all splits of a tile on ONE XCD (tile % 8 == XCC_ID of the block)
//   mode 2: workgroup-scope atomics (no sc1: executed in the XCD's own L2), all splits of a tile on one XCD
//   mode 3: plain stores of the same bytes (the floor)
// Prints the block -> XCD mapping, GB/s of atomic payload and whether the sums are exact.
//   hipcc --offload-arch=gfx950 -O2 tools/micro/atomic_scope.hip -o /tmp/atomic_scope && /tmp/atomic_scope
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

constexpr int TILE = 16384;          // floats per tile (128 x 128)

__device__ __forceinline__ int xcc_id() { return __builtin_amdgcn_s_getreg(20 | (0 << 6) | (3 << 11)) & 15; }

__global__ void which_xcc(int* out) { if (threadIdx.x == 0) out[blockIdx.x] = xcc_id(); }

template <int MODE>
__global__ __launch_bounds__(256) void add_tiles(float* buf, int T, int reps, int* count)
{
    const int b = blockIdx.x;
    const int xcc = xcc_id();
    int tile;
    if (MODE == 0 || MODE == 3) tile = b % T;
    else tile = xcc + 8 * ((b >> 3) % (T / 8));          // a tile of this block's own XCD
    if (threadIdx.x == 0 && count) atomicAdd(&count[tile], reps);
    float* dst = buf + (size_t)tile * TILE;
    for (int r = 0; r < reps; ++r)
        for (int i = threadIdx.x; i < TILE; i += 256) {
            if (MODE == 3) dst[i] = 1.0f;
            else if (MODE == 2) __hip_atomic_fetch_add(dst + i, 1.0f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            else atomicAdd(dst + i, 1.0f);
        }
}

int main()
{
    const int T = 64, NB = 512, REPS = 8;
    float* buf; int *cnt, *xm;
    hipMalloc(&buf, sizeof(float) * T * TILE);
    hipMalloc(&cnt, sizeof(int) * T);
    hipMalloc(&xm, sizeof(int) * NB);
    which_xcc<<<NB, 64>>>(xm);
    std::vector<int> hx(NB);
    hipMemcpy(hx.data(), xm, sizeof(int) * NB, hipMemcpyDeviceToHost);
    int rr = 0; for (int i = 0; i < NB; ++i) rr += hx[i] == (i & 7);
    printf("blocks whose XCC_ID == blockIdx %% 8: %d of %d; first 16:", rr, NB);
    for (int i = 0; i < 16; ++i) printf(" %d", hx[i]);
    printf("\n");
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int mode = 0; mode < 4; ++mode) {
        float best = 1e9f; bool exact = true;
        for (int it = 0; it < 4; ++it) {
            hipMemset(buf, 0, sizeof(float) * T * TILE);
            hipMemset(cnt, 0, sizeof(int) * T);
            hipDeviceSynchronize();
            hipEventRecord(e0);
            if (mode == 0) add_tiles<0><<<NB, 256>>>(buf, T, REPS, cnt);
            if (mode == 1) add_tiles<1><<<NB, 256>>>(buf, T, REPS, cnt);
            if (mode == 2) add_tiles<2><<<NB, 256>>>(buf, T, REPS, cnt);
            if (mode == 3) add_tiles<3><<<NB, 256>>>(buf, T, REPS, cnt);
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            if (ms < best) best = ms;
            if (mode != 3) {
                std::vector<float> h((size_t)T * TILE); std::vector<int> hc(T);
                hipMemcpy(h.data(), buf, sizeof(float) * T * TILE, hipMemcpyDeviceToHost);
                hipMemcpy(hc.data(), cnt, sizeof(int) * T, hipMemcpyDeviceToHost);
                for (int t = 0; t < T && exact; ++t)
                    for (int i = 0; i < TILE; i += 97)
                        if (h[(size_t)t * TILE + i] != (float)hc[t]) { exact = false; printf("  mode %d: tile %d elem %d = %.1f, expected %d\n", mode, t, i, h[(size_t)t * TILE + i], hc[t]); break; }
            }
        }
        const double bytes = (double)NB * REPS * TILE * 4;
        printf("mode %d: %.3f ms  %.0f GB/s payload  %s\n", mode, best, bytes / best / 1e6, mode == 3 ? "" : (exact ? "sums exact" : "SUMS WRONG"));
    }
    return 0;
}
