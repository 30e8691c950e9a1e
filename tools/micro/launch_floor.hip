// Cost of a dependent kernel boundary on one HIP stream (gfx950): N launches of a one-wave kernel / of a grid that fills the
// chip with nothing to do, back to back from C++ (no Python in the loop).  Build + run on the GPU box:
//   hipcc --offload-arch=gfx950 -O2 tools/micro/launch_floor.hip -o /tmp/launch_floor && /tmp/launch_floor
#include <hip/hip_runtime.h>
#include <stdio.h>

__global__ void tiny(int* p) { if (threadIdx.x == 0 && blockIdx.x == 0) p[0] += 1; }
__global__ void empty(int* p) { if (p == nullptr) __builtin_trap(); }          // no memory access at all: the boundary alone

int main()
{
    int* d;
    hipMalloc(&d, 4096);
    hipMemset(d, 0, 4096);
    hipStream_t st;
    hipStreamCreate(&st);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    const int N = 5000;
    const int grids[3] = {1, 256, 2048};
    for (int gi = 0; gi < 3; ++gi) {
        for (int i = 0; i < 100; ++i) tiny<<<grids[gi], 256, 0, st>>>(d);
        hipStreamSynchronize(st);
        hipEventRecord(e0, st);
        for (int i = 0; i < N; ++i) tiny<<<grids[gi], 256, 0, st>>>(d);
        hipEventRecord(e1, st);
        hipEventSynchronize(e1);
        float ms = 0.f;
        hipEventElapsedTime(&ms, e0, e1);
        printf("grid %5d x 256 threads: %.2f us per dependent launch (one lane does a global read-modify-write)\n", grids[gi], ms * 1000.f / N);
        for (int i = 0; i < 100; ++i) empty<<<grids[gi], 256, 0, st>>>(d);
        hipStreamSynchronize(st);
        hipEventRecord(e0, st);
        for (int i = 0; i < N; ++i) empty<<<grids[gi], 256, 0, st>>>(d);
        hipEventRecord(e1, st);
        hipEventSynchronize(e1);
        hipEventElapsedTime(&ms, e0, e1);
        printf("grid %5d x 256 threads: %.2f us per dependent launch (empty kernel)\n", grids[gi], ms * 1000.f / N);
    }
    // The loops above are bounded by the HOST's enqueue rate (~2.5 us per launch from one thread), not by the GPU: the same kernels
    // replayed from a hipGraph (no host work between them) show the device-side cost of a dependent boundary.
    for (int gi = 0; gi < 3; ++gi) {
        hipGraph_t g;
        hipGraphExec_t ge;
        hipStreamBeginCapture(st, hipStreamCaptureModeGlobal);
        for (int i = 0; i < 1000; ++i) empty<<<grids[gi], 256, 0, st>>>(d);
        hipStreamEndCapture(st, &g);
        hipGraphInstantiate(&ge, g, nullptr, nullptr, 0);
        hipGraphLaunch(ge, st);
        hipStreamSynchronize(st);
        hipEventRecord(e0, st);
        for (int r = 0; r < 5; ++r) hipGraphLaunch(ge, st);
        hipEventRecord(e1, st);
        hipEventSynchronize(e1);
        float ms = 0.f;
        hipEventElapsedTime(&ms, e0, e1);
        printf("grid %5d x 256 threads: %.2f us per dependent launch (empty kernel, hipGraph replay of 1000)\n", grids[gi], ms * 1000.f / 5000);
        hipGraphExecDestroy(ge);
        hipGraphDestroy(g);
    }
    return 0;
}
