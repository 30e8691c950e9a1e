// What a cross-stream hand-over costs the PRODUCING stream (gfx950, ROCm 7.2): the backward hands every layer's gz from the main
// stream to the weight-gradient stream and takes the gz buffer back later (engine.py, gz ring) — two synchronisation points per
// layer on the main stream.  Variants, N iterations of  [main: A] -> hand-over -> [side: B] , [main: C] ... :
//   0  no hand-over at all (floor: A, C back to back on main; B never launched)
//   1  hipEventRecord(main) + hipStreamWaitEvent(side)                          (record only: what `wg_begin` does)
//   2  variant 1 + hipStreamWaitEvent(main, event recorded on side 3 iterations ago)   (what `gz_begin` adds)
//   3  events created with hipEventDisableTiming
//   4  hipStreamWriteValue32(main) + hipStreamWaitValue32(side) on signal memory, and the reverse for the take-back
//   5  variant 2 with the take-back wait issued right behind the record (one bubble instead of two?)
//   6  variant 2 with the take-back only every 4th layer (a ring deep enough to allow it)
// Build + run:  hipcc --offload-arch=gfx950 -O2 tools/micro/stream_sync.hip -o /tmp/stream_sync && /tmp/stream_sync
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>

__global__ void work(float* p, int n)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    float v = p[i % 4096];
    for (int k = 0; k < n; ++k) v = v * 1.0001f + 0.5f;
    if (v == 12345.f) p[0] = v;
}

int main()
{
    float* d;
    hipMalloc(&d, 1 << 20);
    hipMemset(d, 0, 1 << 20);
    hipStream_t s0, s1;
    hipStreamCreate(&s0);
    hipStreamCreate(&s1);
    hipEvent_t t0, t1;
    hipEventCreate(&t0);
    hipEventCreate(&t1);
    const int N = 400, RING = 4;
    uint32_t *sigA = nullptr, *sigB = nullptr;              // signal memory is handed out in 8-byte pieces
    hipError_t ea = hipExtMallocWithFlags((void**)&sigA, 8, hipMallocSignalMemory);
    hipError_t eb = hipExtMallocWithFlags((void**)&sigB, 8, hipMallocSignalMemory);
    const bool have_sig = ea == hipSuccess && eb == hipSuccess;
    if (!have_sig) printf("hipMallocSignalMemory: %s / %s\n", hipGetErrorString(ea), hipGetErrorString(eb));
    if (have_sig) { hipMemset(sigA, 0, 8); hipMemset(sigB, 0, 8); }
    for (int variant = 0; variant <= 6; ++variant) {
        if (variant == 4 && !have_sig) { printf("variant 4: no signal memory\n"); continue; }
        std::vector<hipEvent_t> ev_gz(1), ev_wg(RING);
        const unsigned flags = variant == 3 ? hipEventDisableTiming : hipEventDefault;
        hipEventCreateWithFlags(&ev_gz[0], flags);
        for (auto& e : ev_wg) hipEventCreateWithFlags(&e, flags);
        uint32_t seq = 0;
        for (int rep = 0; rep < 2; ++rep) {           // rep 0 warms up
            hipDeviceSynchronize();
            hipEventRecord(t0, s0);
            for (int i = 0; i < N; ++i) {
                if ((variant == 2 || variant == 3) && i >= RING) hipStreamWaitEvent(s0, ev_wg[i % RING], 0);
                if (variant == 6 && i >= RING && i % 4 == 0) hipStreamWaitEvent(s0, ev_wg[0], 0);       // take-back every 4th layer only
                if (variant == 4 && i >= RING) hipStreamWaitValue32(s0, sigB, seq - RING + 1, hipStreamWaitValueGte, 0xffffffffu);
                work<<<512, 256, 0, s0>>>(d, 2000);                  // A: ~20 us, fills the chip
                ++seq;
                if ((variant >= 1 && variant <= 3) || variant >= 5) { hipEventRecord(ev_gz[0], s0); hipStreamWaitEvent(s1, ev_gz[0], 0); }
                if (variant == 5 && i >= RING) hipStreamWaitEvent(s0, ev_wg[(i + 1) % RING], 0);        // take-back wait right behind the record
                if (variant == 4) { hipStreamWriteValue32(s0, sigA, seq, 0); hipStreamWaitValue32(s1, sigA, seq, hipStreamWaitValueGte, 0xffffffffu); }
                if (variant >= 1) work<<<128, 256, 0, s1>>>(d + 8192, 2000);     // B on the side stream
                if (variant == 2 || variant == 3 || variant == 5) hipEventRecord(ev_wg[i % RING], s1);
                if (variant == 6 && i % 4 == 3) hipEventRecord(ev_wg[0], s1);
                if (variant == 4) hipStreamWriteValue32(s1, sigB, seq, 0);
                work<<<512, 256, 0, s0>>>(d + 4096, 2000);           // C
            }
            hipEventRecord(t1, s0);
            hipDeviceSynchronize();
        }
        float ms = 0.f;
        hipEventElapsedTime(&ms, t0, t1);
        printf("variant %d: %.2f us per iteration on the main stream\n", variant, ms * 1000.f / N);
    }
    return 0;
}
