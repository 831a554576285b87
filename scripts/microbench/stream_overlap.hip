// Do kernels of two HIP streams (or of two branches of a hipGraph) execute at the same time on this stack?  Each kernel is 64 blocks (a quarter
// of the CUs, one block each) that spin for a fixed time: two of them fit side by side eight times over.  Build + run on the GPU box:
//   hipcc --offload-arch=gfx950 -O2 -o /tmp/stream_overlap scripts/microbench/stream_overlap.hip && /tmp/stream_overlap
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <chrono>
__global__ __launch_bounds__(256) void spin(unsigned long long ticks, unsigned* sink)
{
    const unsigned long long t0 = wall_clock64(); // 100 MHz
    while (wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(8);
    if (ticks == 0xFFFFFFFFull) sink[0] = 1;
}
static double now_us() { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main()
{
    unsigned* d; hipMalloc(&d, 4);
    hipStream_t s1, s2; hipStreamCreateWithFlags(&s1, hipStreamNonBlocking); hipStreamCreateWithFlags(&s2, hipStreamNonBlocking);
    const int N = 20;
    for (unsigned long long ticks : { 500ull, 2000ull, 5000ull }) { // 5, 20, 50 us
        const double ideal = N * ticks / 100.0;
        // one stream
        spin<<<64, 256, 0, s1>>>(ticks, d); hipStreamSynchronize(s1);
        double t0 = now_us();
        for (int i = 0; i < N; i++) spin<<<64, 256, 0, s1>>>(ticks, d);
        hipStreamSynchronize(s1);
        const double one = now_us() - t0;
        // two streams, N each
        spin<<<64, 256, 0, s2>>>(ticks, d); hipStreamSynchronize(s2);
        t0 = now_us();
        for (int i = 0; i < N; i++) { spin<<<64, 256, 0, s1>>>(ticks, d); spin<<<64, 256, 0, s2>>>(ticks, d); }
        hipStreamSynchronize(s1); hipStreamSynchronize(s2);
        const double two = now_us() - t0;
        // a graph with two parallel chains of N kernels each
        hipGraph_t g; hipGraphCreate(&g, 0);
        hipGraphNode_t prev[2] = { nullptr, nullptr };
        for (int i = 0; i < N; i++)
            for (int b = 0; b < 2; b++) {
                hipKernelNodeParams kp = {};
                void* args[2] = { &ticks, &d };
                kp.func = (void*)spin; kp.gridDim = dim3(64); kp.blockDim = dim3(256); kp.kernelParams = args;
                hipGraphNode_t n;
                hipGraphAddKernelNode(&n, g, prev[b] ? &prev[b] : nullptr, prev[b] ? 1 : 0, &kp);
                prev[b] = n;
            }
        hipGraphExec_t ge; hipGraphInstantiate(&ge, g, nullptr, nullptr, 0);
        hipGraphLaunch(ge, s1); hipStreamSynchronize(s1);
        t0 = now_us();
        hipGraphLaunch(ge, s1); hipStreamSynchronize(s1);
        const double gr = now_us() - t0;
        // a graph with ONE chain of 2N kernels
        hipGraph_t g1; hipGraphCreate(&g1, 0);
        hipGraphNode_t p = nullptr;
        for (int i = 0; i < 2 * N; i++) {
            hipKernelNodeParams kp = {};
            void* args[2] = { &ticks, &d };
            kp.func = (void*)spin; kp.gridDim = dim3(64); kp.blockDim = dim3(256); kp.kernelParams = args;
            hipGraphNode_t n;
            hipGraphAddKernelNode(&n, g1, p ? &p : nullptr, p ? 1 : 0, &kp);
            p = n;
        }
        hipGraphExec_t ge1; hipGraphInstantiate(&ge1, g1, nullptr, nullptr, 0);
        hipGraphLaunch(ge1, s1); hipStreamSynchronize(s1);
        t0 = now_us();
        hipGraphLaunch(ge1, s1); hipStreamSynchronize(s1);
        const double gr1 = now_us() - t0;
        printf("spin %5.0f us x %d: one stream %7.1f us (ideal %6.1f) | two streams, %d each: %7.1f us | graph, two chains of %d: %7.1f us | graph, one chain of %d: %7.1f us\n",
               ticks / 100.0, N, one, ideal, N, two, N, gr, 2 * N, gr1);
    }
    return 0;
}
