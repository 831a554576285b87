// How much wave-slot time is lost between workgroups?  32 400 blocks of 256 threads, 20 KB of LDS each (8 per CU), every wave spins for a
// fixed number of shader cycles.  Ideal duration = ceil(blocks / (8 x CUs)) x spin.  Build + run on the GPU box:
//   hipcc --offload-arch=gfx950 -O2 -o /tmp/dispatch_gap scripts/microbench/dispatch_gap.hip && /tmp/dispatch_gap
#include <hip/hip_runtime.h>
#include <stdio.h>
__global__ __launch_bounds__(256) void spin(unsigned long long cycles, int jitter, unsigned* sink)
{
    __shared__ unsigned lds[5000];
    lds[threadIdx.x] = threadIdx.x;
    __syncthreads();
    unsigned long long want = cycles;
    if (jitter) want = cycles / 2 + (unsigned long long)((blockIdx.x * 2654435761u) >> 16) % cycles; // 0.5 .. 1.5 x, mean 1 x
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    while (__builtin_amdgcn_s_memtime() - t0 < want) __builtin_amdgcn_s_sleep(8);
    if (lds[(threadIdx.x * 7) % 5000] == 0xFFFFFFFFu) sink[0] = 1;
}
int main()
{
    unsigned* d; hipMalloc(&d, 4);
    hipDeviceProp_t p; hipGetDeviceProperties(&p, 0);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    for (int jitter = 0; jitter < 2; jitter++)
        for (unsigned long long cyc : { 5000ull, 20000ull, 80000ull }) {
            for (int blocks : { 2048, 32400 }) {
                spin<<<blocks, 256>>>(cyc, jitter, d);
                hipEventRecord(a);
                for (int i = 0; i < 5; i++) spin<<<blocks, 256>>>(cyc, jitter, d);
                hipEventRecord(b); hipEventSynchronize(b);
                float ms; hipEventElapsedTime(&ms, a, b);
                const double rounds = (double)((blocks + 2047) / 2048);
                printf("jitter %d spin %6llu cycles, %5d blocks: %8.1f us per launch; ideal at %d MHz: %8.1f us (%.0f rounds)\n", jitter, cyc, blocks, ms * 200.0, p.clockRate / 1000,
                       rounds * cyc / (p.clockRate / 1000.0), rounds);
            }
        }
    return 0;
}
