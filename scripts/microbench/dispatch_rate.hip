// How fast does the chip START workgroups?  A grid of G blocks of B threads with L bytes of LDS whose waves each live for ~T shader cycles and do
// nothing else: if the launch lasts longer than G * T / (resident blocks), the dispatcher -- not the work -- is the bound.
// (round 4: k1_tile_cull = 8 800 four-wave blocks of ~7 000 cycles lasts 27 us where its work fills 14: is it the dispatch rate?)
// build: hipcc --offload-arch=gfx950 -O2 -o dispatch_rate.bin dispatch_rate.hip ; run: ./dispatch_rate.bin
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>

template <int LDS>
__global__ void k_spin(unsigned long long cycles, unsigned* sink)
{
    __shared__ unsigned lds[LDS / 4 > 0 ? LDS / 4 : 1];
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    if (threadIdx.x == 0) lds[0] = (unsigned)t0;
    while (__builtin_amdgcn_s_memtime() - t0 < cycles) __builtin_amdgcn_s_sleep(4);
    if (sink && lds[0] == 0xdeadbeefu) *sink = 1;
}

template <int LDS>
static float run(int grid, int block, unsigned long long cycles)
{
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    for (int i = 0; i < 3; i++) hipLaunchKernelGGL(k_spin<LDS>, dim3(grid), dim3(block), 0, 0, cycles, (unsigned*)nullptr);
    hipEventRecord(a);
    const int reps = 20;
    for (int i = 0; i < reps; i++) hipLaunchKernelGGL(k_spin<LDS>, dim3(grid), dim3(block), 0, 0, cycles, (unsigned*)nullptr);
    hipEventRecord(b);
    hipEventSynchronize(b);
    float ms = 0;
    hipEventElapsedTime(&ms, a, b);
    return ms / reps * 1000.0f;
}

int main()
{
    printf("%8s %6s %6s %8s %10s %12s\n", "grid", "block", "lds", "cycles", "us/launch", "ns/workgroup");
    const unsigned long long lives[] = { 0ull, 3000ull, 7000ull };
    for (unsigned long long T : lives)
        for (int block : { 64, 128, 256, 512 })
            for (int grid : { 2048, 8800, 35200 }) {
                if ((long long)grid * block > 35200ll * 256) continue;
                const float us0 = run<0>(grid, block, T), us16 = run<16384>(grid, block, T);
                printf("%8d %6d %6d %8llu %10.2f %12.2f\n", grid, block, 0, T, us0, us0 * 1000.0f / grid);
                printf("%8d %6d %6d %8llu %10.2f %12.2f\n", grid, block, 16384, T, us16, us16 * 1000.0f / grid);
            }
    return 0;
}
