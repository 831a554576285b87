// A probe, not product code: a latency-bound stand-in for the cull chain's blocks, launched beside the shade to see what the block SHAPE costs.
// k_filler: every thread walks `iters` dependent loads through an L2-resident table with `alu` dependent multiply-adds behind each (a cull wave's life: load,
// test, load); `ldsBytes` of dynamic LDS are claimed (and touched) so that the block occupies what a cull block occupies.
//   hipcc --offload-arch=gfx950 -O3 -shared -fPIC -o sailor_amd/csrc/ab/libfiller.so scripts/experiments/filler.hip
#include <hip/hip_runtime.h>
#include <stdint.h>

__global__ void k_filler(const uint32_t* __restrict__ table, uint32_t mask, int iters, int alu, uint32_t* sink)
{
    extern __shared__ uint32_t lds[];
    uint32_t v = blockIdx.x * blockDim.x + threadIdx.x;
    float f = (float)v;
    if (threadIdx.x == 0) lds[0] = v;
    for (int i = 0; i < iters; i++) {
        v = table[(v * 2654435761u + (uint32_t)i) & mask];
        for (int k = 0; k < alu; k++) f = f * 1.0001f + 0.5f;
    }
    if (f == 123.0f && v == 7u) sink[0] = v + lds[0];
}

extern "C" int filler_launch(void* stream, int blocks, int threads, int ldsBytes, int iters, int alu, const uint32_t* table, uint32_t mask, uint32_t* sink)
{
    hipLaunchKernelGGL(k_filler, dim3((unsigned)blocks), dim3((unsigned)threads), (size_t)ldsBytes, (hipStream_t)stream, table, mask, iters, alu, sink);
    return (int)hipGetLastError();
}
