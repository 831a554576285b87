// LDS-pipe micro-benchmark for gfx950: cycles per wave-instruction of the operations a (pixel, light) pair pass would use.
// Build: hipcc --offload-arch=gfx950 -O3 lds_ops.hip -o lds_ops ; run on the GPU box.  One block of 256 threads per CU x 8.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#define ITER 512
template <int MODE>
__global__ __launch_bounds__(256) void k(float* out, const int* perm, int stride)
{
    __shared__ float4 buf[2048]; // 32 KB
    __shared__ float acc[1024];
    const int lane = threadIdx.x & 63;
    for (int i = threadIdx.x; i < 2048; i += 256) buf[i] = make_float4(i, 1, 2, 3);
    for (int i = threadIdx.x; i < 1024; i += 256) acc[i] = 0;
    __syncthreads();
    int p = perm[threadIdx.x];          // random lane 0..63 (per wave)
    int idx = (p * stride) & 2047;
    float s = 0.0f; float v = (float)threadIdx.x;
    long long t0 = clock64();
#pragma unroll 8
    for (int it = 0; it < ITER; it++) {
        if (MODE == 0) { v = __int_as_float(__builtin_amdgcn_ds_bpermute(p << 2, __float_as_int(v))); }                 // dependent bpermute chain
        if (MODE == 1) { s += __int_as_float(__builtin_amdgcn_ds_bpermute(p << 2, __float_as_int(v + it))); }          // independent bpermutes
        if (MODE == 2) { float4 r = buf[(it * 7) & 2047]; s += r.x + r.w; }                                            // broadcast b128
        if (MODE == 3) { float4 r = buf[(idx + it * 5) & 2047]; s += r.x + r.w; }                                      // gather b128 (stride given)
        if (MODE == 4) { __builtin_amdgcn_ds_faddf((__attribute__((address_space(3))) float*)&acc[(threadIdx.x + it) & 1023], v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT, false); } // conflict-free atomic add
        if (MODE == 5) { __builtin_amdgcn_ds_faddf((__attribute__((address_space(3))) float*)&acc[((p >> 2) + it) & 1023], v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT, false); } // 4-way same-address conflicts
        if (MODE == 6) { float r = ((float*)buf)[(idx * 4 + it) & 8191]; s += r; }                                     // gather b32
        if (MODE == 7) { acc[(threadIdx.x * 1 + it) & 1023] = v; }                                                     // b32 store
    }
    long long t1 = clock64();
    out[blockIdx.x * 256 + threadIdx.x] = s + v + acc[lane];
    if (threadIdx.x == 0 && blockIdx.x == 0) out[0] = (float)(t1 - t0) / ITER;
}
template <int MODE> void run(const char* name, float* d, int* dperm, int stride, int blocks)
{
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, d, dperm, stride);
    hipDeviceSynchronize();
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    hipEventRecord(a);
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, d, dperm, stride);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    float c; hipMemcpy(&c, d, 4, hipMemcpyDeviceToHost);
    // per-CU LDS throughput: blocks/256 CUs * 4 waves * ITER instr per block
    double instrPerCU = (double)blocks / 256.0 * 4 * ITER;
    printf("%-44s stride %3d blocks %5d: %8.1f clk/iter (wave view)  %7.3f ms  -> %6.2f ns per wave-instr per CU\n", name, stride, blocks, c, ms, ms * 1e6 / instrPerCU);
}
int main()
{
    float* d; hipMalloc(&d, 256 * 8192 * 4);
    int h[256]; uint32_t x = 12345; for (int i = 0; i < 256; i++) { x = x * 1664525u + 1013904223u; h[i] = (x >> 10) & 63; }
    int* dperm; hipMalloc(&dperm, sizeof h); hipMemcpy(dperm, h, sizeof h, hipMemcpyHostToDevice);
    for (int blocks : { 256, 256 * 4 }) {
        run<0>("ds_bpermute dependent", d, dperm, 1, blocks);
        run<1>("ds_bpermute independent", d, dperm, 1, blocks);
        run<2>("ds_read_b128 broadcast", d, dperm, 1, blocks);
        run<3>("ds_read_b128 gather stride 1 (16 B)", d, dperm, 1, blocks);
        run<3>("ds_read_b128 gather stride 5 (80 B)", d, dperm, 5, blocks);
        run<3>("ds_read_b128 gather stride 3 (48 B)", d, dperm, 3, blocks);
        run<6>("ds_read_b32 gather", d, dperm, 1, blocks);
        run<4>("ds_add_f32 conflict-free", d, dperm, 1, blocks);
        run<5>("ds_add_f32 4-way same address", d, dperm, 1, blocks);
        run<7>("ds_write_b32", d, dperm, 1, blocks);
    }
    return 0;
}
