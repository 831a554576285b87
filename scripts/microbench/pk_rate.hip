// Issue rate of v_pk_{fma,mul,add}_f32 against their scalar forms on gfx950: N independent chains per lane, 8 waves per SIMD, wall time per instruction.
// hipcc --offload-arch=gfx950 -O3 -o pk_rate pk_rate.hip && ./pk_rate
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float v2f __attribute__((ext_vector_type(2)));
#define ITER 4096
template <int MODE> __global__ __launch_bounds__(256) void k(float* out, float s)
{
    float a0 = threadIdx.x * 1e-3f, a1 = a0 + 1.f, a2 = a0 + 2.f, a3 = a0 + 3.f, a4 = a0 + 4.f, a5 = a0 + 5.f, a6 = a0 + 6.f, a7 = a0 + 7.f;
    v2f p0 = { a0, a1 }, p1 = { a2, a3 }, p2 = { a4, a5 }, p3 = { a6, a7 };
    const v2f ss = { s, s * 1.0001f };
    for (int i = 0; i < ITER; i++) {
        if (MODE == 0) { // 8 scalar fma
            asm volatile("v_fma_f32 %0, %0, %8, %0\n v_fma_f32 %1, %1, %8, %1\n v_fma_f32 %2, %2, %8, %2\n v_fma_f32 %3, %3, %8, %3\n"
                         "v_fma_f32 %4, %4, %8, %4\n v_fma_f32 %5, %5, %8, %5\n v_fma_f32 %6, %6, %8, %6\n v_fma_f32 %7, %7, %8, %7"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(s));
        } else if (MODE == 1) { // 4 packed fma (same flops as mode 0) x 2 = 8 instructions
            asm volatile("v_pk_fma_f32 %0, %0, %4, %0\n v_pk_fma_f32 %1, %1, %4, %1\n v_pk_fma_f32 %2, %2, %4, %2\n v_pk_fma_f32 %3, %3, %4, %3\n"
                         "v_pk_fma_f32 %0, %0, %4, %0\n v_pk_fma_f32 %1, %1, %4, %1\n v_pk_fma_f32 %2, %2, %4, %2\n v_pk_fma_f32 %3, %3, %4, %3"
                         : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3) : "v"(ss));
        } else if (MODE == 2) { // 8 packed mul
            asm volatile("v_pk_mul_f32 %0, %0, %4\n v_pk_mul_f32 %1, %1, %4\n v_pk_mul_f32 %2, %2, %4\n v_pk_mul_f32 %3, %3, %4\n"
                         "v_pk_mul_f32 %0, %0, %4\n v_pk_mul_f32 %1, %1, %4\n v_pk_mul_f32 %2, %2, %4\n v_pk_mul_f32 %3, %3, %4"
                         : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3) : "v"(ss));
        } else if (MODE == 3) { // 8 scalar mul
            asm volatile("v_mul_f32 %0, %0, %8\n v_mul_f32 %1, %1, %8\n v_mul_f32 %2, %2, %8\n v_mul_f32 %3, %3, %8\n"
                         "v_mul_f32 %4, %4, %8\n v_mul_f32 %5, %5, %8\n v_mul_f32 %6, %6, %8\n v_mul_f32 %7, %7, %8"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(s));
        } else if (MODE == 4) { // 8 packed add
            asm volatile("v_pk_add_f32 %0, %0, %4\n v_pk_add_f32 %1, %1, %4\n v_pk_add_f32 %2, %2, %4\n v_pk_add_f32 %3, %3, %4\n"
                         "v_pk_add_f32 %0, %0, %4\n v_pk_add_f32 %1, %1, %4\n v_pk_add_f32 %2, %2, %4\n v_pk_add_f32 %3, %3, %4"
                         : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3) : "v"(ss));
        } else if (MODE == 5) { // 8 scalar instructions, 4 vector + 4 scalar-ALU interleaved: do SALU instructions share the VALU's issue?
            asm volatile("v_mul_f32 %0, %0, %8\n s_add_u32 s20, s20, 1\n v_mul_f32 %1, %1, %8\n s_add_u32 s21, s21, 1\n v_mul_f32 %2, %2, %8\n s_add_u32 s22, s22, 1\n v_mul_f32 %3, %3, %8\n s_add_u32 s23, s23, 1"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(s) : "s20", "s21", "s22", "s23", "scc");
        } else if (MODE == 6) { // 4 vector only (reference for mode 5)
            asm volatile("v_mul_f32 %0, %0, %8\n v_mul_f32 %1, %1, %8\n v_mul_f32 %2, %2, %8\n v_mul_f32 %3, %3, %8"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(s));
        }
    }
    out[blockIdx.x * 256 + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + p0.x + p0.y + p1.x + p1.y + p2.x + p2.y + p3.x + p3.y;
}
template <int MODE> float run(float* d, int blocks)
{
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, d, 1.0000001f);
    hipDeviceSynchronize();
    hipEventRecord(a);
    for (int r = 0; r < 5; r++) hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, d, 1.0000001f);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    return ms / 5;
}
int main()
{
    const int blocks = 256 * 8; // 8 blocks of 4 waves per CU = 8 waves per SIMD
    float* d; hipMalloc(&d, blocks * 256 * 4);
    const char* names[] = { "8 x v_fma_f32", "8 x v_pk_fma_f32", "8 x v_pk_mul_f32", "8 x v_mul_f32", "8 x v_pk_add_f32", "4 x v_mul_f32 + 4 x s_add_u32", "4 x v_mul_f32" };
    float t[7] = { run<0>(d, blocks), run<1>(d, blocks), run<2>(d, blocks), run<3>(d, blocks), run<4>(d, blocks), run<5>(d, blocks), run<6>(d, blocks) };
    for (int m = 0; m < 7; m++) {
        // per SIMD: 8 waves x ITER x n instructions
        const int n = m == 6 ? 4 : 8;
        printf("%-34s %8.3f ms  -> %.2f cycles per wave-instruction at 2.4 GHz (per SIMD, 8 waves)\n", names[m], t[m], t[m] * 1e-3 * 2.4e9 / (8.0 * ITER * n));
    }
    return 0;
}
