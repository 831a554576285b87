// Is "v_sqrt_f32 + the two one-ulp corrections" (the compiler's correctly rounded sqrtf WITHOUT its input scaling for x < 2^-96 and WITHOUT its
// class check for 0 / inf) the same function as sqrtf on this chip?  Every one of the 2^32 bit patterns is compared.
//   hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -fno-fast-math -fhip-fp32-correctly-rounded-divide-sqrt -fno-gpu-flush-denormals-to-zero \
//         scripts/microbench/sqrt_exhaustive.hip -o /tmp/sqrt_exhaustive && /tmp/sqrt_exhaustive
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

__device__ __forceinline__ float sqrt_cr(float x)
{
    const float s = __builtin_amdgcn_sqrtf(x);
    const float sd = __uint_as_float(__float_as_uint(s) - 1u), su = __uint_as_float(__float_as_uint(s) + 1u);
    const float rd = fmaf(-sd, s, x), ru = fmaf(-su, s, x);
    float r = (0.0f >= rd) ? sd : s;
    r = (0.0f < ru) ? su : r;
    return r;
}

__global__ void k(unsigned long long* bad, uint32_t* firstBad, unsigned long long* badTiny)
{
    const uint32_t stride = gridDim.x * blockDim.x;
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    for (unsigned long long n = 0; n < (1ull << 32) / stride; n++, i += stride) {
        const float x = __uint_as_float(i);
        const float a = sqrtf(x), b = sqrt_cr(x);
        const bool same = (__float_as_uint(a) == __float_as_uint(b)) || (a != a && b != b);
        if (!same) {
            const unsigned long long k = atomicAdd(bad, 1ull);
            if (k < 16) firstBad[k] = i;
            if ((i & 0x7FFFFFFFu) < 0x0F800000u) atomicAdd(badTiny, 1ull);
        }
    }
}

int main()
{
    unsigned long long *bad, *badTiny; uint32_t* first;
    hipMalloc(&bad, 8); hipMalloc(&badTiny, 8); hipMalloc(&first, 64);
    hipMemset(bad, 0, 8); hipMemset(badTiny, 0, 8); hipMemset(first, 0, 64);
    k<<<4096, 256>>>(bad, first, badTiny);
    unsigned long long hb = 0, ht = 0; uint32_t hf[16];
    hipMemcpy(&hb, bad, 8, hipMemcpyDeviceToHost); hipMemcpy(&ht, badTiny, 8, hipMemcpyDeviceToHost); hipMemcpy(hf, first, 64, hipMemcpyDeviceToHost);
    printf("mismatches over all 2^32 inputs: %llu (of which |x| < 2^-96: %llu)\n", hb, ht);
    for (int i = 0; i < 16 && (unsigned long long)i < hb; i++) printf("  x bits 0x%08x\n", hf[i]);
    return 0;
}
