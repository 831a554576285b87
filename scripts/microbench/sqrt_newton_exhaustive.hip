// Candidates for a cheaper correctly rounded sqrtf on [2^-96, inf): one Newton step on the hardware's v_sqrt_f32 / v_rsq_f32.  All 2^32 inputs.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
__device__ __forceinline__ float sa(float x) { const float s0 = __builtin_amdgcn_sqrtf(x); const float e = fmaf(-s0, s0, x); const float h = 0.5f * __builtin_amdgcn_rsqf(x); return fmaf(e, h, s0); }
__device__ __forceinline__ float sb(float x) { const float y = __builtin_amdgcn_rsqf(x); const float s0 = x * y, h = 0.5f * y; const float e = fmaf(-s0, s0, x); return fmaf(e, h, s0); }
__device__ __forceinline__ float rc(float x) { const float y = __builtin_amdgcn_rsqf(x); const float s0 = x * y, h = 0.5f * y; const float e = fmaf(-s0, s0, x); const float s = fmaf(e, h, s0); const float e2 = fmaf(-s, y, 1.0f); return fmaf(e2, y, y); }
__global__ void k(unsigned long long* bad, uint32_t* first)
{
    const uint32_t stride = gridDim.x * blockDim.x;
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    for (unsigned long long n = 0; n < (1ull << 32) / stride; n++, i += stride) {
        if (i < 0x0F800000u || i >= 0x7F800000u) continue; // only [2^-96, inf)
        const float x = __uint_as_float(i);
        const float a = sqrtf(x);
        const float b[3] = { sa(x), sb(x), __builtin_amdgcn_sqrtf(x) };
        if (__float_as_uint(1.0f / a) != __float_as_uint(rc(x))) atomicAdd(&bad[3], 1ull);
        for (int v = 0; v < 3; v++)
            if (__float_as_uint(a) != __float_as_uint(b[v])) { if (atomicAdd(&bad[v], 1ull) < 4) first[v * 4 + (bad[v] & 3)] = i; }
    }
}
int main()
{
    unsigned long long* bad; uint32_t* first;
    (void)hipMalloc(&bad, 32); (void)hipMalloc(&first, 48); (void)hipMemset(bad, 0, 32); (void)hipMemset(first, 0, 48);
    k<<<4096, 256>>>(bad, first);
    unsigned long long hb[4]; uint32_t hf[12];
    (void)hipMemcpy(hb, bad, 32, hipMemcpyDeviceToHost); (void)hipMemcpy(hf, first, 48, hipMemcpyDeviceToHost);
    printf("sqrt + rsq Newton: %llu mismatches (e.g. 0x%08x 0x%08x); rsq Newton: %llu mismatches (e.g. 0x%08x 0x%08x)\n", hb[0], hf[0], hf[1], hb[1], hf[4], hf[5]);
    printf("1 / sqrt from the same rsq (rsq Newton for the root, one more step on rsq for its reciprocal): %llu mismatches against 1.0f / sqrtf(x)\n", hb[3]);
    printf("(the bare v_sqrt_f32, as a check that the comparison bites: %llu mismatches)\n", hb[2]);
    return 0;
}
