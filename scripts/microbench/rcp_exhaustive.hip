// Is 1.0f / s -- the compiler's IEEE division: div_scale x 2, rcp, five fma, div_fmas, div_fixup -- the same function as one (or two) Newton
// steps on v_rcp_f32 followed by v_div_fixup_f32, for the values a square root can take?  Every one of the 2^32 bit patterns is compared; the
// mismatches are reported by exponent range.
//   hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -fno-fast-math -fhip-fp32-correctly-rounded-divide-sqrt -fno-gpu-flush-denormals-to-zero \
//         scripts/microbench/rcp_exhaustive.hip -o scripts/microbench/rcp_exhaustive.bin
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

__device__ __forceinline__ float rcp_v1(float s)
{
    const float r0 = __builtin_amdgcn_rcpf(s);
    const float e = fmaf(-s, r0, 1.0f);
    const float r1 = fmaf(e, r0, r0);
    return __builtin_amdgcn_div_fixupf(r1, s, 1.0f);
}
__device__ __forceinline__ float rcp_v2(float s)
{
    const float r0 = __builtin_amdgcn_rcpf(s);
    const float e = fmaf(-s, r0, 1.0f);
    const float r1 = fmaf(e, r0, r0);
    const float e2 = fmaf(-s, r1, 1.0f);
    const float r2 = fmaf(e2, r1, r1);
    return __builtin_amdgcn_div_fixupf(r2, s, 1.0f);
}

__global__ void k(unsigned long long* bad, uint32_t* lo, uint32_t* hi)
{
    const uint32_t stride = gridDim.x * blockDim.x;
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    for (unsigned long long n = 0; n < (1ull << 32) / stride; n++, i += stride) {
        const float s = __uint_as_float(i);
        const float a = 1.0f / s;
        const float b[2] = { rcp_v1(s), rcp_v2(s) };
        for (int v = 0; v < 2; v++) {
            const bool same = (__float_as_uint(a) == __float_as_uint(b[v])) || (a != a && b[v] != b[v]);
            if (!same) {
                atomicAdd(&bad[v], 1ull);
                const uint32_t m = i & 0x7FFFFFFFu; // magnitude bits
                atomicMin(&lo[v * 2 + (m < 0x3F800000u ? 0 : 1)], m);
                atomicMax(&hi[v * 2 + (m < 0x3F800000u ? 0 : 1)], m);
            }
        }
    }
}

int main()
{
    unsigned long long* bad; uint32_t *lo, *hi;
    hipMalloc(&bad, 16); hipMalloc(&lo, 16); hipMalloc(&hi, 16);
    hipMemset(bad, 0, 16); hipMemset(lo, 0xFF, 16); hipMemset(hi, 0, 16);
    k<<<4096, 256>>>(bad, lo, hi);
    unsigned long long hb[2]; uint32_t hl[4], hh[4];
    hipMemcpy(hb, bad, 16, hipMemcpyDeviceToHost); hipMemcpy(hl, lo, 16, hipMemcpyDeviceToHost); hipMemcpy(hh, hi, 16, hipMemcpyDeviceToHost);
    for (int v = 0; v < 2; v++)
        printf("variant %d: %llu mismatches; magnitudes below 1: bits [0x%08x, 0x%08x]; from 1 up: bits [0x%08x, 0x%08x]\n", v + 1, hb[v], hl[v * 2], hh[v * 2],
               hl[v * 2 + 1], hh[v * 2 + 1]);
    return 0;
}
