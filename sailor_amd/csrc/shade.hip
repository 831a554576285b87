// K2 + K3 -- PBR shade over per-tile light lists with cascaded-shadow sampling, for gfx950.
//
// Replaces the fragment work of Content/Shaders/Standard.shader (main :377-439, CalculateLighting :259-341) and
// Lighting.glsl (BRDF :39-76, CSM :168-284) for the draws of RenderSceneNode::Process
// (FrameGraph/RenderSceneNode.cpp:109), as compute over a surface buffer (ambient term optional, 8f rank 2).
//
// Shape: one 256-thread block per 16x16 tile -- the unit the light list is defined on -- so the tile's <=128
// light records are fetched from HBM once, derived per-light constants (normalised spot axis, cone width, reach
// thresholds) are computed once by 128 lanes, and the records sit in LDS (128 x 80 B = 10 KB).  Each wave owns one 8x8
// quadrant of the tile (eight 128-byte row segments per float4 plane); radiance goes out as one float4 per lane.  The
// tile list is conservative (sphere vs tile frustum) while the surface is a thin sheet inside that frustum: on the 4K /
// 65 536-light frame only ~3 % of the (pixel, light) pairs are actually lit, and the kernel is bound by vector-instruction
// issue.  So a wave (1) tests its list one LANE per LIGHT against the quadrant's bounding sphere (a point light's reach, a spot light's cone), (2) tests the survivors one
// LANE per PIXEL with a 6-instruction conservative reach test + the facing test and queues the (pixel, light) pairs that
// pass, (3) runs the exact falloff and the BRDF one LANE per PAIR on full waves.  Details at k2_shade_body.  The ambient /
// IBL term of Standard.shader (:343-372) and the BRDF look-up table it samples are the last two sections of this file.
//
// Numerics: the BRDF is tolerance-checked (1e-4 relative), so its well-conditioned parts use v_rcp and explicit
// FMAs.  Two places are NOT well-conditioned and are evaluated in the oracle's exact fp32 order instead:
//   * the shadow factor is a step function of its inputs (cascade select, PCF compares, EVSM's exp(40 z) - moment):
//     K3 uses true divisions and the fixed exp algorithm, bit for bit;
//   * NdfGGX's denominator cosLh^2 (a^2 - 1) + 1 cancels catastrophically at the specular peak of smooth surfaces
//     (a^2 down to 6e-6), amplifying 1 ulp of the half vector ~10^4 times: the chain viewDir -> Lo -> Lh -> cosLh ->
//     denominator uses normalize(v) = v * (1 / sqrt(dot(v, v))) with IEEE sqrt and divide (the Vulkan spec's definition
//     of GLSL.std.450 Normalize) and unfused dot products; likewise distance / bounds.x of the radius window and the
//     spot cone's (theta - cutOff.y) / epsilon, which cancel at the edge of a light's reach.
// The translation unit is compiled with -ffp-contract=off; every fused multiply-add below is written explicitly.
#include "common.h"
#include "shade_body.h"

// make EXTRA=-DSHADE_PROF + scripts/shade_prof.py: start / end of every block of the band kernel on the 100 MHz constant clock, its XCD and hardware id
#ifdef SHADE_PROF
#define SPROF_BLOCKS 262144
__device__ unsigned long long g_shadeProf[SPROF_BLOCKS][4];
extern "C" __attribute__((visibility("default"))) int sailor_hip_debug_read_shade_prof(void* dst, size_t bytes)
{
    return (int)hipMemcpyFromSymbol(dst, HIP_SYMBOL(g_shadeProf), bytes);
}
__device__ __forceinline__ void sprof_mark(const int i)
{
    const unsigned long long t = __builtin_amdgcn_s_memrealtime();
    const unsigned long long h = ((unsigned long long)(__builtin_amdgcn_s_getreg((20 << 0) | (0 << 6) | (3 << 11)) & 0xF) << 32) | __builtin_amdgcn_s_getreg((4 << 0) | (0 << 6) | (31 << 11));
    if (threadIdx.x == 0 && blockIdx.x < SPROF_BLOCKS) { g_shadeProf[blockIdx.x][i] = t; if (i == 0) g_shadeProf[blockIdx.x][2] = h; }
}
#define SPROF_T(i) sprof_mark(i);
// (per WAVE as well: [w] = start, [4 + w] = end, [8 + w] = XCD << 32 | HW_ID of wave w -- scripts/shade_wave_prof.py: how long a finished wave's slot stays empty)
__device__ unsigned long long g_shadeWaveProf[65536][12];
extern "C" __attribute__((visibility("default"))) int sailor_hip_debug_read_shade_wave_prof(void* dst, size_t bytes)
{
    return (int)hipMemcpyFromSymbol(dst, HIP_SYMBOL(g_shadeWaveProf), bytes);
}
template <bool ON> __device__ __forceinline__ void sprof_mark_grid(const int i)   // (the per-tile grid's kernels: by linear block index)
{
    if constexpr (ON) {
        const unsigned long long t = __builtin_amdgcn_s_memrealtime();
        const unsigned long long h = ((unsigned long long)(__builtin_amdgcn_s_getreg((20 << 0) | (0 << 6) | (3 << 11)) & 0xF) << 32) | __builtin_amdgcn_s_getreg((4 << 0) | (0 << 6) | (31 << 11));
        const uint32_t id = blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z);
        if (threadIdx.x == 0 && id < (uint32_t)SPROF_BLOCKS) { g_shadeProf[id][i] = t; if (i == 0) g_shadeProf[id][2] = h; }
        if ((threadIdx.x & 63u) == 0u && id < 65536u) {
            const uint32_t w = threadIdx.x >> 6;
            g_shadeWaveProf[id][(i == 0 ? 0u : 4u) + w] = t;
            if (i == 0) g_shadeWaveProf[id][8u + w] = h;
        }
    }
}
#define SPROF_TG(i, ON) sprof_mark_grid<(ON)>(i);
#else
#define SPROF_T(i)
#define SPROF_TG(i, ON)
#endif

// Entry points: without shadow lookups in it the kernel fits 64 VGPRs, and the register allocator is told to stay there
// (8 waves per SIMD); with the 16-tap PCF inlined it does not, and forcing it would spill.  The ambient term adds a third.
#define SHADE_ENTRY(NAME, ATTR, CSM, IBL, PREP, TL)                                                                                                \
    __global__ __launch_bounds__(256) ATTR void NAME(ShadeArgs A, CsmArgs C, IblArgs I, const float4* __restrict__ surface, size_t planeStride,    \
                                                     const SailorLightShaderData* __restrict__ lights, const SailorLightsGrid* __restrict__ grid, \
                                                     const uint32_t* __restrict__ culled, float4* __restrict__ radiance)                          \
    {                                                                                                                                              \
        __shared__ ShadeLds lds;                                                                                                                   \
        SPROF_TG(0, !(IBL))                                                                                                               \
        k2_shade_body<CSM, IBL, ROLE_TILE, PREP, TL>(lds, A, C, I, surface, planeStride, lights, grid, culled, radiance);                          \
        SPROF_TG(3, !(IBL))                                                                                                               \
    }
#define FORCE_64_VGPRS __attribute__((amdgpu_waves_per_eu(8, 8)))
#ifndef SHADE_HALF_DEFAULT
#define SHADE_HALF_DEFAULT false
#endif
// (the K3 kernels: 64 registers like the others since the shadow look-ups run before the view / material terms -- "K3 first" in shade_body.h; it
// was 80 = six waves per SIMD.  The pin matters: unpinned, the prepared twins come out at 116-134.)
#define CSM_PIN __attribute__((amdgpu_waves_per_eu(8, 8)))
#define FIVE_WAVES __attribute__((amdgpu_waves_per_eu(5, 5))) // (K3 + ambient: 88-91 registers by itself; its prepared twin 134 unpinned)
// Four kernels (plain, K3, ambient, K3 + ambient) x the lights as 112-byte records or as sailor_hip_prepare_lights' staged records (_p: `lights` = the
// staged array) x the lists in the reference's lightsGrid / culledLights or in the cull's per-tile slots (..t: `grid` = tileNum, `culled` = the slots)
#define SHADE_ENTRIES(SUFFIX, PREP, TL)                                          \
    SHADE_ENTRY(k2_shade##SUFFIX, FORCE_64_VGPRS, false, false, PREP, TL)         \
    SHADE_ENTRY(k2_shade_csm##SUFFIX, CSM_PIN, true, false, PREP, TL)             \
    SHADE_ENTRY(k2_shade_ibl##SUFFIX, , false, true, PREP, TL)                    \
    SHADE_ENTRY(k2_shade_csm_ibl##SUFFIX, FIVE_WAVES, true, true, PREP, TL)
SHADE_ENTRIES(, false, false)
SHADE_ENTRIES(_p, true, false)
SHADE_ENTRIES(_t, false, true)
SHADE_ENTRIES(_pt, true, true)

// The same kernels as TWO-wave blocks, one per half tile (round 6; ShadeLdsT<2>): 128 threads, 8.7 KB of LDS -- sixteen blocks per CU.
#define SHADE_ENTRY_H(NAME, ATTR, CSM, PREP, TL)                                                                                                   \
    __global__ __launch_bounds__(128) ATTR void NAME(ShadeArgs A, CsmArgs C, IblArgs I, const float4* __restrict__ surface, size_t planeStride,    \
                                                     const SailorLightShaderData* __restrict__ lights, const SailorLightsGrid* __restrict__ grid, \
                                                     const uint32_t* __restrict__ culled, float4* __restrict__ radiance)                          \
    {                                                                                                                                              \
        __shared__ ShadeLdsT<2> lds;                                                                                                               \
        SPROF_TG(0, true)                                                                                                                          \
        k2_shade_body<CSM, false, ROLE_TILE, PREP, TL, 2>(lds, A, C, I, surface, planeStride, lights, grid, culled, radiance);                     \
        SPROF_TG(3, true)                                                                                                                          \
    }
// (only the forms that keep 64 registers without scratch: prepared lights -- the second staging round is then a copy -- and no shadow maps; the K3 body inside
// the round loop spills nine to eleven registers, the in-kernel staging three)
SHADE_ENTRY_H(k2_shade_h_p, FORCE_64_VGPRS, false, true, false)
SHADE_ENTRY_H(k2_shade_h_pt, FORCE_64_VGPRS, false, true, true)

template <bool PREP, bool TL, bool CSM>
__device__ __forceinline__ void k2_shade_band_body(ShadeLds& lds, const ShadeArgs& A, const CsmArgs& C, int bandTiles, const float4* __restrict__ surface, size_t planeStride,
                                                   const SailorLightShaderData* __restrict__ lights, const SailorLightsGrid* __restrict__ grid,
                                                   const uint32_t* __restrict__ culled, float4* __restrict__ radiance)
{
    if (blockIdx.x >= (unsigned)SPLIT_BLOCKS) {
        const int t = (int)blockIdx.x - SPLIT_BLOCKS, tr = t / A.Tx;
        // (with shadow maps the band's tile rows are taken from the far end: tile rows count up with the distance on a ground plane, the far rows lie in the
        // PCF cascades -- 16 taps against the EVSM cascade's four -- and a band's launch ends with its longest blocks unless they start first; 1-3 % of a
        // C4 band's step)
        const int ty = CSM ? A.bandTileRows - 1 - tr : tr;
        SPROF_T(0)
        k2_shade_body<CSM, false, ROLE_BAND_TILE, PREP, TL>(lds, A, C, IblArgs(), surface, planeStride, lights, grid, culled, radiance, t - tr * A.Tx, ty, 0);
        SPROF_T(3)
        return;
    }
    SPROF_T(0)
    // The band's long tiles, found by the split blocks themselves (round 5): the (tile, quadrant) items are numbered 4 * tile + quadrant and split block b
    // takes the items b, b + SPLIT_BLOCKS, b + 2 * SPLIT_BLOCKS, ...; it reads the list lengths of 64 of them at a time (A.order: one byte per tile, written
    // by k1_tile_cull; one load, the bytes 512 tiles apart), and the ballot of the long ones is what it shades.  Round 4 had k1_tile_cull append the long
    // tiles to a list through two device-scope counters: every long tile one atomic on one word -- +3 us on that kernel on an eighth of the 4K frame, +12.7
    // on half of it.  (Neighbouring tiles' items go to different blocks, a cluster is spread over all of them; all four waves of the block run the
    // same scalar-uniform scan.)
    const uint32_t items = 4u * (uint32_t)bandTiles;
    const uint32_t lane = (uint32_t)__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));
    for (uint32_t base = blockIdx.x; base < items; base += (uint32_t)SPLIT_BLOCKS * 64u) {
        const uint32_t it = base + lane * (uint32_t)SPLIT_BLOCKS;
        const uint32_t num = it < items ? (uint32_t)A.order[it >> 2] : 0u;
        unsigned long long todo = __ballot(num >= (uint32_t)A.splitMin);
        while (todo != 0ull) {
            const uint32_t l = (uint32_t)__builtin_ctzll(todo);
            todo &= todo - 1ull;
            const uint32_t item = base + l * (uint32_t)SPLIT_BLOCKS, tile = item >> 2;
            const uint32_t ty = tile / (uint32_t)A.Tx;
            k2_shade_body<CSM, false, ROLE_BAND_SPLIT, PREP, TL>(lds, A, C, IblArgs(), surface, planeStride, lights, grid, culled, radiance, (int)(tile - ty * (uint32_t)A.Tx), (int)ty,
                                                               (int)(item & 3u));
            __syncthreads(); // the LDS arrays are reused by the block's next tile
        }
    }
    SPROF_T(3)
}

#define SHADE_BAND_ENTRY(NAME, PREP, TL, CSM, WAVES)                                                                                                                   \
    __global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(WAVES, WAVES)))                                                                                   \
    void NAME(ShadeArgs A, CsmArgs C, int bandTiles, const float4* __restrict__ surface, size_t planeStride, const SailorLightShaderData* __restrict__ lights,        \
              const SailorLightsGrid* __restrict__ grid, const uint32_t* __restrict__ culled, float4* __restrict__ radiance)                                         \
    {                                                                                                                                                                \
        __shared__ ShadeLds lds;                                                                                                                                     \
        k2_shade_band_body<PREP, TL, CSM>(lds, A, C, bandTiles, surface, planeStride, lights, grid, culled, radiance);                                                   \
    }
SHADE_BAND_ENTRY(k2_shade_band, false, false, false, 8)
SHADE_BAND_ENTRY(k2_shade_band_p, true, false, false, 8)
SHADE_BAND_ENTRY(k2_shade_band_t, false, true, false, 8)
SHADE_BAND_ENTRY(k2_shade_band_pt, true, true, false, 8)
// (round 4: with shadow maps too -- a band of C4 was its longest tile: 63 us for 25 us of block-slot time, scripts/shade_prof_csm.py.  Two copies of the
// K3 body do not fit 64 registers without scratch (4-5 spilled registers even with round 5's window look-ups).  Seven waves per SIMD = 72 registers, none
// spilled, 94 scalar registers = seven blocks per CU by that file too; round 4's six (80) had more than the body needs: a quarter band of C4 94 -> 90 us,
// an eighth -- held to six blocks by the wave-slot reserve anyway -- unchanged: gpurun r05y, profiles/r05/ab_band_csm_waves.txt)
#ifndef BAND_CSM_WAVES
#define BAND_CSM_WAVES 7
#endif
SHADE_BAND_ENTRY(k2_shade_band_csm, false, false, true, BAND_CSM_WAVES)
SHADE_BAND_ENTRY(k2_shade_band_csm_p, true, false, true, BAND_CSM_WAVES)
SHADE_BAND_ENTRY(k2_shade_band_csm_t, false, true, true, BAND_CSM_WAVES)
SHADE_BAND_ENTRY(k2_shade_band_csm_pt, true, true, true, BAND_CSM_WAVES)

// ---- sailor_hip_prepare_lights: the per-light half of the path, once per UPLOADED light instead of once per frame and list slot ----
// One lane per light of [first, first + count): the cull's 20-byte view of it -- (worldPosition, bounds.x) as a float4 and the type, two dense
// arrays that k01_prepare streams instead of the 112-byte records (7.3 MB for 1.3 at 65 536 lights) -- and the shade's staged record
// (stage_light_record: what every shade block used to derive for every slot of its list).
struct PreparedLayout { size_t offPosRadius, offType, offStaged, total; };
static PreparedLayout prepared_layout(int32_t capacity)
{
    const size_t n = (size_t)(capacity > 0 ? capacity : 1);
    PreparedLayout L;
    size_t o = 0;
    L.offPosRadius = o; o = align_up(o + n * 16, 256);
    L.offType = o; o = align_up(o + n * 4, 256);
    L.offStaged = o; o = align_up(o + n * (LREC * 16), 256);
    L.total = o;
    return L;
}

__global__ __launch_bounds__(256) void k_prepare_lights(const SailorLightShaderData* __restrict__ lights, int first, int count, float4* __restrict__ posRadius,
                                                        uint32_t* __restrict__ type, float4* __restrict__ staged)
{
    const int i = (int)(blockIdx.x * 256 + threadIdx.x);
    if (i >= count) return;
    const int j = first + i;
    const float4* L = reinterpret_cast<const float4*>(lights + j);
    const float4 q0 = L[0], q1 = L[1], q2 = L[2], q3 = L[3], q4 = L[4], q5 = L[5], q6 = L[6];
    posRadius[j] = make_float4(q1.x, q1.y, q1.z, q6.x);
    type[j] = __float_as_uint(q0.x);
    float4 o0, o1, o2, o3, o4;
    stage_light_record(q0, q1, q2, q3, q4, q5, q6, o0, o1, o2, o3, o4);
    float4* o = staged + (size_t)j * LREC;
    o[0] = o0; o[1] = o1; o[2] = o2; o[3] = o3; o[4] = o4;
}

extern "C" size_t sailor_hip_prepared_lights_size(int32_t lightsCapacity)
{
    if (lightsCapacity < 0) return 0;
    return prepared_layout(lightsCapacity).total;
}

extern "C" int sailor_hip_prepare_lights(SailorHipContext* ctx, const SailorLightShaderData* dLights, int32_t firstLight, int32_t count, int32_t lightsCapacity,
                                         void* dPrepared, size_t preparedBytes)
{
    if (!ctx || !dPrepared || firstLight < 0 || count < 0 || lightsCapacity < 0 || (count > 0 && !dLights)) return SAILOR_HIP_ERR_INVALID_ARGUMENT;
    if ((int64_t)firstLight + count > (int64_t)lightsCapacity) return SAILOR_HIP_ERR_INVALID_ARGUMENT;
    if (((uintptr_t)dLights & 15) || ((uintptr_t)dPrepared & 15)) return SAILOR_HIP_ERR_INVALID_ARGUMENT;
    const PreparedLayout L = prepared_layout(lightsCapacity);
    if (preparedBytes < L.total) return SAILOR_HIP_ERR_WORKSPACE_TOO_SMALL;
    if (count == 0) return SAILOR_HIP_OK;
    SAILOR_TRY_HIP(ctx, hipSetDevice(ctx->device));
    char* base = (char*)dPrepared;
    sailor_launch(ctx, k_prepare_lights, dim3((unsigned)((count + 255) / 256)), dim3(256), dLights, firstLight, count,
                  (float4*)(base + L.offPosRadius), (uint32_t*)(base + L.offType), (float4*)(base + L.offStaged));
    SAILOR_CHECK_LAUNCH(ctx, "k_prepare_lights");
    return SAILOR_HIP_OK;
}

// the two arrays the cull reads (light_cull.hip: sailor_hip_light_cull_prepared)
extern "C" int sailor_hip_prepared_lights_views(int32_t lightsCapacity, const void* dPrepared, const void** outPosRadius, const void** outType, const void** outStaged)
{
    if (!dPrepared || lightsCapacity < 0) return SAILOR_HIP_ERR_INVALID_ARGUMENT;
    const PreparedLayout L = prepared_layout(lightsCapacity);
    const char* base = (const char*)dPrepared;
    if (outPosRadius) *outPosRadius = base + L.offPosRadius;
    if (outType) *outType = base + L.offType;
    if (outStaged) *outStaged = base + L.offStaged;
    return SAILOR_HIP_OK;
}

// ---- ComputeBrdfLut.shader:26-71 (Lighting.glsl:27-37 SampleGGX, :65-70 GeometrySchlickGGX_IBL, Math.glsl:285-293) ----
// One lane per texel, 1 024 samples each, the sums in sample order (as the shader's loop).
__global__ __launch_bounds__(256) void k_brdf_lut(float2* __restrict__ lut, int W, int H)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= W * H) return;
    const int gx = i % W, gy = i / W;
    const float TwoPI = 6.283185307179586f;
    const float InvNumSamples = 1.0f / 1024.0f;
    float cosLo = (float)gx / (float)W;
    const float roughness = (float)gy / (float)H;
    cosLo = fmaxf(cosLo, 0.001f);
    const float Lox = sqrtf(1.0f - cosLo * cosLo), Loz = cosLo;
    const float alpha = roughness * roughness, k = (roughness * roughness) / 2.0f;
    const float g1Lo = cosLo / (cosLo * (1.0f - k) + k);
    float DFG1 = 0.0f, DFG2 = 0.0f;
    for (uint32_t n = 0; n < 1024u; n++) {
        const float u1 = (float)n * InvNumSamples;
        const float u2 = (float)__brev(n) * 2.3283064365386963e-10f; // RadicalInverse_VdC == 32-bit reversal
        const float cosTheta = sqrtf((1.0f - u2) / (1.0f + (alpha * alpha - 1.0f) * u2));
        const float sinTheta = sqrtf(1.0f - cosTheta * cosTheta);
        const float phi = TwoPI * u1;
        const float Lhx = sinTheta * cosf(phi), Lhz = cosTheta; // Lh.y only enters Li.y, which is not used
        const float d = (Lox * Lhx + 0.0f * (sinTheta * sinf(phi))) + Loz * Lhz;
        const float cosLi = 2.0f * d * Lhz - Loz, cosLoLh = fmaxf(d, 0.0f);
        if (cosLi > 0.0f) {
            const float G = (cosLi / (cosLi * (1.0f - k) + k)) * g1Lo;
            const float Gv = G * cosLoLh / (Lhz * cosLo);
            const float x = 1.0f - cosLoLh, x2 = x * x, Fc = x2 * x2 * x;
            DFG1 += (1.0f - Fc) * Gv;
            DFG2 += Fc * Gv;
        }
    }
    lut[i] = make_float2(DFG1 * InvNumSamples, DFG2 * InvNumSamples);
}

// ---- sailor_hip_self_check_exact_math: sqrt_exact / rcp_of_sqrt against sqrtf / the IEEE division, every float of their ranges ----
__global__ __launch_bounds__(256) void k_self_check_exact_math(unsigned long long* __restrict__ bad)
{
    const uint32_t stride = gridDim.x * blockDim.x;
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    uint32_t badSqrt = 0u, badRcp = 0u;
    for (uint32_t n = 0; n < (uint32_t)((1ull << 32) / stride); n++, i += stride) {
        const float x = __uint_as_float(i);
        if (i >= 0x0F800000u && i < 0x7F800000u) { // [2^-96, inf): the fast form proper (everything else goes to sqrtf by construction)
            const float y = __builtin_amdgcn_rsqf(x);
            const float s0 = x * y, h = 0.5f * y;
            badSqrt += __float_as_uint(fmaf(fmaf(-s0, s0, x), h, s0)) != __float_as_uint(sqrtf(x)) ? 1u : 0u;
        }
        const uint32_t m = i & 0x7FFFFFFFu;
        if (m >= 0x00800000u && m <= 0x7E800000u) // 2^-126 <= |x| <= 2^126
            badRcp += __float_as_uint(rcp_of_sqrt(x)) != __float_as_uint(1.0f / x) ? 1u : 0u;
    }
    // div_stored (the quotient from a staged reciprocal) cannot be compared exhaustively: 2^10 pseudo-random pairs per thread (2^30 in all), the
    // divisor's exponent in [-40, 40] as the stager guarantees, the numerator's in [-48, 64] (a distance), both signs
    uint32_t badDiv = 0u;
    uint64_t rs = 0x9E3779B97F4A7C15ull * (uint64_t)(blockIdx.x * blockDim.x + threadIdx.x + 1u);
    for (int n = 0; n < 1024; n++) {
        rs ^= rs << 13; rs ^= rs >> 7; rs ^= rs << 17;
        const uint32_t ra = (uint32_t)(rs >> 11), rb = (uint32_t)(rs >> 33) ^ (uint32_t)rs;
        rs ^= rs << 13; rs ^= rs >> 7; rs ^= rs << 17;
        const uint32_t re = (uint32_t)(rs >> 20);
        const float a = __uint_as_float((ra & 0x807FFFFFu) | ((79u + (re & 0xFFFu) % 113u) << 23));
        const float b = __uint_as_float((rb & 0x807FFFFFu) | ((87u + ((re >> 12) & 0xFFFu) % 81u) << 23));
        badDiv += __float_as_uint(div_stored(a, b, rcp_of_sqrt(b))) != __float_as_uint(a / b) ? 1u : 0u;
    }
    if (badSqrt) atomicAdd(&bad[0], (unsigned long long)badSqrt);
    if (badRcp) atomicAdd(&bad[1], (unsigned long long)badRcp);
    if (badDiv) atomicAdd(&bad[2], (unsigned long long)badDiv);
}

extern "C" int sailor_hip_self_check_exact_math(SailorHipContext* ctx, void* dScratch, uint64_t mismatches[3])
{
    if (!ctx || !dScratch || !mismatches || ((uintptr_t)dScratch & 7)) return SAILOR_HIP_ERR_INVALID_ARGUMENT;
    SAILOR_TRY_HIP(ctx, hipSetDevice(ctx->device));
    SAILOR_TRY_HIP(ctx, hipMemsetAsync(dScratch, 0, 24, ctx->stream));
    hipLaunchKernelGGL(k_self_check_exact_math, dim3(4096), dim3(256), 0, ctx->stream, (unsigned long long*)dScratch);
    SAILOR_CHECK_LAUNCH(ctx, "k_self_check_exact_math");
    SAILOR_TRY_HIP(ctx, hipMemcpyAsync(mismatches, dScratch, 24, hipMemcpyDeviceToHost, ctx->stream));
    SAILOR_TRY_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return SAILOR_HIP_OK;
}

extern "C" int sailor_hip_compute_brdf_lut(SailorHipContext* ctx, float* dLut, int32_t width, int32_t height)
{
    if (!ctx || !dLut || width <= 0 || height <= 0 || ((uintptr_t)dLut & 7)) return SAILOR_HIP_ERR_INVALID_ARGUMENT;
    SAILOR_TRY_HIP(ctx, hipSetDevice(ctx->device)); // a host thread may drive several contexts
    hipLaunchKernelGGL(k_brdf_lut, dim3((unsigned)(((size_t)width * height + 255) / 256)), dim3(256), 0, ctx->stream, (float2*)dLut, width, height);
    SAILOR_CHECK_LAUNCH(ctx, "k_brdf_lut");
    return SAILOR_HIP_OK;
}

// dTileNum == null: (dLightsGrid, dCulledLights) in the reference's layout; else dCulledLights = the cull's per-tile slots and dLightsGrid is not read
static int shade_impl(SailorHipContext* ctx, const SailorUboFrameData* frame, const float* dSurface, size_t surfacePlaneStride,
                      const SailorLightShaderData* dLights, int32_t lightsNum,
                      const SailorLightsGrid* dLightsGrid, const uint32_t* dCulledLights, const uint32_t* dTileNum,
                      const SailorCsmDesc* csm, const SailorIblDesc* ibl, float* dRadiance, const SailorBand* band, const uint32_t* dTileOrder,
                      const void* dPreparedLights, int32_t preparedCapacity)
{
    if (!ctx || !frame || !dSurface || (!dLightsGrid && !dTileNum) || !dCulledLights || !dRadiance) return SAILOR_HIP_ERR_INVALID_ARGUMENT;
    if (lightsNum < 0 || (lightsNum > 0 && !dLights && !dPreparedLights)) return SAILOR_HIP_ERR_INVALID_ARGUMENT;
    if (dPreparedLights && (preparedCapacity < lightsNum || ((uintptr_t)dPreparedLights & 15))) return SAILOR_HIP_ERR_INVALID_ARGUMENT;
    const int W = frame->viewportSize[0], H = frame->viewportSize[1];
    if (W <= 0 || H <= 0) return SAILOR_HIP_ERR_INVALID_ARGUMENT;
    SailorBand whole;
    if (!band) { sailor_hip_band_whole_frame(W, H, &whole); band = &whole; }
    const int Ty = (H - 1) / TILE + 1;
    if (!sailor_hip_band_is_valid(W, H, band)) return SAILOR_HIP_ERR_INVALID_ARGUMENT; // tile rows AND their framebuffer rows, as the cull checks them
    if (surfacePlaneStride < (size_t)band->fbRowCount * W) return SAILOR_HIP_ERR_INVALID_ARGUMENT;
    if (((uintptr_t)dSurface & 15) || ((uintptr_t)dRadiance & 15) || (!dPreparedLights && ((uintptr_t)dLights & 15))) return SAILOR_HIP_ERR_INVALID_ARGUMENT;

    SAILOR_TRY_HIP(ctx, hipSetDevice(ctx->device));
    ShadeArgs A;
    memcpy(A.view.m, frame->view, 64);
    A.camX = frame->cameraPosition[0]; A.camY = frame->cameraPosition[1]; A.camZ = frame->cameraPosition[2];
    A.zFar = frame->cameraZNearZFar[1];
    A.vpW = W; A.vpH = H; A.W = W; A.H = H;
    // Standard.shader:413-420: numTiles.x + padding.x == LightCullingNode's numTiles.x
    A.Tx = W / TILE + ((W % TILE) ? 1 : 0);
    A.tileRow0 = band->tileRowBegin;
    A.fbRow0 = band->fbRowBegin;
    A.fbRows = band->fbRowCount;
    A.lightsNum = lightsNum;
    // (A / B knobs, read once; a value outside its range is ignored and says so: [1, 128] -- a list has at most 128 entries)
    static const int splitMinRaw = [] { const char* e = getenv("SAILOR_SPLIT_MIN"); return e ? atoi(e) : -1; }();
    static const bool splitMinSet = getenv("SAILOR_SPLIT_MIN") != nullptr;
    const int splitMinEnv = (splitMinRaw >= 1 && splitMinRaw <= KEEP) ? splitMinRaw : -1;
    if (splitMinSet && splitMinEnv < 0) ctx->lastError = "SAILOR_SPLIT_MIN outside [1, 128]: ignored";
    A.order = reinterpret_cast<const uint8_t*>(dTileOrder); // (the band's list lengths as bytes: sailor_hip_light_cull_tile_order)
    const int bandTiles = (band->tileRowEnd - band->tileRowBegin) * A.Tx;
    if (bandTiles == 0) return SAILOR_HIP_OK;
    A.splitMin = splitMinEnv > 0 ? splitMinEnv : (bandTiles <= SPLIT_SMALL_TILES ? SPLIT_MIN_SMALL : SPLIT_MIN_LARGE); // (SAILOR_SPLIT_MIN=<n>: A / B)

    CsmArgs C;
    memset(&C, 0, sizeof C);
    bool hasCsm = false;
    if (csm) {
        for (int k = 0; k < SAILOR_NUM_CSM_CASCADES; k++) {
            memcpy(C.lightsMatrices[k].m, csm->lightsMatrices[k], 64);
            C.maps[k] = csm->maps[k];
            C.width[k] = csm->width[k]; C.height[k] = csm->height[k]; C.format[k] = csm->format[k];
            C.texelW[k] = 1.0f / (float)csm->width[k]; C.texelH[k] = 1.0f / (float)csm->height[k];
            if (csm->maps[k]) {
                if (csm->width[k] <= 0 || csm->height[k] <= 0 || csm->format[k] < 0 || csm->format[k] > 2) return SAILOR_HIP_ERR_INVALID_ARGUMENT;
                hasCsm = true;
            }
        }
    }
    A.bandTileRows = band->tileRowEnd - band->tileRowBegin;
    // (XCD + 8 x tile within the piece, piece, tile row): see k2_shade_body
    const dim3 grid(8u * (unsigned)((A.Tx + 8 * SHADE_XCD_PIECES - 1) / (8 * SHADE_XCD_PIECES)), (unsigned)SHADE_XCD_PIECES, (unsigned)A.bandTileRows);
    IblArgs I {};
    if (ibl) {
        if (!ibl->irradiance || !ibl->env || !ibl->brdfLut || ibl->irrSize <= 0 || ibl->envSize <= 0 || ibl->envLevels <= 0 || ibl->envLevels > 16 ||
            ibl->lutW <= 0 || ibl->lutH <= 0)
            return SAILOR_HIP_ERR_INVALID_ARGUMENT;
        if (((uintptr_t)ibl->irradiance & 15) || ((uintptr_t)ibl->env & 15) || ((uintptr_t)ibl->brdfLut & 7)) return SAILOR_HIP_ERR_INVALID_ARGUMENT;
        I.irradiance = (const float4*)ibl->irradiance; I.env = (const float4*)ibl->env; I.brdfLut = (const float2*)ibl->brdfLut; I.ao = ibl->ao;
        I.irrSize = ibl->irrSize; I.envSize = ibl->envSize; I.envLevels = ibl->envLevels; I.lutW = ibl->lutW; I.lutH = ibl->lutH;
    }
    const float4* S = (const float4*)dSurface;
    float4* Rd = (float4*)dRadiance;
    // with prepared lights the kernels' `lights` argument is the staged array (LREC float4 per light)
    const SailorLightShaderData* L = dLights;
    if (dPreparedLights) L = reinterpret_cast<const SailorLightShaderData*>((const char*)dPreparedLights + prepared_layout(preparedCapacity).offStaged);
    // (tile lists: the kernels' `grid` argument is tileNum, their `culled` the per-tile slots -- see k2_shade_body)
    const SailorLightsGrid* G = dTileNum ? reinterpret_cast<const SailorLightsGrid*>(dTileNum) : dLightsGrid;
    const char* kname = "k2_shade"; // (the launched variant's name, for sailor_hip_context_launch_log)
    // (bandLds: the wave-slot reserve of a band's shade, below; 0 on the whole frame)
#define LAUNCH_SHADE(K) do { if (dPreparedLights && dTileNum) { kname = #K "_pt"; sailor_launch_lds(ctx, K##_pt, grid, dim3(256), bandLds, A, C, I, S, surfacePlaneStride, L, G, dCulledLights, Rd); } \
                             else if (dPreparedLights) { kname = #K "_p"; sailor_launch_lds(ctx, K##_p, grid, dim3(256), bandLds, A, C, I, S, surfacePlaneStride, L, G, dCulledLights, Rd); } \
                             else if (dTileNum) { kname = #K "_t"; sailor_launch_lds(ctx, K##_t, grid, dim3(256), bandLds, A, C, I, S, surfacePlaneStride, L, G, dCulledLights, Rd); } \
                             else { kname = #K; sailor_launch_lds(ctx, K, grid, dim3(256), bandLds, A, C, I, S, surfacePlaneStride, L, G, dCulledLights, Rd); } } while (0)
    const bool partial = band->tileRowEnd - band->tileRowBegin < Ty;
    const bool splitBand = dTileOrder && partial && !ibl; // a band of a split frame with the cull's length bytes: long tiles are split across four blocks
    // A band's shade does not take the whole chip: SHADE_BAND_RESERVE bytes of dynamic LDS that nobody touches cap it at six blocks (24 of 32 wave slots)
    // per CU.  A split frame is a pipeline of short launches -- the NEXT frame's cull chain runs beside this kernel on another stream -- and with every wave
    // slot taken by shade blocks each of the chain's four launches queues behind them: measured on 1/8 bands of the 4K frame, alone the kernel takes
    // ~3 us longer (30 -> 33 us), the pipelined step ~8 us less (53 -> 45 us).  SAILOR_BAND_SHADE_LDS=<bytes> overrides (0: eight blocks).
    // For bands of up to three rounds of resident blocks (an eighth of the 4K frame is two) -- a larger band is bound by the shade's throughput like the
    // whole frame, and the cap costs it what it costs there (half the 4K frame: 108 -> 120 us per step) -- and for any band under a large light set, whose
    // cull chain is as long as its shade (an eighth of the 8K frame under a million lights: 16 320 tiles, 80 us of cull beside 78 us of shade; 152 -> 134 us).
    static const int bandLdsRaw = [] { const char* e = getenv("SAILOR_BAND_SHADE_LDS"); return e ? atoi(e) : -1; }();
    static const bool bandLdsSet = getenv("SAILOR_BAND_SHADE_LDS") != nullptr;
    // (a launch may claim 64 KB of LDS in all; the kernels' static ShadeLds comes off that)
    const int bandLdsEnv = (bandLdsRaw >= 0 && (size_t)bandLdsRaw + sizeof(ShadeLds) <= 65536u) ? bandLdsRaw : -1;
    if (bandLdsSet && bandLdsEnv < 0) ctx->lastError = "SAILOR_BAND_SHADE_LDS negative or beyond 64 KB minus the kernel's own LDS: ignored";
    // (round 5: NOT for the shadowed band kernels -- seven waves per SIMD by their registers, so a CU's eighth block slot is the chain's already; capped at six
    // an eighth of C4 took 62.2 us per step instead of 56.8: profiles/r05/ab_band_csm_waves.txt, probe Q)
    const bool reserve = (bandTiles <= 3 * 8 * ctx->numCUs || lightsNum >= 131072) && !(hasCsm && splitBand);
    const unsigned bandLds = (!partial || ibl) ? 0u : (bandLdsEnv >= 0 ? (unsigned)bandLdsEnv : (reserve ? (unsigned)SHADE_BAND_RESERVE : 0u));
    // Two-wave blocks, one per half tile (round 6; SAILOR_SHADE_HALF=1): built for the wave-slot gap of the four-wave form (a new block needs a free slot on all
    // four SIMDs at once), parity-green, and measured NEUTRAL on the 4K frame -- kernel 134.6 / 135.3 us against 136.3 / 135.7, step 171.4 / 169.8 against
    // 170.5 / 168.4, same box, alternating runs (profiles/r06/ab_two_wave_blocks.txt): off by default.  Not with the ambient term, shadow maps, in-kernel
    // staging or the band form's split launch.
    static const int halfEnv = [] { const char* e = getenv("SAILOR_SHADE_HALF"); return e ? atoi(e) : -1; }();
    const bool halfBlocks = (halfEnv >= 0 ? halfEnv != 0 : SHADE_HALF_DEFAULT) && !ibl && !splitBand && !hasCsm && dPreparedLights;
    if (halfBlocks) {
        const dim3 gridH(2u * grid.x, grid.y, grid.z);
        if (dTileNum) { kname = "k2_shade_h_pt"; sailor_launch_lds(ctx, k2_shade_h_pt, gridH, dim3(128), bandLds, A, C, I, S, surfacePlaneStride, L, G, dCulledLights, Rd); }
        else { kname = "k2_shade_h_p"; sailor_launch_lds(ctx, k2_shade_h_p, gridH, dim3(128), bandLds, A, C, I, S, surfacePlaneStride, L, G, dCulledLights, Rd); }
    }
    else if (hasCsm && ibl) LAUNCH_SHADE(k2_shade_csm_ibl);
    else if (hasCsm && !splitBand) LAUNCH_SHADE(k2_shade_csm);
    else if (ibl) LAUNCH_SHADE(k2_shade_ibl);
    else if (splitBand) {
        const dim3 bgrid((unsigned)SPLIT_BLOCKS + (unsigned)bandTiles);
#define LAUNCH_BAND(K) do { if (dPreparedLights && dTileNum) { kname = #K "_pt"; sailor_launch_lds(ctx, K##_pt, bgrid, dim3(256), bandLds, A, C, bandTiles, S, surfacePlaneStride, L, G, dCulledLights, Rd); } \
                            else if (dPreparedLights) { kname = #K "_p"; sailor_launch_lds(ctx, K##_p, bgrid, dim3(256), bandLds, A, C, bandTiles, S, surfacePlaneStride, L, G, dCulledLights, Rd); } \
                            else if (dTileNum) { kname = #K "_t"; sailor_launch_lds(ctx, K##_t, bgrid, dim3(256), bandLds, A, C, bandTiles, S, surfacePlaneStride, L, G, dCulledLights, Rd); } \
                            else { kname = #K; sailor_launch_lds(ctx, K, bgrid, dim3(256), bandLds, A, C, bandTiles, S, surfacePlaneStride, L, G, dCulledLights, Rd); } } while (0)
        if (hasCsm) LAUNCH_BAND(k2_shade_band_csm);
        else LAUNCH_BAND(k2_shade_band);
#undef LAUNCH_BAND
    } else LAUNCH_SHADE(k2_shade);
#undef LAUNCH_SHADE
    SAILOR_CHECK_LAUNCH(ctx, kname);
    return SAILOR_HIP_OK;
}

extern "C" int sailor_hip_shade_prepared(SailorHipContext* ctx, const SailorUboFrameData* frame, const float* dSurface, size_t surfacePlaneStride,
                                         const SailorLightShaderData* dLights, int32_t lightsNum,
                                         const SailorLightsGrid* dLightsGrid, const uint32_t* dCulledLights,
                                         const SailorCsmDesc* csm, const SailorIblDesc* ibl, float* dRadiance, const SailorBand* band, const uint32_t* dTileOrder,
                                         const void* dPreparedLights, int32_t preparedCapacity)
{
    if (!dLightsGrid) return SAILOR_HIP_ERR_INVALID_ARGUMENT;
    return shade_impl(ctx, frame, dSurface, surfacePlaneStride, dLights, lightsNum, dLightsGrid, dCulledLights, nullptr, csm, ibl, dRadiance, band, dTileOrder, dPreparedLights,
                      preparedCapacity);
}

extern "C" int sailor_hip_shade_tile_lists(SailorHipContext* ctx, const SailorUboFrameData* frame, const float* dSurface, size_t surfacePlaneStride,
                                           const SailorLightShaderData* dLights, int32_t lightsNum, const uint32_t* dTileNum, const uint32_t* dTileLists,
                                           const SailorCsmDesc* csm, const SailorIblDesc* ibl, float* dRadiance, const SailorBand* band, const uint32_t* dTileOrder,
                                           const void* dPreparedLights, int32_t preparedCapacity)
{
    if (!dTileNum || !dTileLists) return SAILOR_HIP_ERR_INVALID_ARGUMENT;
    return shade_impl(ctx, frame, dSurface, surfacePlaneStride, dLights, lightsNum, nullptr, dTileLists, dTileNum, csm, ibl, dRadiance, band, dTileOrder, dPreparedLights,
                      preparedCapacity);
}

extern "C" int sailor_hip_shade_ex(SailorHipContext* ctx, const SailorUboFrameData* frame, const float* dSurface, size_t surfacePlaneStride,
                                   const SailorLightShaderData* dLights, int32_t lightsNum,
                                   const SailorLightsGrid* dLightsGrid, const uint32_t* dCulledLights,
                                   const SailorCsmDesc* csm, const SailorIblDesc* ibl, float* dRadiance, const SailorBand* band, const uint32_t* dTileOrder)
{
    return sailor_hip_shade_prepared(ctx, frame, dSurface, surfacePlaneStride, dLights, lightsNum, dLightsGrid, dCulledLights, csm, ibl, dRadiance, band, dTileOrder, nullptr, 0);
}

extern "C" int sailor_hip_shade(SailorHipContext* ctx, const SailorUboFrameData* frame, const float* dSurface, size_t surfacePlaneStride,
                                const SailorLightShaderData* dLights, int32_t lightsNum,
                                const SailorLightsGrid* dLightsGrid, const uint32_t* dCulledLights,
                                const SailorCsmDesc* csm, float* dRadiance, const SailorBand* band)
{
    return sailor_hip_shade_ex(ctx, frame, dSurface, surfacePlaneStride, dLights, lightsNum, dLightsGrid, dCulledLights, csm, nullptr, dRadiance, band, nullptr);
}
