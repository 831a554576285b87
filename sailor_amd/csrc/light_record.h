// The staged light record and the exact short forms of sqrt / reciprocal it is built with -- shared by the shade kernels (shade_body.h), by
// sailor_hip_prepare_lights (shade.hip) and by the cull's fused per-frame preparation (light_cull.hip: SAILOR_CULL_PREPARE_LIGHTS): one definition,
// so a record staged in any of the three places has the same bits.
#pragma once
#include "common.h"

// sqrtf for the squared lengths of this file -- the same bits, five instructions.  The compiler's correctly rounded sqrtf is v_sqrt_f32 plus two
// one-ulp corrections, wrapped in a 2^32 scaling for x < 2^-96 and a class check for 0 / inf: 21 instructions and four hazard nops.  One Newton
// step on v_rsq_f32 -- s0 = x y, s = s0 + (x - s0^2) (y / 2), the residual by fma -- gives sqrtf's bits for EVERY x in [2^-96, inf) (all
// 1.9 x 10^9 of them compared on the chip: scripts/microbench/sqrt_newton_exhaustive.hip; the two corrections alone do as well, at nine
// instructions: sqrt_exhaustive.hip).  A wave-uniform check sends a wave that holds anything else -- 0, a squared length below 2^-96, inf, NaN --
// to sqrtf itself.
__device__ __forceinline__ float sqrt_exact(float x)
{
    const float y = __builtin_amdgcn_rsqf(x);
    const float s0 = x * y, h = 0.5f * y;
    float s = fmaf(fmaf(-s0, s0, x), h, s0);
    if (__builtin_expect(__ballot(__float_as_uint(x) - 0x0F800000u >= 0x70000000u) != 0ull, 0)) s = sqrtf(x); // (an `if` without an `else`: one branch)
    return s;
}
// 1.0f / s for s = a square root -- the same bits, four instructions instead of twelve.  The compiler's IEEE division scales its operands, refines
// v_rcp_f32 and the quotient with five fma and undoes the scaling (div_scale x 2, div_fmas, div_fixup).  For a numerator of 1 and a denominator
// in [2^-126, 2^126] one Newton step on v_rcp_f32 plus v_div_fixup_f32 (0, inf, NaN) gives the same bits on every one of the 2^32 inputs
// (scripts/microbench/rcp_exhaustive.hip: the mismatches are the denormals and |s| > 2^126, where the quotient is a denormal), and the square
// root of a float lies in [2^-74.5, 2^64] or is 0, inf or NaN.
__device__ __forceinline__ float rcp_of_sqrt(float s)
{
    const float r0 = __builtin_amdgcn_rcpf(s);
    const float e = fmaf(-s, r0, 1.0f);
    return __builtin_amdgcn_div_fixupf(fmaf(e, r0, r0), s, 1.0f);
}


#define LREC 5 // float4 per staged light
#define LIGHT_SLOW_SHIFT 24 // see stage_light_record

// One light's record (SailorLightShaderData as seven float4) -> its staged form (the five float4 described at "Staged light record" below).
// Used by k2_shade's own staging and by sailor_hip_prepare_lights (shade.hip), which runs it once per uploaded light instead of once per
// (tile, list slot): the same instructions, so the same bits either way.
__device__ __forceinline__ void stage_light_record(const float4 q0, const float4 q1, const float4 q2, const float4 q3, const float4 q4, const float4 q5, const float4 q6,
                                                   float4& o0, float4& o1, float4& o2, float4& o3, float4& o4)
{
    const uint32_t type = __float_as_uint(q0.x), shadowType = __float_as_uint(q0.y);
    const float ndx = -q2.x, ndy = -q2.y, ndz = -q2.z;            // Li = -light.direction (:309)
    const float len = sqrt_exact(dot3f(ndx, ndy, ndz, ndx, ndy, ndz)); // normalize(-light.direction) (:298)
    const float linv = rcp_of_sqrt(len);
    // A zero factor only annihilates a FINITE product (inf * 0 = NaN in the reference): a light with ANY non-finite parameter --
    // intensity, but also position, direction, attenuation, cone or radius, which reach the product through the falloff -- is never
    // skipped.
    // (kept short and free of branches: x * 0 is 0 for a finite x and NaN otherwise, two fma chains collect the fifteen parameters, one
    // compare reads the result)
    const float z0 = fmaf(q3.x, 0.0f, fmaf(q3.y, 0.0f, fmaf(q3.z, 0.0f, fmaf(q1.x, 0.0f, fmaf(q1.y, 0.0f, fmaf(q1.z, 0.0f, fmaf(q2.x, 0.0f, q2.y * 0.0f)))))));
    const float z1 = fmaf(q2.z, 0.0f, fmaf(q4.x, 0.0f, fmaf(q4.y, 0.0f, fmaf(q4.z, 0.0f, fmaf(q5.x, 0.0f, fmaf(q5.y, 0.0f, q6.x * 0.0f))))));
    const float zz = z0 + z1;
    const bool finite = zz == zz;
    // Conservative "out of reach" threshold of a point light: d^2 > r^2 (1 + 1e-5) => fl(dist / r) >= 1 => the radius
    // window (:290) is exactly 0.  Only for r > 0 (a negative radius clamps to the FULL window in the reference).
    const float r = q6.x;
    const bool isPointLight = type == 1u;
    const float raFinite = isPointLight ? (r > 0.0f ? (r * r) * 1.00001f : __builtin_inff()) : -(q5.y - 1e-5f);
    const float ra = finite ? raFinite : __builtin_inff();
    const float rb = isPointLight ? r : q5.x - q5.y;
    // The staged KIND (low byte of rec1.w): 1 point, 2 spot, 3 any other type (no falloff: all the reference's branches pass it by), 0 the lights
    // that are shaded one lane per PIXEL behind the pair queue: the directional ones, and (bits 24-25 = the type) a point / spot light whose
    // divisor B (bounds.x / the cone's epsilon) lies outside [2^-40, 2^40], where the staged reciprocal cannot stand in for the IEEE division
    // (see div_stored) -- the pair pass, which every ordinary pair runs, then needs no check for such a light.
    const uint32_t rbExp = (__float_as_uint(rb) >> 23) & 0xFFu;
    const bool storedOk = rbExp - 87u <= 80u;
    const bool slow = (type == 1u || type == 2u) && !storedOk;
    const uint32_t kind = slow ? 0u : (type < 3u ? type : 3u);
    const uint32_t bits = kind | ((shadowType < 255u ? shadowType : 255u) << 8) | (finite ? 0x10000u : 0u) | (slow ? type << LIGHT_SLOW_SHIFT : 0u);
    o0 = make_float4(q1.x, q1.y, q1.z, ra);
    o1 = make_float4(ndx * linv, ndy * linv, ndz * linv, __uint_as_float(bits));
    o2 = make_float4(q4.x, q4.y, q4.z, rb);
    o3 = make_float4(ndx, ndy, ndz, q5.y);
    o4 = make_float4(q3.x, q3.y, q3.z, storedOk ? rcp_of_sqrt(rb) : __builtin_nanf("")); // (rcp_of_sqrt: RN(1 / x) for every |x| in [2^-126, 2^126])
}
