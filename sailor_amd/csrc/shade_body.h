// The device side of K2 + K3 (and the ambient / IBL term): argument structs, the canonical shadow-map helpers, and k2_shade_body -- one 16 x 16 tile
// per 256-thread block.  In a header so that shade.hip (one kernel per variant) and pipeline.hip (shade slices fused with the next frame's cull
// stages) instantiate the same code.  Design notes: shade.hip.
#pragma once
#include "common.h"
#include "sampling.h"
#include "light_record.h"
#include <hip/hip_fp16.h>
#include <type_traits>

struct CsmArgs {
    Mat4 lightsMatrices[SAILOR_NUM_CSM_CASCADES];
    const void* maps[SAILOR_NUM_CSM_CASCADES];
    int width[SAILOR_NUM_CSM_CASCADES];
    int height[SAILOR_NUM_CSM_CASCADES];
    int format[SAILOR_NUM_CSM_CASCADES];
    float texelW[SAILOR_NUM_CSM_CASCADES], texelH[SAILOR_NUM_CSM_CASCADES]; // 1.0f / width, 1.0f / height (Lighting.glsl:171 texelSize), divided once on the host: the same IEEE quotient
};

struct IblArgs { // SailorIblDesc by value
    const float4* irradiance; const float4* env; const float2* brdfLut; const float* ao;
    int irrSize, envSize, envLevels, lutW, lutH;
};

struct ShadeArgs {
    Mat4 view;
    float camX, camY, camZ;
    float zFar;
    int vpW, vpH;     // frame.viewportSize
    int W;            // surface width in pixels
    int H;            // full frame height
    int Tx;
    int tileRow0;     // band.tileRowBegin
    int fbRow0;       // band.fbRowBegin
    int fbRows;       // band.fbRowCount
    int bandTileRows; // band.tileRowEnd - band.tileRowBegin
    int lightsNum;
    int splitMin;     // band form: a tile with at least this many lights goes to the split blocks (SPLIT_MIN_* by the band's size)
    const uint8_t* order;  // sailor_hip_light_cull_tile_order (the band's list lengths as bytes, one per tile) or null
};

// ---- K3: canonical-order helpers (must match oracle/sailor_oracle.c bit for bit) -----------------------------
__device__ __forceinline__ float texel_r(const void* __restrict__ map, int fmt, int W, int x, int y)
{
    const size_t i = (size_t)y * W + x;
    if (fmt == SAILOR_SHADOWMAP_R16_SFLOAT) return __half2float(reinterpret_cast<const __half*>(map)[i]);
    if (fmt == SAILOR_SHADOWMAP_R32_SFLOAT) return reinterpret_cast<const float*>(map)[i];
    return reinterpret_cast<const float*>(map)[i * 4];
}

__device__ __forceinline__ float sample_r(const void* __restrict__ map, int fmt, int W, int H, float u, float v)
{
    const BilinearTaps t = bilinear_taps(W, H, u, v);
    return lerp2(texel_r(map, fmt, W, t.x0, t.y0), texel_r(map, fmt, W, t.x1, t.y0),
                 texel_r(map, fmt, W, t.x0, t.y1), texel_r(map, fmt, W, t.x1, t.y1), t.ax, t.ay);
}

__device__ __forceinline__ float4 sample_rgba(const void* __restrict__ map, int fmt, int W, int H, float u, float v)
{
    const BilinearTaps t = bilinear_taps(W, H, u, v);
    if (fmt != SAILOR_SHADOWMAP_R32G32B32A32_SFLOAT) {
        const float r = lerp2(texel_r(map, fmt, W, t.x0, t.y0), texel_r(map, fmt, W, t.x1, t.y0),
                              texel_r(map, fmt, W, t.x0, t.y1), texel_r(map, fmt, W, t.x1, t.y1), t.ax, t.ay);
        return make_float4(r, 0.0f, 0.0f, 1.0f);
    }
    const float4* m = reinterpret_cast<const float4*>(map);
    const float4 a = m[(size_t)t.y0 * W + t.x0], b = m[(size_t)t.y0 * W + t.x1];
    const float4 c = m[(size_t)t.y1 * W + t.x0], d = m[(size_t)t.y1 * W + t.x1];
    return make_float4(lerp2(a.x, b.x, c.x, d.x, t.ax, t.ay), lerp2(a.y, b.y, c.y, d.y, t.ax, t.ay),
                       lerp2(a.z, b.z, c.z, d.z, t.ax, t.ay), lerp2(a.w, b.w, c.w, d.w, t.ax, t.ay));
}

// The one exp() of the path (Lighting.glsl:277-278): fixed fp32 algorithm shared with the oracle.
__device__ __forceinline__ float canonical_expf(float x)
{
    if (x > 88.0f) x = 88.0f;
    if (x < -87.0f) x = -87.0f;
    const float n = floorf(x * 1.44269504088896341f + 0.5f);
    float r = x - n * 0.693359375f;
    r = r - n * -2.12194440e-4f;
    const float z = r * r;
    float p = 1.9875691500e-4f;
    p = p * r + 1.3981999507e-3f;
    p = p * r + 8.3334519073e-3f;
    p = p * r + 4.1665795894e-2f;
    p = p * r + 1.6666665459e-1f;
    p = p * r + 5.0000001201e-1f;
    const float y = (p * z + r) + 1.0f;
    return ldexpf(y, (int)n);
}

__constant__ float kPoissonDisk[16][2] = { // Lighting.glsl:176-185
    { -0.94201624f, -0.39906216f }, { 0.94558609f, -0.76890725f }, { -0.094184101f, -0.92938870f }, { 0.34495938f, 0.29387760f },
    { -0.91588581f, 0.45771432f }, { -0.81544232f, -0.87912464f }, { -0.38277543f, 0.27676845f }, { 0.97484398f, 0.75648379f },
    { 0.44323325f, -0.97511554f }, { 0.53742981f, -0.47373420f }, { -0.26496911f, -0.41893023f }, { 0.79197514f, 0.19090188f },
    { -0.24188840f, 0.99706507f }, { -0.81409955f, 0.91437590f }, { 0.19984126f, 0.78641367f }, { 0.14383161f, -0.14100790f }
};

// One bilinear footprint of an R16F map: the two texels of a footprint row are neighbours, so each row is ONE 4-byte request (x1 = x0 + 1
// unless the footprint is clamped at the left / right edge; the pair is then read at the nearest in-row position and the halves are
// selected).  Same texels, same lerp2 as sample_r: same bits.
__device__ __forceinline__ float sample_r16_pairs(const __half* __restrict__ map, int W, int H, float u, float v)
{
    const BilinearTaps t = bilinear_taps(W, H, u, v);
    const int base = min(t.x0, W - 2);
    typedef uint32_t __attribute__((aligned(2))) u32_a2;
    const uint32_t r0 = *reinterpret_cast<const u32_a2*>(map + (size_t)t.y0 * W + base);
    const uint32_t r1 = *reinterpret_cast<const u32_a2*>(map + (size_t)t.y1 * W + base);
    const bool lo0 = t.x0 == base, lo1 = t.x1 == base;
    const float a00 = __half2float(__ushort_as_half((unsigned short)(lo0 ? r0 & 0xFFFFu : r0 >> 16)));
    const float a10 = __half2float(__ushort_as_half((unsigned short)(lo1 ? r0 & 0xFFFFu : r0 >> 16)));
    const float a01 = __half2float(__ushort_as_half((unsigned short)(lo0 ? r1 & 0xFFFFu : r1 >> 16)));
    const float a11 = __half2float(__ushort_as_half((unsigned short)(lo1 ? r1 & 0xFFFFu : r1 >> 16)));
    return lerp2(a00, a10, a01, a11, t.ax, t.ay);
}

// The 16 taps of an R16F cascade out of ONE 6 x 6-texel window (round 5).  The Poisson offsets are +-2 texels (Lighting.glsl:171-185: disk * 2 * texelSize),
// so the sixteen 2 x 2 footprints of a pixel all lie in the texels [cx - 2, cx + 3] x [cy - 2, cy + 3] around the centre's own footprint: six 12-byte row
// reads per pixel in place of thirty-two 4-byte ones -- the kernel is bound by the texel path's address rate (its lanes' footprints are ~100 texels apart:
// every lane of every request is a line of its own), not by bytes or arithmetic.  A tap's footprint origin is floor(frac + offset) texels from the
// centre's, i.e. one of TWO compile-time positions per axis: the selection is four conditional moves on row dwords and one funnel shift per row.  Same
// texels, same weights (the tap's own bilinear_taps arithmetic), same lerp2 as sample_r: same bits.  The sum of sixteen 0 / 1 terms is exact in any order.
// Lanes whose window would cross the map's left / right edge and maps larger than PCF_WINDOW_MAX_SIZE take the tap-by-tap path below.  There is NO per-tap
// guard on the origin: that it is one of the two positions is arithmetic (see PCF_WINDOW_MAX_SIZE), and the size limit is what the arithmetic needs.
// Register budget (the K3 kernels sit at exactly 64): the window is walked in two phases of four rows -- taps whose footprints start in window rows 0-1, then
// rows 2-3 -- so twelve dwords are live, not eighteen (the tap-by-tap path holds sixteen).  The disk happens to put each of the four column origins exactly
// once into each of the four row origins: ordered that way, tap j of either phase has the same compile-time origin (column j & 3, row j >> 2 of the phase's
// four rows) and only its offset differs -- the phases are one rolled loop reading the offsets as scalars.
struct __attribute__((packed, aligned(2))) PcfRow { uint32_t d[3]; };
__constant__ float kPcfDisk[2][8][2] = { // kPoissonDisk in the order { 5, 2, 8, 1, 0, 10, 15, 9 }, { 4, 6, 3, 11, 13, 12, 14, 7 }: by (row origin, column origin)
    { { -0.81544232f, -0.87912464f }, { -0.094184101f, -0.92938870f }, { 0.44323325f, -0.97511554f }, { 0.94558609f, -0.76890725f }, { -0.94201624f, -0.39906216f }, { -0.26496911f, -0.41893023f }, { 0.14383161f, -0.14100790f }, { 0.53742981f, -0.47373420f } },
    { { -0.91588581f, 0.45771432f }, { -0.38277543f, 0.27676845f }, { 0.34495938f, 0.29387760f }, { 0.79197514f, 0.19090188f }, { -0.81409955f, 0.91437590f }, { -0.24188840f, 0.99706507f }, { 0.19984126f, 0.78641367f }, { 0.97484398f, 0.75648379f } },
};

// Where a tap's footprint starts.  With x_c = RN(RN(px W) - 0.5) the centre's coordinate and x_t = RN(RN(RN(px + RN(2 p RN(1 / W))) W) - 0.5) the tap's, the
// roundings move x_t - x_c at most 3e-7 W texels away from the offset 2 p; no 2 p of the disk is nearer than 0.0059 to an integer, so for W <= 8192
// (2.5e-3 texels) floor(x_t) - floor(x_c) is floor(2 p) or floor(2 p) + 1 -- the window's two compile-time origins -- and nothing else.  Larger maps take the
// tap-by-tap path.
#define PCF_WINDOW_MAX_SIZE 8192

// (Measured with it and dropped, round 5: the window's least and greatest texel -- taken on the halves' bits, six row reads and ~110 instructions -- decide all
// sixteen compares of a lane whose reference depth is above / below every value a tap can take, and a wave whose 64 lanes are all decided skips the taps'
// ~45 instructions each.  Bit-exact on every map of tests/test_shade_gpu.py's hostile set; on C4 it buys nothing -- k2_shade_csm_pt 215 us with it, 212
// without (profiles/r05/ab_pcf_window.txt): the synthetic surface's depth noise puts a wave's 64 pixels ~100 texels apart in the map, some 5 % of the map lies
// within a window of an occluder's edge, and ONE undecided lane keeps its wave in the loop (0.95^64: a few waves in a hundred skip it).  A scene with
// coherent depth is where it would pay.)
// (addresses: the map's base plus a 32-bit byte offset -- a map of at most 8192 x 8192 halves is 128 MB -- so that a wave-uniform base stays scalar)
__device__ __forceinline__ PcfRow pcf_row(const __half* __restrict__ m16, uint32_t colBytes, int y, int W, int H)
{
    const uint32_t off = (uint32_t)min(max(y, 0), H - 1) * (uint32_t)W * 2u + colBytes;
    return *reinterpret_cast<const PcfRow*>(reinterpret_cast<const char*>(m16) + off);
}

__device__ __forceinline__ float shadow_pcf_window(const __half* __restrict__ m16, float cxf, float cyf, int W, int H, const float tsx, const float tsy,
                                                   float px, float py, float ref)
{
    // (the window's origin is kept as the two floats the taps compare against; the rows' addresses are put together from them where a row is read -- two
    // integer copies of the same numbers held across the taps are two registers the 64-register kernels spill)
    uint32_t w[4][3];
#pragma unroll
    for (int r = 0; r < 2; r++) {
        const PcfRow row = pcf_row(m16, (uint32_t)((int)cxf - 2) * 2u, (int)cyf - 2 + r, W, H);
        w[2 + r][0] = row.d[0]; w[2 + r][1] = row.d[1]; w[2 + r][2] = row.d[2];
    }
    int count = 0;
#pragma unroll 1
    for (int h = 0; h < 2; h++) {
#pragma unroll
        for (int r = 0; r < 2; r++) {
            w[r][0] = w[2 + r][0]; w[r][1] = w[2 + r][1]; w[r][2] = w[2 + r][2];
            const PcfRow row = pcf_row(m16, (uint32_t)((int)cxf - 2) * 2u, (int)cyf + 2 * h + r, W, H);
            w[2 + r][0] = row.d[0]; w[2 + r][1] = row.d[1]; w[2 + r][2] = row.d[2];
        }
        const float rowf = cyf + (float)(2 * h - 2); // the phase's first row as a float (small integers: exact)
#pragma unroll
        for (int j = 0; j < 8; j++) {
            const float ox = kPcfDisk[h][j][0] * 2.0f * tsx, oy = kPcfDisk[h][j][1] * 2.0f * tsy;
            const float x = (px + ox) * (float)W - 0.5f, y = (py + oy) * (float)H - 0.5f; // bilinear_taps' own arithmetic
            const float fx = floorf(x), fy = floorf(y);
            const float ax = x - fx, ay = y - fy;
            const int KX = j & 3, KY = j >> 2; // (compile-time once unrolled)
            const bool sx = fx - cxf != (float)(KX - 2), sy = fy - rowf != (float)KY; // the footprint starts one texel right of / below its first origin
            const int d0 = KX >> 1;
            const uint32_t aLo = sy ? w[KY + 1][d0] : w[KY][d0], aHi = sy ? w[KY + 1][d0 + 1] : w[KY][d0 + 1];
            const uint32_t bLo = sy ? w[KY + 2][d0] : w[KY + 1][d0], bHi = sy ? w[KY + 2][d0 + 1] : w[KY + 1][d0 + 1];
            const uint32_t aMid = __builtin_amdgcn_alignbit(aHi, aLo, 16), bMid = __builtin_amdgcn_alignbit(bHi, bLo, 16);
            const uint32_t pa = (KX & 1) ? (sx ? aHi : aMid) : (sx ? aMid : aLo);
            const uint32_t pb = (KX & 1) ? (sx ? bHi : bMid) : (sx ? bMid : bLo);
            const float a00 = __half2float(__ushort_as_half((unsigned short)(pa & 0xFFFFu))), a10 = __half2float(__ushort_as_half((unsigned short)(pa >> 16)));
            const float a01 = __half2float(__ushort_as_half((unsigned short)(pb & 0xFFFFu))), a11 = __half2float(__ushort_as_half((unsigned short)(pb >> 16)));
            const float d = lerp2(a00, a10, a01, a11, ax, ay) * 0.5f + 0.5f;
            count += (ref > d) ? 1 : 0;
        }
    }
    return (float)count;
}

// Lighting.glsl:242-261 ShadowCalculation_Pcf + :168-197 ManualPCF
__device__ __forceinline__ float shadow_pcf(const void* __restrict__ map, int fmt, int W, int H, const float tsx, const float tsy, float4 lp, float bias)
{
    float px = lp.x, py = lp.y, pz = lp.z;
    if (__ballot(lp.w != 1.0f) != 0ull) { px = px / lp.w; py = py / lp.w; pz = pz / lp.w; } // (x / 1 == x: see shadow_evsm)
    px = px * 0.5f + 0.5f; py = py * 0.5f + 0.5f; pz = pz * 0.5f + 0.5f;
    py = 1.0f - py;
    if (px > 1.0f || py > 1.0f || px < 0.0f || py < 0.0f || pz < 0.5f) return 1.0f;
    float shadow = 0.0f;
    if (fmt == SAILOR_SHADOWMAP_R16_SFLOAT && W >= 2) {
        const __half* m16 = reinterpret_cast<const __half*>(map);
#ifndef PCF_NO_WINDOW
        const float cxf = floorf(px * (float)W - 0.5f), cyf = floorf(py * (float)H - 0.5f);
        const int wx0 = (int)cxf - 2;
        if (wx0 >= 0 && wx0 + 5 <= W - 1 && max(W, H) <= PCF_WINDOW_MAX_SIZE)
            return shadow_pcf_window(m16, cxf, cyf, W, H, tsx, tsy, px, py, pz + bias) / 16.0f;
        // (the lanes at a map's left / right edge: one tap at a time -- what this loop needs in registers, the whole kernel needs)
#pragma unroll 1
        for (int i = 0; i < 16; i++) {
            const float ox = kPoissonDisk[i][0] * 2.0f * tsx, oy = kPoissonDisk[i][1] * 2.0f * tsy;
            shadow += (pz + bias > sample_r16_pairs(m16, W, H, px + ox, py + oy) * 0.5f + 0.5f) ? 1.0f : 0.0f;
        }
        return shadow / 16.0f;
#endif
        // the cascades' own format (ECS/LightingECS.h:57-58), tap by tap: 2 requests per tap, 8 taps = 16 requests in flight
#pragma unroll 1
        for (int i0 = 0; i0 < 16; i0 += 8) {
            float d[8];
#pragma unroll
            for (int j = 0; j < 8; j++) {
                const float ox = kPoissonDisk[i0 + j][0] * 2.0f * tsx, oy = kPoissonDisk[i0 + j][1] * 2.0f * tsy;
                d[j] = sample_r16_pairs(m16, W, H, px + ox, py + oy) * 0.5f + 0.5f;
            }
#pragma unroll
            for (int j = 0; j < 8; j++) shadow += (pz + bias > d[j]) ? 1.0f : 0.0f;
        }
        return shadow / 16.0f;
    }
#pragma unroll 1
    for (int i = 0; i < 16; i++) {
        const float ox = kPoissonDisk[i][0] * 2.0f * tsx, oy = kPoissonDisk[i][1] * 2.0f * tsy;
        const float pcfDepth = sample_r(map, fmt, W, H, px + ox, py + oy) * 0.5f + 0.5f;
        shadow += (pz + bias > pcfDepth) ? 1.0f : 0.0f;
    }
    return shadow / 16.0f;
}

// Lighting.glsl:218-240
__device__ __forceinline__ float chebyshev(float m0, float m1, float currentDepth, float minVariance, float lin)
{
    const float d = currentDepth - m0;
    if (d < 0.0f) return 1.0f;
    const float variance = fmaxf(minVariance, m1 - m0 * m0);
    const float pmax = variance / (variance + d * d);
    return fminf(fmaxf((pmax - lin) / (1.0f - lin), 0.0f), 1.0f);
}

// Lighting.glsl:263-284 ShadowCalculation_Evsm
__device__ float shadow_evsm(const void* __restrict__ map, int fmt, int W, int H, float4 lp, float bias, int cascade)
{
    float px = lp.x, py = lp.y, pz = lp.z;
    // an orthographic cascade: every lane's w is exactly 1 and x / 1 == x -- the three correctly rounded divisions (36 instructions) are skipped
    // (a wave-uniform branch around them and nothing else: round 2's form of this idea took the kernel from 80 to 103 registers; 275 -> 271 us at C4)
    if (__ballot(lp.w != 1.0f) != 0ull) { px = px / lp.w; py = py / lp.w; pz = pz / lp.w; }
    px = px * 0.5f + 0.5f; py = py * 0.5f + 0.5f;
    py = 1.0f - py;
    if (px > 1.0f || py > 1.0f || px < 0.0f || py < 0.0f || pz < 0.0f) return 1.0f;
    const float4 s = sample_rgba(map, fmt, W, H, px, py);
    float p05 = 1.0f;
    for (int i = 0; i < cascade; i++) p05 = p05 * 0.5f;
    const float currentDepth = canonical_expf(40.0f * (pz + 0.003f * bias * p05));
#ifndef EVSM_NO_EARLY_OUT
    // (round 6) Chebyshev returns exactly 1 where the fragment lies behind the stored moment (d < 0), and 1 - max(1, anything) clamps to exactly +0 -- a NaN on
    // the other side included (fmax returns the number).  A wave in which EVERY lane is such a fragment needs neither the second exponential nor the two
    // correctly rounded divisions; likewise for the negative pair once its exponential is there.  Wave-uniform branches, the same bits.
    if (__ballot(!(currentDepth - s.x < 0.0f)) == 0ull) return 0.0f;
#endif
    const float negCurrentDepth = -canonical_expf(-40.0f * (pz + 0.0001f * bias));
#ifndef EVSM_NO_EARLY_OUT
    if (cascade <= 2 && __ballot(!(negCurrentDepth - s.z < 0.0f)) == 0ull) return 0.0f;
#endif
    const float posValue = chebyshev(s.x, s.y, currentDepth, 0.01f, 0.0f);
    const float negValue = chebyshev(s.z, s.w, negCurrentDepth, 0.0f, 0.0f) * (cascade > 2 ? 0.0f : 1.0f);
    return fminf(fmaxf(1.0f - fmaxf(posValue, negValue), 0.0f), 1.0f);
}

// Standard.shader:266-283 + Lighting.glsl:200-216 SelectCascade
// The look-up in one cascade.  `cascade` is either the wave's common cascade (a scalar: the matrix, the map and its size then come by scalar loads
// from the argument segment and the matrix multiplies read them as scalar operands) or the lane's own (per-lane loads from the segment).
template <bool UNIFORM>
__device__ __forceinline__ float cascade_shadow(const CsmArgs& C, const int cascade, const uint32_t shadowType, const float ndl, const float wx, const float wy, const float wz)
{
    const void* map = C.maps[cascade];
    if (!map) return 1.0f;
    const float4 lp = glsl_mul(C.lightsMatrices[cascade], wx, wy, wz, 1.0f);
    if (shadowType == 2u && cascade == 0) {
        const float bias = (1.0f - ndl) * (float)(1 + cascade);
        return shadow_evsm(map, C.format[cascade], C.width[cascade], C.height[cascade], lp, bias, cascade);
    }
    const float bias = fmaxf(0.000075f * (1.0f - ndl), 0.000005f);
    return shadow_pcf(map, C.format[cascade], C.width[cascade], C.height[cascade], C.texelW[cascade], C.texelH[cascade], lp, bias);
}

__device__ float directional_shadow(const ShadeArgs& A, const CsmArgs& C, uint32_t shadowType,
                                    float dirX, float dirY, float dirZ, float nx, float ny, float nz, float wx, float wy, float wz)
{
    const float4 pv = glsl_mul(A.view, wx, wy, wz, 1.0f);
    // (a view matrix is affine: w is exactly 1 in every lane and z / 1 == z -- the correctly rounded division, 12 instructions, only where it is not)
    float depthValue = pv.z;
    if (__ballot(pv.w != 1.0f) != 0ull) depthValue = pv.z / pv.w;
    depthValue = fabsf(depthValue);
    int cascade = SAILOR_NUM_CSM_CASCADES;
    const float levels[4] = { 0.05f, 0.1f, 0.333333f, 0.5f }; // Constants.glsl:24
#pragma unroll
    for (int i = SAILOR_NUM_CSM_CASCADES - 1; i >= 0; i--)
        if (depthValue < A.zFar * levels[i]) cascade = i;
    cascade = min(cascade, SAILOR_NUM_CSM_CASCADES - 1);
    const float ndl = dot3f(nx, ny, nz, dirX, dirY, dirZ);
    // a wave is an 8 x 8-pixel quadrant: nearly always inside one cascade
    // (Measured and dropped, round 5: ONE copy of the look-up code, run once per cascade present in the wave with that cascade's lanes alone active -- always a
    // scalar cascade, a fifth less code.  The loop keeps fourteen more values live than the kernel has registers for: 52 bytes of scratch a pixel, and a
    // kernel that moves 1 GB at 4.7 TB/s has no room for 0.9 GB of spill traffic -- k2_shade_csm_pt 213 -> 287 us.)
    const int common = __builtin_amdgcn_readfirstlane(cascade);
    if (__ballot(cascade != common) == 0ull) return cascade_shadow<true>(C, common, shadowType, ndl, wx, wy, wz);
    return cascade_shadow<false>(C, cascade, shadowType, ndl, wx, wy, wz);
}

// ---- ambient / IBL term (Standard.shader:343-372), canonical samplers == oracle/sailor_oracle.c (tolerance-checked) ----
__device__ __forceinline__ float2 lut_sample(const float2* __restrict__ lut, int W, int H, float u, float v)
{
    const BilinearTaps b = bilinear_taps(W, H, u, v);
    const float2 a = lut[(size_t)b.y0 * W + b.x0], c = lut[(size_t)b.y0 * W + b.x1];
    const float2 d = lut[(size_t)b.y1 * W + b.x0], e = lut[(size_t)b.y1 * W + b.x1];
    return make_float2(lerp2(a.x, c.x, d.x, e.x, b.ax, b.ay), lerp2(a.y, c.y, d.y, e.y, b.ax, b.ay));
}

// ---- K2 ---------------------------------------------------------------------------------------------------------
__device__ __forceinline__ float rcp_fast(float x) { return __builtin_amdgcn_rcpf(x); }
__device__ __forceinline__ float rsq_fast(float x) { return __builtin_amdgcn_rsqf(x); }
// a / b with y = RN(1 / b) staged beside b:  q0 = a y;  e = fma(-b, q0, a) (exact);  q = fma(e, y, q0)  -- Markstein's correction step: with y
// the correctly rounded reciprocal and q0 within an ulp of the quotient, q IS the correctly rounded quotient as long as nothing over- or
// underflows on the way (4 instructions with v_div_fixup_f32 for 0 / inf / NaN operands against the IEEE division's 12; no mismatch in 2^36
// random pairs and 36 special divisors x every significand, round 2).  "Nothing overflows" is the stager's business: it leaves NaN instead of
// a reciprocal when |b| is outside [2^-40, 2^40] (the numerators here are a distance in [2^-48, 2^64] and a cosine difference) and marks the
// light LIGHT_KIND_SLOW: its pairs take the IEEE division on a branch of their own (stage_light_record; was: a wave-uniform check in front
// of every division -- nine instructions per batch of pairs for a light that practically never occurs).
__device__ __forceinline__ float div_stored(float a, float b, float y)
{
    const float q0 = a * y;
    return __builtin_amdgcn_div_fixupf(fmaf(fmaf(-b, q0, a), y, q0), b, a);
}

typedef float v2f_ __attribute__((ext_vector_type(2)));
__device__ __forceinline__ v2f_ fma2(const v2f_ a, const v2f_ b, const v2f_ c) { return __builtin_elementwise_fma(a, b, c); }
// dot3f on a float pair + a float: the same three products and two sums in the same association ((ax bx + ay by) + az bz), the two products of
// the pair as ONE v_pk_mul_f32 (IEEE per component, like v_mul_f32): four instructions for five.
__device__ __forceinline__ float dot3_pk(const v2f_ axy, const float az, const v2f_ bxy, const float bz)
{
    const v2f_ p = axy * bxy;
    return (p.x + p.y) + az * bz;
}

#ifndef SHADE_XCD_PIECES
#define SHADE_XCD_PIECES 10 // pieces of a tile row per XCD (see the grid mapping in k2_shade_body)
#endif
#define PENDK 4  // queued pairs per pixel in one window (their queue positions ride in one register, 7 bits each under a sentinel bit)
#define QMAX 128 // queued pairs per wave in one window (two lights that reach every pixel fit; positions are 7 bits; 17.4 KB of LDS per block, 9 blocks per CU)
typedef float v2f __attribute__((ext_vector_type(2)));
typedef v2f_ p2f;

// The falloff of a point / spot light at one surface point (Standard.shader:286-307), exact: the oracle's op order where it is ill-conditioned.
// R = the staged record, r0 / r1 = its first two float4, cutY = rec3.w, (wxy, wz) = the surface point.  SLOW: IEEE divisions instead of div_stored.
template <typename SlowTag>
__device__ __forceinline__ float exact_falloff(SlowTag, const bool isPoint, const float4* R, const float4 r0, const float4 r1, const float cutY, const p2f wxy, const float wz)
{
    constexpr bool SLOW = SlowTag::value;
    const float4 r2 = R[2];
    const float binv = reinterpret_cast<const float*>(R)[19]; // rec4.w = RN(1 / r2.w) (NaN for a SLOW light)
    const p2f dxy = p2f{ r0.x, r0.y } - wxy;
    const float dz = r0.z - wz;
    const float d2 = dot3_pk(dxy, dz, dxy, dz);
    const float dist = sqrt_exact(d2);                                       // exact: feeds 1 - (dist / bounds.x)^2
    const float att = rcp_fast(fmaf(r2.z, d2, fmaf(r2.y, dist, r2.x))); // 1/(a.x + a.y d + a.z d^2) (:289,:300)
    // point: dist / bounds.x (:290); spot: 1 / dist (normalize, :298)
    const float xPoint = SLOW ? dist / r2.w : div_stored(dist, r2.w, binv), xSpot = rcp_of_sqrt(dist);
    const float x = isPoint ? xPoint : xSpot;
    if (isPoint) {
        const float q = fminf(fmaxf(x, 0.0f), 1.0f);
        return att * (1.0f - q * q);                                     // (:290)
    }
    const float theta = dot3_pk(dxy * x, dz * x, p2f{ r1.x, r1.y }, r1.z); // dot(normalize(pos - wp), normalize(-dir))
    const float t = SLOW ? (theta - cutY) / r2.w : div_stored(theta - cutY, r2.w, binv);
    const float f = att * fminf(fmaxf(t, 0.0f), 1.0f);                    // (:301); exact: cancels at the cone edge
    return theta < cutY ? 0.0f : f;                                       // (:303-306)
}

// max of a non-negative (or NaN) float's bits over the wave as unsigned integers; the value of lane 63 (which holds the result) is returned.
// (xor-1, xor-2, mirror within 8, mirror within 16, then the gfx9 row broadcasts 15 -> row+1 and 31 -> rows 2,3; a DPP read of a VGPR
// needs 2 wait states after the VALU write.)
__device__ __forceinline__ uint32_t wave_max_u32(uint32_t v)
{
    // (as intrinsics, not inline assembly: the compiler folds each move into a v_max_u32_dpp, knows the two wait states a DPP read needs after
    // the write of its source and fills them with the per-pixel invariants instead of s_nop)
#define MAX_STEP(CTRL, ROWS) v = max(v, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, CTRL, ROWS, 0xf, false)) /* (0 = the identity of max: lanes the control leaves out) */
    MAX_STEP(0xB1, 0xf);  // quad_perm:[1,0,3,2]
    MAX_STEP(0x4E, 0xf);  // quad_perm:[2,3,0,1]
    MAX_STEP(0x141, 0xf); // row_half_mirror
    MAX_STEP(0x140, 0xf); // row_mirror
    MAX_STEP(0x142, 0xa); // row_bcast:15 into rows 1 and 3
    MAX_STEP(0x143, 0xc); // row_bcast:31 into rows 2 and 3
#undef MAX_STEP
    return (uint32_t)__builtin_amdgcn_readlane((int)v, 63);
}

// Staged light record (LDS, 5 float4):
//   rec0 = (worldPosition.xyz, A)   the conservative reach test of both types is  !(v > A):
//                                     point: v = d^2,    A = r^2 (1 + 1e-5)      (out of reach => exact-zero radius window)
//                                     spot : v ~ -theta, A = -(cutOff.y - 1e-5)  (outside the cone => falloff exactly 0)
//                                     A = +inf: never reject
//   rec1 = (normalize(-direction).xyz, bits: type | shadowType << 8 | finite << 16)
//   rec2 = (attenuation.xyz, B)     B = point: bounds.x          spot: epsilon = cutOff.x - cutOff.y (:297)
//   rec3 = (Li = -direction.xyz, cutOff.y)
//   rec4 = (intensity.xyz, RN(1 / B) if 2^-40 <= |B| <= 2^40, else NaN: see div_stored)
//
// The kernel is VALU-bound (rocprofv3: SQ_INSTS_VALU x 4 cycles on 1024 SIMDs == the duration of the loop-per-light version it
// replaced), so the shape below is about vector instructions per wave: light kind, finiteness and "survived the box test" are
// wave-uniform 64-bit masks (scalar registers, scalar branches, six static segments -- no per-light type branches), the reach
// test feeds the scalar branch directly, and the ~130-instruction exact falloff + BRDF only ever runs on queued pairs.
//
// BAND (split frames: sailor_hip_shade_ex on a sub-band with the cull's per-tile list lengths): a band of a split frame has too few
// tiles to hide its longest one -- a tile in the middle of a light cluster (128 lights reaching all 256 pixels = 128 pair passes
// per wave) kept its block busy for ~65 us while the rest of a 1/8 band took 25.  The tiles with >= ShadeArgs::splitMin lights are taken by "split"
// blocks, one per (tile, 8x8 quadrant): the block's four waves take every fourth list slot each over the SAME 64 pixels and add
// their partial sums up through LDS (wave 0 + 1 + 2 + 3, a fixed order).  The grid is 1-D: SPLIT_BLOCKS split blocks first (split block b
// looks at the (tile, quadrant) items b, b + SPLIT_BLOCKS, ... -- the lengths of up to 64 of them in one load, a ballot of the long ones --
// so the long tiles start first), then one ordinary block per tile, which returns at once if the tile belongs to the split blocks.
// The loop over the point / spot lights of one list half that may reach a quadrant, for the usual quadrant (every pixel inside the frame, none
// with roughness 0), by hand: the wave is bound by instruction issue -- a scalar instruction costs what a vector one costs -- and the
// compiler's version of this loop spends more instructions on getting around it than on the tests.  Per light: the conservative reach test on
// rec0 (and rec1 for a cone), out if no pixel passes; the facing test on rec3, out if no pixel passes both; the two overflow checks (queue,
// pairs per pixel), which end the window with the light still in `rest`; the append of (slot << 6 | lane) for the lanes that
// passed, exec set to them.  Registers v56-v62 hold rec0 and rec1 / rec3 (inline assembly cannot name the parts of a register tuple, so the
// tuples are fixed ones, and they double as the loop's temporaries); everything else is the compiler's choice.  Hazards (the compiler does
// not look inside): a v_pk result is not read by the next instruction, a v_rsq result not by the next one either.  The arithmetic is instruction for instruction what the
// C++ loop beside it compiles to.  (The in / out operands are early-clobber: an input that happens to hold the same value -- the
// records' base address and the count are both 0 at the first light -- would otherwise share the register.)
#define SHADE_TEST_POINT \
    "v_mul_f32 v58, v58, v58\n\t" \
    "v_fmac_f32 v58, v57, v57\n\t" \
    "v_fmac_f32 v58, v56, v56\n\t" \
    "v_cmp_ngt_f32 vcc, v58, v59\n\t"
#define SHADE_TEST_SPOT \
    "v_mul_f32 v62, v58, v62\n\t" \
    "v_fmac_f32 v62, v57, v61\n\t" \
    "v_mul_f32 v61, v58, v58\n\t" \
    "v_fmac_f32 v61, v57, v57\n\t" \
    "v_fmac_f32 v61, v56, v56\n\t" \
    "v_rsq_f32 v61, v61\n\t" \
    "v_fmac_f32 v62, v56, v60\n\t"      /* (the dot product's last term in the wait state of the v_rsq result) */ \
    "v_mul_f32_e64 v62, -v62, v61\n\t" \
    "v_cmp_ngt_f32 vcc, v62, v59\n\t"
#define SHADE_LIGHT_LOOP(H_LINE, LOADS, TEST) \
    asm volatile("s_mov_b64 %[ex], exec\n" \
                 "0:\n\t" \
                 "s_ff1_i32_b64 %[bit], %[todo]\n\t" \
                 H_LINE \
                 "s_lshl2_add_u32 %[n], %[bit], %[bit]\n\t"        /* 5 bit */ \
                 "s_lshl4_add_u32 %[n], %[n], %[base]\n\t"         /* the record's LDS address: 80 bit + base (LREC * 16 = 80) */ \
                 "v_mov_b32 v62, %[n]\n\t" \
                 "ds_read_b128 v[56:59], v62\n\t" \
                 LOADS \
                 "s_waitcnt lgkmcnt(0)\n\t" \
                 "v_pk_add_f32 v[56:57], v[56:57], %[wxy] neg_lo:[0,1] neg_hi:[0,1]\n\t" \
                 "v_sub_f32 v58, v58, %[wz]\n\t" \
                 TEST \
                 "s_cbranch_vccz 2f\n\t" \
                 "v_mov_b32 v60, %[n]\n\t" \
                 "ds_read_b96 v[60:62], v60 offset:48\n\t" \
                 "s_waitcnt lgkmcnt(0)\n\t" \
                 "v_mul_f32 v60, %[nx], v60\n\t" \
                 "v_mul_f32 v61, %[ny], v61\n\t" \
                 "v_mul_f32 v62, %[nz], v62\n\t" \
                 "v_add_f32 v60, v60, v61\n\t" \
                 "v_add_f32 v60, v60, v62\n\t" \
                 "v_cmp_lt_f32_e64 %[m], 0, v60\n\t" \
                 "s_and_b64 vcc, %[m], vcc\n\t" \
                 "s_cbranch_scc0 2f\n\t" \
                 "s_bcnt1_i32_b64 %[n], vcc\n\t" \
                 "s_add_i32 %[n], %[n], %[cnt]\n\t" \
                 "s_cmp_gt_u32 %[n], %[qmax]\n\t" \
                 "s_cbranch_scc1 1f\n\t" \
                 "v_cmp_lt_u32_e64 %[m], %[lim], %[pc]\n\t" \
                 "s_and_b64 %[m], %[m], vcc\n\t" \
                 "s_cbranch_scc1 1f\n\t" \
                 "s_mov_b64 exec, vcc\n\t" \
                 "v_mov_b32 v56, %[cnt]\n\t" \
                 "v_mbcnt_lo_u32_b32 v56, vcc_lo, v56\n\t" \
                 "v_mbcnt_hi_u32_b32 v56, vcc_hi, v56\n\t" \
                 "v_lshl_or_b32 %[pc], %[pc], 7, v56\n\t" \
                 "v_lshl_add_u32 v56, v56, 1, %[q]\n\t" \
                 "v_lshl_or_b32 v57, %[bit], 6, %[lane]\n\t" \
                 "ds_write_b16 v56, v57\n\t" \
                 "s_mov_b64 exec, %[ex]\n\t" \
                 "s_mov_b32 %[cnt], %[n]\n" \
                 "2:\n\t" \
                 "s_bitset0_b64 %[todo], %[bit]\n\t" \
                 "s_cmp_lg_u64 %[todo], 0\n\t" \
                 "s_cbranch_scc1 0b\n\t" \
                 "s_branch 3f\n" \
                 "1:\n\t" \
                 "s_mov_b64 %[rest], %[todo]\n" \
                 "3:" \
                 : [todo] "+&s"(todo), [cnt] "+&s"(cnt), [pc] "+&v"(pc), [rest] "+&s"(rest), [bit] "=&s"(bit), [n] "=&s"(n), [m] "=&s"(m), [ex] "=&s"(ex) \
                 : [wxy] "v"(wxy), [wz] "v"(wz), [nx] "v"(nx), [ny] "v"(ny), [nz] "v"(nz), [q] "v"(qAddr), [base] "s"(sLAddr), \
                   [lim] "s"((1u << (7 * PENDK)) - 1u), [qmax] "n"(QMAX), [lane] "v"(lane) \
                 : "v56", "v57", "v58", "v59", "v60", "v61", "v62", "vcc", "scc", "memory")

// A tile with at least ShadeArgs::splitMin lights goes to the split blocks.  By the band's size (round 5; profiles/r05/README.md: block timelines and the step
// of the frame pipeline, same box): a band IS its longest block only while it is a few rounds of blocks; on a larger band the split blocks' extra work (four
// waves load the same 64 pixels: +20 % block-slot time over the tiles they take) and the slots they hold while the tile blocks wait cost more than the tail
// they cut, and only the tiles that WOULD be the tail -- 96 lights and more: 16-42 us as one block -- are split.  Step of an eighth / a quarter / the lower
// half of the 4K frame at thresholds 40 / 64 / 96 / 128: 53.3 / 48.6 / 50.9 / 54.4, 71.9 / 68.4 / 68.8 / 70.6, 110.4 / 105.8 / 103.2 / 103.9 us.
#define SPLIT_MIN_SMALL 64   // bands of up to SPLIT_SMALL_TILES tiles (round 4: 40 everywhere)
#define SPLIT_MIN_LARGE 96   // larger bands
#define SPLIT_SMALL_TILES 12000
#define SHADE_BAND_RESERVE 9000 // bytes of untouched dynamic LDS per block of k2_shade_band*: six blocks per CU instead of eight (see its launch)
#ifndef SPLIT_BLOCKS
#define SPLIT_BLOCKS 2048 // one round of resident blocks (8 per CU)
#endif
static_assert(QMAX <= 128 && 7 * PENDK < 32 && 3 * QMAX >= 192, "queue positions are 7 bits each under a sentinel bit; the split blocks park 3 x 64 partial sums in a wave's slots");
// WAVES = 4: one block per 16 x 16 tile, all <= 128 records staged at once.  WAVES = 2 (round 6): one block per HALF tile (two 8 x 8 quadrants side by side),
// the records staged 64 at a time (a second round for the few tiles with more): 8.7 KB a block -- sixteen two-wave blocks fit a CU where eight four-wave
// ones did, and a new block needs a free wave slot on two SIMDs, not on all four at once.
template <int WAVES>
struct ShadeLdsT {
    float4 sL[(WAVES == 2 ? 64 : KEEP) * LREC];
    float sRes[WAVES * 3 * QMAX]; // per wave: [3 colours][queue position]
    uint16_t sQ[WAVES * QMAX];
    uint32_t sEnd[4];
};
typedef ShadeLdsT<4> ShadeLds;
#define ROLE_TILE 0       // one block per tile, grid (tiles per row, tile rows)
#define ROLE_BAND_TILE 1  // the same inside k2_shade_band: returns at once on a tile of the split blocks
#define ROLE_BAND_SPLIT 2 // one block per (long tile, quadrant)
// PREPARED: `lights` points at the staged records sailor_hip_prepare_lights wrote (LREC float4 per light, indexed like the `light` SSBO) instead of
// at the SSBO itself: a list slot is then five 16-byte loads and five LDS stores -- no arithmetic on the path the block's other waves wait for.
// TILE_LISTS: `grid` / `culled` are not the reference's lightsGrid / culledLights but the cull's own per-tile form (sailor_hip_light_cull_tile_lists):
// `grid` points at tileNum (uint32 per band tile), the list of band tile t is culled[128 t ..] -- the same entries in the same order, available as
// soon as k1_tile_cull has run (k1_pack is then off the frame's critical path).  A template parameter, not a run-time switch: as a kernel-argument
// branch it cost every wave 12 scalar + 4 vector instructions (SQ counters, round 4).
// (Round 5, measured and taken out again -- scripts/shade_wave_prof.py, profiles/r05/shade_waves_*.txt: a 256-thread block's four waves sit one on each
// SIMD of a CU, so a new block needs a free wave slot on ALL four at once; on the 4K frame a finished wave's slot then stays empty for 1.5 us of a 6.8 us
// wave life -- 7.3 of 8 blocks alive per CU, their waves busy 89 % of the block's life: 26 of 32 wave slots in use.  A form with ONE wave per block -- a
// 64-thread block per 8x8 quadrant that reads the list itself, tests it from registers and stages only the records that pass, 32 at a time, into 2.5 KB
// of LDS of its own: no barrier, nobody's slowest quadrant to sit out -- fills 90 % of the slots (the gap shrinks to 0.8 us) but every wave then does the
// list's loads and the compaction itself: wave life 7.44 us, kernel 140 against 135 us, and the frame pipeline's step 176 against 166 us because the next
// frame's cull blocks no longer find four free slots beside it.  Oracle parity was green; profiles/r05/ab_quad_form.txt.)
template <bool HAS_CSM, bool HAS_IBL, int ROLE = ROLE_TILE, bool PREPARED = false, bool TILE_LISTS = false, int WAVES = 4>
__device__ __forceinline__ void k2_shade_body(ShadeLdsT<WAVES>& lds, const ShadeArgs& A, const CsmArgs& C, const IblArgs& I, const float4* __restrict__ surface, size_t planeStride,
                                                 const SailorLightShaderData* __restrict__ lights,
                                                 const SailorLightsGrid* __restrict__ grid, const uint32_t* __restrict__ culled,
                                                 float4* __restrict__ radiance, int selTx = 0, int selTy = 0, int selQuad = 0)
{
    float4* const sL = lds.sL;
    float* const sRes = lds.sRes;
    uint16_t* const sQ = lds.sQ;
    uint32_t* const sEnd = lds.sEnd;
    constexpr bool BAND = ROLE != ROLE_TILE;
    constexpr bool splitRole = ROLE == ROLE_BAND_SPLIT;
    constexpr bool HALFT = WAVES == 2;                 // a half-tile block: two waves, staging rounds of 64 records
    constexpr int HN = HALFT ? 1 : 2;                  // 64-slot halves of the staged records
    static_assert(WAVES == 4 || (WAVES == 2 && ROLE == ROLE_TILE), "two-wave blocks exist in the per-tile grid only");
    int tid = threadIdx.x;
    // (the band kernel holds two copies of this body at 64 registers each, and the thread id -- live from the entry through both -- is what the
    // allocator spills: seven reloads from scratch in the prologue of every ordinary tile.  Put together again from the wave's number, a
    // scalar, and the lane's, two instructions, it need not live across anything.)
    if (BAND) tid = (__builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6) << 6) | (int)__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));
    // (a split block walks several tiles: keep everything derived from the lane id inside the loop body -- hoisted out of the
    // loop those values stay live across the whole body and the 64-register budget spills)
    if (splitRole) asm volatile("" : "+v"(tid));

    // grid = (8 x tiles per piece, pieces, tile rows): no division.  The hardware deals consecutive linear block ids to the eight XCDs in turn, and
    // with a multiple of 8 as the fastest dimension blockIdx.x & 7 IS the XCD (blockIdx.x >> 3 = the tile within the piece).  Every tile row is cut into 8 * SHADE_XCD_PIECES pieces of a few tiles; XCD x takes the pieces
    // ((x - row) mod 8) + 8 c of row `row`: every XCD gets a share of EVERY row, a different one from row to row (a light cluster is spread over all eight -- whole rows per
    // XCD, x, x + 8, ..., measured 1-2 % slower; PAIRS of rows per XCD 7 % slower: some XCDs then hold two cluster rows, others one), a piece's
    // tiles (neighbours share most of their lights) read their records through one L2, and the blocks an XCD runs at a time form a compact patch
    // of the frame.  Pieces of three tiles at 4K (10 per XCD and row); pieces of 6, 15, 30 or single tiles were all within 1.5 %.
    int bty = (int)blockIdx.z;
    // (the K3 kernels do NOT rotate the pieces with the row: an XCD then owns the same columns in every row, and the shadow-map texels of
    // vertically adjacent tiles meet in one L2 -- k2_shade_csm_p 234 -> 225 us on C4; the plain kernel, which has no texels to share, pays 4.6 %
    // for that: the light cluster's columns land on three of the eight XCDs.)
    constexpr unsigned rot = HAS_CSM ? 0u : 1u;
    // (two-wave blocks: the fastest grid dimension counts HALF tiles, 8 x 2 x tiles per piece -- the two halves of a tile are neighbours on one XCD)
    int btx = ((int)((blockIdx.x - (unsigned)bty * rot) & 7u) + 8 * (int)blockIdx.y) * (int)(gridDim.x >> (HALFT ? 4 : 3)) + (int)(blockIdx.x >> (HALFT ? 4 : 3));
    if (ROLE == ROLE_TILE && btx >= A.Tx) return;
    const int lane = tid & 63, wave = tid >> 6;
    int quad = HALFT ? 2 * (int)((blockIdx.x >> 3) & 1u) + wave : wave;
    if (BAND) { btx = selTx; bty = selTy; if (splitRole) quad = selQuad; }
    const int tx = btx, ty = A.tileRow0 + bty;
    const int bandTile = bty * A.Tx + btx;
    // each wave shades one 8x8 quadrant of the tile: the most compact 64-pixel footprint, so that "no pixel of the wave
    // is within reach of this light" holds as often as possible
    const int gx = tx * TILE + (quad & 1) * 8 + (lane & 7);
    const int gy = ty * TILE + (quad >> 1) * 8 + (lane >> 3);
    const int py = A.H - 1 - gy;            // framebuffer row (Standard.shader:414: screenUv.y = H - fragY)
    const unsigned long long activeMask = __ballot(gx < A.W) & __ballot(py >= 0); // (masks of single compares, the per-lane flag from the mask: a ballot of `a && b`, or a flag kept as a bool, costs a VGPR 0 / 1 and a compare per use)
    const bool active = __builtin_amdgcn_inverse_ballot_w64(activeMask);
    const size_t pix = active ? ((size_t)(py - A.fbRow0) * A.W + gx) : 0;

    // Prologue, ordered by what waits for what.  The list is a chain of three dependent round trips (grid entry -> culledLights indices -> light
    // records), the surface is one; a wave that sits through them one after the other spends a third of its life waiting.  So: the grid entry
    // (a scalar load) and the surface loads go out together; the list indices follow as soon as the grid entry is there; the per-pixel
    // invariants -- which wait for the surface only -- are computed while the light records are in flight; and the workgroup barrier that
    // publishes the staged records waits for LDS only, not for outstanding global loads.
    SailorLightsGrid g; // Standard.shader:422-423
    if constexpr (TILE_LISTS) { g.offset = (uint32_t)bandTile * (uint32_t)KEEP; g.num = reinterpret_cast<const uint32_t*>(grid)[bandTile]; }
    else g = grid[bandTile];
    // unconditional surface loads (lanes outside the frame read pixel 0 of the band and are masked out of every ballot and of the store)
    // (streamed once: non-temporal, so that what every tile reads again -- lists, light records, the next frame's depth and masks -- keeps its
    // place in L2)
#define NT_LOAD4(P) make_float4(__builtin_nontemporal_load(&(P).x), __builtin_nontemporal_load(&(P).y), __builtin_nontemporal_load(&(P).z), __builtin_nontemporal_load(&(P).w))
    const float4 P0 = NT_LOAD4(surface[pix]);
    const float4 P1 = NT_LOAD4(surface[planeStride + pix]);
    const float4 P2 = NT_LOAD4(surface[2 * planeStride + pix]);
#undef NT_LOAD4
    if (ROLE == ROLE_BAND_TILE && g.num >= (uint32_t)A.splitMin) return; // a tile of the split blocks
    const uint32_t listNum = g.num < (uint32_t)KEEP ? g.num : (uint32_t)KEEP;
    const unsigned long long haveMask = __ballot((uint32_t)tid < listNum); // (the masks where the compares are: a flag carried across a branch as a bool is a VGPR 0 / 1)
    const bool haveLight = __builtin_amdgcn_inverse_ballot_w64(haveMask);
    uint32_t index = 0xFFFFFFFFu;
    if (haveLight) index = culled[g.offset + tid];
    // Standard.shader:430-433 "index == uint(-1) -> break" (and out-of-range guard): the list ends at the first such slot
    const unsigned long long stagedMask = __ballot(index < (uint32_t)A.lightsNum); // (a lane without a list slot holds uint(-1))
    const bool staged = __builtin_amdgcn_inverse_ballot_w64(stagedMask);
    float4 q0, q1, q2, q3, q4, q5, q6;
    // (two-wave blocks: wave 0 stages slots 0-63 now; wave 1's slots 64-127 wait for the second staging round -- their indices only decide where the list ends)
    const bool stager = !HALFT || __builtin_amdgcn_readfirstlane(wave) == 0;
    if (staged && stager) {
        if constexpr (PREPARED) {
            const float4* L = reinterpret_cast<const float4*>(lights) + (size_t)index * LREC;
            q0 = L[0]; q1 = L[1]; q2 = L[2]; q3 = L[3]; q4 = L[4];
        } else {
            const float4* L = reinterpret_cast<const float4*>(lights + index);
            q0 = L[0]; q1 = L[1]; q2 = L[2]; q3 = L[3]; q4 = L[4]; q5 = L[5]; q6 = L[6];
        }
    }

    // ---- per-pixel invariants (Standard.shader:379-401) ----
    // (x / y of every vector and two of the three colour channels as float pairs: see the pair pass)
    const float wx = P0.x, wy = P0.y, wz = P0.z;
    const v2f wxy = { wx, wy };
    const p2f wxyp = { wx, wy };
    const p2f nxy = { P1.x, P1.y };
    const float nx = nxy.x, ny = nxy.y, nz = P1.z, roughness = P1.w;
    float accX = 0.0f, accY = 0.0f, accZ = 0.0f;
    // The view vector, F0, kd * albedo and the Schlick-GGX terms of the view direction.  Computed HERE, while the light records are in flight, by
    // the kernels without shadow maps; by the K3 kernels only after the shadow look-ups of the tile's directional lights ("K3 first" below): the
    // look-up -- four 16-byte texels or eight PCF taps in flight -- and these 13 values are then never live together, and the K3 kernels fit the
    // 64 registers of 8 waves per SIMD like the others (was: 80 registers, 6 waves).
    p2f Loxy, F0xy, kdAxy;
    float Lox, Loy, Loz, cosLo, F0x, F0y, F0z, kdAx, kdAy, kdAz, k, oneMinusK, g1Lo;
    auto view_and_material = [&](const float camX, const float camY, const float camZ, const float one) {
        const float metallic = P2.w;
        const p2f vxy = wxyp - p2f{ camX, camY };
        const float vz = wz - camZ;
        const float vinv = rcp_of_sqrt(sqrt_exact(dot3_pk(vxy, vz, vxy, vz)));          // exact chain (see header)
        Loxy = -(vxy * vinv);                                                            // Lo = -viewDirection
        Lox = Loxy.x; Loy = Loxy.y; Loz = -(vz * vinv);
        cosLo = fmaxf(0.0f, dot3_pk(nxy, nz, Loxy, Loz));
        const float oneMinusMetal = one - metallic;
        // F0 = mix(0.04, albedo, metallic) and, below, F = F0 + (1 - F0) x5 in the oracle's own operations, unfused: the diffuse term is (1 - F) kd albedo,
        // and on a bright metal (F0 -> 1) a half-ulp difference in F is 1e-4 of 1 - F (scripts/fuzz_parity.py found the pixel: roughness 0, so no
        // specular term to hide it behind)
        const float dielectric = 0.04f * oneMinusMetal;
        F0xy = dielectric + p2f{ P2.x, P2.y } * metallic;
        F0x = F0xy.x; F0y = F0xy.y; F0z = dielectric + P2.z * metallic;
        kdAxy = oneMinusMetal * p2f{ P2.x, P2.y };                                      // kd = (1 - F)(1 - metallic)
        kdAx = kdAxy.x; kdAy = kdAxy.y; kdAz = oneMinusMetal * P2.z;
        const float rr = roughness + one;
        k = (rr * rr) * 0.125f; oneMinusK = one - k;
        g1Lo = cosLo * rcp_fast(fmaf(cosLo, oneMinusK, k)); // GeometrySchlickG1(cosLo, k)
    };
    constexpr bool K3_FIRST = HAS_CSM && !HAS_IBL; // (the ambient term keeps the view and material terms live to the very end: with them the K3 + IBL kernels need 91 registers one way, 136 the other)
    if constexpr (!K3_FIRST) view_and_material(A.camX, A.camY, A.camZ, 1.0f);
    {
        const unsigned long long bad = haveMask & ~stagedMask;
        if (lane == 0) sEnd[wave] = bad ? (uint32_t)(wave * 64 + __builtin_ctzll(bad)) : 0xFFFFFFFFu;
    }
    if (staged && stager) {
        // (staging is the block's critical path -- the other three waves wait at the barrier for the first; with PREPARED records it is a copy)
        float4* o = sL + tid * LREC;
        if constexpr (PREPARED) { o[0] = q0; o[1] = q1; o[2] = q2; o[3] = q3; o[4] = q4; }
        else {
            float4 o0, o1, o2, o3, o4;
            stage_light_record(q0, q1, q2, q3, q4, q5, q6, o0, o1, o2, o3, o4);
            o[0] = o0; o[1] = o1; o[2] = o2; o[3] = o3; o[4] = o4;
        }
    }
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); // the records are in LDS (global loads may stay in flight)
    const uint32_t numLightsAll = HALFT ? min(listNum, min(sEnd[0], sEnd[1])) : min(min(listNum, sEnd[0]), min(sEnd[1], min(sEnd[2], sEnd[3])));
    const float alpha = roughness * roughness, alphaSq = alpha * alpha;

    // ---- which lights can reach this quadrant at all?  One LANE per LIGHT against the bounding SPHERE of the quadrant's 64 surface
    // points: centre = the pixel in its middle (lane 27), radius^2 = the largest squared distance to it (one max-reduction over the
    // bits: a non-negative float orders like its bits, and a NaN wins and keeps every light).  |L - c| > sqrt(A) + R  =>  every pixel
    // has d^2 > A = r^2 (1 + 1e-5), i.e. fails the reach test of the per-pixel loop below, i.e. an exact-zero radius window.  (Was: the
    // quadrant's bounding box by six min / max reductions -- 12.1 instead of 13.3 surviving lights per quadrant on the 4K frame
    // (scripts/analysis/shade_trips.py), for three times the instructions.)
    const float scx = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(wx), 27));
    const float scy = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(wy), 27));
    const float scz = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(wz), 27));
    float sphereR;
    {
        const float ex = wx - scx, ey = wy - scy, ez = wz - scz;
        const float d2c = fmaf(ex, ex, fmaf(ey, ey, ez * ez));
        sphereR = __builtin_amdgcn_sqrtf(__uint_as_float(wave_max_u32(active ? __float_as_uint(d2c) : 0u))) * 1.0001f;
    }
    const unsigned long long forceMask = activeMask & __ballot(!(alphaSq > 0.0f)); // roughness 0: such pixels must see every light (0 * NaN)
    // ---- one staging round: the staged records -> this quadrant's lit pairs -> the sums.  Four-wave blocks stage the whole list at once and run this once;
    // a two-wave block runs it once per 64 list slots (a second time for the few tiles with more than 64 lights: the records of slots 64-127 are
    // staged by wave 1 behind a barrier that waits for both waves to be done with the first 64).
    uint32_t roundBase = 0u;
    for (;;) {
        const uint32_t numLights = HALFT ? min(numLightsAll - roundBase, 64u) : numLightsAll;
        // survivors by kind: [0,1] finite point lights, [2,3] finite spot lights, [4,5] the rest (directional, unknown type,
        // non-finite intensity: every pixel is a pair) -- for list slots 0..63 and 64..127
        unsigned long long seg[8] = { 0ull, 0ull, 0ull, 0ull, 0ull, 0ull, 0ull, 0ull }; // [6,7]: directional lights (type 0), see the loop behind the queue
    #pragma unroll
        for (int h = 0; h < HN; h++) {
            if ((uint32_t)(h * 64) >= numLights) break;
            // (every condition as a wave mask of ONE simple compare, the combinations as scalar mask arithmetic: a bool that is assigned in branches
            // lives in a VGPR as 0 / 1 and costs a v_cndmask + v_cmp per use)
            const uint32_t li = (uint32_t)(h * 64 + lane);
            const uint32_t lc = li < numLights ? li : 0u; // (lanes past the list read slot 0 and are masked out)
            const float4 c0 = sL[lc * LREC + 0];
            const float4 c1 = sL[lc * LREC + 1];
            const uint32_t bits = __float_as_uint(c1.w);
            const unsigned long long mIn = __ballot(li < numLights);
            const unsigned long long mFin = __ballot((bits & 0x10000u) != 0u);
            const unsigned long long mPoint = __ballot((bits & 0xFFu) == 1u), mSpot = __ballot((bits & 0xFFu) == 2u), mDir = __ballot((bits & 0xFFu) == 0u);
            const float ex = c0.x - scx, ey = c0.y - scy, ez = c0.z - scz;
            const float t = __builtin_amdgcn_sqrtf(c0.w) * 1.0001f + sphereR; // c0.w = r^2 (1 + 1e-5) (+inf: never reject)
            const float e2 = fmaf(ex, ex, fmaf(ey, ey, ez * ez));
            const unsigned long long mFar = __ballot(e2 > t * t); // only meaningful for finite point lights
            // A spot light whose cone misses the sphere: seen from the light the sphere spans the angle delta = asin(R / |e|) around the direction
            // to its centre, which makes the angle A with the cone's axis; no pixel can do better than cos(A - delta) = cosA cosd + sinA sind, and
            // the per-pixel test passes from c = cutOff.y - 1e-5 (= -rec0.w) up.  Approximate arithmetic (v_rsq / v_sqrt), hence the 1e-4; a NaN
            // anywhere (the light inside the sphere: sind > 1; a zero axis) fails the compares and keeps the light.  On the 4K frame 6.0 spot
            // lights per quadrant come this far, 1.6 reach a pixel, 3.8 pass this test (scripts/analysis/shade_trips.py).
            const float rinv = rsq_fast(e2);
            const float sind = sphereR * rinv;
            const float cosA = fmaf(ex, c1.x, fmaf(ey, c1.y, ez * c1.z)) * rinv;
            const float cosd = __builtin_amdgcn_sqrtf(fmaf(-sind, sind, 1.0f)), sinA = __builtin_amdgcn_sqrtf(fmaf(-cosA, cosA, 1.0f));
            const unsigned long long mOut = __ballot(fmaf(sinA, sind, cosA * cosd) < -c0.w - 1e-4f) & __ballot(cosA < cosd); // only meaningful for finite spot lights
            const unsigned long long dropped = forceMask == 0ull ? (mFin & ((mPoint & mFar) | (mSpot & mOut))) : 0ull;
            const unsigned long long all = mIn & ~dropped;
            seg[h] = all & mFin & mPoint;
            seg[2 + h] = all & mFin & mSpot;
            seg[6 + h] = mIn & mDir;
            seg[4 + h] = all & ~(seg[h] | seg[2 + h] | seg[6 + h]);
        }

        if (BAND && splitRole) { // this wave's share of the list: every fourth slot
            const unsigned long long share = 0x1111111111111111ull << __builtin_amdgcn_readfirstlane(wave); // (scalar: the masks stay in SGPRs)
    #pragma unroll
            for (int q = 0; q < 8; q++) seg[q] &= share;
        }

        // ---- directional lights (staged kind 0, with the odd point / spot light of stage_light_record): every pixel is a pair, so they are shaded one
        // LANE per PIXEL from the pixel's own registers -- no queue, no pulls -- in list order, nothing skipped (cosLi = 0 and non-finite intensities
        // take their natural course).  `shadow` = the light's K3 factor (Standard.shader:266-283), or the exact falloff of an odd light.
        auto shade_directional = [&](const float4* R, const float shadow) {
            const float4 r3 = R[3], r4 = R[4];
            // ---- Cook-Torrance (Standard.shader:309-340), as in the pair pass ----
            const float Lix = r3.x, Liy = r3.y, Liz = r3.z;
            float hx = Lix + Lox, hy = Liy + Loy, hz = Liz + Loz;
            const float hinv = rcp_of_sqrt(sqrt_exact(dot3f(hx, hy, hz, hx, hy, hz)));          // exact chain: Lh = normalize(Li + Lo)
            hx *= hinv; hy *= hinv; hz *= hinv;
            const float cosLi = fmaxf(0.0f, dot3f(nx, ny, nz, Lix, Liy, Liz));
            const float cosLh = fmaxf(0.0f, dot3f(nx, ny, nz, hx, hy, hz));
            const float x1 = 1.0f - fmaxf(0.0f, dot3f(hx, hy, hz, Lox, Loy, Loz));
            const float x2 = x1 * x1, x5 = x2 * x2 * x1;
            const float dn = (cosLh * cosLh) * (alphaSq - 1.0f) + 1.0f;
            const float D = alphaSq * rcp_fast(3.14159265359f * dn * dn);
            const float G = cosLi * rcp_fast(fmaf(cosLi, oneMinusK, k)) * g1Lo;
            const float spec = D * G * rcp_fast(fmaxf(0.00001f, 4.0f * cosLi * cosLo));
            const float scale = shadow * cosLi; // falloff = 1 (:287)
            const float Fx = F0x + (1.0f - F0x) * x5, Fy = F0y + (1.0f - F0y) * x5, Fz = F0z + (1.0f - F0z) * x5;
            accX += (fmaf(1.0f - Fx, kdAx, Fx * spec) * r4.x) * scale;
            accY += (fmaf(1.0f - Fy, kdAy, Fy * spec) * r4.y) * scale;
            accZ += (fmaf(1.0f - Fz, kdAz, Fz * spec) * r4.z) * scale;
        };
        // the factor of the light in list slot `slot`: its shadow look-up (K3), or the IEEE falloff of a light from far outside the staged reciprocal's range
        auto directional_factor = [&](const int slot) -> float {
            const float4* R = sL + (uint32_t)slot * LREC;
            const uint32_t lbits = __builtin_amdgcn_readfirstlane(__float_as_uint(R[1].w));
            if (__builtin_expect((lbits >> LIGHT_SLOW_SHIFT) != 0u, 0))
                return exact_falloff(std::true_type{}, (lbits >> LIGHT_SLOW_SHIFT) == 1u, R, R[0], R[1], R[3].w, wxyp, wz);
            if constexpr (HAS_CSM) {
                const float4 r3 = R[3];
                return directional_shadow(A, C, (lbits >> 8) & 0xFFu, -r3.x, -r3.y, -r3.z, nx, ny, nz, wx, wy, wz);
            }
            return 1.0f;
        };
        if constexpr (K3_FIRST) {
            // ---- K3 first: per directional light its factor (one register), THEN the view / material terms, then the light's Cook-Torrance term.
            // One round in practice (a tile with several directional lights repeats it and recomputes the terms: the camera position and the
            // constant 1 they are computed from are made opaque, so that the compiler cannot hoist them out of the loop and across the look-up).
            unsigned long long d0 = seg[6], d1 = seg[7];
            do {
                int slot = -1;
                float f = 1.0f;
                if ((d0 | d1) != 0ull) {
                    slot = d0 != 0ull ? __builtin_ctzll(d0) : 64 + __builtin_ctzll(d1);
                    if (d0 != 0ull) d0 &= d0 - 1ull; else d1 &= d1 - 1ull;
                    f = directional_factor(slot);
                }
                float camX = A.camX, camY = A.camY, camZ = A.camZ, one = 1.0f;
                asm volatile("" : "+s"(camX), "+s"(camY), "+s"(camZ), "+s"(one));
                view_and_material(camX, camY, camZ, one);
                if (slot >= 0) shade_directional(sL + (uint32_t)slot * LREC, f);
            } while ((d0 | d1) != 0ull);
        }

        // ---- 2 + 3. queue the (pixel, light) pairs that can be lit, then shade them one LANE per PAIR ----
        // Window = up to QMAX queued pairs, at most PENDK per pixel; a light whose pairs do not fit ends the window (it is
        // tested again in the next one -- rare: a quadrant of the 4K frame queues ~50 pairs).  A pair's result goes to slot
        // [colour][its position in the queue]; each pixel keeps the positions of its own pairs (7 bits each, in the order queued, under a
        // sentinel bit) and adds their results up afterwards: no atomics (ds_add_f32 is serialised per lane on this LDS: ~170 cycles
        // per wave instruction, scripts/microbench/lds_ops.hip).  (Was: slots [colour][ordinal of the pair among its pixel's][pixel] --
        // 9 KB per block instead of 6, a cap of three pairs per pixel and window instead of four, and an address of three instructions in the pair
        // pass instead of one.  More blocks per CU were NOT what it bought: with room for ten the kernel takes what it takes with eight -- the
        // 32 wave slots of a CU are the cap.)
        uint16_t* Q = sQ + wave * QMAX;
        const uint32_t qAddr = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) uint16_t*)Q; // its LDS byte address, for the hand-written append below
        const uint32_t sLAddr = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) float4*)sL;
        static_assert(LREC == 5, "SHADE_LIGHT_LOOP computes 80 x slot as (5 x slot) << 4");
        float* res = sRes + wave * (3 * QMAX); // this wave's [3 colours][QMAX queue positions] slots
        for (;;) {
            uint32_t cnt = 0u;      // queued pairs (wave-uniform)
            uint32_t pc = 1u;       // the queue positions of this pixel's pairs: 1 (sentinel), then 7 bits per pair, the first one queued on top
            // The wave is bound by instruction issue of every kind (a scalar instruction costs what a vector one costs: measured), and the loop
            // around a light is mostly scalar mask arithmetic.  In the usual quadrant -- every pixel inside the frame, none with roughness 0 -- the
            // "force" and "active" masks are the identity, so that case gets its own copy of the loops without them (PLAIN): m = reach & facing.
            auto light_loop = [&](auto plainTag, auto kindTag, auto hTag) -> bool {
                constexpr bool PLAIN = decltype(plainTag)::value;
                constexpr int kind = decltype(kindTag)::value, h = decltype(hTag)::value;
                unsigned long long todo = seg[kind * 2 + h];
                unsigned long long rest = 0ull; // on overflow: what is left, this light included
                if constexpr (PLAIN && kind < 2) {
                    // The usual case by hand (see SHADE_LIGHT_LOOP above): the same tests, the same append, 15 instructions around a light out of
                    // reach where the compiler's control flow takes 21.
                    if (todo != 0ull) {
                        uint32_t bit, n;
                        unsigned long long m, ex;
                        if constexpr (kind == 0 && h == 0) SHADE_LIGHT_LOOP("", "", SHADE_TEST_POINT);
                        if constexpr (kind == 0 && h == 1) SHADE_LIGHT_LOOP("s_or_b32 %[bit], %[bit], 64\n\t", "", SHADE_TEST_POINT);
                        if constexpr (kind == 1 && h == 0) SHADE_LIGHT_LOOP("", "ds_read_b96 v[60:62], v62 offset:16\n\t", SHADE_TEST_SPOT);
                        if constexpr (kind == 1 && h == 1) SHADE_LIGHT_LOOP("s_or_b32 %[bit], %[bit], 64\n\t", "ds_read_b96 v[60:62], v62 offset:16\n\t", SHADE_TEST_SPOT);
                    }
                } else {
                    // (one way out of the loop, through its condition: with a `break` in the middle the loop is no single-exit region of its own, falls
                    // into the region of the divergent pair pass below and is structurised along with it -- see the append)
                    while (todo) {
                        const int bit = __builtin_ctzll(todo);
                        const uint32_t s = (uint32_t)(h * 64 + bit);
                        unsigned long long m = activeMask; // "the rest": every pixel is a pair
                        if (kind < 2) {
                            const float4* R = sL + s * LREC;
                            const float4 r0 = R[0];
                            const v2f dxy = v2f{ r0.x, r0.y } - wxy;
                            const float dz = r0.z - wz;
                            const float d2 = fmaf(dxy.x, dxy.x, fmaf(dxy.y, dxy.y, dz * dz));
                            float v = d2;
                            if (kind == 1) {
                                // spot: falloff is exactly 0 iff theta < cutOff.y (:303-306); theta ~ dot(d, axis) / |d| to a few ulp
                                const float4 r1 = R[1];
                                v = -(fmaf(dxy.x, r1.x, fmaf(dxy.y, r1.y, dz * r1.z)) * rsq_fast(d2));
                            }
                            const unsigned long long reach = __ballot(!(v > r0.w));
                            m = 0ull;
                            if ((PLAIN ? reach : ((reach | forceMask) & activeMask)) != 0ull) {
                                // ... and facing it?  cosLi = max(0, n . Li) = 0 zeroes both the specular G term and the final product.
                                const float4 r3 = R[3];
                                const unsigned long long facing = __ballot(dot3f(nx, ny, nz, r3.x, r3.y, r3.z) > 0.0f);
                                m = PLAIN ? (reach & facing) : (((reach & facing) | forceMask) & activeMask);
                            }
                        }
                        if (m != 0ull) {
                            // (masks of single compares combined as scalars: a ballot of `mine && ...` goes through a VGPR 0 / 1 and back)
                            if (cnt + (uint32_t)__popcll(m) > (uint32_t)QMAX || (m & __ballot(pc >= (1u << (7 * PENDK)))) != 0ull) { rest = todo; todo = 0ull; continue; }
                            // The lanes of m append (s << 6 | lane) to the queue and note the position.  Written out with the exec mask set by hand: as
                            // `if (lane in m) { ... }` this is the only divergent branch of the loops around it, and with it the compiler
                            // structurises them -- a state variable, three more branches and five more scalar instructions per light.  Every
                            // lane is live here (the waves are full and nothing above has diverged), so exec goes back to all ones.
                            uint32_t t0, t1;
                            asm volatile("s_mov_b64 exec, %[m]\n\t"
                                         "v_mov_b32 %[t0], %[cnt]\n\t"
                                         "v_mbcnt_lo_u32_b32 %[t0], %[mlo], %[t0]\n\t"
                                         "v_mbcnt_hi_u32_b32 %[t0], %[mhi], %[t0]\n\t"   // the count so far rides in as mbcnt's addend
                                         "v_lshl_or_b32 %[pc], %[pc], 7, %[t0]\n\t"
                                         "v_lshl_add_u32 %[t0], %[t0], 1, %[q]\n\t"
                                         "v_lshl_or_b32 %[t1], %[s], 6, %[lane]\n\t"
                                         "ds_write_b16 %[t0], %[t1]\n\t"
                                         "s_mov_b64 exec, -1"
                                         : [t0] "=&v"(t0), [t1] "=&v"(t1), [pc] "+&v"(pc)
                                         : [m] "s"(m), [mlo] "s"((uint32_t)m), [mhi] "s"((uint32_t)(m >> 32)), [cnt] "s"(cnt), [s] "s"(s), [q] "v"(qAddr), [lane] "v"(lane)
                                         : "memory");
                            cnt += (uint32_t)__popcll(m);
                        }
                        asm("s_bitset0_b64 %0, %1" : "+s"(todo) : "s"(bit)); // todo &= todo - 1 in one scalar instruction instead of three
                    }
                }
                seg[kind * 2 + h] = rest; // what the next window still has to look at
                return rest != 0ull;
            };
            auto fill_window = [&](auto plainTag) -> bool {
                using K0 = std::integral_constant<int, 0>; using K1 = std::integral_constant<int, 1>; using K2 = std::integral_constant<int, 2>;
                if constexpr (HALFT) // (a staging round holds 64 records: there is no second half)
                    return light_loop(plainTag, K0{}, K0{}) || light_loop(plainTag, K1{}, K0{}) || light_loop(plainTag, K2{}, K0{});
                else
                    return light_loop(plainTag, K0{}, K0{}) || light_loop(plainTag, K0{}, K1{}) || light_loop(plainTag, K1{}, K0{}) || light_loop(plainTag, K1{}, K1{}) ||
                           light_loop(plainTag, K2{}, K0{}) || light_loop(plainTag, K2{}, K1{});
            };
            const bool overflow = (forceMask == 0ull && activeMask == ~0ull) ? fill_window(std::true_type{}) : fill_window(std::false_type{});
            if (cnt == 0u) break;
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            for (uint32_t base = 0u; base < cnt; base += 64u) {
                // one LANE per PAIR.  The pixel's invariants are pulled from its lane's registers (ds_bpermute: the LDS crossbar,
                // no LDS memory), the light record from LDS.  All 64 lanes execute the pulls (a disabled source lane returns 0).
                // (the lanes past the last pair repeat the batch's FIRST pair and store the result in their own slot, which no pixel looks at: every
                // lane runs the same straight code -- no `if (valid)` around the three stretches of arithmetic, which was 13 instructions of
                // exec-mask bookkeeping and zero-initialisation per batch)
                const uint32_t qi = base + (uint32_t)lane;
                const uint32_t e = Q[qi < cnt ? qi : base];
                constexpr bool valid = true;
                const int pa = (int)((e & 63u) << 2);
                const uint32_t s = (e >> 6) & (uint32_t)(KEEP - 1);
                const float4* R = sL + s * LREC;
    #define PULL(x) __int_as_float(__builtin_amdgcn_ds_bpermute(pa, __float_as_int(x)))
                // (float pairs where the arithmetic comes in pairs -- x / y of a vector, two of three colour channels: v_pk_{add,mul,fma}_f32 round each
                // component like the scalar instruction, so the bits are those of the scalar form, at one issue slot per pair instead of two.  The pulled
                // values and the LDS reads land in adjacent registers by construction, which is what the compiler's own SLP pairing could not arrange.)
                float falloff = 1.0f;
                const float4 r3 = R[3];
                const p2f pwxy = { PULL(wx), PULL(wy) };
                const float pwz = PULL(wz);
                if (valid) {
                    const float4 r0 = R[0], r1 = R[1];
                    const uint32_t type = __float_as_uint(r1.w) & 0xFFu;
                    if (type == 1u || type == 2u) // exact falloff (the oracle's op order where it is ill-conditioned)
                        falloff = exact_falloff(std::false_type{}, type == 1u, R, r0, r1, r3.w, pwxy, pwz);
                }
                // (the pulls are spread out so that at most ten pulled values are live at a time: 64 VGPRs = 8 waves per SIMD)
                float spec = 0.0f, x5 = 0.0f, scale = 0.0f;
                {
                    const p2f pnxy = { PULL(nx), PULL(ny) };
                    const float pnz = PULL(nz);
                    const p2f pLoxy = { PULL(Lox), PULL(Loy) };
                    const float pLoz = PULL(Loz);
                    const float pcosLo = PULL(cosLo), pg1Lo = PULL(g1Lo), palphaSq = PULL(alphaSq);
                    const float pkr = HAS_IBL ? PULL(roughness) : PULL(k); // the ambient term at the end needs the roughness itself: pull it, derive k
                    const float pk = HAS_IBL ? ((pkr + 1.0f) * (pkr + 1.0f)) * 0.125f : pkr;
                    if (valid) {
                        // ---- Cook-Torrance (Standard.shader:309-340) ----
                        const p2f Lixy = { r3.x, r3.y };
                        const float Liz = r3.z;
                        p2f hxy = Lixy + pLoxy;
                        float hz = Liz + pLoz;
                        const float hinv = rcp_of_sqrt(sqrt_exact(dot3_pk(hxy, hz, hxy, hz)));          // exact chain: Lh = normalize(Li + Lo)
                        hxy *= hinv; hz *= hinv;
                        const float cosLi = fmaxf(0.0f, dot3_pk(pnxy, pnz, Lixy, Liz));
                        const float cosLh = fmaxf(0.0f, dot3_pk(pnxy, pnz, hxy, hz));
                        const float x1 = 1.0f - fmaxf(0.0f, dot3_pk(hxy, hz, pLoxy, pLoz));
                        const float x2 = x1 * x1;
                        x5 = x2 * x2 * x1;                                                        // pow(1 - cosTheta, 5)
                        const float dn = (cosLh * cosLh) * (palphaSq - 1.0f) + 1.0f;                // exact: the cancelling denominator
                        const float D = palphaSq * rcp_fast(3.14159265359f * dn * dn);              // NdfGGX
                        const float G = cosLi * rcp_fast(fmaf(cosLi, 1.0f - pk, pk)) * pg1Lo;      // GeometrySchlickGGX
                        spec = D * G * rcp_fast(fmaxf(0.00001f, 4.0f * cosLi * pcosLo));
                        scale = cosLi * falloff; // (shadow = 1: only directional lights are shadowed, and they do not come through the queue)
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
                {
                    const p2f pF0xy = { PULL(F0x), PULL(F0y) };
                    const float pF0z = PULL(F0z);
                    const p2f pkdAxy = { PULL(kdAx), PULL(kdAy) };
                    const float pkdAz = PULL(kdAz);
                    if (valid) {
                        const p2f Fxy = pF0xy + (1.0f - pF0xy) * x5;
                        const float Fz = pF0z + (1.0f - pF0z) * x5;
                        const float4 r4 = R[4];
                        // shadow * ((kd*albedo + F*D*G/denom) * Lradiance * cosLi) * falloff ; kd = mix(1 - F, 0, metallic)
                        float* o = res + (base + (uint32_t)lane);
                        const p2f oxy = (fma2(1.0f - Fxy, pkdAxy, Fxy * spec) * p2f{ r4.x, r4.y }) * scale;
                        o[0] = oxy.x;
                        o[QMAX] = oxy.y;
                        o[2 * QMAX] = (fmaf(1.0f - Fz, pkdAz, Fz * spec) * r4.z) * scale;
                    }
                }
    #undef PULL
            }
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            // each pixel adds up the results of its own pairs, in the order they were queued (the position list shifted up against the sentinel:
            // the first pair's position then sits in bits 30..24, the next one in 23..17, ...)
            const uint32_t pcTop = pc << __builtin_clz(pc);
    #pragma unroll
            for (uint32_t j = 0; j < (uint32_t)PENDK; j++) {
                if (__ballot(pc >= (1u << (7u * (j + 1u)))) == 0ull) break;
                if (pc >= (1u << (7u * (j + 1u)))) {
                    const float* r = res + ((pcTop >> (24u - 7u * j)) & 127u);
                    accX += r[0];
                    accY += r[QMAX];
                    accZ += r[2 * QMAX];
                }
            }
            if (!overflow) break;
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        }
        if constexpr (!K3_FIRST) {
            if ((seg[6] | seg[7]) != 0ull) {
    #pragma unroll
                for (int h = 0; h < HN; h++) {
                    unsigned long long todo = seg[6 + h];
                    while (todo) {
                        const int slot = h * 64 + __builtin_ctzll(todo);
                        shade_directional(sL + (uint32_t)slot * LREC, directional_factor(slot));
                        todo &= todo - 1ull;
                    }
                }
            }
        }
        if constexpr (!HALFT) break;
        else {
            roundBase += 64u;
            if (roundBase >= numLightsAll) break;   // (block-uniform: both waves read the same numLightsAll)
            asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); // both waves are done with the records of the round before
            if (__builtin_amdgcn_readfirstlane(wave) == 1) {
                const uint32_t slot = roundBase + (uint32_t)lane;
                if (slot < numLightsAll) {
                    const uint32_t idx2 = culled[g.offset + slot];   // (read again: held from the prologue it would be a register live across the whole first round)
                    float4* o = sL + lane * LREC;
                    if constexpr (PREPARED) {
                        const float4* Lp = reinterpret_cast<const float4*>(lights) + (size_t)idx2 * LREC;
                        const float4 a0 = Lp[0], a1 = Lp[1], a2 = Lp[2], a3 = Lp[3], a4 = Lp[4];
                        o[0] = a0; o[1] = a1; o[2] = a2; o[3] = a3; o[4] = a4;
                    } else {
                        const float4* Lp = reinterpret_cast<const float4*>(lights + idx2);
                        float4 o0, o1, o2, o3, o4;
                        stage_light_record(Lp[0], Lp[1], Lp[2], Lp[3], Lp[4], Lp[5], Lp[6], o0, o1, o2, o3, o4);
                        o[0] = o0; o[1] = o1; o[2] = o2; o[3] = o3; o[4] = o4;
                    }
                }
            }
            asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
        }
    }
    if (BAND && splitRole) { // the four partial sums of each pixel: wave 0 + 1 + 2 + 3
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        float* res = sRes + wave * (3 * QMAX);
        res[lane] = accX; res[64 + lane] = accY; res[128 + lane] = accZ; // (in the wave's own slots: another wave may still be reading its own)
        __syncthreads();
        if (wave != 0) return;
#pragma unroll
        for (int w = 1; w < 4; w++) {
            accX += sRes[w * (3 * QMAX) + lane]; accY += sRes[w * (3 * QMAX) + 64 + lane]; accZ += sRes[w * (3 * QMAX) + 128 + lane];
        }
    }
    if (HAS_IBL) {
        float ambX, ambY, ambZ;
        // outColor.xyz = AmbientLighting(material, F0, Lr, normal, cosLo) (Standard.shader:425, :343-372); Lr = 2 cosLo n + viewDirection (:396)
        const float Lrx = fmaf(2.0f * cosLo, nx, -Lox), Lry = fmaf(2.0f * cosLo, ny, -Loy), Lrz = fmaf(2.0f * cosLo, nz, -Loz);
        // Evaluated LAST, when only the pixel's invariants and the three sums are live, and one texture at a time (scheduling
        // barriers): the pair pass sits exactly at 64 VGPRs, and three more live values across it -- or sixteen float4 gathers
        // in flight here -- push the kernel to ~100 VGPRs = half the occupancy (measured: 0.40 instead of 0.19 ms).
        const float x1 = 1.0f - cosLo, x2 = x1 * x1, x5 = x2 * x2 * x1;                                               // :352 FresnelSchlick(F0, cosLo)
        const float Fx = F0x + (1.0f - F0x) * x5, Fy = F0y + (1.0f - F0y) * x5, Fz = F0z + (1.0f - F0z) * x5;
        {
            int face; float cs, ct;
            cube_face_st(nx, ny, nz, face, cs, ct);
            const float4 irr = cube_sample_level(I.irradiance, I.irrSize, 0, face, cs, ct);                           // :346
            ambX = (1.0f - Fx) * kdAx * irr.x; ambY = (1.0f - Fy) * kdAy * irr.y; ambZ = (1.0f - Fz) * kdAz * irr.z;  // :358 kd albedo irradiance
        }
        __builtin_amdgcn_sched_barrier(0);
        const float2 dfg = lut_sample(I.brdfLut, I.lutW, I.lutH, cosLo, roughness);                                    // :365
        const float sx = fmaf(F0x, dfg.x, dfg.y), sy = fmaf(F0y, dfg.x, dfg.y), sz = fmaf(F0z, dfg.x, dfg.y);         // :368 F0 A + B
        __builtin_amdgcn_sched_barrier(0);
        {
            int face; float cs, ct;
            cube_face_st(Lrx, Lry, Lrz, face, cs, ct);
            const float maxLod = (float)(I.envLevels - 1);
            float lod = roughness * (float)I.envLevels;                                                               // :361-362
            lod = lod < 0.0f ? 0.0f : (lod > maxLod ? maxLod : lod);
            const float fl = floorf(lod), f = lod - fl;
            const int l0 = (int)fl, l1 = min(l0 + 1, I.envLevels - 1);
            const float4 a = cube_sample_level(I.env, I.envSize, l0, face, cs, ct);
            ambX = fmaf(sx * (1.0f - f), a.x, ambX); ambY = fmaf(sy * (1.0f - f), a.y, ambY); ambZ = fmaf(sz * (1.0f - f), a.z, ambZ);
            __builtin_amdgcn_sched_barrier(0);
            const float4 b = cube_sample_level(I.env, I.envSize, l1, face, cs, ct);
            ambX = fmaf(sx * f, b.x, ambX); ambY = fmaf(sy * f, b.y, ambY); ambZ = fmaf(sz * f, b.z, ambZ);
        }
        __builtin_amdgcn_sched_barrier(0);
        const float ao = I.ao ? I.ao[pix] : 1.0f;                                                                     // :386
        accX = fmaf(ambX, ao, accX); accY = fmaf(ambY, ao, accY); accZ = fmaf(ambZ, ao, accZ);                              // :371, :425 ambient + sum over lights
        __builtin_amdgcn_sched_barrier(0);
    }
    if (active) { // outColor.a = material.albedo.a (:438); written once and read by nobody here: non-temporal, like the surface loads
        // (the pixel's coordinates put together AGAIN from the lane's number -- two instructions -- and the scalars of the tile and the quadrant: held from
        // the prologue to here they are two registers live across the whole kernel, which at 64 registers and no scratch is two too many for the K3 kernels)
        int lane2 = (int)__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));
        asm volatile("" : "+v"(lane2));
        const int quadS = __builtin_amdgcn_readfirstlane(quad);
        const int gx2 = tx * TILE + (quadS & 1) * 8 + (lane2 & 7);
        const int py2 = A.H - 1 - (ty * TILE + (quadS >> 1) * 8 + (lane2 >> 3));
        float4* o = radiance + ((size_t)(py2 - A.fbRow0) * A.W + gx2);
        __builtin_nontemporal_store(accX, &o->x); __builtin_nontemporal_store(accY, &o->y);
        __builtin_nontemporal_store(accZ, &o->z); __builtin_nontemporal_store(P0.w, &o->w);
    }
}

