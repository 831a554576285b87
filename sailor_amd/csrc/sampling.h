// Canonical texture sampling shared by shade.hip (K3 shadow lookups, ambient / IBL term) and ibl_prefilter.hip: manual bilinear taps
// (texel centres at (i + 0.5) / size, clamp-to-edge), the Vulkan cube face table (ties z over y over x) and the cube mip chain
// layout -- level-major, then face, then size x size float4 texels.  Must match oracle/sailor_oracle.c bit for bit.
#pragma once
#include "common.h"

struct BilinearTaps { int x0, x1, y0, y1; float ax, ay; };

__device__ __forceinline__ BilinearTaps bilinear_taps(int W, int H, float u, float v)
{
    BilinearTaps t;
    const float x = u * (float)W - 0.5f, y = v * (float)H - 0.5f;
    const float fx = floorf(x), fy = floorf(y);
    t.ax = x - fx; t.ay = y - fy;
    int x0 = (int)fx, y0 = (int)fy, x1 = x0 + 1, y1 = y0 + 1;
    t.x0 = min(max(x0, 0), W - 1); t.x1 = min(max(x1, 0), W - 1);
    t.y0 = min(max(y0, 0), H - 1); t.y1 = min(max(y1, 0), H - 1);
    return t;
}

__device__ __forceinline__ float lerp2(float t00, float t10, float t01, float t11, float ax, float ay)
{
    const float top = t00 * (1.0f - ax) + t10 * ax;
    const float bot = t01 * (1.0f - ax) + t11 * ax;
    return top * (1.0f - ay) + bot * ay;
}

__device__ __forceinline__ void cube_face_st(float rx, float ry, float rz, int& face, float& s, float& t)
{
    const float ax = fabsf(rx), ay = fabsf(ry), az = fabsf(rz);
    float sc, tc, ma;
    if (az >= ax && az >= ay) { face = rz < 0.0f ? 5 : 4; sc = rz < 0.0f ? -rx : rx; tc = -ry; ma = az; }
    else if (ay >= ax)        { face = ry < 0.0f ? 3 : 2; sc = rx; tc = ry < 0.0f ? -rz : rz; ma = ay; }
    else                      { face = rx < 0.0f ? 1 : 0; sc = rx < 0.0f ? rz : -rz; tc = -ry; ma = ax; }
    s = 0.5f * (sc / ma + 1.0f);
    t = 0.5f * (tc / ma + 1.0f);
}

__device__ __forceinline__ float4 bilinear_f4(const float4* __restrict__ tex, int size, float s, float t)
{
    const BilinearTaps b = bilinear_taps(size, size, s, t);
    const float4 a = tex[(size_t)b.y0 * size + b.x0], c = tex[(size_t)b.y0 * size + b.x1];
    const float4 d = tex[(size_t)b.y1 * size + b.x0], e = tex[(size_t)b.y1 * size + b.x1];
    return make_float4(lerp2(a.x, c.x, d.x, e.x, b.ax, b.ay), lerp2(a.y, c.y, d.y, e.y, b.ax, b.ay),
                       lerp2(a.z, c.z, d.z, e.z, b.ax, b.ay), lerp2(a.w, c.w, d.w, e.w, b.ax, b.ay));
}

__device__ __forceinline__ float4 cube_sample_level(const float4* __restrict__ cube, int size0, int level, int face, float s, float t)
{
    size_t off = 0;
    for (int l = 0; l < level; l++) { const int sz = max(size0 >> l, 1); off += (size_t)6 * sz * sz; }
    const int size = max(size0 >> level, 1);
    return bilinear_f4(cube + off + (size_t)face * size * size, size, s, t);
}

__device__ __forceinline__ float4 cube_sample_lod(const float4* __restrict__ cube, int size0, int levels, float rx, float ry, float rz, float lod)
{
    int face; float s, t;
    cube_face_st(rx, ry, rz, face, s, t);
    const float maxLod = (float)(levels - 1);
    lod = lod < 0.0f ? 0.0f : (lod > maxLod ? maxLod : lod);
    const float fl = floorf(lod);
    const int l0 = (int)fl, l1 = min(l0 + 1, levels - 1);
    const float f = lod - fl;
    const float4 a = cube_sample_level(cube, size0, l0, face, s, t), b = cube_sample_level(cube, size0, l1, face, s, t);
    return make_float4(a.x * (1.0f - f) + b.x * f, a.y * (1.0f - f) + b.y * f, a.z * (1.0f - f) + b.z * f, a.w * (1.0f - f) + b.w * f);
}

