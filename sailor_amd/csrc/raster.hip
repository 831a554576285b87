// The shadow-map producer: ShadowPrepassNode's caster draws (FrameGraph/ShadowPrepassNode.cpp:219-262, Content/Shaders/ShadowCaster.shader) as a
// compute rasteriser for gfx950.  Semantics = oracle/sailor_oracle.c oracle_raster_depth (Vulkan's rasterisation rules with the freedoms pinned:
// 1/256-pixel snapping, 64-bit integer edge functions, both windings, top-left rule, unfused fp32 depth interpolation, GREATER against a depth
// buffer cleared to 0 -- reversed Z).  The winning depth of a texel does not depend on the order the fragments arrive in, so the depth buffer is
// a plain atomicMax over float bits (depths are > 0) and the result is bit-exact against the sequential oracle.
//
// Work distribution: one LANE per (instance, triangle) sets its triangle up.  Triangles whose pixel box is small are filled by their own lane;
// the others are handed round the wave one after the other (the set-up travels by lane broadcast): the wave looks at the box as 64 x 64-texel
// superblocks, one per lane, then at the 8 x 8-texel BLOCKS of each surviving superblock, one per lane -- a (super)block entirely outside an edge is
// dropped, and so is one whose coarse depth says nothing of this triangle can still win there -- and fills the surviving blocks one lane per texel.
//
// Coarse depth (optional workspace, one word per 8 x 8 block and, behind those, one per 64 x 64 superblock): a LOWER BOUND of every depth stored there.  A triangle that covers a whole
// block raises it to the smallest depth it wrote there; a triangle (or a block of one) whose largest possible depth does not exceed it is
// skipped.  Bounds only -- stale values are merely less effective -- so the depth buffer is the same with and without it; shadow casters overdraw
// each texel hundreds of times (every box along the light direction lands on it), and this is what makes the passes finish.
#include "common.h"
#include <hip/hip_fp16.h>

#define RASTER_SMALL_BOX 64 // pixels a lane fills on its own
#ifdef RASTER_STATS
__device__ unsigned long long gStats[8];
#define STAT(i, v) atomicAdd(&gStats[i], (unsigned long long)(v))
#else
#define STAT(i, v)
#endif

struct RasterTri { long long x0, y0, x1, y1, x2, y2; float z0, z1, z2; int i0, i1, j0, j1; bool valid; };

// clip = (lightMatrix * model) * vec4(p, 1): GLSL order
__device__ __forceinline__ Mat4 raster_mul(const Mat4& a, const float* __restrict__ b)
{
    Mat4 o;
#pragma unroll
    for (int j = 0; j < 4; j++) {
        const float4 c = glsl_mul(a, b[4 * j + 0], b[4 * j + 1], b[4 * j + 2], b[4 * j + 3]);
        o.m[4 * j + 0] = c.x; o.m[4 * j + 1] = c.y; o.m[4 * j + 2] = c.z; o.m[4 * j + 3] = c.w;
    }
    return o;
}

__device__ __forceinline__ long long raster_floor_div256(long long a) { return a >= 0 ? a / 256 : -((-a + 255) / 256); }

// A vertex beyond the near plane of the reversed depth range (z_clip > w_clip, which includes everything behind the eye) cannot be projected:
// such a triangle is cut against w - z = 0 in clip space first.  The new vertex on an edge is always computed from its inside end,
// P = I + (O - I) * (dI / (dI - dO)) with d = w - z, one rounding per operation.  One vertex inside (A; B, C follow it in the triangle's
// own order): the triangle (A, AB, AC).  Two inside (A, B; C outside follows them): the quad A, B, BC, AC as the triangles
// (A, B, BC) = part 0 and (A, BC, AC) = part 1.  Triangles wholly inside are untouched (their fragments beyond the plane fail z <= 1).
__device__ __forceinline__ float4 raster_cut(const float4& I, float dI, const float4& O, float dO)
{
    const float t = dI / (dI - dO);
    return make_float4(I.x + (O.x - I.x) * t, I.y + (O.y - I.y) * t, I.z + (O.z - I.z) * t, I.w + (O.w - I.w) * t);
}

// hasView: clip = projection * (view * (model * position)) (DepthOnly.shader:51, LM = projection); else clip = (lightMatrix * model) * position
__device__ __forceinline__ RasterTri raster_setup(const Mat4& LM, bool hasView, const Mat4& V, const float* __restrict__ model, const float* __restrict__ positions,
                                                   const uint32_t* __restrict__ tri, int W, int H, bool cullBack, int part, bool& hasSecond)
{
    RasterTri t;
    t.valid = false;
    hasSecond = false;
    float4 c[3];
#pragma unroll
    for (int k = 0; k < 3; k++) {
        const float* p = positions + 3 * (size_t)tri[k];
        if (hasView) {
            Mat4 M;
#pragma unroll
            for (int q = 0; q < 16; q++) M.m[q] = model[q];
            const float4 a = glsl_mul(M, p[0], p[1], p[2], 1.0f);
            const float4 bq = glsl_mul(V, a.x, a.y, a.z, a.w);
            c[k] = glsl_mul(LM, bq.x, bq.y, bq.z, bq.w);
        } else c[k] = glsl_mul(LM, p[0], p[1], p[2], 1.0f);
    }
    const float d0 = c[0].w - c[0].z, d1 = c[1].w - c[1].z, d2 = c[2].w - c[2].z;
    const int mask = (d0 >= 0.0f ? 1 : 0) | (d1 >= 0.0f ? 2 : 0) | (d2 >= 0.0f ? 4 : 0);
    if (mask == 0) return t;
    if (mask != 7) {
        const bool one = (mask & (mask - 1)) == 0;
        const int r = one ? (mask == 1 ? 0 : (mask == 2 ? 1 : 2)) : (mask == 6 ? 1 : (mask == 5 ? 2 : 0)); // the rotation that brings A to the front
        const float4 A = r == 0 ? c[0] : (r == 1 ? c[1] : c[2]), B = r == 0 ? c[1] : (r == 1 ? c[2] : c[0]), C = r == 0 ? c[2] : (r == 1 ? c[0] : c[1]);
        const float dA = r == 0 ? d0 : (r == 1 ? d1 : d2), dB = r == 0 ? d1 : (r == 1 ? d2 : d0), dC = r == 0 ? d2 : (r == 1 ? d0 : d1);
        if (one) {
            if (part) return t;
            c[0] = A; c[1] = raster_cut(A, dA, B, dB); c[2] = raster_cut(A, dA, C, dC);
        } else {
            const float4 BC = raster_cut(B, dB, C, dC);
            hasSecond = true;
            c[0] = A;
            if (part == 0) { c[1] = B; c[2] = BC; }
            else { c[1] = BC; c[2] = raster_cut(A, dA, C, dC); }
        }
    } else if (part) return t;
    long long X[3], Y[3];
    float Z[3];
#pragma unroll
    for (int k = 0; k < 3; k++) {
        const float4 clip = c[k];
        if (!(clip.w > 0.0f)) return t;
        const float nx = clip.x / clip.w, ny = clip.y / clip.w, nz = clip.z / clip.w;
        const float xf = (nx + 1.0f) * ((float)W * 0.5f);
        const float yf = (ny + 1.0f) * ((float)H * -0.5f) + (float)H;
        const float sx = xf * 256.0f, sy = yf * 256.0f;
        if (!(fabsf(sx) < 1.0e9f) || !(fabsf(sy) < 1.0e9f)) return t;
        X[k] = (long long)rintf(sx); Y[k] = (long long)rintf(sy); Z[k] = nz;
    }
    const long long area2 = (X[1] - X[0]) * (Y[2] - Y[0]) - (X[2] - X[0]) * (Y[1] - Y[0]);
    if (area2 == 0) return t;
    if (cullBack && area2 > 0) return t; // Vulkan's signed area is -area2 / 2, front = counter-clockwise = positive (see the oracle)
    if (area2 < 0) { long long s = X[1]; X[1] = X[2]; X[2] = s; s = Y[1]; Y[1] = Y[2]; Y[2] = s; const float z = Z[1]; Z[1] = Z[2]; Z[2] = z; }
    t.x0 = X[0]; t.y0 = Y[0]; t.x1 = X[1]; t.y1 = Y[1]; t.x2 = X[2]; t.y2 = Y[2];
    t.z0 = Z[0]; t.z1 = Z[1]; t.z2 = Z[2];
    const long long minx = min(X[0], min(X[1], X[2])), maxx = max(X[0], max(X[1], X[2]));
    const long long miny = min(Y[0], min(Y[1], Y[2])), maxy = max(Y[0], max(Y[1], Y[2]));
    long long i0 = raster_floor_div256(minx - 128 + 255), i1 = raster_floor_div256(maxx - 128);
    long long j0 = raster_floor_div256(miny - 128 + 255), j1 = raster_floor_div256(maxy - 128);
    if (i0 < 0) i0 = 0;
    if (j0 < 0) j0 = 0;
    if (i1 > W - 1) i1 = W - 1;
    if (j1 > H - 1) j1 = H - 1;
    if (i1 < i0 || j1 < j0) return t;
    t.i0 = (int)i0; t.i1 = (int)i1; t.j0 = (int)j0; t.j1 = (int)j1;
    t.valid = true;
    return t;
}

__device__ __forceinline__ long long raster_edge(long long ax, long long ay, long long bx, long long by, long long px, long long py)
{
    return (bx - ax) * (py - ay) - (by - ay) * (px - ax);
}
__device__ __forceinline__ bool raster_top_left(long long ax, long long ay, long long bx, long long by)
{
    const long long dx = bx - ax, dy = by - ay;
    return (dy == 0 && dx > 0) || dy < 0;
}

__device__ __forceinline__ void raster_pixel(const RasterTri& t, float area, bool tl0, bool tl1, bool tl2, int i, int j, int W, unsigned int* __restrict__ depthBits)
{
    const long long px = 256ll * i + 128, py = 256ll * j + 128;
    const long long e0 = raster_edge(t.x1, t.y1, t.x2, t.y2, px, py), e1 = raster_edge(t.x2, t.y2, t.x0, t.y0, px, py), e2 = raster_edge(t.x0, t.y0, t.x1, t.y1, px, py);
    if (e0 < 0 || e1 < 0 || e2 < 0) return;
    if ((e0 == 0 && !tl0) || (e1 == 0 && !tl1) || (e2 == 0 && !tl2)) return;
    const float w1 = (float)e1 / area, w2 = (float)e2 / area;
    const float z = (t.z0 + (t.z1 - t.z0) * w1) + (t.z2 - t.z0) * w2;
    if (!(z > 0.0f && z <= 1.0f)) return; // z == 0 never passes GREATER against the cleared 0 either
    atomicMax(depthBits + (size_t)j * W + i, __float_as_uint(z)); // positive floats order like their bits
}

__device__ __forceinline__ long long bcast64(long long v, int src)
{
    const int lo = __shfl((int)(unsigned int)(unsigned long long)v, src, 64), hi = __shfl((int)((unsigned long long)v >> 32), src, 64);
    return (long long)(((unsigned long long)(unsigned int)hi << 32) | (unsigned int)lo);
}

#define RASTER_Z_MARGIN 1.0e-6f // covers the rounding of the per-texel interpolation when a bound is derived from other points of the same plane

// z of the triangle's plane at texel (i, j), the per-texel formula (edge functions need not be non-negative here)
__device__ __forceinline__ float raster_plane_z(const RasterTri& t, float area, int i, int j)
{
    const long long px = 256ll * i + 128, py = 256ll * j + 128;
    const long long e1 = raster_edge(t.x2, t.y2, t.x0, t.y0, px, py), e2 = raster_edge(t.x0, t.y0, t.x1, t.y1, px, py);
    return (t.z0 + (t.z1 - t.z0) * ((float)e1 / area)) + (t.z2 - t.z0) * ((float)e2 / area);
}

// Bounds over the texel rectangle [xa, xb] x [ya, yb]: an edge function and the depth plane are affine, so their extremes sit at corner texels.
// m* = largest value of each edge function (all >= 0 <=> the rectangle may touch the triangle), n* = smallest (all > 0 <=> it lies inside it).
__device__ __forceinline__ void raster_box_bounds(const RasterTri& t, float area, int xa, int ya, int xb, int yb, long long& m0, long long& m1, long long& m2,
                                                  long long& n0, long long& n1, long long& n2, float& zhi, float& zlo)
{
    m0 = m1 = m2 = -0x7FFFFFFFFFFFFFFFll; n0 = n1 = n2 = 0x7FFFFFFFFFFFFFFFll;
    zhi = -3.0e38f; zlo = 3.0e38f;
#pragma unroll
    for (int c = 0; c < 4; c++) {
        const int cx = (c & 1) ? xb : xa, cy = (c & 2) ? yb : ya;
        const long long px = 256ll * cx + 128, py = 256ll * cy + 128;
        const long long e0 = raster_edge(t.x1, t.y1, t.x2, t.y2, px, py), e1 = raster_edge(t.x2, t.y2, t.x0, t.y0, px, py), e2 = raster_edge(t.x0, t.y0, t.x1, t.y1, px, py);
        m0 = max(m0, e0); m1 = max(m1, e1); m2 = max(m2, e2);
        n0 = min(n0, e0); n1 = min(n1, e1); n2 = min(n2, e2);
        const float z = (t.z0 + (t.z1 - t.z0) * ((float)e1 / area)) + (t.z2 - t.z0) * ((float)e2 / area);
        zhi = fmaxf(zhi, z); zlo = fminf(zlo, z);
    }
}

__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(4, 4))) void k_raster_depth(Mat4 L, Mat4 V, int hasView, int cullBack, const float* __restrict__ positions, const uint32_t* __restrict__ indices, uint32_t numTriangles,
                                                       const float* __restrict__ models, const uint32_t* __restrict__ instanceIds, uint32_t numDrawn, int W, int H,
                                                       unsigned int* __restrict__ depthBits, unsigned int* __restrict__ coarse)
{
    const unsigned long long total = (unsigned long long)numDrawn * numTriangles;
    const unsigned long long id = (unsigned long long)blockIdx.x * 256 + threadIdx.x;
    const int lane = threadIdx.x & 63;
    const int CW = (W + 7) >> 3, SW = (W + 63) >> 6;
    unsigned int* coarse2 = coarse ? coarse + (size_t)CW * ((H + 7) >> 3) : nullptr; // level 2 behind level 1 in the same workspace
    // a triangle cut by the near plane can leave a quad: its second half is a second trip through the same code, taken only by waves that hold one
    bool again = false;
    for (int part = 0; part < 2; part++) {
        if (part && !__any(again)) break;
        RasterTri t;
        t.valid = false;
        bool second = false;
        if (id < total) {
            const uint32_t d = (uint32_t)(id / numTriangles), tri = (uint32_t)(id - (unsigned long long)d * numTriangles);
            const uint32_t inst = instanceIds ? instanceIds[d] : d;
            const Mat4 LM = hasView ? L : raster_mul(L, models + 16 * (size_t)inst);
            t = raster_setup(LM, hasView != 0, V, models + 16 * (size_t)inst, positions, indices + 3 * (size_t)tri, W, H, cullBack != 0, part, second);
        }
        const float zmaxTri = fmaxf(t.z0, fmaxf(t.z1, t.z2)) + RASTER_Z_MARGIN; // inside the triangle z is a convex combination of the vertices'
        if (t.valid && !(zmaxTri > 0.0f)) t.valid = false;                        // nothing of it can pass z > 0
        if (t.valid && fminf(t.z0, fminf(t.z1, t.z2)) - RASTER_Z_MARGIN > 1.0f) t.valid = false; // ... or z <= 1
        const bool small = t.valid && (long long)(t.i1 - t.i0 + 1) * (t.j1 - t.j0 + 1) <= RASTER_SMALL_BOX;
        if (small) {
            bool hidden = false;
            if (coarse && (t.i0 >> 3) == (t.i1 >> 3) && (t.j0 >> 3) == (t.j1 >> 3))
                hidden = zmaxTri <= __uint_as_float(coarse[(size_t)(t.j0 >> 3) * CW + (t.i0 >> 3)]);
            if (!hidden) {
                const float area = (float)raster_edge(t.x0, t.y0, t.x1, t.y1, t.x2, t.y2);
                const bool tl0 = raster_top_left(t.x1, t.y1, t.x2, t.y2), tl1 = raster_top_left(t.x2, t.y2, t.x0, t.y0), tl2 = raster_top_left(t.x0, t.y0, t.x1, t.y1);
                for (int j = t.j0; j <= t.j1; j++)
                    for (int i = t.i0; i <= t.i1; i++) raster_pixel(t, area, tl0, tl1, tl2, i, j, W, depthBits);
            }
        }
        // the large ones: the whole wave on one triangle at a time
        unsigned long long todo = __ballot(t.valid && !small);
        while (todo) {
            const int src = __builtin_ctzll(todo);
            todo &= todo - 1ull;
            RasterTri b;
            b.x0 = bcast64(t.x0, src); b.y0 = bcast64(t.y0, src); b.x1 = bcast64(t.x1, src); b.y1 = bcast64(t.y1, src); b.x2 = bcast64(t.x2, src); b.y2 = bcast64(t.y2, src);
            b.z0 = __shfl(t.z0, src, 64); b.z1 = __shfl(t.z1, src, 64); b.z2 = __shfl(t.z2, src, 64);
            b.i0 = __shfl(t.i0, src, 64); b.i1 = __shfl(t.i1, src, 64); b.j0 = __shfl(t.j0, src, 64); b.j1 = __shfl(t.j1, src, 64);
            const float zmaxB = __shfl(zmaxTri, src, 64);
            const float area = (float)raster_edge(b.x0, b.y0, b.x1, b.y1, b.x2, b.y2);
            const bool tl0 = raster_top_left(b.x1, b.y1, b.x2, b.y2), tl1 = raster_top_left(b.x2, b.y2, b.x0, b.y0), tl2 = raster_top_left(b.x0, b.y0, b.x1, b.y1);
            // ---- level 2: 64 x 64-texel superblocks, one per lane ----
            const int si0 = b.i0 >> 6, sj0 = b.j0 >> 6, sw = (b.i1 >> 6) - si0 + 1, sh = (b.j1 >> 6) - sj0 + 1;
            const int ns = sw * sh;
            for (int sbase = 0; sbase < ns; sbase += 64) {
                const int sblk = sbase + lane;
                bool salive = sblk < ns;
                int si = 0, sj = 0;
                if (salive) {
                    sj = sblk / sw; si = sblk - sj * sw;
                    si += si0; sj += sj0;
                    long long m0, m1, m2, n0, n1, n2;
                    float zhi, zlo;
                    raster_box_bounds(b, area, si * 64, sj * 64, si * 64 + 63, sj * 64 + 63, m0, m1, m2, n0, n1, n2, zhi, zlo);
                    salive = m0 >= 0 && m1 >= 0 && m2 >= 0 && zhi + RASTER_Z_MARGIN > 0.0f && !(zlo - RASTER_Z_MARGIN > 1.0f); // ... and not clipped away as a whole
                    if (salive && coarse2) {
                        unsigned int* c2 = coarse2 + (size_t)sj * SW + si;
                        salive = fminf(zmaxB, zhi + RASTER_Z_MARGIN) > __uint_as_float(*c2);
                        // the triangle covers the whole superblock (every corner texel strictly inside every edge) with depths in (0, 1]: once its
                        // texels are written, nothing below the smallest of them can win anywhere in the superblock
                        if (salive && n0 > 0 && n1 > 0 && n2 > 0 && zlo - RASTER_Z_MARGIN > 0.0f && zhi + RASTER_Z_MARGIN <= 1.0f && si * 64 + 63 < W && sj * 64 + 63 < H)
                            atomicMax(c2, __float_as_uint(zlo - RASTER_Z_MARGIN));
                    }
                }
                unsigned long long slive = __ballot(salive);
                if (lane == 0) { STAT(0, min(64, ns - sbase)); STAT(1, __popcll(slive)); }
                while (slive) {
                    const int s1 = __builtin_ctzll(slive);
                    slive &= slive - 1ull;
                    const int csi = __shfl(si, s1, 64), csj = __shfl(sj, s1, 64);
                    // ---- level 1: the superblock's 8 x 8 blocks, one per lane ----
                    const int bi = csi * 8 + (lane & 7), bj = csj * 8 + (lane >> 3);
                    bool alive = bi >= (b.i0 >> 3) && bi <= (b.i1 >> 3) && bj >= (b.j0 >> 3) && bj <= (b.j1 >> 3);
                    float c1 = 3.0e38f; // this block's coarse depth (blocks beyond the map do not exist)
                    if (coarse && bi < CW && bj < ((H + 7) >> 3)) c1 = __uint_as_float(coarse[(size_t)bj * CW + bi]);
                    if (alive) {
                        long long m0, m1, m2, n0, n1, n2;
                        float zhi, zlo;
                        raster_box_bounds(b, area, bi * 8, bj * 8, bi * 8 + 7, bj * 8 + 7, m0, m1, m2, n0, n1, n2, zhi, zlo);
                        alive = m0 >= 0 && m1 >= 0 && m2 >= 0 && zhi + RASTER_Z_MARGIN > 0.0f && !(zlo - RASTER_Z_MARGIN > 1.0f);
                        if (alive && coarse) alive = fminf(zmaxB, zhi + RASTER_Z_MARGIN) > c1;
                    }
                    if (coarse2) { // the smallest of the 64 block bounds is a bound for the superblock: keeps level 2 as tight as level 1 has become
                        float cmin = c1;
    #pragma unroll
                        for (int d = 32; d > 0; d >>= 1) cmin = fminf(cmin, __shfl_xor(cmin, d, 64));
                        if (lane == 0 && cmin > 0.0f && cmin < 3.0e38f) atomicMax(coarse2 + (size_t)csj * SW + csi, __float_as_uint(cmin));
                    }
                    unsigned long long live = __ballot(alive);
                    if (lane == 0) { STAT(2, 64); STAT(3, __popcll(live)); }
                    // ---- the surviving blocks, one lane per texel ----
                    while (live) {
                        const int s2 = __builtin_ctzll(live);
                        live &= live - 1ull;
                        const int cbi = csi * 8 + (s2 & 7), cbj = csj * 8 + (s2 >> 3);
                        const int i = cbi * 8 + (lane & 7), j = cbj * 8 + (lane >> 3);
                        bool wrote = false;
                        float z = 2.0f;
                        if (i >= b.i0 && i <= b.i1 && j >= b.j0 && j <= b.j1) {
                            const long long px = 256ll * i + 128, py = 256ll * j + 128;
                            const long long e0 = raster_edge(b.x1, b.y1, b.x2, b.y2, px, py), e1 = raster_edge(b.x2, b.y2, b.x0, b.y0, px, py),
                                            e2 = raster_edge(b.x0, b.y0, b.x1, b.y1, px, py);
                            const bool in = !(e0 < 0 || e1 < 0 || e2 < 0) && !((e0 == 0 && !tl0) || (e1 == 0 && !tl1) || (e2 == 0 && !tl2));
                            if (in) {
                                z = (b.z0 + (b.z1 - b.z0) * ((float)e1 / area)) + (b.z2 - b.z0) * ((float)e2 / area);
                                if (z > 0.0f && z <= 1.0f) {
                                    atomicMax(depthBits + (size_t)j * W + i, __float_as_uint(z)); // positive floats order like their bits
                                    wrote = true;
                                }
                            }
                        }
    #ifdef RASTER_STATS
                        { const unsigned long long wb = __ballot(wrote), ib = __ballot(z < 2.0f); if (lane == 0) { STAT(4, __popcll(ib)); STAT(5, __popcll(wb)); STAT(6, wb == ~0ull); } }
    #endif
                        if (coarse && __ballot(wrote) == ~0ull) { // the whole block now holds depths >= the smallest one written here
                            float zmin = z;
    #pragma unroll
                            for (int d = 32; d > 0; d >>= 1) zmin = fminf(zmin, __shfl_xor(zmin, d, 64));
                            if (lane == 0) atomicMax(coarse + (size_t)cbj * CW + cbi, __float_as_uint(zmin));
                        }
                    }
                }
            }
        }
        if (part == 0) again = second;
    }
}

// ShadowCaster.shader:66-78 on the winning depth; canonical exp == shade.hip / the oracle (polynomial, no fused operations)
__device__ __forceinline__ float raster_expf(float x)
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

__global__ __launch_bounds__(256) void k_shadow_resolve(const float* __restrict__ depth, size_t texels, int format, void* __restrict__ out)
{
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= texels) return;
    const float z = depth[i];
    if (format == SAILOR_SHADOWMAP_R32G32B32A32_SFLOAT) {
        float4 m = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
        if (z > 0.0f) {
            m.x = raster_expf(40.0f * z); m.y = m.x * m.x;
            m.z = -raster_expf(-40.0f * z); m.w = m.z * m.z;
        }
        reinterpret_cast<float4*>(out)[i] = m;
    } else if (format == SAILOR_SHADOWMAP_R16_SFLOAT) {
        reinterpret_cast<__half*>(out)[i] = __float2half_rn(z);
    } else {
        reinterpret_cast<float*>(out)[i] = z;
    }
}

extern "C" {

#ifdef RASTER_STATS
__attribute__((visibility("default"))) void sailor_hip_raster_stats(unsigned long long* out, int reset)
{
    hipMemcpyFromSymbol(out, HIP_SYMBOL(gStats), 64);
    if (reset) { unsigned long long z[8] = {0}; hipMemcpyToSymbol(HIP_SYMBOL(gStats), z, 64); }
}
#endif

size_t sailor_hip_raster_coarse_words(int32_t width, int32_t height)
{
    if (width <= 0 || height <= 0) return 0;
    return (size_t)((width + 7) / 8) * ((height + 7) / 8) + (size_t)((width + 63) / 64) * ((height + 63) / 64);
}

static int raster_depth_launch(SailorHipContext* ctx, const float* lightMatrix, const float* viewMatrix, const float* dPositions, const uint32_t* dIndices,
                               uint32_t numTriangles, const float* dModels, const uint32_t* dInstanceIds, uint32_t numDrawn, int32_t width, int32_t height, float* dDepth,
                               uint32_t flags, uint32_t* dCoarseDepth)
{
    const bool clear = (flags & SAILOR_RASTER_CLEAR) != 0;
    if (!ctx || !lightMatrix || !dDepth || width <= 0 || height <= 0 || width > 32768 || height > 32768) return SAILOR_HIP_ERR_INVALID_ARGUMENT;
    SAILOR_TRY_HIP(ctx, hipSetDevice(ctx->device)); // a host thread may drive several contexts
    if (clear) {
        SAILOR_TRY_HIP(ctx, hipMemsetAsync(dDepth, 0, (size_t)width * height * 4, ctx->stream));
        if (dCoarseDepth)
            SAILOR_TRY_HIP(ctx, hipMemsetAsync(dCoarseDepth, 0, ((size_t)((width + 7) / 8) * ((height + 7) / 8) + (size_t)((width + 63) / 64) * ((height + 63) / 64)) * 4, ctx->stream));
    }
    if (numTriangles == 0 || numDrawn == 0) return SAILOR_HIP_OK;
    if (!dPositions || !dIndices || !dModels) return SAILOR_HIP_ERR_INVALID_ARGUMENT;
    Mat4 L, V;
    memcpy(L.m, lightMatrix, 64);
    memset(V.m, 0, 64);
    if (viewMatrix) memcpy(V.m, viewMatrix, 64);
    const unsigned long long total = (unsigned long long)numDrawn * numTriangles;
    if ((total + 255) / 256 > 0x7FFFFFFFull) return SAILOR_HIP_ERR_INVALID_ARGUMENT;
    hipLaunchKernelGGL(k_raster_depth, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, ctx->stream, L, V, viewMatrix ? 1 : 0, (flags & SAILOR_RASTER_CULL_BACK) ? 1 : 0, dPositions, dIndices, numTriangles, dModels, dInstanceIds,
                       numDrawn, width, height, (unsigned int*)dDepth, (unsigned int*)dCoarseDepth);
    SAILOR_CHECK_LAUNCH(ctx, "k_raster_depth");
    return SAILOR_HIP_OK;
}

int sailor_hip_raster_depth(SailorHipContext* ctx, const float* lightMatrix, const float* dPositions, const uint32_t* dIndices, uint32_t numTriangles,
                            const float* dModels, const uint32_t* dInstanceIds, uint32_t numDrawn, int32_t width, int32_t height, float* dDepth, uint32_t flags,
                            uint32_t* dCoarseDepth)
{
    return raster_depth_launch(ctx, lightMatrix, nullptr, dPositions, dIndices, numTriangles, dModels, dInstanceIds, numDrawn, width, height, dDepth, flags, dCoarseDepth);
}

int sailor_hip_raster_depth_camera(SailorHipContext* ctx, const SailorUboFrameData* frame, const float* dPositions, const uint32_t* dIndices, uint32_t numTriangles,
                                   const float* dModels, const uint32_t* dInstanceIds, uint32_t numDrawn, int32_t width, int32_t height, float* dDepth, uint32_t flags,
                                   uint32_t* dCoarseDepth)
{
    if (!frame) return SAILOR_HIP_ERR_INVALID_ARGUMENT;
    return raster_depth_launch(ctx, frame->projection, frame->view, dPositions, dIndices, numTriangles, dModels, dInstanceIds, numDrawn, width, height, dDepth, flags,
                               dCoarseDepth);
}

int sailor_hip_shadow_resolve(SailorHipContext* ctx, const float* dDepth, int32_t width, int32_t height, int32_t format, void* dShadowMap)
{
    if (!ctx || !dDepth || !dShadowMap || width <= 0 || height <= 0 || format < 0 || format > 2) return SAILOR_HIP_ERR_INVALID_ARGUMENT;
    SAILOR_TRY_HIP(ctx, hipSetDevice(ctx->device)); // a host thread may drive several contexts
    const size_t texels = (size_t)width * height;
    hipLaunchKernelGGL(k_shadow_resolve, dim3((unsigned)((texels + 255) / 256)), dim3(256), 0, ctx->stream, dDepth, texels, format, dShadowMap);
    SAILOR_CHECK_LAUNCH(ctx, "k_shadow_resolve");
    return SAILOR_HIP_OK;
}

} // extern "C"
