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

#ifndef RASTER_SMALL_BOX
#define RASTER_SMALL_BOX 64 // pixels a lane fills on its own
#endif
// (RASTER_PRETEST builds only) how the depth test reads the stored depth in front of its atomic: a plain load (this XCD's L2: possibly stale, never too high) or --
// RASTER_PRETEST_COHERENT -- a relaxed device-scope atomic load (the memory side: current, dearer)
#ifdef RASTER_PRETEST_COHERENT
#define RASTER_PRETEST_LOAD(p) __hip_atomic_load((p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)
#else
#define RASTER_PRETEST_LOAD(p) (*(p))
#endif
#if defined(RASTER_STATS) || defined(RASTER_WAVE_TIME)
__device__ unsigned long long gStats[16]; // [8..]: per-wave durations on the 100 MHz clock -- sum, max, waves, waves above 100 us, above 1 ms
#ifdef RASTER_STATS
#define STAT(i, v) atomicAdd(&gStats[i], (unsigned long long)(v))
#else
#define STAT(i, v) // (RASTER_WAVE_TIME alone: only the per-wave durations -- the counters' own atomics, thousands per wave on eight words, would be what is timed)
#endif
#define WSTAT(i, v) atomicAdd(&gStats[i], (unsigned long long)(v))
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

// The depth test of one fragment: atomicMax over float bits (depths are > 0).  (Round 6, measured and left off -- RASTER_PRETEST: the stored depth read first, the
// atomic only for a fragment that beats it.  The buffer only rises, so a stale value is merely too low and the test is safe; but the load is a round trip in front of
// every block of a wave that fills its blocks one after the other, and the caster draws are bound by exactly that chain, not by the atomics' throughput: the four
// cascades of the million-box scene 26.8 -> 34.3 ms with it, plain or device-coherent load alike -- profiles/r06/raster_variants.txt.)
__device__ __forceinline__ void raster_depth_test(unsigned int* __restrict__ p, const float z)
{
    const unsigned int zb = __float_as_uint(z); // positive floats order like their bits
#ifdef RASTER_PRETEST
    if (zb > RASTER_PRETEST_LOAD(p))
#endif
        atomicMax(p, zb);
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

// ---- the large triangles' inner levels, shared by k_raster_depth (a wave works through its own large triangles) and k_raster_giant (sixty-four waves share one) ----
struct RasterEdges { float area; bool tl0, tl1, tl2; };
__device__ __forceinline__ RasterEdges raster_edges(const RasterTri& b)
{
    RasterEdges e;
    e.area = (float)raster_edge(b.x0, b.y0, b.x1, b.y1, b.x2, b.y2);
    e.tl0 = raster_top_left(b.x1, b.y1, b.x2, b.y2); e.tl1 = raster_top_left(b.x2, b.y2, b.x0, b.y0); e.tl2 = raster_top_left(b.x0, b.y0, b.x1, b.y1);
    return e;
}

// level 2: can anything of the triangle still win in superblock (si, sj)?  (also raises the superblock's bound when the triangle covers all of it)
__device__ __forceinline__ bool raster_superblock_alive(const RasterTri& b, const RasterEdges& E, float zmaxB, int si, int sj, int W, int H, unsigned int* __restrict__ coarse2, int SW)
{
    long long m0, m1, m2, n0, n1, n2;
    float zhi, zlo;
    raster_box_bounds(b, E.area, si * 64, sj * 64, si * 64 + 63, sj * 64 + 63, m0, m1, m2, n0, n1, n2, zhi, zlo);
    bool salive = m0 >= 0 && m1 >= 0 && m2 >= 0 && zhi + RASTER_Z_MARGIN > 0.0f && !(zlo - RASTER_Z_MARGIN > 1.0f); // ... and not clipped away as a whole
    if (salive && coarse2) {
        unsigned int* c2 = coarse2 + (size_t)sj * SW + si;
        salive = fminf(zmaxB, zhi + RASTER_Z_MARGIN) > __uint_as_float(*c2);
        // the triangle covers the whole superblock (every corner texel strictly inside every edge) with depths in (0, 1]: once its
        // texels are written, nothing below the smallest of them can win anywhere in the superblock
        if (salive && n0 > 0 && n1 > 0 && n2 > 0 && zlo - RASTER_Z_MARGIN > 0.0f && zhi + RASTER_Z_MARGIN <= 1.0f && si * 64 + 63 < W && sj * 64 + 63 < H)
            atomicMax(c2, __float_as_uint(zlo - RASTER_Z_MARGIN));
    }
    return salive;
}

// level 1 and the fill: superblock (csi, csj) of triangle b by the whole wave -- its 8 x 8 blocks one per lane, the surviving blocks one lane per texel
__device__ __forceinline__ void raster_superblock(const RasterTri& b, const RasterEdges& E, float zmaxB, int csi, int csj, int lane, int W, int H,
                                                  unsigned int* __restrict__ depthBits, unsigned int* __restrict__ coarse, unsigned int* __restrict__ coarse2, int CW, int SW)
{
    const float area = E.area;
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
#ifdef RASTER_PROBE_NOFILL
    live = 0ull; // (timing probe: everything but the texel-level fill -- WRONG results)
#endif
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
            const bool in = !(e0 < 0 || e1 < 0 || e2 < 0) && !((e0 == 0 && !E.tl0) || (e1 == 0 && !E.tl1) || (e2 == 0 && !E.tl2));
            if (in) {
                z = (b.z0 + (b.z1 - b.z0) * ((float)e1 / area)) + (b.z2 - b.z0) * ((float)e2 / area);
                if (z > 0.0f && z <= 1.0f) {
                    raster_depth_test(depthBits + (size_t)j * W + i, z);
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

// ---- round 6: the GIANT triangles -- visible, large on the map, thousands of 8 x 8 blocks to fill -- leave the wave that set them up ------------------------------
// A wave fills its large triangles one after the other, block by block, and each block is a round trip of its own (coarse word, atomics): a box face that covers
// a quarter of a 4096^2 cascade keeps ONE wave busy for milliseconds while the other sixty thousand are long done -- per-wave clocks on the million-box scene:
// mean 92 us, longest 9.7 ms of the launch's 10 (profiles/r06/README.md).  Such a triangle -- its box spans more than 64 superblocks, or more than
// RASTER_GIANT_ALIVE of them survive the coarse test -- is appended to a queue behind the coarse depth instead (its set-up, 24 words), and k_raster_giant, launched
// right behind k_raster_depth, gives every queued triangle SIXTY-FOUR waves, wave w taking the superblocks w, w + 64, ...  A full queue sends the triangle back to
// the ordinary path.  The giants are drawn last, whatever the draw's order: the depth buffer does not depend on it.
#ifndef RASTER_GIANT_ALIVE
#define RASTER_GIANT_ALIVE 4
#endif
#define RASTER_GIANT_WORDS 24 // x0 y0 x1 y1 x2 y2 (int64 each), z0 z1 z2, i0 i1 j0 j1, zmax, pad
#define RASTER_GIANT_GRID 1024u // queue entries that get blocks of their own in one launch of k_raster_giant (x 16 blocks of four waves each)
__device__ __forceinline__ bool raster_giant_push(unsigned int* __restrict__ giants, unsigned int giantCap, const RasterTri& b, float zmaxB, int lane)
{
    unsigned int slot = 0;
    if (lane == 0) slot = atomicAdd(giants, 1u);
    slot = (unsigned int)__shfl((int)slot, 0, 64);
    if (slot >= giantCap) return false; // (the count keeps running past the capacity; the reader clamps it)
    unsigned int* e = giants + 4 + (size_t)slot * RASTER_GIANT_WORDS;
    if (lane == 0) {
        const long long xy[6] = { b.x0, b.y0, b.x1, b.y1, b.x2, b.y2 };
#pragma unroll
        for (int k = 0; k < 6; k++) { e[2 * k] = (unsigned int)(unsigned long long)xy[k]; e[2 * k + 1] = (unsigned int)((unsigned long long)xy[k] >> 32); }
        e[12] = __float_as_uint(b.z0); e[13] = __float_as_uint(b.z1); e[14] = __float_as_uint(b.z2);
        e[15] = (unsigned int)b.i0; e[16] = (unsigned int)b.i1; e[17] = (unsigned int)b.j0; e[18] = (unsigned int)b.j1;
        e[19] = __float_as_uint(zmaxB);
    }
    return true;
}

__global__ __launch_bounds__(256) void k_raster_giant(const unsigned int* __restrict__ giants, unsigned int giantCap, int W, int H, unsigned int* __restrict__ depthBits,
                                                      unsigned int* __restrict__ coarse)
{
    const unsigned int count = min(giants[0], giantCap);
    // (the grid holds at most RASTER_GIANT_GRID entries' worth of blocks -- a launch of the queue's full capacity is 131 072 blocks at 4096^2, 35 us of empty
    // blocks behind every chunk of every draw -- and walks the queue in strides)
    for (unsigned int entry = blockIdx.x; entry < count; entry += gridDim.x) {
    const unsigned int* e = giants + 4 + (size_t)entry * RASTER_GIANT_WORDS;
    RasterTri b;
    long long xy[6];
#pragma unroll
    for (int k = 0; k < 6; k++) xy[k] = (long long)(((unsigned long long)e[2 * k + 1] << 32) | e[2 * k]);
    b.x0 = xy[0]; b.y0 = xy[1]; b.x1 = xy[2]; b.y1 = xy[3]; b.x2 = xy[4]; b.y2 = xy[5];
    b.z0 = __uint_as_float(e[12]); b.z1 = __uint_as_float(e[13]); b.z2 = __uint_as_float(e[14]);
    b.i0 = (int)e[15]; b.i1 = (int)e[16]; b.j0 = (int)e[17]; b.j1 = (int)e[18];
    b.valid = true;
    const float zmaxB = __uint_as_float(e[19]);
    const RasterEdges E = raster_edges(b);
    const int lane = threadIdx.x & 63;
    const int CW = (W + 7) >> 3, SW = (W + 63) >> 6;
    unsigned int* coarse2 = coarse + (size_t)CW * ((H + 7) >> 3);
    const int si0 = b.i0 >> 6, sj0 = b.j0 >> 6, sw = (b.i1 >> 6) - si0 + 1, sh = (b.j1 >> 6) - sj0 + 1;
    const int ns = sw * sh;
    // wave w of the triangle's 64 (blockIdx.y, threadIdx.x >> 6): superblocks w, w + 64, ...; the level-2 test is the same in every lane
    for (int sblk = (int)(blockIdx.y * 4 + (threadIdx.x >> 6)); sblk < ns; sblk += 64) {
        const int sj = sblk / sw, si = sblk - sj * sw;
        bool alive = false;
        if (lane == 0) alive = raster_superblock_alive(b, E, zmaxB, si + si0, sj + sj0, W, H, coarse2, SW);
        if (__shfl((int)alive, 0, 64)) raster_superblock(b, E, zmaxB, si + si0, sj + sj0, lane, W, H, depthBits, coarse, coarse2, CW, SW);
    }
    }
}

// ---- round 6: a whole INSTANCE against the coarse depth, before any of its triangles is set up ------------------------------------------------------------------
// The caster draws of a near cascade overdraw every texel hundreds of times; drawn front to back, most instances are hidden as a whole by what is already there.
// k_mesh_bounds: the box of the mesh's referenced vertices (once per draw call, eight words behind the coarse depth).  raster_instance_hidden: its eight corners
// through the instance's matrix -- clip-space x / w, y / w and z / w are projective in the position, so over a box that lies in front of the eye (every corner's
// w > 0) and inside the near plane their extremes sit at corners -- give the instance's texel rectangle and its largest possible depth; the instance is skipped
// when the rectangle misses the map altogether, when nothing of it can pass z > 0, or when -- the rectangle touching at most 2 x 2 blocks / superblocks of the
// coarse depth -- its largest depth does not exceed the lower bound of what those hold.  Bounds only: the depth buffer is the same with and without it.
#define RASTER_INST_Z_MARGIN 4.0e-6f // (the corners' z / w here and the triangles' own vertex depths are rounded separately: a few ulp of 1)
__global__ __launch_bounds__(1024) void k_mesh_bounds(const float* __restrict__ positions, const uint32_t* __restrict__ indices, uint32_t numIndices, float* __restrict__ bounds)
{
    __shared__ float sMin[3][16], sMax[3][16];
    float mn[3] = { 3.0e38f, 3.0e38f, 3.0e38f }, mx[3] = { -3.0e38f, -3.0e38f, -3.0e38f };
    bool nan = false;
    for (uint32_t i = threadIdx.x; i < numIndices; i += 1024u) {
        const float* p = positions + 3 * (size_t)indices[i];
#pragma unroll
        for (int a = 0; a < 3; a++) { mn[a] = fminf(mn[a], p[a]); mx[a] = fmaxf(mx[a], p[a]); nan |= !(p[a] == p[a]); }
    }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int a = 0; a < 3; a++) {
        if (nan) { mn[a] = -3.0e38f; mx[a] = 3.0e38f; } // (a NaN vertex: no bound -- the box of everything)
#pragma unroll
        for (int d = 32; d > 0; d >>= 1) { mn[a] = fminf(mn[a], __shfl_xor(mn[a], d, 64)); mx[a] = fmaxf(mx[a], __shfl_xor(mx[a], d, 64)); }
        if (lane == 0) { sMin[a][wave] = mn[a]; sMax[a][wave] = mx[a]; }
    }
    __syncthreads();
    if (threadIdx.x < 3) {
        float lo = sMin[threadIdx.x][0], hi = sMax[threadIdx.x][0];
        for (int w = 1; w < 16; w++) { lo = fminf(lo, sMin[threadIdx.x][w]); hi = fmaxf(hi, sMax[threadIdx.x][w]); }
        bounds[threadIdx.x] = lo; bounds[3 + threadIdx.x] = hi;
    }
}

__device__ __forceinline__ bool raster_instance_hidden(const Mat4& LM, bool hasView, const Mat4& V, const float* __restrict__ model, const float* __restrict__ mb, int W, int H,
                                                       const unsigned int* __restrict__ coarse, const unsigned int* __restrict__ coarse2, int CW, int SW)
{
    float xlo = 3.0e38f, xhi = -3.0e38f, ylo = 3.0e38f, yhi = -3.0e38f, zhi = -3.0e38f;
    bool ok = true;
#pragma unroll
    for (int c = 0; c < 8; c++) {
        const float px = mb[(c & 1) ? 3 : 0], py = mb[(c & 2) ? 4 : 1], pz = mb[(c & 4) ? 5 : 2];
        float4 clip;
        if (hasView) {
            Mat4 M;
#pragma unroll
            for (int q = 0; q < 16; q++) M.m[q] = model[q];
            const float4 a = glsl_mul(M, px, py, pz, 1.0f);
            const float4 b = glsl_mul(V, a.x, a.y, a.z, a.w);
            clip = glsl_mul(LM, b.x, b.y, b.z, b.w);
        } else clip = glsl_mul(LM, px, py, pz, 1.0f);
        ok = ok && clip.w > 0.0f && clip.w - clip.z >= 0.0f; // in front of the eye and inside the near plane of the reversed range: projectable, extremes at corners
        const float rw = 1.0f / clip.w;
        const float xf = (clip.x * rw + 1.0f) * ((float)W * 0.5f), yf = (clip.y * rw + 1.0f) * ((float)H * -0.5f) + (float)H;
        xlo = fminf(xlo, xf); xhi = fmaxf(xhi, xf); ylo = fminf(ylo, yf); yhi = fmaxf(yhi, yf);
        zhi = fmaxf(zhi, clip.z * rw);
    }
    // (a NaN anywhere fails a comparison below and keeps the instance)
    if (!ok || !(fabsf(xlo) < 1.0e9f) || !(fabsf(xhi) < 1.0e9f) || !(fabsf(ylo) < 1.0e9f) || !(fabsf(yhi) < 1.0e9f) || !(zhi == zhi)) return false;
    // the texels whose centres the box can cover, one texel of margin for the roundings (a texel centre is at i + 0.5)
    const float fi0 = floorf(xlo - 1.5f), fi1 = floorf(xhi + 0.5f), fj0 = floorf(ylo - 1.5f), fj1 = floorf(yhi + 0.5f);
    if (fi1 < 0.0f || fj1 < 0.0f || fi0 > (float)(W - 1) || fj0 > (float)(H - 1)) return true; // misses the map
    if (!(zhi + RASTER_INST_Z_MARGIN > 0.0f)) return true;                                       // nothing of it passes z > 0
    if (!coarse) return false;
    const int i0 = max((int)fi0, 0), i1 = min((int)fi1, W - 1), j0 = max((int)fj0, 0), j1 = min((int)fj1, H - 1);
    const float zin = zhi + RASTER_INST_Z_MARGIN;
    if ((i1 >> 3) - (i0 >> 3) <= 1 && (j1 >> 3) - (j0 >> 3) <= 1) { // at most 2 x 2 blocks
        const unsigned int* r0 = coarse + (size_t)(j0 >> 3) * CW, *r1 = coarse + (size_t)(j1 >> 3) * CW;
        const float m = fminf(fminf(__uint_as_float(r0[i0 >> 3]), __uint_as_float(r0[i1 >> 3])), fminf(__uint_as_float(r1[i0 >> 3]), __uint_as_float(r1[i1 >> 3])));
        return zin <= m;
    }
    if (coarse2 && (i1 >> 6) - (i0 >> 6) <= 1 && (j1 >> 6) - (j0 >> 6) <= 1) { // at most 2 x 2 superblocks
        const unsigned int* r0 = coarse2 + (size_t)(j0 >> 6) * SW, *r1 = coarse2 + (size_t)(j1 >> 6) * SW;
        const float m = fminf(fminf(__uint_as_float(r0[i0 >> 6]), __uint_as_float(r0[i1 >> 6])), fminf(__uint_as_float(r1[i0 >> 6]), __uint_as_float(r1[i1 >> 6])));
        return zin <= m;
    }
    return false;
}

__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(4, 4))) void k_raster_depth(Mat4 L, Mat4 V, int hasView, int cullBack, const float* __restrict__ positions, const uint32_t* __restrict__ indices, uint32_t numTriangles,
                                                       const float* __restrict__ models, const uint32_t* __restrict__ instanceIds, uint32_t firstDrawn, uint32_t numDrawn, int W, int H,
                                                       unsigned int* __restrict__ depthBits, unsigned int* __restrict__ coarse, const float* __restrict__ meshBounds, int interleave,
                                                       unsigned int* __restrict__ giants, unsigned int giantCap)
{
    const unsigned long long total = (unsigned long long)numDrawn * numTriangles;
    const int lane = threadIdx.x & 63;
#if defined(RASTER_STATS) || defined(RASTER_WAVE_TIME)
    const unsigned long long statT0 = __builtin_amdgcn_s_memrealtime();
#endif
    const unsigned long long waveId = ((unsigned long long)blockIdx.x * 256 + threadIdx.x) >> 6;
    const int CW = (W + 7) >> 3, SW = (W + 63) >> 6;
    unsigned int* coarse2 = coarse ? coarse + (size_t)CW * ((H + 7) >> 3) : nullptr; // level 2 behind level 1 in the same workspace
    // One round of the wave: every lane with `have` holds one (instance, triangle).  testInstance: hold the instance's box against the coarse depth first.
    auto process = [&](const bool have, const uint32_t inst, const uint32_t tri, const bool testInstance) {
    // a triangle cut by the near plane can leave a quad: its second half is a second trip through the same code, taken only by waves that hold one
    bool again = false;
    for (int part = 0; part < 2; part++) {
        if (part && !__any(again)) break;
        RasterTri t;
        t.valid = false;
        bool second = false;
        if (have) {
            const Mat4 LM = hasView ? L : raster_mul(L, models + 16 * (size_t)inst);
            // (part 1 only re-runs lanes whose triangle was cut in two: their instance was not hidden)
            const bool hidden = testInstance && part == 0 && raster_instance_hidden(LM, hasView != 0, V, models + 16 * (size_t)inst, meshBounds, W, H, coarse, coarse2, CW, SW);
            if (hidden) STAT(7, 1);
            if (!hidden) t = raster_setup(LM, hasView != 0, V, models + 16 * (size_t)inst, positions, indices + 3 * (size_t)tri, W, H, cullBack != 0, part, second);
        }
        const float zmaxTri = fmaxf(t.z0, fmaxf(t.z1, t.z2)) + RASTER_Z_MARGIN; // inside the triangle z is a convex combination of the vertices'
        if (t.valid && !(zmaxTri > 0.0f)) t.valid = false;                        // nothing of it can pass z > 0
        if (t.valid && fminf(t.z0, fminf(t.z1, t.z2)) - RASTER_Z_MARGIN > 1.0f) t.valid = false; // ... or z <= 1
        const bool small = t.valid && (long long)(t.i1 - t.i0 + 1) * (t.j1 - t.j0 + 1) <= RASTER_SMALL_BOX;
        if (small) {
            // (its box against the coarse depth of the blocks it touches, where those are at most 2 x 2 -- round 6; one block through round 5)
            bool hidden = false;
            if (coarse && (t.i1 >> 3) - (t.i0 >> 3) <= 1 && (t.j1 >> 3) - (t.j0 >> 3) <= 1) {
                const unsigned int* r0 = coarse + (size_t)(t.j0 >> 3) * CW, *r1 = coarse + (size_t)(t.j1 >> 3) * CW;
                hidden = zmaxTri <= fminf(fminf(__uint_as_float(r0[t.i0 >> 3]), __uint_as_float(r0[t.i1 >> 3])), fminf(__uint_as_float(r1[t.i0 >> 3]), __uint_as_float(r1[t.i1 >> 3])));
            }
            if (!hidden) {
                const float area = (float)raster_edge(t.x0, t.y0, t.x1, t.y1, t.x2, t.y2);
                const bool tl0 = raster_top_left(t.x1, t.y1, t.x2, t.y2), tl1 = raster_top_left(t.x2, t.y2, t.x0, t.y0), tl2 = raster_top_left(t.x0, t.y0, t.x1, t.y1);
                // (round 6: the three edge functions walked texel by texel -- exact integer steps of 256 sub-pixels, -256 (by - ay) along x and 256 (bx - ax)
                // along y -- instead of six 64-bit multiplications per texel: the far cascades are made of these triangles)
                const long long px0 = 256ll * t.i0 + 128, py0 = 256ll * t.j0 + 128;
                long long r0 = raster_edge(t.x1, t.y1, t.x2, t.y2, px0, py0), r1 = raster_edge(t.x2, t.y2, t.x0, t.y0, px0, py0), r2 = raster_edge(t.x0, t.y0, t.x1, t.y1, px0, py0);
                const long long dx0 = -256ll * (t.y2 - t.y1), dx1 = -256ll * (t.y0 - t.y2), dx2 = -256ll * (t.y1 - t.y0);
                const long long dy0 = 256ll * (t.x2 - t.x1), dy1 = 256ll * (t.x0 - t.x2), dy2 = 256ll * (t.x1 - t.x0);
                for (int j = t.j0; j <= t.j1; j++, r0 += dy0, r1 += dy1, r2 += dy2) {
                    long long e0 = r0, e1 = r1, e2 = r2;
                    for (int i = t.i0; i <= t.i1; i++, e0 += dx0, e1 += dx1, e2 += dx2) {
                        if (e0 < 0 || e1 < 0 || e2 < 0) continue;
                        if ((e0 == 0 && !tl0) || (e1 == 0 && !tl1) || (e2 == 0 && !tl2)) continue;
                        const float z = (t.z0 + (t.z1 - t.z0) * ((float)e1 / area)) + (t.z2 - t.z0) * ((float)e2 / area);
                        if (z > 0.0f && z <= 1.0f) raster_depth_test(depthBits + (size_t)j * W + i, z); // (z == 0 never passes GREATER against the cleared 0 either)
                    }
                }
            }
        }
        // the large ones: the whole wave on one triangle at a time
        unsigned long long todo = __ballot(t.valid && !small);
#ifdef RASTER_PROBE_NOLARGE
        todo = 0ull; // (timing probe: what the launch costs without its large triangles -- WRONG results)
#endif
        while (todo) {
            const int src = __builtin_ctzll(todo);
            todo &= todo - 1ull;
            RasterTri b;
            b.x0 = bcast64(t.x0, src); b.y0 = bcast64(t.y0, src); b.x1 = bcast64(t.x1, src); b.y1 = bcast64(t.y1, src); b.x2 = bcast64(t.x2, src); b.y2 = bcast64(t.y2, src);
            b.z0 = __shfl(t.z0, src, 64); b.z1 = __shfl(t.z1, src, 64); b.z2 = __shfl(t.z2, src, 64);
            b.i0 = __shfl(t.i0, src, 64); b.i1 = __shfl(t.i1, src, 64); b.j0 = __shfl(t.j0, src, 64); b.j1 = __shfl(t.j1, src, 64);
            const float zmaxB = __shfl(zmaxTri, src, 64);
            const RasterEdges E = raster_edges(b);
            // ---- level 2: 64 x 64-texel superblocks, one per lane ----
            const int si0 = b.i0 >> 6, sj0 = b.j0 >> 6, sw = (b.i1 >> 6) - si0 + 1, sh = (b.j1 >> 6) - sj0 + 1;
            const int ns = sw * sh;
            // (measured and dropped, round 6: a triangle inside ONE superblock sent straight to its blocks, without the superblock's own test -- 15.4 -> 16.2 ms over
            // the four cascades: the one coarse word of level 2 spares most of these triangles the sixty-four of level 1)
            // (round 6) a triangle with many superblocks left after the coarse test -- visible, large on the map, thousands of blocks to fill -- is not this wave's
            // to fill: see k_raster_giant.  One that spans several batches of superblocks has them counted first (the test alone: a round trip per batch).
            if (giants && ns > 64) {
                int aliveAll = 0;
                for (int sbase = 0; sbase < ns && aliveAll <= RASTER_GIANT_ALIVE; sbase += 64) {
                    const int sblk = sbase + lane;
                    bool salive = false;
                    if (sblk < ns) { const int sj = sblk / sw, si = sblk - sj * sw; salive = raster_superblock_alive(b, E, zmaxB, si + si0, sj + sj0, W, H, coarse2, SW); }
                    aliveAll += __popcll(__ballot(salive));
                }
                if (aliveAll > RASTER_GIANT_ALIVE && raster_giant_push(giants, giantCap, b, zmaxB, lane)) continue;
            }
            for (int sbase = 0; sbase < ns; sbase += 64) {
                const int sblk = sbase + lane;
                bool salive = sblk < ns;
                int si = 0, sj = 0;
                if (salive) {
                    sj = sblk / sw; si = sblk - sj * sw;
                    si += si0; sj += sj0;
                    salive = raster_superblock_alive(b, E, zmaxB, si, sj, W, H, coarse2, SW);
                }
                unsigned long long slive = __ballot(salive);
                if (lane == 0) { STAT(0, min(64, ns - sbase)); STAT(1, __popcll(slive)); }
                // ... and neither is one with many superblocks left after the coarse test: visible, large on the map, thousands of blocks to fill
                if (giants && ns <= 64 && __popcll(slive) > RASTER_GIANT_ALIVE && raster_giant_push(giants, giantCap, b, zmaxB, lane)) break;
                while (slive) {
                    const int s1 = __builtin_ctzll(slive);
                    slive &= slive - 1ull;
                    raster_superblock(b, E, zmaxB, __shfl(si, s1, 64), __shfl(sj, s1, 64), lane, W, H, depthBits, coarse, coarse2, CW, SW);
                }
            }
        }
        if (part == 0) again = second;
    }
    }; // process

    // A lane per (instance, triangle), INTERLEAVED (round 6): lane l of wave w takes item l * waves + w, not the 64 consecutive items 64 w + l.  A wave works
    // through its large triangles one after the other, and a draw sorted front to back begins with the instances that are visible and large on the map: as
    // consecutive items they all sat in the first waves (cascade 1 of the million-box scene: 15.0 -> 8.9 ms by this alone).  Interleaved, every wave gets a sample
    // of the whole depth range and walks it front to back.
    // (Measured and taken out again: an INSTANCE-major form -- a lane per instance holds the box against the coarse depth once, the survivors' triangles go through
    // the lanes 64 at a time: a twelfth of the instance tests, a fifth of the set-ups on the far cascades -- 15.9 -> 33.5 ms: a wave then carries up to 64 x 12
    // triangles through its serial large-triangle loop, and the early chunks have a twelfth of the waves.  The draw is bound by its longest chains, not by work.)
    {
        const unsigned long long waves = (total + 63) / 64;
        const unsigned long long id = !interleave ? (unsigned long long)blockIdx.x * 256 + threadIdx.x : (waveId < waves ? (unsigned long long)lane * waves + waveId : total);
        const uint32_t d = (uint32_t)(id / numTriangles), tri = (uint32_t)(id - (unsigned long long)d * numTriangles);
        const bool have = id < total;
        const uint32_t inst = have ? (instanceIds ? instanceIds[firstDrawn + d] : firstDrawn + d) : 0u;
        process(have, inst, tri, meshBounds != nullptr);
    }
#if defined(RASTER_STATS) || defined(RASTER_WAVE_TIME)
    if (lane == 0) {
        const unsigned long long dt = __builtin_amdgcn_s_memrealtime() - statT0;
        WSTAT(8, dt); atomicMax(&gStats[9], dt); WSTAT(10, 1); WSTAT(11, dt > 10000ull); WSTAT(12, dt > 100000ull);
    }
#endif
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

// the workspace behind a depth buffer: level 1 (one word per 8 x 8 block), level 2 (one per 64 x 64 superblock), and -- round 6 -- eight words for the box of the
// mesh of the draw in flight (k_mesh_bounds)
#define RASTER_BOUNDS_WORDS 8
#define RASTER_CHUNKS_MAX 6
static size_t raster_coarse_levels(int32_t width, int32_t height)
{
    return (size_t)((width + 7) / 8) * ((height + 7) / 8) + (size_t)((width + 63) / 64) * ((height + 63) / 64);
}
// ... and the queue of giant triangles (k_raster_giant): a count (4 words) + two entries per superblock of the map, at least 64
static unsigned int raster_giant_capacity(int32_t width, int32_t height)
{
    const size_t n = 2 * (size_t)((width + 63) / 64) * ((height + 63) / 64);
    return (unsigned int)(n < 64 ? 64 : n);
}

extern "C" {

#if defined(RASTER_STATS) || defined(RASTER_WAVE_TIME)
__attribute__((visibility("default"))) void sailor_hip_raster_stats(unsigned long long* out, int reset)
{
    (void)hipMemcpyFromSymbol(out, HIP_SYMBOL(gStats), 128);
    if (reset) { unsigned long long z[16] = {0}; (void)hipMemcpyToSymbol(HIP_SYMBOL(gStats), z, 128); }
}
#endif

size_t sailor_hip_raster_coarse_words(int32_t width, int32_t height)
{
    if (width <= 0 || height <= 0) return 0;
    return raster_coarse_levels(width, height) + RASTER_BOUNDS_WORDS + 4 + (size_t)raster_giant_capacity(width, height) * RASTER_GIANT_WORDS;
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
        if (dCoarseDepth) SAILOR_TRY_HIP(ctx, hipMemsetAsync(dCoarseDepth, 0, raster_coarse_levels(width, height) * 4, ctx->stream));
    }
    if (numTriangles == 0 || numDrawn == 0) return SAILOR_HIP_OK;
    if (!dPositions || !dIndices || !dModels) return SAILOR_HIP_ERR_INVALID_ARGUMENT;
    Mat4 L, V;
    memcpy(L.m, lightMatrix, 64);
    memset(V.m, 0, 64);
    if (viewMatrix) memcpy(V.m, viewMatrix, 64);
    const unsigned long long total = (unsigned long long)numDrawn * numTriangles;
    if ((total + 255) / 256 > 0x7FFFFFFFull) return SAILOR_HIP_ERR_INVALID_ARGUMENT;
    // Round 6, for draws that have a coarse depth (any order of drawing gives the same depth buffer; the environment switches are for A / B runs).
    // (a) The giant triangles' queue: k_raster_giant follows every launch (raster_giant_push).
    // (b) A draw of many instances goes out in CHUNKS of growing size (2 048, x 4, at most six launches): a launch keeps half a million triangles in flight at once,
    // so inside ONE launch neither the coarse depth nor the giants -- drawn behind the launch -- can hide an instance from the thousands that started with it; a
    // caller that draws front to back has its first few thousand instances and their giants fill the coarse depth, and the launches behind them find most of
    // theirs hidden.  WITHOUT the queue chunks lose (every launch ends in its own long waves: 37.6 -> 73.7 ms); with it the four cascades of the million-box scene
    // take 15.9 ms for 26.8.
    // (c) The instance test (raster_instance_hidden) for such draws: the mesh's box first, eight words behind the coarse levels.
    static const bool chunksOn = [] { const char* e = getenv("SAILOR_RASTER_CHUNKS"); return !e || atoi(e) != 0; }();
    static const uint32_t chunkFirst = [] { const char* e = getenv("SAILOR_RASTER_CHUNK_FIRST"); const int v = e ? atoi(e) : 0; return (uint32_t)(v >= 64 ? v : 2048); }();
    static const uint32_t chunkGrowth = [] { const char* e = getenv("SAILOR_RASTER_CHUNK_GROWTH"); const int v = e ? atoi(e) : 0; return (uint32_t)(v >= 2 && v <= 1024 ? v : 4); }();
    static const bool interleaveOn = [] { const char* e = getenv("SAILOR_RASTER_INTERLEAVE"); return !e || atoi(e) != 0; }();
    static const bool giantsOn = [] { const char* e = getenv("SAILOR_RASTER_GIANTS"); return !e || atoi(e) != 0; }();
    unsigned int* giants = (dCoarseDepth && giantsOn) ? (unsigned int*)dCoarseDepth + raster_coarse_levels(width, height) + RASTER_BOUNDS_WORDS : nullptr;
    const unsigned int giantCap = raster_giant_capacity(width, height);

    static const bool instanceTestOn = [] { const char* e = getenv("SAILOR_RASTER_INSTANCE_TEST"); return !e || atoi(e) != 0; }();
    const bool manyInstances = dCoarseDepth && numDrawn >= 4096u;
    const float* meshBounds = nullptr;
    if (manyInstances && instanceTestOn) {
        float* b = reinterpret_cast<float*>(dCoarseDepth + raster_coarse_levels(width, height));
        hipLaunchKernelGGL(k_mesh_bounds, dim3(1), dim3(1024), 0, ctx->stream, dPositions, dIndices, numTriangles * 3u, b);
        meshBounds = b;
    }
    static const uint32_t chunkMax = [] { const char* e = getenv("SAILOR_RASTER_CHUNK_MAX"); const int v = e ? atoi(e) : 0; return (uint32_t)(v >= 1 ? v : RASTER_CHUNKS_MAX); }();
    uint32_t first = 0, chunk = (manyInstances && chunksOn && giants) ? chunkFirst : numDrawn, launches = 0;
    // (Measured and not kept: letting the DEVICE decide after the first chunk whether chunks pay -- its giants' count against a threshold, the second launch sized
    // for everything that is left.  Only the farthest cascade of the million-box scene prefers one launch, by 0.4 of its 5 ms; the other three want every chunk:
    // cascade 1 3.4 -> 9.4 ms, cascade 2 4.8 -> 6.4 without them -- profiles/r06/raster_experiments.txt, blocks 8-10.)
    while (first < numDrawn) {
        // (the last of at most chunkMax launches takes whatever is left)
        const uint32_t n = (numDrawn - first < chunk || ++launches >= chunkMax) ? numDrawn - first : chunk;
        if (giants) SAILOR_TRY_HIP(ctx, hipMemsetAsync(giants, 0, 16, ctx->stream));
        const unsigned long long t = (unsigned long long)n * numTriangles;
        hipLaunchKernelGGL(k_raster_depth, dim3((unsigned)((t + 255) / 256)), dim3(256), 0, ctx->stream, L, V, viewMatrix ? 1 : 0, (flags & SAILOR_RASTER_CULL_BACK) ? 1 : 0,
                           dPositions, dIndices, numTriangles, dModels, dInstanceIds, first, n, width, height, (unsigned int*)dDepth, (unsigned int*)dCoarseDepth, meshBounds,
                           interleaveOn ? 1 : 0, giants, giantCap);
        // (a chunk's giants behind the chunk: what they write -- and what they make of the coarse depth -- is there for the next chunk)
        if (giants) hipLaunchKernelGGL(k_raster_giant, dim3(giantCap < RASTER_GIANT_GRID ? giantCap : RASTER_GIANT_GRID, 16), dim3(256), 0, ctx->stream, giants, giantCap, width, height, (unsigned int*)dDepth, (unsigned int*)dCoarseDepth);
        first += n;
        if (chunk < 0x40000000u / chunkGrowth) chunk *= chunkGrowth;
    }
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
