// K0 + K1 -- screen-tile light cull for gfx950.
//
// Replaces Content/Shaders/ComputeLightCulling.shader (+ Math.glsl:116-173,224-239) as dispatched by
// LightCullingNode::Process (FrameGraph/LightCullingNode.cpp:74-77), under the canonical sequential semantics of
// SURVEY.md Appendix A.  Not a translation of the GLSL: the reference runs one 16x16 workgroup per tile that
// re-reads and re-transforms every light, appends with LDS atomics, bubble-sorts on one thread and allocates
// output space with a global atomic.  Here:
//
//   k0_light_view      once per light: view-space position + radius into a float4 SoA (same fp32 op sequence as
//                      ComputeLightCulling.shader:164-169, so bits are identical), light type into a u32 SoA
//   k1_tile_setup      streaming pass over the linear-depth image (the only large HBM stream of the cull):
//                      16 tiles per 256-thread block, 1 KiB coalesced float4 row segments, wave shuffles + a
//                      4-entry LDS combine for min/max, then 16 lanes build the 16 tile frusta
//   k1_macro_setup     8x8-tile macro tiles: union depth window + macro frustum
//   k1_macro_cull      conservative pre-filter: (macro tile, 2048-light chunk) blocks, wave-ballot ordered
//                      compaction into LDS, one bump allocation per block from a shared pool
//   k1_tile_cull       one 64-lane wave per tile walks its macro tile's survivor lists in ascending light index,
//                      exact test, ballot/popcount ordered append into LDS, rank-based nearest-128 selection
//   k1_scan / k1_pack  canonical offsets = prefix sum over tiles in tile-index order, then compaction
//
// Bit-exactness: the pre-filter only ever removes lights that every member tile's exact test would reject
// (depth window: the very same fp32 expressions compared against the max/min of the member tiles' windows;
// side planes: only for spheres entirely in front of the eye, with a relative margin 1000x the fp32 error), so
// the candidate sequence of every tile -- and therefore its first 196, its selection and its order -- is the
// sequence the brute-force walk produces.  tests/test_light_cull_gpu.py checks default == brute force == oracle.
//
// No MFMA: sphere/plane tests and compaction, not a contraction.  Compiled with -ffp-contract=off.
#include "common.h"

#define MACRO 8            // tiles per macro-tile edge
#define CHUNK 2048         // lights per macro-cull block
#define SEG_RAW 0xFFFFFFFFu // segment marker: pool overflow, walk the raw light range instead

struct CullLayout {
    int Tx, Ty, bandTiles, macroX, macroY0, macroRows, numMacros, numChunks;
    size_t poolEntries;
    size_t offLightView, offLightType, offTileInfo, offMacroInfo, offSegTable, offCursor, offTileNum, offTileList, offPool, total;
};

static CullLayout make_layout(int W, int H, int N, const SailorBand& band)
{
    CullLayout L;
    L.Tx = (W - 1) / TILE + 1;
    L.Ty = (H - 1) / TILE + 1;
    const int rows = band.tileRowEnd - band.tileRowBegin;
    L.bandTiles = rows * L.Tx;
    L.macroX = (L.Tx + MACRO - 1) / MACRO;
    L.macroY0 = band.tileRowBegin / MACRO;
    const int macroY1 = rows > 0 ? (band.tileRowEnd - 1) / MACRO + 1 : L.macroY0;
    L.macroRows = macroY1 - L.macroY0;
    L.numMacros = L.macroX * L.macroRows;
    L.numChunks = (N + CHUNK - 1) / CHUNK;
    if (L.numChunks < 1) L.numChunks = 1;
    const size_t n = (size_t)(N > 0 ? N : 1);
    // Survivor pool: the expected load is a few percent of numMacros*N; 48*N entries (+ slack) covers scenes far
    // denser than the reference's cap; overflow degrades to the raw walk per segment, never to wrong results.
    L.poolEntries = n * 48 + (size_t)L.numMacros * 64 + 4096;
    size_t o = 0;
    L.offLightView = o; o = align_up(o + n * 16, 256);
    L.offLightType = o; o = align_up(o + n * 4, 256);
    L.offTileInfo = o; o = align_up(o + (size_t)(L.bandTiles > 0 ? L.bandTiles : 1) * 64, 256);
    L.offMacroInfo = o; o = align_up(o + (size_t)(L.numMacros > 0 ? L.numMacros : 1) * 80, 256);
    L.offSegTable = o; o = align_up(o + (size_t)(L.numMacros > 0 ? L.numMacros : 1) * L.numChunks * 8, 256);
    L.offCursor = o; o = align_up(o + 256, 256);
    L.offTileNum = o; o = align_up(o + (size_t)(L.bandTiles > 0 ? L.bandTiles : 1) * 4, 256);
    L.offTileList = o; o = align_up(o + (size_t)(L.bandTiles > 0 ? L.bandTiles : 1) * KEEP * 4, 256);
    L.offPool = o; o = align_up(o + L.poolEntries * 4, 256);
    L.total = o;
    return L;
}

// ------------------------------------------------------------------------------------------------------------
// K0: ComputeLightCulling.shader:164-169 hoisted out of the per-tile loop (it does not depend on the tile)
// ------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k0_light_view(Mat4 view, const SailorLightShaderData* __restrict__ lights, int N,
                                                      float4* __restrict__ lightView, uint32_t* __restrict__ lightType)
{
    const int j = blockIdx.x * 256 + threadIdx.x;
    if (j >= N) return;
    const SailorLightShaderData* L = lights + j;
    const float x = L->worldPosition[0], y = L->worldPosition[1], z = L->worldPosition[2];
    float4 p = glsl_mul(view, x, y, z, 1.0f);
    const float w = p.w;
    p.x = p.x / w; p.y = p.y / w; p.z = p.z / w;
    p.z = p.z * -1.0f; // "Reverse Z"
    lightView[j] = make_float4(p.x, p.y, p.z, L->bounds[0]);
    lightType[j] = L->type;
}

// ------------------------------------------------------------------------------------------------------------
// Frustum of a screen rectangle: ComputeLightCulling.shader:57-95 CreateFrustum / Math.glsl:185-222 CreateViewFrustum
// via Math.glsl:164-173 ScreenSpaceToViewSpace, :143-154 ClipSpaceToViewSpace, :122-134 ComputePlane.
// ------------------------------------------------------------------------------------------------------------
struct Frustum4 { float n[4][3]; float cx, cy; };

__device__ __forceinline__ void screen_to_view(const Mat4& invProj, float sx, float sy, float sz, float sw, float vpW, float vpH, float* o)
{
    const float tx = sx / vpW, ty = sy / vpH;
    float4 v = glsl_mul(invProj, tx * 2.0f - 1.0f, ty * 2.0f - 1.0f, sz, sw);
    const float w = v.w;
    o[0] = v.x / w; o[1] = v.y / w; o[2] = (v.z / w) * -1.0f;
}

__device__ __forceinline__ void compute_plane_normal(const float* p1, const float* p2, float* n)
{
    // eye = 0: v0 = p1 - 0, v2 = p2 - 0 (exact); plane.w = dot(n, 0) = +-0 and x - (+-0) == x, so w is dropped
    const float cx = p1[1] * p2[2] - p2[1] * p1[2];
    const float cy = p1[2] * p2[0] - p2[2] * p1[0];
    const float cz = p1[0] * p2[1] - p2[0] * p1[1];
    const float len = sqrtf(dot3f(cx, cy, cz, cx, cy, cz));
    n[0] = cx / len; n[1] = cy / len; n[2] = cz / len;
}

__device__ void frustum_from_rect(const Mat4& invProj, float x0, float y0, float x1, float y1, int vpW, int vpH, Frustum4& f)
{
    const float fw = (float)vpW, fh = (float)vpH;
    float vs[5][3];
    screen_to_view(invProj, x0, y0, -1.0f, 1.0f, fw, fh, vs[0]);
    screen_to_view(invProj, x1, y0, -1.0f, 1.0f, fw, fh, vs[1]);
    screen_to_view(invProj, x0, y1, -1.0f, 1.0f, fw, fh, vs[2]);
    screen_to_view(invProj, x1, y1, -1.0f, 1.0f, fw, fh, vs[3]);
    // screenSpace[4] = (screenSpace[0] + screenSpace[3]) * 0.5
    screen_to_view(invProj, (x0 + x1) * 0.5f, (y0 + y1) * 0.5f, (-1.0f + -1.0f) * 0.5f, (1.0f + 1.0f) * 0.5f, fw, fh, vs[4]);
    compute_plane_normal(vs[2], vs[0], f.n[0]); // left
    compute_plane_normal(vs[1], vs[3], f.n[1]); // right
    compute_plane_normal(vs[0], vs[1], f.n[2]); // top
    compute_plane_normal(vs[3], vs[2], f.n[3]); // bottom
    f.cx = vs[4][0];
    f.cy = vs[4][1];
}

// ------------------------------------------------------------------------------------------------------------
// K1a: depth bounds (ComputeLightCulling.shader:119-128) + tile frustum, 16 tiles per block.
// tileInfo[t] = { (n0, cx), (n1, cy), (n2, zNear'), (n3, zFar') } with the near/far swap of :171-177 applied.
// ------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ uint32_t depth_bits(const float* __restrict__ depth, int W, int H, int bandRow0, int gx, int gy)
{
    int col = gx < W - 1 ? gx : W - 1;
    int row = H - 1 - gy;
    row = row < 0 ? 0 : (row > H - 1 ? H - 1 : row);
    return __float_as_uint(depth[(size_t)(row - bandRow0) * W + col]);
}

__global__ __launch_bounds__(256) void k1_tile_setup(Mat4 invProj, int vpW, int vpH, const float* __restrict__ depth, int W, int H,
                                                      int Tx, int tileRow0, int bandRow0, int stripsPerRow, float4* __restrict__ tileInfo)
{
    __shared__ uint32_t sMin[4][16], sMax[4][16];
    const int strip = blockIdx.x % stripsPerRow;
    const int tyLocal = blockIdx.x / stripsPerRow;
    const int ty = tileRow0 + tyLocal;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int gx0 = strip * 256 + lane * 4; // 4 pixels per lane, 4 lanes per tile
    uint32_t mn = 0xFFFFFFFFu, mx = 0u;
    const bool vec = ((W & 3) == 0) && (gx0 + 3 < W);
#pragma unroll
    for (int k = 0; k < 4; k++) {
        const int gy = ty * TILE + wave * 4 + k;
        if (vec) {
            int row = H - 1 - gy;
            row = row < 0 ? 0 : row;
            const float4 d = *reinterpret_cast<const float4*>(depth + (size_t)(row - bandRow0) * W + gx0);
            const uint32_t a = __float_as_uint(d.x), b = __float_as_uint(d.y), c = __float_as_uint(d.z), e = __float_as_uint(d.w);
            mn = min(min(mn, a), min(b, min(c, e)));
            mx = max(max(mx, a), max(b, max(c, e)));
        } else {
#pragma unroll
            for (int q = 0; q < 4; q++) {
                const uint32_t a = depth_bits(depth, W, H, bandRow0, gx0 + q, gy);
                mn = min(mn, a);
                mx = max(mx, a);
            }
        }
    }
    // 4 lanes share a tile
    mn = min(mn, (uint32_t)__shfl_xor((int)mn, 1)); mx = max(mx, (uint32_t)__shfl_xor((int)mx, 1));
    mn = min(mn, (uint32_t)__shfl_xor((int)mn, 2)); mx = max(mx, (uint32_t)__shfl_xor((int)mx, 2));
    if ((lane & 3) == 0) { sMin[wave][lane >> 2] = mn; sMax[wave][lane >> 2] = mx; }
    __syncthreads();
    if (threadIdx.x < 16) {
        const int tx = strip * 16 + threadIdx.x;
        if (tx < Tx) {
            const int i = threadIdx.x;
            const uint32_t bmn = min(min(sMin[0][i], sMin[1][i]), min(sMin[2][i], sMin[3][i]));
            const uint32_t bmx = max(max(sMax[0][i], sMax[1][i]), max(sMax[2][i], sMax[3][i]));
            float zFar = __uint_as_float(bmx), zNear = __uint_as_float(bmn);
            const float diff = zFar - zNear; // "Add extra bounds" (:174-177): swaps near and far in fp32
            zFar -= diff;
            zNear += diff;
            Frustum4 f;
            frustum_from_rect(invProj, (float)(tx * TILE), (float)(ty * TILE), (float)((tx + 1) * TILE), (float)((ty + 1) * TILE), vpW, vpH, f);
            float4* o = tileInfo + (size_t)(tyLocal * Tx + tx) * 4;
            o[0] = make_float4(f.n[0][0], f.n[0][1], f.n[0][2], f.cx);
            o[1] = make_float4(f.n[1][0], f.n[1][1], f.n[1][2], f.cy);
            o[2] = make_float4(f.n[2][0], f.n[2][1], f.n[2][2], zNear);
            o[3] = make_float4(f.n[3][0], f.n[3][1], f.n[3][2], zFar);
        }
    }
}

// ------------------------------------------------------------------------------------------------------------
// K1a2: macro tiles.  macroInfo[m] = 5 float4: 4 x (plane normal, -) + (maxZNear', minZFar', -, -)
// ------------------------------------------------------------------------------------------------------------
__global__ void k1_macro_setup(Mat4 invProj, int vpW, int vpH, int Tx, int tileRow0, int tileRow1, int macroX, int macroY0, int numMacros,
                               const float4* __restrict__ tileInfo, float4* __restrict__ macroInfo)
{
    const int m = blockIdx.x * blockDim.x + threadIdx.x;
    if (m >= numMacros) return;
    const int mx = m % macroX, my = macroY0 + m / macroX;
    float zN = -__builtin_inff(), zF = __builtin_inff();
    bool any = false;
    for (int ty = max(my * MACRO, tileRow0); ty < min(my * MACRO + MACRO, tileRow1); ty++)
        for (int tx = mx * MACRO; tx < min(mx * MACRO + MACRO, Tx); tx++) {
            const float4* ti = tileInfo + (size_t)((ty - tileRow0) * Tx + tx) * 4;
            const float a = ti[2].w, b = ti[3].w;
            // NaN windows never reject in the exact test (comparisons false) => widen to "never reject"
            zN = (a != a) ? __builtin_inff() : fmaxf(zN, a);
            zF = (b != b) ? -__builtin_inff() : fminf(zF, b);
            any = true;
        }
    if (!any) { zN = __builtin_inff(); zF = -__builtin_inff(); }
    Frustum4 f;
    frustum_from_rect(invProj, (float)(mx * MACRO * TILE), (float)(my * MACRO * TILE), (float)((mx + 1) * MACRO * TILE), (float)((my + 1) * MACRO * TILE), vpW, vpH, f);
    float4* o = macroInfo + (size_t)m * 5;
    for (int k = 0; k < 4; k++) o[k] = make_float4(f.n[k][0], f.n[k][1], f.n[k][2], 0.0f);
    o[4] = make_float4(zN, zF, 0.0f, 0.0f);
}

// Conservative macro-tile test: true = "some member tile might accept this light".
__device__ __forceinline__ bool macro_may_overlap(const float4 lv, const float4* __restrict__ mi, float planeMargin)
{
    const float r = lv.w;
    const float zlo = lv.z - r, zhi = lv.z + r; // the exact test's own expressions (Math.glsl:226)
    const float4 zw = mi[4];
    if (zlo > zw.x || zhi < zw.y) return false; // beyond every member tile's window
    // Side planes: valid as an outer bound only for spheres entirely in front of the eye; margin >> fp32 error.
    const float m = planeMargin * ((fabsf(lv.x) + fabsf(lv.y)) + (fabsf(lv.z) + fabsf(r)));
    if (zlo > m) {
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const float4 n = mi[k];
            if (dot3f(n.x, n.y, n.z, lv.x, lv.y, lv.z) < -(r + m)) return false;
        }
    }
    return true;
}

__device__ __forceinline__ uint64_t lanemask_lt()
{
    const uint32_t lane = threadIdx.x & 63;
    return lane == 0 ? 0ull : (~0ull >> (64 - lane));
}

// ------------------------------------------------------------------------------------------------------------
// K1b: macro cull.  grid = (numChunks, numMacros); survivors of the chunk, ascending, -> pool segment.
// Pool entries: light index | (directional ? 1<<31 : 0).
// ------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k1_macro_cull(const float4* __restrict__ lightView, const uint32_t* __restrict__ lightType, int N,
                                                      const float4* __restrict__ macroInfo, int numChunks, float planeMargin,
                                                      uint32_t* __restrict__ pool, uint32_t poolEntries, uint32_t* __restrict__ cursor,
                                                      uint2* __restrict__ segTable)
{
    __shared__ uint32_t sSurv[CHUNK];
    __shared__ uint32_t sWave[4];
    __shared__ uint32_t sBase;
    const int chunk = blockIdx.x, m = blockIdx.y;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    __shared__ float4 sInfo[5];
    if (threadIdx.x < 5) sInfo[threadIdx.x] = macroInfo[(size_t)m * 5 + threadIdx.x];
    __syncthreads();
    uint32_t count = 0;
    const int j0 = chunk * CHUNK;
    const int jEnd = min(j0 + CHUNK, N);
    for (int base = j0; base < jEnd; base += 256) {
        const int j = base + threadIdx.x;
        bool pass = false;
        uint32_t entry = 0;
        if (j < jEnd) {
            const float4 lv = lightView[j];
            const uint32_t type = lightType[j];
            if (type == 0u) { pass = true; entry = (uint32_t)j | 0x80000000u; }
            else { pass = macro_may_overlap(lv, sInfo, planeMargin); entry = (uint32_t)j; }
        }
        const uint64_t mask = __ballot(pass);
        if (lane == 0) sWave[wave] = (uint32_t)__popcll(mask);
        __syncthreads();
        uint32_t before = 0, total = 0;
#pragma unroll
        for (int w = 0; w < 4; w++) { const uint32_t c = sWave[w]; before += (w < wave) ? c : 0u; total += c; }
        if (pass) sSurv[count + before + (uint32_t)__popcll(mask & lanemask_lt())] = entry;
        count += total;
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        uint32_t b = count ? atomicAdd(cursor, count) : 0u;
        if (count && (b > poolEntries || count > poolEntries - b)) b = SEG_RAW;
        sBase = b;
        segTable[(size_t)m * numChunks + chunk] = make_uint2(b, count);
    }
    __syncthreads();
    const uint32_t b = sBase;
    if (b != SEG_RAW)
        for (uint32_t i = threadIdx.x; i < count; i += 256) pool[b + i] = sSurv[i];
}

// ------------------------------------------------------------------------------------------------------------
// K1c: exact per-tile cull, one wave per tile.
// ------------------------------------------------------------------------------------------------------------
struct TileCtx {
    float n[4][3];
    float cx, cy, cz, zNear, zFar;
};

// Math.glsl:224-239 SphereFrustumOverlaps + ComputeLightCulling.shader:187 impact
__device__ __forceinline__ bool tile_test(const TileCtx& t, const float4 lv, float& impact)
{
    const float r = lv.w;
    if (lv.z - r > t.zNear || lv.z + r < t.zFar) return false;
#pragma unroll
    for (int k = 0; k < 4; k++)
        if (dot3f(t.n[k][0], t.n[k][1], t.n[k][2], lv.x, lv.y, lv.z) < -r) return false;
    const float dx = lv.x - t.cx, dy = lv.y - t.cy, dz = lv.z - t.cz;
    impact = sqrtf(dot3f(dx, dy, dz, dx, dy, dz));
    return true;
}

__device__ __forceinline__ void wave_append(bool pass, uint32_t j, float impact, uint32_t& count, uint32_t* sIdx, float* sImp)
{
    const uint64_t mask = __ballot(pass);
    const uint32_t pos = count + (uint32_t)__popcll(mask & lanemask_lt());
    if (pass && pos < CAND) { sIdx[pos] = j; sImp[pos] = impact; }
    count += (uint32_t)__popcll(mask);
}

// walk the raw light range [j0, j1) in ascending order (brute force / overflowed segment)
__device__ __forceinline__ void walk_raw(const TileCtx& t, const float4* __restrict__ lightView, const uint32_t* __restrict__ lightType,
                                         int j0, int j1, uint32_t& count, uint32_t* sIdx, float* sImp)
{
    const int lane = threadIdx.x & 63;
    for (int base = j0; base < j1 && count < CAND; base += 64) {
        const int j = base + lane;
        bool pass = false;
        float impact = 0.0f;
        if (j < j1) {
            const float4 lv = lightView[j];
            if (lightType[j] == 0u) pass = true; // directional: always a candidate, impact 0 (:153-162)
            else pass = tile_test(t, lv, impact);
        }
        wave_append(pass, (uint32_t)j, impact, count, sIdx, sImp);
    }
}

template <bool BRUTE>
__global__ __launch_bounds__(64) void k1_tile_cull(const float4* __restrict__ lightView, const uint32_t* __restrict__ lightType, int N,
                                                    const float4* __restrict__ tileInfo, int Tx, int tileRow0, int macroX, int macroY0,
                                                    int numChunks, const uint2* __restrict__ segTable, const uint32_t* __restrict__ pool,
                                                    uint32_t* __restrict__ tileNum, uint32_t* __restrict__ tileList)
{
    __shared__ uint32_t sIdx[CAND];
    __shared__ float sImp[CAND];
    const int bandTile = blockIdx.x;
    const int lane = threadIdx.x;
    TileCtx t;
    {
        const float4* ti = tileInfo + (size_t)bandTile * 4;
        const float4 a = ti[0], b = ti[1], c = ti[2], d = ti[3];
        t.n[0][0] = a.x; t.n[0][1] = a.y; t.n[0][2] = a.z; t.cx = a.w;
        t.n[1][0] = b.x; t.n[1][1] = b.y; t.n[1][2] = b.z; t.cy = b.w;
        t.n[2][0] = c.x; t.n[2][1] = c.y; t.n[2][2] = c.z; t.zNear = c.w;
        t.n[3][0] = d.x; t.n[3][1] = d.y; t.n[3][2] = d.z; t.zFar = d.w;
        t.cz = (t.zFar + t.zNear) * 0.5f;
    }
    uint32_t count = 0;
    if (BRUTE) {
        walk_raw(t, lightView, lightType, 0, N, count, sIdx, sImp);
    } else {
        const int tx = bandTile % Tx, ty = tileRow0 + bandTile / Tx;
        const int m = (ty / MACRO - macroY0) * macroX + tx / MACRO;
        const uint2* segs = segTable + (size_t)m * numChunks;
        for (int c = 0; c < numChunks && count < CAND; c++) {
            const uint2 seg = segs[c];
            if (seg.y == 0u) continue;
            if (seg.x == SEG_RAW) {
                walk_raw(t, lightView, lightType, c * CHUNK, min((c + 1) * CHUNK, N), count, sIdx, sImp);
                continue;
            }
            const uint32_t* __restrict__ list = pool + seg.x;
            for (uint32_t base = 0; base < seg.y && count < CAND; base += 64) {
                const uint32_t i = base + lane;
                bool pass = false;
                float impact = 0.0f;
                uint32_t j = 0;
                if (i < seg.y) {
                    const uint32_t e = list[i];
                    j = e & 0x7FFFFFFFu;
                    if (e & 0x80000000u) pass = true;
                    else pass = tile_test(t, lightView[j], impact);
                }
                wave_append(pass, j, impact, count, sIdx, sImp);
            }
        }
    }
    const uint32_t n = count < CAND ? count : CAND;
    const uint32_t num = n < KEEP ? n : KEEP;
    __syncthreads(); // single wave: orders the LDS writes above with the reads below
    uint32_t* out = tileList + (size_t)bandTile * KEEP;
    if (n <= KEEP) {
        // :235-238 culledLights.indices[offset + i] = candidateIndices[numCandidates - i - 1]
        for (uint32_t i = lane; i < num; i += 64) out[i] = sIdx[n - 1 - i];
    } else {
        // :198-225 partial bubble sort == rank under (impact ascending, candidate position descending); keep rank < 128
        for (uint32_t k = lane; k < n; k += 64) {
            const float f = sImp[k];
            uint32_t rank = 0;
            for (uint32_t q = 0; q < n; q++) {
                const float g = sImp[q];
                rank += (g < f || (g == f && q > k)) ? 1u : 0u;
            }
            if (rank < KEEP) out[rank] = sIdx[k];
        }
    }
    if (lane == 0) tileNum[bandTile] = num;
}

// ------------------------------------------------------------------------------------------------------------
// K1d: canonical offsets (Appendix A step 6) and compaction
// ------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(1024) void k1_scan(const uint32_t* __restrict__ tileNum, int T, SailorLightsGrid* __restrict__ grid, uint32_t* __restrict__ culled)
{
    __shared__ uint32_t sWave[16];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int per = (T + 1023) / 1024;
    const int b = tid * per, e = min(b + per, T);
    uint32_t sum = 0;
    for (int i = b; i < e; i++) sum += tileNum[i];
    uint32_t incl = sum;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const uint32_t v = (uint32_t)__shfl_up((int)incl, d);
        if (lane >= d) incl += v;
    }
    if (lane == 63) sWave[wave] = incl;
    __syncthreads();
    uint32_t waveBase = 0, total = 0;
#pragma unroll
    for (int w = 0; w < 16; w++) { const uint32_t c = sWave[w]; waveBase += (w < wave) ? c : 0u; total += c; }
    uint32_t run = waveBase + incl - sum;
    for (int i = b; i < e; i++) {
        const uint32_t n = tileNum[i];
        grid[i].offset = run + 1u;
        grid[i].num = n;
        run += n;
    }
    if (tid == 0) culled[0] = total;
}

__global__ __launch_bounds__(256) void k1_pack(const SailorLightsGrid* __restrict__ grid, const uint32_t* __restrict__ tileList, int T,
                                                uint32_t* __restrict__ culled, uint32_t capacity)
{
    const int tile = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (tile >= T) return;
    const int lane = threadIdx.x & 63;
    const SailorLightsGrid g = grid[tile];
    const uint32_t* src = tileList + (size_t)tile * KEEP;
    for (uint32_t i = lane; i < g.num; i += 64)
        if (g.offset + i < capacity) culled[g.offset + i] = src[i];
}

__global__ void k_grid_rebase(SailorLightsGrid* grid, int T, uint32_t base)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < T) grid[i].offset += base;
}

// ------------------------------------------------------------------------------------------------------------
// host entry points
// ------------------------------------------------------------------------------------------------------------
static bool band_valid(int W, int H, const SailorBand* b)
{
    if (!b) return false;
    const int Ty = (H - 1) / TILE + 1;
    if (b->tileRowBegin < 0 || b->tileRowEnd > Ty || b->tileRowBegin > b->tileRowEnd) return false;
    int lo = H - TILE * b->tileRowEnd, hi = H - TILE * b->tileRowBegin;
    if (lo < 0) lo = 0;
    if (hi > H) hi = H;
    if (hi < lo) hi = lo;
    return b->fbRowBegin == lo && b->fbRowCount == hi - lo;
}

extern "C" {

size_t sailor_hip_light_cull_workspace_size(int32_t width, int32_t height, int32_t lightsCapacity, const SailorBand* band)
{
    if (width <= 0 || height <= 0 || lightsCapacity < 0) return 0;
    SailorBand whole;
    if (!band) { sailor_hip_band_whole_frame(width, height, &whole); band = &whole; }
    if (!band_valid(width, height, band)) return 0;
    return make_layout(width, height, lightsCapacity, *band).total;
}

int sailor_hip_light_cull(SailorHipContext* ctx, const SailorUboFrameData* frame, const SailorLightCullPushConstants* pc,
                          const SailorLightShaderData* dLights, const float* dLinearDepth,
                          SailorLightsGrid* dLightsGrid, uint32_t* dCulledLights, size_t culledCapacity,
                          void* dWorkspace, size_t workspaceBytes, const SailorBand* band, uint32_t flags)
{
    if (!ctx || !frame || !pc || !dLinearDepth || !dLightsGrid || !dCulledLights || !dWorkspace) return SAILOR_HIP_ERR_INVALID_ARGUMENT;
    const int W = pc->viewportSize[0], H = pc->viewportSize[1], N = pc->lightsNum;
    if (W <= 0 || H <= 0 || N < 0 || (N > 0 && !dLights)) return SAILOR_HIP_ERR_INVALID_ARGUMENT;
    // Appendix A: the depth extent (push constants) and the window viewport (frame UBO) must agree
    if (frame->viewportSize[0] != W || frame->viewportSize[1] != H) return SAILOR_HIP_ERR_UNSUPPORTED;
    SailorBand whole;
    if (!band) { sailor_hip_band_whole_frame(W, H, &whole); band = &whole; }
    if (!band_valid(W, H, band)) return SAILOR_HIP_ERR_INVALID_ARGUMENT;
    const CullLayout L = make_layout(W, H, N, *band);
    if (pc->numTiles[0] != L.Tx || pc->numTiles[1] != L.Ty) return SAILOR_HIP_ERR_INVALID_ARGUMENT;
    if (workspaceBytes < L.total) return SAILOR_HIP_ERR_WORKSPACE_TOO_SMALL;
    if (culledCapacity < 1) return SAILOR_HIP_ERR_INVALID_ARGUMENT;
    if (((uintptr_t)dWorkspace & 255) != 0) return SAILOR_HIP_ERR_INVALID_ARGUMENT;

    hipStream_t s = ctx->stream;
    char* ws = (char*)dWorkspace;
    float4* lightView = (float4*)(ws + L.offLightView);
    uint32_t* lightType = (uint32_t*)(ws + L.offLightType);
    float4* tileInfo = (float4*)(ws + L.offTileInfo);
    float4* macroInfo = (float4*)(ws + L.offMacroInfo);
    uint2* segTable = (uint2*)(ws + L.offSegTable);
    uint32_t* cursor = (uint32_t*)(ws + L.offCursor);
    uint32_t* tileNum = (uint32_t*)(ws + L.offTileNum);
    uint32_t* tileList = (uint32_t*)(ws + L.offTileList);
    uint32_t* pool = (uint32_t*)(ws + L.offPool);

    if (L.bandTiles == 0) {
        SAILOR_TRY_HIP(ctx, hipMemsetAsync(dCulledLights, 0, 4, s));
        return SAILOR_HIP_OK;
    }

    Mat4 view, invProj;
    memcpy(view.m, frame->view, 64);
    memcpy(invProj.m, frame->invProjection, 64);

    if (N > 0) {
        hipLaunchKernelGGL(k0_light_view, dim3((N + 255) / 256), dim3(256), 0, s, view, dLights, N, lightView, lightType);
        SAILOR_CHECK_LAUNCH(ctx, "k0_light_view");
    }
    const int stripsPerRow = (L.Tx + 15) / 16;
    const int rows = band->tileRowEnd - band->tileRowBegin;
    hipLaunchKernelGGL(k1_tile_setup, dim3(stripsPerRow * rows), dim3(256), 0, s, invProj, frame->viewportSize[0], frame->viewportSize[1],
                       dLinearDepth, W, H, L.Tx, band->tileRowBegin, band->fbRowBegin, stripsPerRow, tileInfo);
    SAILOR_CHECK_LAUNCH(ctx, "k1_tile_setup");

    // The side-plane margin argument needs a sane perspective (|N_tile| / |N_macro| bounded); otherwise brute force.
    const float p00 = fabsf(frame->projection[0]), p11 = fabsf(frame->projection[5]);
    const bool sane = p00 > 1e-2f && p11 > 1e-2f && p00 < 1e4f && p11 < 1e4f;
    const bool brute = (flags & SAILOR_CULL_BRUTE_FORCE) || !sane || N < 4 * CHUNK / 8;
    if (brute) {
        hipLaunchKernelGGL(k1_tile_cull<true>, dim3(L.bandTiles), dim3(64), 0, s, lightView, lightType, N, tileInfo, L.Tx, band->tileRowBegin,
                           L.macroX, L.macroY0, L.numChunks, segTable, pool, tileNum, tileList);
        SAILOR_CHECK_LAUNCH(ctx, "k1_tile_cull<brute>");
    } else {
        SAILOR_TRY_HIP(ctx, hipMemsetAsync(cursor, 0, 4, s));
        hipLaunchKernelGGL(k1_macro_setup, dim3((L.numMacros + 63) / 64), dim3(64), 0, s, invProj, frame->viewportSize[0], frame->viewportSize[1],
                           L.Tx, band->tileRowBegin, band->tileRowEnd, L.macroX, L.macroY0, L.numMacros, tileInfo, macroInfo);
        SAILOR_CHECK_LAUNCH(ctx, "k1_macro_setup");
        const float planeMargin = 1e-3f;
        hipLaunchKernelGGL(k1_macro_cull, dim3(L.numChunks, L.numMacros), dim3(256), 0, s, lightView, lightType, N, macroInfo, L.numChunks, planeMargin,
                           pool, (uint32_t)(L.poolEntries > 0xFFFFFFF0u ? 0xFFFFFFF0u : L.poolEntries), cursor, segTable);
        SAILOR_CHECK_LAUNCH(ctx, "k1_macro_cull");
        hipLaunchKernelGGL(k1_tile_cull<false>, dim3(L.bandTiles), dim3(64), 0, s, lightView, lightType, N, tileInfo, L.Tx, band->tileRowBegin,
                           L.macroX, L.macroY0, L.numChunks, segTable, pool, tileNum, tileList);
        SAILOR_CHECK_LAUNCH(ctx, "k1_tile_cull");
    }
    hipLaunchKernelGGL(k1_scan, dim3(1), dim3(1024), 0, s, tileNum, L.bandTiles, dLightsGrid, dCulledLights);
    SAILOR_CHECK_LAUNCH(ctx, "k1_scan");
    hipLaunchKernelGGL(k1_pack, dim3((L.bandTiles + 3) / 4), dim3(256), 0, s, dLightsGrid, tileList, L.bandTiles, dCulledLights,
                       (uint32_t)(culledCapacity > 0xFFFFFFFFull ? 0xFFFFFFFFull : culledCapacity));
    SAILOR_CHECK_LAUNCH(ctx, "k1_pack");
    return SAILOR_HIP_OK;
}

int sailor_hip_light_grid_rebase(SailorHipContext* ctx, SailorLightsGrid* dLightsGrid, int32_t numTiles, uint32_t globalBase)
{
    if (!ctx || !dLightsGrid || numTiles < 0) return SAILOR_HIP_ERR_INVALID_ARGUMENT;
    if (numTiles == 0 || globalBase == 0) return SAILOR_HIP_OK;
    hipLaunchKernelGGL(k_grid_rebase, dim3((numTiles + 255) / 256), dim3(256), 0, ctx->stream, dLightsGrid, numTiles, globalBase);
    SAILOR_CHECK_LAUNCH(ctx, "k_grid_rebase");
    return SAILOR_HIP_OK;
}

} // extern "C"
