// K0 + K1 -- screen-tile light cull for gfx950.
//
// Replaces Content/Shaders/ComputeLightCulling.shader (+ Math.glsl:116-173,224-239) as dispatched by
// LightCullingNode::Process (FrameGraph/LightCullingNode.cpp:74-77), under the canonical sequential semantics of
// SURVEY.md Appendix A.  Not a translation of the GLSL: the reference runs one 16x16 workgroup per tile that
// re-reads and re-transforms every light, appends with LDS atomics, bubble-sorts on one thread and allocates
// output space with a global atomic.  Here:
//
//   (k01_prepare runs the next two in one launch: they are independent)
//   k0_light_view      once per light: view-space position + radius into a float4 SoA (same fp32 op sequence as
//                      ComputeLightCulling.shader:164-169, so bits are identical), light type into a u32 SoA
//   k1_tile_setup      streaming pass over the linear-depth image (the only large HBM stream of the cull):
//                      16 tiles per 256-thread block, 1 KiB coalesced float4 row segments, wave shuffles + a
//                      4-entry LDS combine for min/max, then 16 lanes build the 16 tile frusta
//   k1_band_masks      conservative pre-filter as BITMASKS: one 64-bit ballot per (64 lights, column of 4x4-tile groups)
//                      and per (64 lights, row of groups): "this light's sphere may reach this 64-pixel band".  Pure
//                      streaming, no atomics, no compaction, no inter-block order: 3 MB of masks at 4K / 65 536 lights
//   k1_group_lists     one block per 4x4-tile group: (its column's mask) AND (its row's mask), 16 384 lights per
//                      step; the few surviving bits become an ordered, contiguous candidate list (ballot, readlane, mbcnt)
//   k1_tile_cull       one 256-thread block per 2x2 quarter of a group, one wave per tile: the group's candidate records
//                      are staged in LDS 1024 at a time, each wave streams them through the exact test, 64 per step;
//                      ballot/popcount ordered append; rank-based nearest-128 selection
//                      (groups denser than 2048 candidates fall back to walking the two masks of the tile itself)
//   k1_block_sums / k1_pack  canonical offsets = prefix sum over tiles in tile-index order, then compaction
//
// Bit-exactness: the pre-filter only ever removes lights that every tile of the column (row) would reject by its own
// left/right (top/bottom) plane: the band's planes are the same planes through the eye (screen x = const, resp.
// y = const), used only for spheres entirely in front of the eye and with a relative margin ~1000x the fp32 error;
// depth is left entirely to the exact test.  So the candidate sequence of every tile -- and therefore its first 196,
// its selection and its order -- is the sequence the brute-force walk produces.  tests/test_light_cull_gpu.py checks
// default == brute force == oracle.
//
// No MFMA: sphere/plane tests and compaction, not a contraction.  Compiled with -ffp-contract=off.
#include "common.h"
#include <vector>

#define BANDS_PER_GROUP 24   // group columns / rows handled per k1_band_masks wave (grid.y = ceil(bands / 24))
#define QCAP 128             // LDS candidate queue of k1_tile_cull (ring buffer: < 64 pending + <= 64 new)
#define SCAN_BLOCK 1024      // tiles per k1_block_sums block
#define GROUP 4              // k1_group_lists: tiles per group edge (4x4 tiles share one candidate list)
#define CAPG 2048            // entries per group list; a denser group falls back to walking the masks per tile
#define GROUP_OVERFLOW 0xFFFFFFFFu
#define CHUNK 512            // group candidates staged in LDS per step of k1_tile_cull (16 KB of LDS per block: 8 blocks per CU)

struct CullLayout {
    int Tx, Ty, bandRows, bandTiles, numBands, words, sumBlocks, groupsX, groupsY, numGroups;
    size_t offLightView, offLightType, offTileInfo, offBandPlanes, offMasks, offDirWords, offTileNum, offTilePrefix, offTileList, offBlockSums, offGroupCount, offGroupList,
        offClassPrefix, offClassSums, offTileOrder, total;
};

static CullLayout make_layout(int W, int H, int N, const SailorBand& band)
{
    CullLayout L;
    L.Tx = (W - 1) / TILE + 1;
    L.Ty = (H - 1) / TILE + 1;
    L.bandRows = band.tileRowEnd - band.tileRowBegin;
    L.bandTiles = L.bandRows * L.Tx;
    L.words = (N + 63) / 64;
    if (L.words < 1) L.words = 1;
    const size_t n = (size_t)(N > 0 ? N : 1);
    const size_t tiles = (size_t)(L.bandTiles > 0 ? L.bandTiles : 1);
    L.sumBlocks = (int)((tiles + SCAN_BLOCK - 1) / SCAN_BLOCK);
    L.groupsX = (L.Tx + GROUP - 1) / GROUP;
    L.groupsY = (L.bandRows + GROUP - 1) / GROUP;
    L.numGroups = L.groupsX * L.groupsY;
    L.numBands = L.groupsX + L.groupsY;      // group columns first, then the band's group rows (4 tiles wide / high)
    const size_t groups = (size_t)(L.numGroups > 0 ? L.numGroups : 1);
    size_t o = 0;
    L.offLightView = o; o = align_up(o + n * 16, 256);
    L.offLightType = o; o = align_up(o + n * 4, 256);
    L.offTileInfo = o; o = align_up(o + tiles * 64, 256);
    L.offBandPlanes = o; o = align_up(o + (size_t)(L.numBands > 0 ? L.numBands : 1) * 32, 256);
    L.offMasks = o; o = align_up(o + (size_t)(L.numBands > 0 ? L.numBands : 1) * L.words * 8, 256);
    L.offDirWords = o; o = align_up(o + (size_t)L.words * 8, 256);
    L.offTileNum = o; o = align_up(o + tiles * 4, 256);
    L.offTilePrefix = o; o = align_up(o + tiles * 4, 256);
    L.offTileList = o; o = align_up(o + tiles * KEEP * 4, 256);
    L.offBlockSums = o; o = align_up(o + (size_t)L.sumBlocks * 4, 256);
    L.offGroupCount = o; o = align_up(o + groups * 4, 256);
    L.offGroupList = o; o = align_up(o + groups * CAPG * 4, 256);
    L.offClassPrefix = o; o = align_up(o + tiles * 4, 256);
    L.offClassSums = o; o = align_up(o + (size_t)L.sumBlocks * 4, 256);
    L.offTileOrder = o; o = align_up(o + (tiles + 1) * 4, 256); // + the number of class A and B tiles
    L.total = o;
    return L;
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
// K0: ComputeLightCulling.shader:164-169 hoisted out of the per-tile loop (it does not depend on the tile).
// The tail blocks of the same launch build the conservative band planes (one thread per column / row of tile groups).
// ------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ void k0_light_view(int block, const Mat4& view, const SailorLightShaderData* __restrict__ lights, int N, int lightBlocks,
                                              float4* __restrict__ lightView, uint32_t* __restrict__ lightType,
                                              const Mat4& invProj, int vpW, int vpH, int Tx, int Ty, int tileRow0, int bandRows, int groupsX, int numBands,
                                              float4* __restrict__ bandPlanes)
{
    if (block >= lightBlocks) {
        const int b = (block - lightBlocks) * 256 + threadIdx.x;
        if (b >= numBands) return;
        Frustum4 f;
        if (b < groupsX) { // group column b (tile columns 4b .. 4b+3): planes through the eye and the screen lines x = 64 b, x = 64 (b + 1)
            frustum_from_rect(invProj, (float)(b * GROUP * TILE), 0.0f, (float)(min((b + 1) * GROUP, Tx) * TILE), (float)(Ty * TILE), vpW, vpH, f);
            bandPlanes[2 * b + 0] = make_float4(f.n[0][0], f.n[0][1], f.n[0][2], 0.0f);
            bandPlanes[2 * b + 1] = make_float4(f.n[1][0], f.n[1][1], f.n[1][2], 0.0f);
        } else {           // group row: the band's tile rows 4 gy .. 4 gy + 3
            const int gy = b - groupsX;
            const int ty = tileRow0 + gy * GROUP, tyEnd = tileRow0 + min((gy + 1) * GROUP, bandRows);
            frustum_from_rect(invProj, 0.0f, (float)(ty * TILE), (float)(Tx * TILE), (float)(tyEnd * TILE), vpW, vpH, f);
            bandPlanes[2 * b + 0] = make_float4(f.n[2][0], f.n[2][1], f.n[2][2], 0.0f);
            bandPlanes[2 * b + 1] = make_float4(f.n[3][0], f.n[3][1], f.n[3][2], 0.0f);
        }
        return;
    }
    const int j = block * 256 + threadIdx.x;
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

__device__ __forceinline__ void k1_tile_setup(int block, const Mat4& invProj, int vpW, int vpH, const float* __restrict__ depth, int W, int H,
                                              int Tx, int tileRow0, int bandRow0, int stripsPerRow, int vecOK, int rawDepth, float zNearCam,
                                              float4* __restrict__ tileInfo, unsigned char* __restrict__ lds)
{
    uint32_t (*sMin)[16] = reinterpret_cast<uint32_t (*)[16]>(lds), (*sMax)[16] = reinterpret_cast<uint32_t (*)[16]>(lds + 256); // [4][16] each
    const int strip = block % stripsPerRow;
    const int tyLocal = block / stripsPerRow;
    const int ty = tileRow0 + tyLocal;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int gx0 = strip * 256 + lane * 4; // 4 pixels per lane, 4 lanes per tile
    uint32_t mn = 0xFFFFFFFFu, mx = 0u;
    const bool vec = vecOK && (gx0 + 3 < W);
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
            if (rawDepth) {
                // SAILOR_CULL_RAW_DEPTH: the image holds the reversed-Z attachment.  x -> fl(zNear / x) is monotone
                // non-increasing on x >= 0, so the largest linear depth of the tile is the linearised smallest raw value
                // and vice versa -- bit for bit what min / max over the linearised texels give (LinearizeDepth.shader:70,74).
                const float lo = -(-zNearCam / zFar), hi = -(-zNearCam / zNear);
                zNear = lo; zFar = hi;
            }
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

// One launch for the two independent preparation passes: blocks [0, setupBlocks) stream the depth image (K1a, the long
// pole: they are dispatched first), the rest transform the lights and build the band planes (K0).
struct PrepareArgs {
    Mat4 view, invProj;
    const SailorLightShaderData* lights;
    const float* depth;
    float4* lightView; uint32_t* lightType; float4* tileInfo; float4* bandPlanes;
    int N, lightBlocks, setupBlocks, vpW, vpH, W, H, Tx, Ty, tileRow0, bandRow0, bandRows, groupsX, numBands, stripsPerRow, vecOK, rawDepth;
    float zNearCam;
};

// Every stage below is a device function over (block index, LDS) plus a thin __global__ wrapper: the same bodies run inside the fused
// "shade slice of frame k + cull stage of frame k+1" launches at the end of this file.
#define LDS_K01_PREPARE 512
__device__ __forceinline__ void k01_prepare_body(const int b, unsigned char* __restrict__ lds, const PrepareArgs& a)
{
    if (b < a.setupBlocks)
        k1_tile_setup(b, a.invProj, a.vpW, a.vpH, a.depth, a.W, a.H, a.Tx, a.tileRow0, a.bandRow0, a.stripsPerRow, a.vecOK, a.rawDepth, a.zNearCam, a.tileInfo, lds);
    else
        k0_light_view(b - a.setupBlocks, a.view, a.lights, a.N, a.lightBlocks, a.lightView, a.lightType, a.invProj, a.vpW, a.vpH, a.Tx, a.Ty, a.tileRow0,
                      a.bandRows, a.groupsX, a.numBands, a.bandPlanes);
}
__global__ __launch_bounds__(256) void k01_prepare(PrepareArgs a)
{
    __shared__ __attribute__((aligned(16))) unsigned char lds[LDS_K01_PREPARE];
    k01_prepare_body((int)blockIdx.x, lds, a);
}

// ------------------------------------------------------------------------------------------------------------
// K1b: band masks.  masks[b][word] bit k = "light 64*word + k may reach group column / group row b".
// A light is dropped from a band only if its sphere is entirely in front of the eye AND entirely outside one of the
// band's two planes by more than the margin; directional lights and everything doubtful stay in.
// ------------------------------------------------------------------------------------------------------------
#define LDS_K1_BAND_MASKS (2 * BANDS_PER_GROUP * 16)
__device__ __forceinline__ void k1_band_masks_body(const unsigned bx, const unsigned by, unsigned char* __restrict__ lds, const float4* __restrict__ lightView,
                                                   const uint32_t* __restrict__ lightType, int N, int words, const float4* __restrict__ bandPlanes, int numBands,
                                                   float planeMargin, unsigned long long* __restrict__ masks, unsigned long long* __restrict__ dirWords)
{
    float4* sPl = reinterpret_cast<float4*>(lds); // [2 * BANDS_PER_GROUP]
    const int b0 = by * BANDS_PER_GROUP;
    const int nb = min(BANDS_PER_GROUP, numBands - b0);
    if ((int)threadIdx.x < 2 * nb) sPl[threadIdx.x] = bandPlanes[2 * b0 + threadIdx.x]; // this block's planes, read once
    __syncthreads();
    const int word = bx * 4 + (threadIdx.x >> 6);
    if (word >= words) return;
    const int lane = threadIdx.x & 63;
    const int j = word * 64 + lane;
    const bool valid = j < N;
    float4 lv = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
    uint32_t type = 1u;
    if (valid) { lv = lightView[j]; type = lightType[j]; }
    const float r = lv.w;
    const float m = planeMargin * ((fabsf(lv.x) + fabsf(lv.y)) + (fabsf(lv.z) + fabsf(r)));
    const bool inFront = (lv.z - r) > m; // false for NaN => never plane-culled
    const float thr = -(r + m);
    const bool keepAlways = valid && (type == 0u || !inFront);
    if (by == 0) { // one bit per light: "directional" (rides along into the group lists, saves a gather per candidate)
        const unsigned long long dm = __ballot(valid && type == 0u);
        if (lane == 0) dirWords[word] = dm;
    }
    for (int b = 0; b < nb; b++) {
        const float4 nA = sPl[2 * b + 0], nB = sPl[2 * b + 1]; // LDS broadcast reads
        const bool out = dot3f(nA.x, nA.y, nA.z, lv.x, lv.y, lv.z) < thr || dot3f(nB.x, nB.y, nB.z, lv.x, lv.y, lv.z) < thr;
        const unsigned long long mask = __ballot(keepAlways || (valid && !out));
        if (lane == 0) masks[(size_t)(b0 + b) * words + word] = mask;
    }
}
__global__ __launch_bounds__(256) void k1_band_masks(const float4* __restrict__ lightView, const uint32_t* __restrict__ lightType, int N, int words,
                                                      const float4* __restrict__ bandPlanes, int numBands, float planeMargin,
                                                      unsigned long long* __restrict__ masks, unsigned long long* __restrict__ dirWords)
{
    __shared__ __attribute__((aligned(16))) unsigned char lds[LDS_K1_BAND_MASKS];
    k1_band_masks_body(blockIdx.x, blockIdx.y, lds, lightView, lightType, N, words, bandPlanes, numBands, planeMargin, masks, dirWords);
}

__device__ __forceinline__ uint64_t lanemask_lt()
{
    const uint32_t lane = threadIdx.x & 63;
    return lane == 0 ? 0ull : (~0ull >> (64 - lane));
}

// ------------------------------------------------------------------------------------------------------------
// K1b2: candidate lists of 4x4-tile groups.  One 256-thread block per group ANDs the masks of the group's column and
// row (256 words = 16 384 lights per step) and writes the set bits -- ascending light index -- as a
// contiguous list: a block-wide prefix sum of the words' popcounts gives every word its output position, so the
// sparse-bits -> dense-list conversion is fully parallel.  Done once per 16 tiles, not per tile (profiles/r01: the
// per-tile scalar version saturated the CUs' scalar ALUs; a per-group scalar version was tail-bound by cluster groups).
// ------------------------------------------------------------------------------------------------------------
#define LDS_K1_GROUP_LISTS 16
#define GL_WPT 4 // words per thread and round in k1_group_lists
__device__ __forceinline__ void k1_group_lists_body(const unsigned bx, unsigned char* __restrict__ lds, const unsigned long long* __restrict__ masks,
                                                    const unsigned long long* __restrict__ dirWords, int words, int Tx, int bandRows, int groupsX,
                                                    uint32_t* __restrict__ groupCount, uint32_t* __restrict__ groupList)
{
    uint32_t* sW = reinterpret_cast<uint32_t*>(lds); // [4]
    const int g = bx, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const unsigned long long* __restrict__ c = masks + (size_t)(g % groupsX) * words;
    const unsigned long long* __restrict__ r = masks + (size_t)(groupsX + g / groupsX) * words;
    uint32_t* __restrict__ list = groupList + (size_t)g * CAPG;
    uint32_t base = 0; // entries written by earlier chunks (block-uniform)
    // GL_WPT consecutive words per thread and round: one block-wide scan (and its two barriers) per 256 * GL_WPT words -- at 1 M lights a group walks
    // 16 384 words, and the scan, not the 24 bytes per word, was what a round cost
    for (int w0 = 0; w0 < words; w0 += 256 * GL_WPT) {
        const int wFirst = w0 + (int)threadIdx.x * GL_WPT;
        unsigned long long m[GL_WPT], dm[GL_WPT];
        uint32_t cnt = 0;
        if (wFirst + GL_WPT <= words && (words & (GL_WPT - 1)) == 0) { // 8 * GL_WPT contiguous bytes per thread and array as 16-byte loads, coalesced across the wave
            const ulonglong2* c2 = reinterpret_cast<const ulonglong2*>(c + wFirst);
            const ulonglong2* r2 = reinterpret_cast<const ulonglong2*>(r + wFirst);
            const ulonglong2* d2 = reinterpret_cast<const ulonglong2*>(dirWords + wFirst);
#pragma unroll
            for (int j = 0; j < GL_WPT / 2; j++) {
                const ulonglong2 cv = c2[j], rv = r2[j], dv = d2[j];
                m[2 * j] = cv.x & rv.x; m[2 * j + 1] = cv.y & rv.y;
                dm[2 * j] = dv.x; dm[2 * j + 1] = dv.y;
            }
        } else {
#pragma unroll
            for (int j = 0; j < GL_WPT; j++) {
                const int w = wFirst + j;
                m[j] = 0ull; dm[j] = 0ull;
                if (w < words) {
                    m[j] = c[w] & r[w];
                    dm[j] = dirWords[w];
                }
            }
        }
#pragma unroll
        for (int j = 0; j < GL_WPT; j++) cnt += (uint32_t)__popcll(m[j]);
        // block-wide exclusive prefix sum of the popcounts: word order == light order
        uint32_t incl = cnt;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const uint32_t v = (uint32_t)__shfl_up((int)incl, d);
            if (lane >= d) incl += v;
        }
        if (lane == 63) sW[wave] = incl;
        __syncthreads();
        uint32_t before = 0, total = 0;
#pragma unroll
        for (int k = 0; k < 4; k++) { const uint32_t v = sW[k]; before += (k < wave) ? v : 0u; total += v; }
        uint32_t pos = base + before + incl - cnt;
#pragma unroll
        for (int j = 0; j < GL_WPT; j++) {
            const uint32_t first = (uint32_t)(wFirst + j) * 64u;
            unsigned long long mm = m[j];
            const unsigned long long dd = dm[j];
            while (mm != 0ull) { // this word's set bits, ascending
                const int bit = __builtin_ctzll(mm);
                mm &= mm - 1ull;
                if (pos < CAPG) list[pos] = (first + (uint32_t)bit) | (uint32_t)((dd >> bit) & 1ull) << 31; // bit 31 = directional
                pos++;
            }
        }
        base += total;
        __syncthreads();
    }
    if (threadIdx.x == 0) groupCount[g] = base > CAPG ? GROUP_OVERFLOW : base;
}
__global__ __launch_bounds__(256) void k1_group_lists(const unsigned long long* __restrict__ masks, const unsigned long long* __restrict__ dirWords,
                                                       int words, int Tx, int bandRows, int groupsX,
                                                       uint32_t* __restrict__ groupCount, uint32_t* __restrict__ groupList)
{
    __shared__ __attribute__((aligned(16))) unsigned char lds[LDS_K1_GROUP_LISTS];
    k1_group_lists_body(blockIdx.x, lds, masks, dirWords, words, Tx, bandRows, groupsX, groupCount, groupList);
}

// ------------------------------------------------------------------------------------------------------------
// K1c: exact per-tile cull, one wave per tile.
// ------------------------------------------------------------------------------------------------------------
struct TileCtx {
    float n[4][3];
    float cx, cy, cz, zNear, zFar;
};

// Math.glsl:224-239 SphereFrustumOverlaps + ComputeLightCulling.shader:187 impact
__device__ __forceinline__ bool tile_test(const TileCtx& t, const float4 lv)
{
    const float r = lv.w;
    if (lv.z - r > t.zNear || lv.z + r < t.zFar) return false;
#pragma unroll
    for (int k = 0; k < 4; k++)
        if (dot3f(t.n[k][0], t.n[k][1], t.n[k][2], lv.x, lv.y, lv.z) < -r) return false;
    return true;
}

// ComputeLightCulling.shader:187 impact = distance(light, frustum centre).  Only the 196 -> 128 selection reads it, so it
// is computed afterwards, for the <= 196 candidates of the few tiles that need a selection, not for every tested light.
__device__ __forceinline__ float tile_impact(const TileCtx& t, const float4 lv)
{
    const float dx = lv.x - t.cx, dy = lv.y - t.cy, dz = lv.z - t.cz;
    return sqrtf(dot3f(dx, dy, dz, dx, dy, dz));
}

// j carries the light index, bit 31 = directional (impact 0, :153-162)
__device__ __forceinline__ void wave_append(bool pass, uint32_t j, uint32_t& count, uint32_t* sIdx)
{
    const uint64_t mask = __ballot(pass);
    const uint32_t pos = count + (uint32_t)__popcll(mask & lanemask_lt());
    if (pass && pos < CAND) sIdx[pos] = j;
    count += (uint32_t)__popcll(mask);
}

// exact test of one candidate per lane (ascending light index across lanes), ordered append
__device__ __forceinline__ void test_candidates(const TileCtx& t, const float4* __restrict__ lightView, const uint32_t* __restrict__ lightType,
                                                bool have, uint32_t j, uint32_t& count, uint32_t* sIdx)
{
    bool pass = false;
    uint32_t dir = 0u;
    if (have) {
        const float4 lv = lightView[j];
        if (lightType[j] == 0u) { pass = true; dir = 0x80000000u; } // directional: always a candidate, impact 0 (:153-162)
        else pass = tile_test(t, lv);
    }
    wave_append(pass, j | dir, count, sIdx);
}

// LDS hand-off between the lanes of ONE wave: the wave's DS operations execute in order, so only the compiler has to be
// told not to move accesses across this point (and to wait for outstanding DS results).
#define WAVE_SYNC() __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup")

#define LDS_K1_TILE_CULL (CHUNK * 16 + CHUNK * 4 + 4 * CAND * 4 + 4 * CAND * 4)
template <bool BRUTE>
__device__ __forceinline__ void k1_tile_cull_body(const unsigned bx, unsigned char* __restrict__ lds, const float4* __restrict__ lightView,
                                                  const uint32_t* __restrict__ lightType, int N, int words, const float4* __restrict__ tileInfo, int Tx, int bandRows,
                                                  const unsigned long long* __restrict__ masks, const uint32_t* __restrict__ groupCount,
                                                  const uint32_t* __restrict__ groupList, int groupsX, uint32_t* __restrict__ tileNum, uint32_t* __restrict__ tileList)
{
    // One 256-thread block per 2x2 QUARTER of a 4x4-tile group, one wave per tile.  The group's candidate records are
    // staged in LDS, CHUNK at a time, by the four waves together (list entries first, then the dependent 16-byte
    // gathers, four of each in flight per thread), and every tile streams them out of LDS.  Four blocks per group, not
    // one of sixteen waves: a cluster group (2048 candidates, four 196 -> 128 selections per SIMD) used to keep ONE CU
    // busy for ~40 us while the rest of the chip idled -- the kernel's tail -- and at 26 KB of LDS six blocks fit a CU.
    float4* sLV = reinterpret_cast<float4*>(lds);                                                         // [CHUNK] candidate (view pos, radius)
    uint32_t* sE = reinterpret_cast<uint32_t*>(lds + CHUNK * 16);                                         // [CHUNK] candidate light index | directional << 31
    uint32_t (*sIdxAll)[CAND] = reinterpret_cast<uint32_t (*)[CAND]>(lds + CHUNK * 20);                   // [4][CAND]
    float (*sImpAll)[CAND] = reinterpret_cast<float (*)[CAND]>(lds + CHUNK * 20 + 4 * CAND * 4);          // [4][CAND], 16-byte aligned (CAND * 4 = 784)
    const int g = bx >> 2, quarter = bx & 3;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    uint32_t* sIdx = sIdxAll[wave];
    float* sImp = sImpAll[wave];
    const int tx = (g % groupsX) * GROUP + (quarter & 1) * 2 + (wave & 1), tyLocal = (g / groupsX) * GROUP + (quarter >> 1) * 2 + (wave >> 1);
    const bool active = tx < Tx && tyLocal < bandRows;
    const int bandTile = tyLocal * Tx + tx;
    TileCtx t;
    if (active) {
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
        if (!active) return;
        for (int base = 0; base < N && count < CAND; base += 64) {
            const int j = base + lane;
            test_candidates(t, lightView, lightType, j < N, (uint32_t)j, count, sIdx);
        }
    } else {
        // The kernel is bound by the latency of its dependent loads (a block has ~1 us of work behind 2-3 round trips to
        // L2 / HBM), so the first chunk's list entries are requested together with the group's count, not after it:
        // slots past the count hold stale entries of an earlier frame or nothing at all, hence the clamp to N - 1.
        const uint32_t* __restrict__ list = groupList + (size_t)g * CAPG;
        uint32_t e[CHUNK / 256];
#pragma unroll
        for (int k = 0; k < CHUNK / 256; k++) e[k] = list[threadIdx.x + 256u * k];
        const uint32_t gn = groupCount[g];
        if (gn != GROUP_OVERFLOW) {
            for (uint32_t c0 = 0; c0 < gn; c0 += CHUNK) {
                const uint32_t cn = min((uint32_t)CHUNK, gn - c0);
                if (c0) {
                    __syncthreads(); // every wave is done with the previous chunk
#pragma unroll
                    for (int k = 0; k < CHUNK / 256; k++) { const uint32_t i = threadIdx.x + 256u * k; e[k] = i < cn ? list[c0 + i] : 0u; }
                }
                float4 lv[CHUNK / 256];
#pragma unroll
                for (int k = 0; k < CHUNK / 256; k++) lv[k] = lightView[min(e[k] & 0x7FFFFFFFu, (uint32_t)(N - 1))];
#pragma unroll
                for (int k = 0; k < CHUNK / 256; k++) { const uint32_t i = threadIdx.x + 256u * k; if (i < cn) { sE[i] = e[k]; sLV[i] = lv[k]; } }
                __syncthreads();
                if (active) {
                    for (uint32_t base = 0; base < cn && count < CAND; base += 64u) {
                        const uint32_t i = base + (uint32_t)lane;
                        bool pass = false;
                        uint32_t ee = 0u;
                        if (i < cn) {
                            ee = sE[i];
                            if (ee & 0x80000000u) pass = true; // directional: always a candidate, impact 0 (:153-162)
                            else pass = tile_test(t, sLV[i]);
                        }
                        wave_append(pass, ee, count, sIdx);
                    }
                }
            }
            if (!active) return;
        } else {
        // overflowed group (very dense region / lights around the eye): every tile walks its own two masks
        if (!active) return;
        uint32_t* sQ = reinterpret_cast<uint32_t*>(sLV) + wave * QCAP; // sLV is unused on this path
        const unsigned long long* __restrict__ col = masks + (size_t)(g % groupsX) * words;
        const unsigned long long* __restrict__ row = masks + (size_t)(groupsX + g / groupsX) * words;
        uint32_t qHead = 0, qTail = 0; // ring buffer indices (wave-uniform)
        unsigned long long next = (lane < words) ? (col[lane] & row[lane]) : 0ull;
        for (int w0 = 0; w0 < words && count < CAND; w0 += 64) {
            const unsigned long long m = next;
            const int wn = w0 + 64 + lane;
            next = (wn < words) ? (col[wn] & row[wn]) : 0ull; // prefetch the next 4096 lights' masks
            // Scalar walk over the non-empty words of this step, in ascending order.  Each word's set bits go to the
            // ordered queue with one mbcnt (bit k of word L = light 64 (w0 + L) + k lands behind the k' < k bits).
            unsigned long long nz = __ballot(m != 0ull);
            while (nz != 0ull && count < CAND) {
                const int L = __builtin_ctzll(nz);
                nz &= nz - 1ull;
                const uint32_t lo = __builtin_amdgcn_readlane((uint32_t)m, L);
                const uint32_t hi = __builtin_amdgcn_readlane((uint32_t)(m >> 32), L);
                const unsigned long long mk = ((unsigned long long)hi << 32) | lo;
                if ((mk >> lane) & 1ull)
                    sQ[(qTail + (uint32_t)__popcll(mk & lanemask_lt())) & (QCAP - 1)] = (uint32_t)(w0 + L) * 64u + (uint32_t)lane;
                qTail += (uint32_t)__popcll(mk);
                if (qTail - qHead >= 64u) { // a full wave of candidates is waiting: exact-test them
                    WAVE_SYNC();
                    const uint32_t j = sQ[(qHead + lane) & (QCAP - 1)];
                    test_candidates(t, lightView, lightType, true, j, count, sIdx);
                    qHead += 64u;
                    WAVE_SYNC();
                }
            }
        }
        WAVE_SYNC();
        if (qTail != qHead && count < CAND) { // final partial round (< 64 pending)
            const uint32_t n = qTail - qHead;
            const uint32_t j = sQ[(qHead + lane) & (QCAP - 1)];
            test_candidates(t, lightView, lightType, (uint32_t)lane < n, j, count, sIdx);
        }
        }
    }
    const uint32_t n = count < CAND ? count : CAND;
    const uint32_t num = n < KEEP ? n : KEEP;
    WAVE_SYNC(); // orders the wave's LDS writes above with the reads below
    uint32_t* out = tileList + (size_t)bandTile * KEEP;
    if (n <= KEEP) {
        // :235-238 culledLights.indices[offset + i] = candidateIndices[numCandidates - i - 1]
        for (uint32_t i = lane; i < num; i += 64) out[i] = sIdx[n - 1 - i] & 0x7FFFFFFFu;
    } else {
        bool nanHere = false;
        for (uint32_t k = lane; k < n; k += 64) {
            const uint32_t e = sIdx[k];
            const float imp = (e & 0x80000000u) ? 0.0f : tile_impact(t, lightView[e]);
            sImp[k] = imp;
            nanHere = nanHere || imp != imp;
        }
        if (__ballot(nanHere) != 0ull) {
            // A NaN impact (tile with nothing drawn: depth +inf makes the frustum centre NaN; or a non-finite light) has no
            // rank: the shader's compare-and-swap (:207) is simply false next to it.  Literal semantics then: if no adjacent
            // pair can swap at all -- the sky-tile case, every impact NaN or a directional 0 -- the bubble sort is a no-op;
            // otherwise one lane replays it (NaNs act as walls; rare and slow, but the reference's answer).
            WAVE_SYNC();
            bool swapHere = false;
            for (uint32_t k = lane; k + 1 < n; k += 64) swapHere = swapHere || sImp[k] < sImp[k + 1];
            if (__ballot(swapHere) != 0ull) {
                if (lane == 0) {
                    uint32_t numSorted = KEEP;
                    for (uint32_t i = 0; i + 1 < n; i++) {
                        for (uint32_t j = 0; j < n - i - 1; j++) {
                            const float a = sImp[j], b = sImp[j + 1];
                            if (a < b) {
                                sImp[j] = b; sImp[j + 1] = a;
                                const uint32_t x = sIdx[j]; sIdx[j] = sIdx[j + 1]; sIdx[j + 1] = x;
                            }
                        }
                        if (--numSorted == 0) break;
                    }
                }
                WAVE_SYNC();
            }
            for (uint32_t i = lane; i < num; i += 64) out[i] = sIdx[n - 1 - i] & 0x7FFFFFFFu; // :235-238
            if (lane == 0) tileNum[bandTile] = num;
            return;
        }
        // :198-225 partial bubble sort == rank under (impact ascending, candidate position descending); keep rank < 128
        // each lane ranks up to 4 candidates (k = lane + 64 i) against all n, 4 impacts per LDS read.  Candidates q of
        // an earlier 64-block all have q < k (count g < f), of a later block all have q > k (count g <= f); only the
        // lane's own block needs the position tie-break.  Slots past n are padded with +inf and never counted.
        if (lane < 4 && (n & ~3u) + lane >= n && (n & ~3u) + lane < (uint32_t)CAND) sImp[(n & ~3u) + lane] = __builtin_inff();
        WAVE_SYNC();
        float f[4];
        uint32_t rank[4] = { 0u, 0u, 0u, 0u };
#pragma unroll
        for (int i = 0; i < 4; i++) { const uint32_t k = lane + 64u * i; f[i] = (k < n) ? sImp[k] : -1.0f; }
        const float4* sImp4 = reinterpret_cast<const float4*>(sImp);
        const uint32_t n4 = (n + 3u) / 4u;
#pragma unroll
        for (int jb = 0; jb < 4; jb++) {
            const uint32_t qEnd = min(n4, (uint32_t)(jb + 1) * 16u);
            for (uint32_t q4 = (uint32_t)jb * 16u; q4 < qEnd; q4++) {
                const float4 gv = sImp4[q4];
                const float g[4] = { gv.x, gv.y, gv.z, gv.w };
#pragma unroll
                for (int e = 0; e < 4; e++) {
#pragma unroll
                    for (int i = 0; i < 4; i++) {
                        if (jb < i) rank[i] += (g[e] < f[i]) ? 1u : 0u;
                        else if (jb > i) rank[i] += (g[e] <= f[i]) ? 1u : 0u;
                        else {
                            const uint32_t q = q4 * 4u + e, k = lane + 64u * i;
                            rank[i] += (g[e] < f[i] || (g[e] == f[i] && q > k)) ? 1u : 0u;
                        }
                    }
                }
            }
        }
#pragma unroll
        for (int i = 0; i < 4; i++) {
            const uint32_t k = lane + 64u * i;
            if (k < n && rank[i] < KEEP) out[rank[i]] = sIdx[k] & 0x7FFFFFFFu;
        }
    }
    if (lane == 0) tileNum[bandTile] = num;
}
template <bool BRUTE>
__global__ __launch_bounds__(256) void k1_tile_cull(const float4* __restrict__ lightView, const uint32_t* __restrict__ lightType, int N, int words,
                                                    const float4* __restrict__ tileInfo, int Tx, int bandRows,
                                                    const unsigned long long* __restrict__ masks,
                                                    const uint32_t* __restrict__ groupCount, const uint32_t* __restrict__ groupList, int groupsX,
                                                    uint32_t* __restrict__ tileNum, uint32_t* __restrict__ tileList)
{
    __shared__ __attribute__((aligned(16))) unsigned char lds[LDS_K1_TILE_CULL];
    k1_tile_cull_body<BRUTE>(blockIdx.x, lds, lightView, lightType, N, words, tileInfo, Tx, bandRows, masks, groupCount, groupList, groupsX, tileNum, tileList);
}

// ------------------------------------------------------------------------------------------------------------
// K1d: canonical offsets (Appendix A step 6) and compaction: per-1024-tile block sums, then every tile's wave
// rebuilds its own exclusive prefix (block sums before its block + tiles before it inside the block).
// ------------------------------------------------------------------------------------------------------------
// Tile classes for the shading ORDER hint (sailor_hip_light_cull_tile_order): A = lists of >= CLASS_A entries, B = >= CLASS_B, C = the rest.
// Their counts ride through the same scan, packed A | B << 16 (<= 1024 tiles per block: no carry).
#define CLASS_A 96u
#define CLASS_B 40u
__device__ __forceinline__ uint32_t tile_class_bits(uint32_t num) { return num >= CLASS_A ? 1u : (num >= CLASS_B ? 0x10000u : 0u); }

__global__ __launch_bounds__(1024) void k1_block_sums(const uint32_t* __restrict__ tileNum, int T, uint32_t* __restrict__ tilePrefix,
                                                       uint32_t* __restrict__ blockSums, uint32_t* __restrict__ classPrefix, uint32_t* __restrict__ classSums)
{
    __shared__ uint32_t sW[16], sC[16];
    const int t = blockIdx.x * SCAN_BLOCK + threadIdx.x;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const uint32_t v = (t < T) ? tileNum[t] : 0u;
    const bool classes = classPrefix != nullptr; // the order hint is only produced for split frames (see the host side)
    const uint32_t c = (classes && t < T) ? tile_class_bits(v) : 0u;
    uint32_t incl = v, cincl = c;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const uint32_t u = (uint32_t)__shfl_up((int)incl, d);
        if (lane >= d) incl += u;
        if (classes) { const uint32_t uc = (uint32_t)__shfl_up((int)cincl, d); if (lane >= d) cincl += uc; }
    }
    if (lane == 63) { sW[wave] = incl; sC[wave] = cincl; }
    __syncthreads();
    uint32_t before = 0, total = 0, cbefore = 0, ctotal = 0;
#pragma unroll
    for (int w = 0; w < 16; w++) {
        const uint32_t x = sW[w], y = sC[w];
        before += (w < wave) ? x : 0u; total += x;
        cbefore += (w < wave) ? y : 0u; ctotal += y;
    }
    if (t < T) tilePrefix[t] = before + incl - v; // exclusive prefix inside the 1024-tile block
    if (threadIdx.x == 0) blockSums[blockIdx.x] = total;
    if (classes) {
        if (t < T) classPrefix[t] = cbefore + cincl - c;
        if (threadIdx.x == 0) classSums[blockIdx.x] = ctotal;
    }
}

__device__ __forceinline__ void k1_pack_body(const unsigned bx, const uint32_t* __restrict__ tileNum, const uint32_t* __restrict__ tilePrefix,
                                             const uint32_t* __restrict__ blockSums, int sumBlocks, const uint32_t* __restrict__ tileList, int T,
                                             SailorLightsGrid* __restrict__ grid, uint32_t* __restrict__ culled, uint32_t capacity,
                                             const uint32_t* __restrict__ classPrefix, const uint32_t* __restrict__ classSums, int Tx, uint32_t* __restrict__ tileOrder)
{
    const int tile = bx * 4 + (threadIdx.x >> 6);
    if (tile >= T) return;
    const int lane = threadIdx.x & 63;
    const int blk = tile / SCAN_BLOCK;
    // everything this wave reads is addressed by the tile index alone: issue it all up front (the staging list speculatively, both halves
    // of its KEEP slots), so that the only dependent step is the store at the tile's offset
    const uint32_t* src = tileList + (size_t)tile * KEEP;
    const uint32_t e0 = src[lane], e1 = src[lane + 64];
    const uint32_t num = tileNum[tile];
    const uint32_t prefixInBlock = tilePrefix[tile];
    const bool classes = classPrefix != nullptr;
    uint32_t s = 0, tot = 0, aB = 0, aT = 0, bB = 0, bT = 0; // sums over earlier blocks / all blocks: entries, class A tiles, class B tiles
    for (int i = lane; i < sumBlocks; i += 64) {
        const uint32_t b = blockSums[i];
        tot += b; s += (i < blk) ? b : 0u;
        if (classes) {
            const uint32_t c = classSums[i];
            aT += c & 0xFFFFu; bT += c >> 16;
            if (i < blk) { aB += c & 0xFFFFu; bB += c >> 16; }
        }
    }
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) { s += (uint32_t)__shfl_xor((int)s, d); tot += (uint32_t)__shfl_xor((int)tot, d); }
    if (classes) {
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) {
            aB += (uint32_t)__shfl_xor((int)aB, d); aT += (uint32_t)__shfl_xor((int)aT, d);
            bB += (uint32_t)__shfl_xor((int)bB, d); bT += (uint32_t)__shfl_xor((int)bT, d);
        }
    }
    const uint32_t offset = s + prefixInBlock + 1u;
    if (lane == 0) { grid[tile].offset = offset; grid[tile].num = num; }
    if (classes && lane == 0) {
        // the order hint: long lists first (their blocks run longest; started last they are the tail of the shade launch), each class in
        // tile order (neighbouring tiles share light records in L2)
        const uint32_t cp = classPrefix[tile];
        const uint32_t aBefore = aB + (cp & 0xFFFFu), bBefore = bB + (cp >> 16);
        const uint32_t cls = tile_class_bits(num);
        const uint32_t pos = cls == 1u ? aBefore : (cls ? aT + bBefore : aT + bT + ((uint32_t)tile - aBefore - bBefore));
        tileOrder[pos] = (uint32_t)(tile % Tx) | ((uint32_t)(tile / Tx) << 16);
        if (tile == 0) tileOrder[T] = aT + bT; // how many entries of the order hold >= CLASS_B lights (the shade's split blocks)
    }
    if ((uint32_t)lane < num && offset + lane < capacity) culled[offset + lane] = e0;
    if ((uint32_t)lane + 64u < num && offset + lane + 64u < capacity) culled[offset + lane + 64u] = e1;
    if (tile == 0 && lane == 0) culled[0] = tot;
}
__global__ __launch_bounds__(256) void k1_pack(const uint32_t* __restrict__ tileNum, const uint32_t* __restrict__ tilePrefix,
                                                const uint32_t* __restrict__ blockSums, int sumBlocks,
                                                const uint32_t* __restrict__ tileList, int T, SailorLightsGrid* __restrict__ grid,
                                                uint32_t* __restrict__ culled, uint32_t capacity,
                                                const uint32_t* __restrict__ classPrefix, const uint32_t* __restrict__ classSums, int Tx, uint32_t* __restrict__ tileOrder)
{
    k1_pack_body(blockIdx.x, tileNum, tilePrefix, blockSums, sumBlocks, tileList, T, grid, culled, capacity, classPrefix, classSums, Tx, tileOrder);
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
    if (N > 0x3FFFFFFF) return SAILOR_HIP_ERR_UNSUPPORTED; // bit 31 of a candidate entry carries the "directional" flag
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
    float4* bandPlanes = (float4*)(ws + L.offBandPlanes);
    unsigned long long* masks = (unsigned long long*)(ws + L.offMasks);
    unsigned long long* dirWords = (unsigned long long*)(ws + L.offDirWords);
    uint32_t* tileNum = (uint32_t*)(ws + L.offTileNum);
    uint32_t* tilePrefix = (uint32_t*)(ws + L.offTilePrefix);
    uint32_t* tileList = (uint32_t*)(ws + L.offTileList);
    uint32_t* blockSums = (uint32_t*)(ws + L.offBlockSums);
    uint32_t* groupCount = (uint32_t*)(ws + L.offGroupCount);
    uint32_t* groupList = (uint32_t*)(ws + L.offGroupList);

    if (L.bandTiles == 0) {
        SAILOR_TRY_HIP(ctx, hipMemsetAsync(dCulledLights, 0, 4, s));
        return SAILOR_HIP_OK;
    }

    Mat4 view, invProj;
    memcpy(view.m, frame->view, 64);
    memcpy(invProj.m, frame->invProjection, 64);

    // The side-plane margin argument needs a sane perspective; tiny light counts are cheaper brute force.
    const float p00 = fabsf(frame->projection[0]), p11 = fabsf(frame->projection[5]);
    const bool sane = p00 > 1e-2f && p11 > 1e-2f && p00 < 1e4f && p11 < 1e4f;
    const bool brute = (flags & SAILOR_CULL_BRUTE_FORCE) || !sane || N < 512;

    const int lightBlocks = (N + 255) / 256;
    const int bandBlocks = brute ? 0 : (L.numBands + 255) / 256;
    const int stripsPerRow = (L.Tx + 15) / 16;
    PrepareArgs pa;
    pa.view = view; pa.invProj = invProj;
    pa.lights = dLights; pa.depth = dLinearDepth;
    pa.lightView = lightView; pa.lightType = lightType; pa.tileInfo = tileInfo; pa.bandPlanes = bandPlanes;
    pa.N = N; pa.lightBlocks = lightBlocks; pa.setupBlocks = stripsPerRow * L.bandRows;
    pa.vpW = frame->viewportSize[0]; pa.vpH = frame->viewportSize[1]; pa.W = W; pa.H = H; pa.Tx = L.Tx; pa.Ty = L.Ty;
    pa.tileRow0 = band->tileRowBegin; pa.bandRow0 = band->fbRowBegin; pa.bandRows = L.bandRows; pa.groupsX = L.groupsX; pa.numBands = L.numBands;
    pa.stripsPerRow = stripsPerRow;
    pa.vecOK = (((uintptr_t)dLinearDepth & 15) == 0 && (W & 3) == 0) ? 1 : 0;
    pa.rawDepth = (flags & SAILOR_CULL_RAW_DEPTH) ? 1 : 0;
    pa.zNearCam = frame->cameraZNearZFar[0];
    hipLaunchKernelGGL(k01_prepare, dim3(pa.setupBlocks + lightBlocks + bandBlocks), dim3(256), 0, s, pa);
    SAILOR_CHECK_LAUNCH(ctx, "k01_prepare");

    if (brute) {
        hipLaunchKernelGGL(k1_tile_cull<true>, dim3(L.numGroups * 4), dim3(256), 0, s, lightView, lightType, N, L.words, tileInfo, L.Tx, L.bandRows, masks,
                           groupCount, groupList, L.groupsX, tileNum, tileList);
        SAILOR_CHECK_LAUNCH(ctx, "k1_tile_cull<brute>");
    } else {
        const float planeMargin = 1e-3f;
        const int groups = (L.numBands + BANDS_PER_GROUP - 1) / BANDS_PER_GROUP;
        hipLaunchKernelGGL(k1_band_masks, dim3((L.words + 3) / 4, groups), dim3(256), 0, s, lightView, lightType, N, L.words, bandPlanes, L.numBands,
                           planeMargin, masks, dirWords);
        SAILOR_CHECK_LAUNCH(ctx, "k1_band_masks");
        hipLaunchKernelGGL(k1_group_lists, dim3(L.numGroups), dim3(256), 0, s, masks, dirWords, L.words, L.Tx, L.bandRows, L.groupsX, groupCount, groupList);
        SAILOR_CHECK_LAUNCH(ctx, "k1_group_lists");
        hipLaunchKernelGGL(k1_tile_cull<false>, dim3(L.numGroups * 4), dim3(256), 0, s, lightView, lightType, N, L.words, tileInfo, L.Tx, L.bandRows, masks,
                           groupCount, groupList, L.groupsX, tileNum, tileList);
        SAILOR_CHECK_LAUNCH(ctx, "k1_tile_cull");
    }
    // The order hint pays for itself on split frames (a band's shade launch is bounded by its longest tile; starting those first took
    // 14 us off the slowest band of an 8-way split); on the whole frame it only costs (the scan grows by 5 us, the shade does not
    // get faster: 16 rounds of blocks hide the tail), so it is not produced there.
    const bool splitFrame = L.bandRows < L.Ty;
    uint32_t* classPrefix = splitFrame ? (uint32_t*)(ws + L.offClassPrefix) : nullptr;
    uint32_t* classSums = (uint32_t*)(ws + L.offClassSums);
    uint32_t* tileOrder = (uint32_t*)(ws + L.offTileOrder);
    hipLaunchKernelGGL(k1_block_sums, dim3(L.sumBlocks), dim3(SCAN_BLOCK), 0, s, tileNum, L.bandTiles, tilePrefix, blockSums, classPrefix, classSums);
    SAILOR_CHECK_LAUNCH(ctx, "k1_block_sums");
    hipLaunchKernelGGL(k1_pack, dim3((L.bandTiles + 3) / 4), dim3(256), 0, s, tileNum, tilePrefix, blockSums, L.sumBlocks, tileList, L.bandTiles, dLightsGrid,
                       dCulledLights, (uint32_t)(culledCapacity > 0xFFFFFFFFull ? 0xFFFFFFFFull : culledCapacity), classPrefix, classSums, L.Tx, tileOrder);
    SAILOR_CHECK_LAUNCH(ctx, "k1_pack");
    return SAILOR_HIP_OK;
}

// ------------------------------------------------------------------------------------------------------------
// LinearizeDepth: Content/Shaders/LinearizeDepth.shader:61-73 under REVERSE_Z_INF_FAR_PLANE, as
// drawn by FrameGraph/LinearizeDepthNode.cpp:22-109.  Pure stream: 4 bytes in, 4 bytes out per texel, float4 per lane.
// (K1 does not need this pass at all -- see SAILOR_CULL_RAW_DEPTH -- it exists for the node's other consumers.)
// ------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_linearize_depth(const float* __restrict__ raw, float* __restrict__ out, size_t count, float zNear, int vec)
{
    const size_t stride = (size_t)gridDim.x * 256;
    if (vec) {
        const size_t n4 = count / 4;
        const float4* __restrict__ r4 = reinterpret_cast<const float4*>(raw);
        float4* __restrict__ o4 = reinterpret_cast<float4*>(out);
        for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += stride) {
            const float4 d = r4[i];
            o4[i] = make_float4(-(-zNear / d.x), -(-zNear / d.y), -(-zNear / d.z), -(-zNear / d.w));
        }
        for (size_t i = n4 * 4 + (size_t)blockIdx.x * 256 + threadIdx.x; i < count; i += stride) out[i] = -(-zNear / raw[i]);
    } else {
        for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < count; i += stride) out[i] = -(-zNear / raw[i]);
    }
}

int sailor_hip_linearize_depth(SailorHipContext* ctx, const SailorUboFrameData* frame, const float* dRawDepth, float* dLinearDepth, int32_t width, int32_t rows)
{
    if (!ctx || !frame || !dRawDepth || !dLinearDepth || width <= 0 || rows < 0) return SAILOR_HIP_ERR_INVALID_ARGUMENT;
    const size_t count = (size_t)width * (size_t)rows;
    if (count == 0) return SAILOR_HIP_OK;
    const int vec = ((((uintptr_t)dRawDepth | (uintptr_t)dLinearDepth) & 15) == 0) ? 1 : 0;
    size_t blocks = (count / 4 + 255) / 256;
    if (blocks > 256 * 32) blocks = 256 * 32; // 32 blocks per CU, grid-stride beyond
    if (blocks < 1) blocks = 1;
    hipLaunchKernelGGL(k_linearize_depth, dim3((unsigned)blocks), dim3(256), 0, ctx->stream, dRawDepth, dLinearDepth, count, frame->cameraZNearZFar[0], vec);
    SAILOR_CHECK_LAUNCH(ctx, "k_linearize_depth");
    return SAILOR_HIP_OK;
}

const uint32_t* sailor_hip_light_cull_tile_order(int32_t width, int32_t height, int32_t lightsCapacity, const SailorBand* band, const void* dWorkspace)
{
    if (!dWorkspace || width <= 0 || height <= 0 || lightsCapacity < 0) return nullptr;
    SailorBand whole;
    if (!band) { sailor_hip_band_whole_frame(width, height, &whole); band = &whole; }
    if (!band_valid(width, height, band)) return nullptr;
    if (width > 16 * 65535 || height > 16 * 65535) return nullptr; // packed as two 16-bit tile coordinates
    if (band->tileRowEnd - band->tileRowBegin >= (height - 1) / TILE + 1) return nullptr; // whole frame: no hint is produced (raster order)
    return (const uint32_t*)((const char*)dWorkspace + make_layout(width, height, lightsCapacity, *band).offTileOrder);
}

// Diagnostics for benchmarks / tuning (synchronises): density of the band masks and of the group candidate lists left in
// `dWorkspace` by the last sailor_hip_light_cull with the same geometry.
// out[0] = numBands, out[1] = mask bits set (all bands), out[2] = numGroups, out[3] = sum of group list lengths,
// out[4] = overflowed groups, out[5] = longest group list, out[6] = words per band, out[7] = bits set in column masks only
int sailor_hip_light_cull_diagnostics(SailorHipContext* ctx, int32_t width, int32_t height, int32_t lightsNum, const SailorBand* band,
                                      const void* dWorkspace, uint64_t* out8)
{
    if (!ctx || !dWorkspace || !out8 || width <= 0 || height <= 0 || lightsNum < 0) return SAILOR_HIP_ERR_INVALID_ARGUMENT;
    SailorBand whole;
    if (!band) { sailor_hip_band_whole_frame(width, height, &whole); band = &whole; }
    if (!band_valid(width, height, band)) return SAILOR_HIP_ERR_INVALID_ARGUMENT;
    const CullLayout L = make_layout(width, height, lightsNum, *band);
    std::vector<unsigned long long> masks((size_t)L.numBands * L.words);
    std::vector<uint32_t> counts((size_t)L.numGroups);
    SAILOR_TRY_HIP(ctx, hipStreamSynchronize(ctx->stream));
    SAILOR_TRY_HIP(ctx, hipMemcpy(masks.data(), (const char*)dWorkspace + L.offMasks, masks.size() * 8, hipMemcpyDeviceToHost));
    SAILOR_TRY_HIP(ctx, hipMemcpy(counts.data(), (const char*)dWorkspace + L.offGroupCount, counts.size() * 4, hipMemcpyDeviceToHost));
    uint64_t bits = 0, colBits = 0, sum = 0, over = 0, longest = 0;
    for (size_t i = 0; i < masks.size(); i++) {
        const uint64_t c = (uint64_t)__builtin_popcountll(masks[i]);
        bits += c;
        if (i < (size_t)L.groupsX * L.words) colBits += c;
    }
    for (uint32_t c : counts) {
        if (c == GROUP_OVERFLOW) { over++; continue; }
        sum += c;
        if (c > longest) longest = c;
    }
    out8[0] = (uint64_t)L.numBands; out8[1] = bits; out8[2] = (uint64_t)L.numGroups; out8[3] = sum;
    out8[4] = over; out8[5] = longest; out8[6] = (uint64_t)L.words; out8[7] = colBits;
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
