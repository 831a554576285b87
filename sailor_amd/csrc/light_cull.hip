// K0 + K1 -- screen-tile light cull for gfx950.
//
// Replaces Content/Shaders/ComputeLightCulling.shader (+ Math.glsl:116-173,224-239) as dispatched by
// LightCullingNode::Process (FrameGraph/LightCullingNode.cpp:74-77), under the canonical sequential semantics of
// SURVEY.md Appendix A.  Not a translation of the GLSL: the reference runs one 16x16 workgroup per tile that
// re-reads and re-transforms every light, appends with LDS atomics, bubble-sorts on one thread and allocates
// output space with a global atomic.  Here, four launches:
//
//   k01_prepare        three independent roles in one launch.
//                      lights : once per light, view-space position + radius into a float4 SoA (same fp32 op sequence as
//                               ComputeLightCulling.shader:164-169, so bits are identical), then -- same wave, the record
//                               still in registers -- the conservative pre-filter as BITMASKS: one 64-bit ballot per
//                               (64 lights, column of 4x4-tile groups) and per (64 lights, row of groups): "this light's
//                               sphere may reach this 64-pixel band".  No atomics, no compaction, no inter-block order.
//                      frusta : one lane per tile: the four side planes + centre (they do not depend on the depth)
//                      depth  : streaming pass over the linear-depth image (the only large HBM stream of the cull):
//                               32 tiles per 256-thread block, 1 KiB coalesced float4 row segments, wave shuffles + a
//                               4-entry LDS combine for min/max
//                      Above 262 144 lights a block holds all bands and finds each light's band INTERVALS by bisection on the
//                      same plane tests instead (k0_lights; SAILOR_CULL_INTERVAL_MASKS forces that form)
//   k1_group_lists     one block per 4x4-tile group: (its column's mask) AND (its row's mask), 16 384 lights per
//                      step; the few surviving bits become an ordered, contiguous candidate list (ballot, readlane, mbcnt)
//   k1_group_lists_wide  the same lists for >= 4 096 mask words (a million lights): one wave per group queues the non-empty
//                      words and drains them one lane per word; four groups per block share the row band's mask through LDS
//   k1_tile_cull       one 256-thread block per run of four tiles of a tile row (= one row of a group), one wave per tile:
//                      the group's candidate records are staged in LDS 512 at a time, each wave streams them through the
//                      exact test, 64 per step; ballot/popcount ordered append; rank-based nearest-128 selection
//                      (groups denser than 2048 candidates fall back to walking the two masks of the tile itself).
//                      The tiles of a LIGHT CLUSTER (a group with more than 512 candidates -- 384 on a band of a split frame --, listed by k1_group_lists) get a whole
//                      block each at the front of the grid: four waves share the tile's candidates and its 196 -> 128 selection
//                      (round 4; one wave took ~20 us for such a tile -- the launch's tail and a cluster band's whole cull).
//                      Every tile leaves its list in its own fixed 128-entry slot of the workspace (`tile lists`) and its length in
//                      tileNum
//   k1_pack            canonical offsets (Appendix A step 6: prefix sum of the list lengths in tile order) and compaction
//                      in ONE streaming launch without a scan chain: a pack block owns 64 consecutive tiles, adds up the lengths
//                      of all tiles in front of them itself (tileNum: L2-resident), then moves its tiles' lists through LDS into
//                      culledLights as one contiguous run.  Since round 4 NOTHING on
//                      the path needs its output: the shade reads a tile's list from its slot (sailor_hip_shade_tile_lists), so the
//                      launch can run beside the shade on another stream (SAILOR_CULL_DEFER_PACK + sailor_hip_light_cull_pack) and
//                      still produces the reference's lightsGrid / culledLights bit for bit.  (Tried in round 2: a decoupled
//                      look-back inside k1_tile_cull -- bit-exact, 95 us instead of 33.)
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
#include "light_record.h"
#include <vector>

#define BANDS_PER_BLOCK 24   // group columns / rows whose masks one light block of k01_prepare builds when the lights alone do not fill the chip
#define MAX_BANDS_PER_BLOCK 256 // ... and at most (LDS for their planes); with >= 1024 light blocks a block builds all of its lights' masks
#define QCAP 128             // LDS candidate queue of the overflow path of k1_tile_cull (ring buffer: < 64 pending + <= 64 new)
#define GROUP 4              // k1_group_lists: tiles per group edge (4x4 tiles share one candidate list)
#define CAPG 2048            // entries per group list; a denser group falls back to walking the masks per tile
#define GROUP_OVERFLOW 0xFFFFFFFFu
#define GROUP_OVERFLOW_LISTED 0xFFFFFFFEu // ... the same for a group that is in k1_group_lists' cluster list
#define GROUP_LISTED 0x40000000u // flag in a group's count: a light cluster k1_group_lists found room for in its list (k1_tile_cull starts with those)
// A group with more candidates than this is listed as a light cluster: its tiles get a block each.  Measured 512 / 384 / 256: whole frame 25.5 / 26.1 /
// 30.6 us for k1_tile_cull, a cluster band of an 8-way split 15.4 / 9.7 / 8.0 -- the whole frame hides a long block behind its other 8 000, a band
// IS its longest block; below 384 too many groups qualify.  So: by the size of the band.
#define HEAVY_MIN_FRAME 512
#define HEAVY_MIN_BAND 384
#define HEAVY_MAX 96            // ... and that list's room (16 head blocks of k1_tile_cull per entry: the empty ones are dispatched in front of everything else -- 4 096 of them cost ~3 us)
#define CHUNK 512            // group candidates staged in LDS per step of k1_tile_cull (16 KB of LDS per block: 8 blocks per CU)

#define PACK_TILES 64        // tiles per k1_pack block
#define SEL_LIGHTS 1024      // lights per block of k0_band_count / k0_band_scatter
#define CL_STEPS (CAPG / 4 / 64) // cluster tiles: 64-candidate steps per wave for the longest listable group (8)

#define BAND_FORM_MAX_TILES 12000 // (see layout_has_hint)

struct CullLayout {
    int Tx, Ty, bandRows, bandTiles, numBands, words, groupsX, groupsY, numGroups, packBlocks;
    size_t offLightView, offLightType, offTileInfo, offMasks, offDirWords, offGroupCount, offGroupList, offTileNum, offTileNum8, offTileLists, offDirFlag, offHeavy,
           offLightMap, offSelState, offSelKeep, total;
    int selBlocks;
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
    L.groupsX = (L.Tx + GROUP - 1) / GROUP;
    L.groupsY = (L.bandRows + GROUP - 1) / GROUP;
    L.numGroups = L.groupsX * L.groupsY;
    L.numBands = L.groupsX + L.groupsY;      // group columns first, then the band's group rows (4 tiles wide / high)
    L.packBlocks = (L.bandTiles + PACK_TILES - 1) / PACK_TILES;
    const size_t groups = (size_t)(L.numGroups > 0 ? L.numGroups : 1);
    // Sections whose size depends on the geometry only come first, those that scale with the light count last: the offset of anything a later
    // call looks up from (width, height, band) alone -- the per-tile lists, their lengths -- is then the same for every lightsNum <= the capacity the
    // workspace was sized for (a cull may run with fewer lights than the capacity; round 2 computed the hint's address from the capacity and
    // the cull's own layout from lightsNum, which only agree when the two are equal).
    size_t o = 0;
    L.offTileInfo = o; o = align_up(o + tiles * 64, 256);
    L.offGroupCount = o; o = align_up(o + groups * 4, 256);
    L.offDirFlag = o; o = align_up(o + 4, 256);
    L.offHeavy = o; o = align_up(o + (1 + HEAVY_MAX) * 4, 256); // [0] = light-cluster groups listed by k1_group_lists (zeroed by k01_prepare), then their indices
    L.offGroupList = o; o = align_up(o + groups * CAPG * 4, 256);
    L.offTileNum = o; o = align_up(o + tiles * 4, 256);
    L.offTileNum8 = o; o = align_up(o + tiles + 64, 256);      // the same lengths (<= 128) as bytes: what k1_pack adds up for its base (32 KB at 4K instead of 130)
    L.offTileLists = o; o = align_up(o + tiles * KEEP * 4, 256); // one fixed 128-entry slot per tile (only the list's own bytes are ever touched)
    L.offLightView = o; o = align_up(o + n * 16, 256);
    L.offLightType = o; o = align_up(o + n * 4, 256);
    L.offMasks = o; o = align_up(o + (size_t)(L.numBands > 0 ? L.numBands : 1) * L.words * 8, 256);
    L.offDirWords = o; o = align_up(o + (size_t)L.words * 8, 256);
    // the band's own light set (k0_band_count / k0_band_scatter: split frames with large light sets): compact index -> light index; [0] the number of
    // selected lights, [2 + b] block b's count; block b's sixteen ballot words
    L.selBlocks = (int)((n + SEL_LIGHTS - 1) / SEL_LIGHTS);
    L.offLightMap = o; o = align_up(o + n * 4, 256);
    L.offSelState = o; o = align_up(o + (size_t)(2 + L.selBlocks) * 4, 256);
    L.offSelKeep = o; o = align_up(o + (size_t)L.selBlocks * 16 * 8, 256);
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

// One launch for the three independent preparation passes: blocks [0, lightRoleBlocks) transform the lights and build the band
// masks (K0 + K1b), the next few build the tile frusta (latency / ALU work, dispatched first so that it runs beside the stream), the
// rest stream the depth image (K1a).
struct PrepareArgs {
    Mat4 view, invProj;
    const SailorLightShaderData* lights;
    const float4* soaPosRadius; const uint32_t* soaType; // sailor_hip_prepare_lights' 20-byte view of the lights, or null: read the 112-byte records
    float4* prepPosRadius; uint32_t* prepType; float4* prepStaged; // SAILOR_CULL_PREPARE_LIGHTS: the prepared views, WRITTEN here from the records (else null)
    const float* depth;
    float4* lightView; uint32_t* lightType; float4* tileInfo;
    unsigned long long* masks; unsigned long long* dirWords;
    uint32_t* heavy;   // [0]: k1_group_lists' count of light-cluster groups, zeroed here (one launch ahead of it)
    const uint32_t* selCount; // the band selection ran (null: it did not): the light role works on lightView / lightType [0, *selCount), already in view space
    uint32_t* dirFlag; // "some light may be directional": set here, read by k1_group_lists_wide, cleared by k1_tile_cull (unknown before the first cull: then merely conservative)
    int N, words, lightBlocks, lightRoleBlocks, frustumBlocks, bandsPerBlock, setupBlocks, vpW, vpH, W, H, Tx, Ty, tileRow0, bandRow0, bandRows, groupsX, numBands,
        stripsPerRow, vecOK, rawDepth, intervals;
    float zNearCam, planeMargin;
};

// 64 x 64 bit-matrix transpose across a wave: lane l holds row l, returns column l (bit i = bit l of lane i's row).  Recursive block swap:
// at stage k the lanes l and l ^ k exchange the off-diagonal k x k blocks.
__device__ __forceinline__ unsigned long long transpose_64x64(unsigned long long r)
{
    const uint32_t lane = threadIdx.x & 63;
    const unsigned long long keep[6] = { 0x00000000FFFFFFFFull, 0x0000FFFF0000FFFFull, 0x00FF00FF00FF00FFull,
                                         0x0F0F0F0F0F0F0F0Full, 0x3333333333333333ull, 0x5555555555555555ull }; // columns whose bit k is clear
#pragma unroll
    for (int s = 0; s < 6; s++) {
        const int k = 32 >> s;
        const uint32_t plo = (uint32_t)__shfl_xor((int)(uint32_t)r, k, 64), phi = (uint32_t)__shfl_xor((int)(uint32_t)(r >> 32), k, 64);
        const unsigned long long p = ((unsigned long long)phi << 32) | plo;
        const unsigned long long m = keep[s];
        r = (lane & (uint32_t)k) ? ((r & ~m) | ((p & ~m) >> k)) : ((r & m) | ((p & m) << k));
    }
    return r;
}

// ------------------------------------------------------------------------------------------------------------
// K0: ComputeLightCulling.shader:164-169 hoisted out of the per-tile loop (it does not depend on the tile), and
// K1b: the band masks.  masks[b][word] bit k = "light 64*word + k may reach group column / group row b".
// A light is dropped from a band only if its sphere is entirely in front of the eye AND entirely outside one of the
// band's two planes by more than the margin; directional lights and everything doubtful stay in.
// The block first builds the planes of its <= BANDS_PER_BLOCK bands (one thread per band; planes through the eye and the
// screen lines x = 64 b, x = 64 (b + 1), resp. y): ~1 us, hidden behind the depth stream of the other role.
// ------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ void k0_lights(const int lb, unsigned char* __restrict__ lds, const PrepareArgs& a)
{
    float4* sPl = reinterpret_cast<float4*>(lds); // [2 * bandsPerBlock]
    const int wordBlock = lb % a.lightBlocks, split = lb / a.lightBlocks;
    const int b0 = split * a.bandsPerBlock;
    const int nb = min(a.bandsPerBlock, a.numBands - b0); // <= 0: no pre-filter (brute-force walk)
    if (a.selCount && wordBlock * 4 >= min((int)(((*a.selCount + 63u) / 64u + 1u) & ~1u), a.words)) return; // (behind the band selection: a block past the selected lights -- the grid is sized for all of them)
    if ((int)threadIdx.x < nb) {
        const int b = b0 + (int)threadIdx.x;
        Frustum4 f;
        if (b < a.groupsX) { // group column b (tile columns 4b .. 4b+3)
            frustum_from_rect(a.invProj, (float)(b * GROUP * TILE), 0.0f, (float)(min((b + 1) * GROUP, a.Tx) * TILE), (float)(a.Ty * TILE), a.vpW, a.vpH, f);
            sPl[2 * threadIdx.x + 0] = make_float4(f.n[0][0], f.n[0][1], f.n[0][2], 0.0f);
            sPl[2 * threadIdx.x + 1] = make_float4(f.n[1][0], f.n[1][1], f.n[1][2], 0.0f);
        } else {             // group row: the band's tile rows 4 gy .. 4 gy + 3
            const int gy = b - a.groupsX;
            const int ty = a.tileRow0 + gy * GROUP, tyEnd = a.tileRow0 + min((gy + 1) * GROUP, a.bandRows);
            frustum_from_rect(a.invProj, 0.0f, (float)(ty * TILE), (float)(a.Tx * TILE), (float)(tyEnd * TILE), a.vpW, a.vpH, f);
            sPl[2 * threadIdx.x + 0] = make_float4(f.n[2][0], f.n[2][1], f.n[2][2], 0.0f);
            sPl[2 * threadIdx.x + 1] = make_float4(f.n[3][0], f.n[3][1], f.n[3][2], 0.0f);
        }
    }
    const int word = wordBlock * 4 + (int)(threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    const int j = word * 64 + lane;
    // (behind the band selection the light set is the selected one: *selCount lights, compact, ascending light index, view-space records in lightView; the mask
    // words behind the last one are not written -- nobody reads them -- but for the odd word that completes a 16-byte pair, which gets zeros)
    const int nEff = a.selCount ? (int)*a.selCount : a.N;
    const int wordsEff = a.selCount ? min(((nEff + 63) / 64 + 1) & ~1, a.words) : a.words; // (never past the row: with an odd stride and every light selected there is no pair to complete)
    const bool valid = j < nEff;
    float4 lv = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
    uint32_t type = 1u;
    if (valid && a.selCount) { lv = a.lightView[j]; type = a.lightType[j]; }
    else if (valid) {
        float x, y, z, radius;
        if (a.prepStaged) {   // every light is dirty (SAILOR_CULL_PREPARE_LIGHTS): ONE pass over the 112-byte records derives the prepared views too --
            // what sailor_hip_prepare_lights(0, N) in front of this cull does in a launch of its own (k_prepare_lights: the same instructions,
            // the same bits), without reading the records a second time and the 20-byte view back
            const float4* L = reinterpret_cast<const float4*>(a.lights + j);
            const float4 q0 = L[0], q1 = L[1], q2 = L[2], q3 = L[3], q4 = L[4], q5 = L[5], q6 = L[6];
            x = q1.x; y = q1.y; z = q1.z; radius = q6.x;
            type = __float_as_uint(q0.x);
            if (split == 0) {
                a.prepPosRadius[j] = make_float4(x, y, z, radius);
                a.prepType[j] = type;
                float4 o0, o1, o2, o3, o4;
                stage_light_record(q0, q1, q2, q3, q4, q5, q6, o0, o1, o2, o3, o4);
                float4* o = a.prepStaged + (size_t)j * LREC;
                o[0] = o0; o[1] = o1; o[2] = o2; o[3] = o3; o[4] = o4;
            }
        } else if (a.soaPosRadius) { // two dense arrays, 20 bytes per light
            const float4 pr = a.soaPosRadius[j];
            x = pr.x; y = pr.y; z = pr.z; radius = pr.w;
            type = a.soaType[j];
        } else {              // the `light` SSBO itself: 20 of every 112 bytes
            const SailorLightShaderData* L = a.lights + j;
            x = L->worldPosition[0]; y = L->worldPosition[1]; z = L->worldPosition[2]; radius = L->bounds[0];
            type = L->type;
        }
        float4 p = glsl_mul(a.view, x, y, z, 1.0f);
        const float w = p.w;
        p.x = p.x / w; p.y = p.y / w; p.z = p.z / w;
        p.z = p.z * -1.0f; // "Reverse Z"
        lv = make_float4(p.x, p.y, p.z, radius);
        if (split == 0) { a.lightView[j] = lv; a.lightType[j] = type; }
    }
    __syncthreads(); // the planes are in LDS
    if (nb <= 0 || word >= wordsEff) return;
    const float r = lv.w;
    const float m = a.planeMargin * ((fabsf(lv.x) + fabsf(lv.y)) + (fabsf(lv.z) + fabsf(r)));
    const bool inFront = (lv.z - r) > m; // false for NaN => never plane-culled
    const float thr = -(r + m);
    const bool keepAlways = valid && (type == 0u || !inFront);
    if (split == 0) { // one bit per light: "directional" (rides along into the group lists, saves a gather per candidate)
        const unsigned long long dm = __ballot(valid && type == 0u);
        if (lane == 0) { a.dirWords[word] = dm; if (dm != 0ull) atomicOr(a.dirFlag, 1u); }
    }
    if (a.intervals) {
        // Large light sets (the block holds ALL bands): the 2 x numBands plane tests per light were the launch -- 188 bands x ~25 instructions per
        // wave at C5, 115 of prepare's 134 us.  The bands a sphere in front of the eye can reach are an interval of group columns times an
        // interval of group rows, so each lane bisects for its light's four interval ends on the SAME plane tests (per-lane LDS reads of the
        // planes), and the masks are ballots of "band inside the lane's interval".  What makes dropping a band safe is unchanged: a band is only
        // dropped on the strength of an evaluated test that put the sphere entirely beyond a plane between it and the band (beyond the far
        // plane of band L => beyond every band up to L; the half-spaces of a family of planes through the eye and parallel screen lines nest
        // for a sphere in front of the eye), so the bisection needs no monotonicity of the rounded test to stay conservative.
        int lo[2] = { 0, 0 }, hi[2] = { 0, 0 };
#pragma unroll
        for (int axis = 0; axis < 2; axis++) {
            const int first = axis == 0 ? 0 : a.groupsX, count = axis == 0 ? a.groupsX : nb - a.groupsX;
            // "far" plane (right / bottom, sPl[2 b + 1]): the bands the sphere lies entirely beyond form a prefix 0 .. L
            int L = -1, R = count;
            while (R - L > 1) {
                const int mid = (L + R) >> 1;
                const float4 nB = sPl[2 * (first + mid) + 1];
                if (dot3f(nB.x, nB.y, nB.z, lv.x, lv.y, lv.z) < thr) L = mid; else R = mid;
            }
            lo[axis] = L + 1;
            // "near" plane (left / top, sPl[2 b]): the bands the sphere lies entirely in front of form a suffix R .. count - 1
            L = lo[axis] - 1; R = count;
            while (R - L > 1) {
                const int mid = (L + R) >> 1;
                const float4 nA = sPl[2 * (first + mid) + 0];
                if (dot3f(nA.x, nA.y, nA.z, lv.x, lv.y, lv.z) < thr) R = mid; else L = mid;
            }
            hi[axis] = R - 1; // (lo > hi: no band of this axis)
        }
        // The mask words of 64 bands at a time.  A lane's interval IS its row of the (64 lights x 64 bands) bit matrix -- a run of ones, two
        // shifts -- and the mask words are the matrix's columns: one 64 x 64 bit transpose across the wave (six butterfly stages) replaces 64
        // ballots of "band inside the lane's interval" (and their 64 single-lane stores: lane j ends up holding band j's word and all
        // sixty-four leave in ONE store instruction -- a single-lane store costs the CU's memory pipeline 16 cycles like any other, and
        // 188 per wave, 3 M per launch at C5, were most of the role's time once the plane tests were gone).
#pragma unroll
        for (int axis = 0; axis < 2; axis++) {
            const int first = axis == 0 ? 0 : a.groupsX, count = axis == 0 ? a.groupsX : nb - a.groupsX;
            for (int w0 = 0; w0 < count; w0 += 64) {
                const int n = min(64, count - w0);                                     // bands in this window
                const unsigned long long window = n == 64 ? ~0ull : ((1ull << n) - 1ull);
                unsigned long long row = 0ull;
                if (keepAlways) row = window;
                else if (valid) {
                    const int x = max(lo[axis], w0) - w0, y = min(hi[axis], w0 + 63) - w0; // the interval inside the window: bits x .. y
                    if (x <= y) row = ((~0ull) >> (63 - (y - x))) << x;
                }
                const unsigned long long col = transpose_64x64(row) ;                 // lane j: bit i = light i of this wave reaches band w0 + j
                if (lane < n) a.masks[(size_t)(b0 + first + w0 + lane) * a.words + word] = col;
            }
        }
        return;
    }
    // (the masks of up to 64 bands are parked in the lanes of a register pair and leave as one store instruction, as above)
    uint32_t accLo = 0u, accHi = 0u;
    for (int bb = 0; bb < nb; bb += 64) {
        const int cnt = min(64, nb - bb);
#pragma unroll 4
        for (int k = 0; k < cnt; k++) {
            const float4 nA = sPl[2 * (bb + k) + 0], nB = sPl[2 * (bb + k) + 1]; // LDS broadcast reads
            const bool out = dot3f(nA.x, nA.y, nA.z, lv.x, lv.y, lv.z) < thr || dot3f(nB.x, nB.y, nB.z, lv.x, lv.y, lv.z) < thr;
            const unsigned long long mask = __ballot(keepAlways || (valid && !out));
            if (lane == k) { accLo = (uint32_t)mask; accHi = (uint32_t)(mask >> 32); }
        }
        if (lane < cnt) a.masks[(size_t)(b0 + bb + lane) * a.words + word] = ((unsigned long long)accHi << 32) | accLo;
    }
}

// ------------------------------------------------------------------------------------------------------------
// K1a: depth bounds (ComputeLightCulling.shader:119-128) + tile frustum, 32 tiles per block.
// tileInfo[t] = { (n0, cx), (n1, cy), (n2, zNear'), (n3, zFar') } with the near/far swap of :171-177 applied.
// ------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ uint32_t depth_bits(const float* __restrict__ depth, int W, int H, int bandRow0, int gx, int gy)
{
    int col = gx < W - 1 ? gx : W - 1;
    int row = H - 1 - gy;
    row = row < 0 ? 0 : (row > H - 1 ? H - 1 : row);
    return __float_as_uint(depth[(size_t)(row - bandRow0) * W + col]);
}

#define SETUP_STRIPS 2 // 16-tile strips per block: the whole role is one round of resident blocks at 4K, 8 float4 loads in flight per lane
__device__ __forceinline__ void k1_tile_setup(const int block, unsigned char* __restrict__ lds, const PrepareArgs& a)
{
    uint32_t (*sMin)[16 * SETUP_STRIPS] = reinterpret_cast<uint32_t (*)[16 * SETUP_STRIPS]>(lds);                       // [4][32]
    uint32_t (*sMax)[16 * SETUP_STRIPS] = reinterpret_cast<uint32_t (*)[16 * SETUP_STRIPS]>(lds + 256 * SETUP_STRIPS); // [4][32]
    const int strip0 = (block % a.stripsPerRow) * SETUP_STRIPS;
    const int tyLocal = block / a.stripsPerRow;
    const int ty = a.tileRow0 + tyLocal;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int W = a.W, H = a.H;
    float4 d[SETUP_STRIPS][4];
    bool vec[SETUP_STRIPS];
#pragma unroll
    for (int s = 0; s < SETUP_STRIPS; s++) {
        const int gx0 = (strip0 + s) * 256 + lane * 4; // 4 pixels per lane, 4 lanes per tile
        vec[s] = a.vecOK && (gx0 + 3 < W);
        if (vec[s]) {
#pragma unroll
            for (int k = 0; k < 4; k++) {
                int row = H - 1 - (ty * TILE + wave * 4 + k);
                row = row < 0 ? 0 : row;
                d[s][k] = *reinterpret_cast<const float4*>(a.depth + (size_t)(row - a.bandRow0) * W + gx0);
            }
        }
    }
    const int tx = strip0 * 16 + (int)threadIdx.x;
    const bool owner = threadIdx.x < 16 * SETUP_STRIPS && tx < a.Tx;
#pragma unroll
    for (int s = 0; s < SETUP_STRIPS; s++) {
        uint32_t mn = 0xFFFFFFFFu, mx = 0u;
        if (vec[s]) {
#pragma unroll
            for (int k = 0; k < 4; k++) {
                const uint32_t p = __float_as_uint(d[s][k].x), q = __float_as_uint(d[s][k].y), r = __float_as_uint(d[s][k].z), w = __float_as_uint(d[s][k].w);
                mn = min(min(mn, p), min(q, min(r, w)));
                mx = max(max(mx, p), max(q, max(r, w)));
            }
        } else if ((strip0 + s) * 16 < a.Tx) {
            const int gx0 = (strip0 + s) * 256 + lane * 4;
#pragma unroll
            for (int k = 0; k < 4; k++) {
                const int gy = ty * TILE + wave * 4 + k;
#pragma unroll
                for (int q = 0; q < 4; q++) {
                    const uint32_t v = depth_bits(a.depth, W, H, a.bandRow0, gx0 + q, gy);
                    mn = min(mn, v);
                    mx = max(mx, v);
                }
            }
        }
        // 4 lanes share a tile
        mn = min(mn, (uint32_t)__shfl_xor((int)mn, 1)); mx = max(mx, (uint32_t)__shfl_xor((int)mx, 1));
        mn = min(mn, (uint32_t)__shfl_xor((int)mn, 2)); mx = max(mx, (uint32_t)__shfl_xor((int)mx, 2));
        if ((lane & 3) == 0) { sMin[wave][s * 16 + (lane >> 2)] = mn; sMax[wave][s * 16 + (lane >> 2)] = mx; }
    }
    __syncthreads();
    if (owner) {
        const int i = threadIdx.x;
        const uint32_t bmn = min(min(sMin[0][i], sMin[1][i]), min(sMin[2][i], sMin[3][i]));
        const uint32_t bmx = max(max(sMax[0][i], sMax[1][i]), max(sMax[2][i], sMax[3][i]));
        float zFar = __uint_as_float(bmx), zNear = __uint_as_float(bmn);
        if (a.rawDepth) {
            // SAILOR_CULL_RAW_DEPTH: the image holds the reversed-Z attachment.  x -> fl(zNear / x) is monotone
            // non-increasing on x >= 0, so the largest linear depth of the tile is the linearised smallest raw value
            // and vice versa -- bit for bit what min / max over the linearised texels give (LinearizeDepth.shader:70,74).
            const float lo = -(-a.zNearCam / zFar), hi = -(-a.zNearCam / zNear);
            zNear = lo; zFar = hi;
        }
        const float diff = zFar - zNear; // "Add extra bounds" (:174-177): swaps near and far in fp32
        zFar -= diff;
        zNear += diff;
        float* o = reinterpret_cast<float*>(a.tileInfo + (size_t)(tyLocal * a.Tx + tx) * 4);
        o[11] = zNear; // the .w of the record's last two float4s; the frustum role writes the other fourteen floats
        o[15] = zFar;
    }
}

// K1a': the tile frusta (ComputeLightCulling.shader:57-95), one LANE per tile.  They do not depend on the depth image, so they are not
// computed by 16 lanes of a streaming block while its other 240 wait at a barrier: full waves in blocks of their own, beside the stream.
__device__ __forceinline__ void k1_tile_frusta(const int block, const PrepareArgs& a)
{
    const int i = block * 256 + (int)threadIdx.x;
    if (i >= a.bandRows * a.Tx) return;
    const int tyLocal = i / a.Tx, tx = i - tyLocal * a.Tx, ty = a.tileRow0 + tyLocal;
    Frustum4 f;
    frustum_from_rect(a.invProj, (float)(tx * TILE), (float)(ty * TILE), (float)((tx + 1) * TILE), (float)((ty + 1) * TILE), a.vpW, a.vpH, f);
    float4* o = a.tileInfo + (size_t)i * 4;
    o[0] = make_float4(f.n[0][0], f.n[0][1], f.n[0][2], f.cx);
    o[1] = make_float4(f.n[1][0], f.n[1][1], f.n[1][2], f.cy);
    float* o2 = reinterpret_cast<float*>(o + 2);
    o2[0] = f.n[2][0]; o2[1] = f.n[2][1]; o2[2] = f.n[2][2];
    o2[4] = f.n[3][0]; o2[5] = f.n[3][1]; o2[6] = f.n[3][2];
}

// ------------------------------------------------------------------------------------------------------------
// K0': the light set of a BAND (split frames, large light sets).  Every rank of a split frame used to transform all N lights and build masks over all N
// for a band that an eighth of them can reach: at C5 (1 M lights) a band's k01_prepare and k1_group_lists_wide were 74 of its 117 us.  These two kernels
// keep the lights whose sphere is not entirely beyond the band's top or bottom plane -- the SAME test, on the same planes, that keeps a light out of
// the band's first / last row mask, so no light a tile of the band could list is dropped -- and write them, transformed, in ascending light index:
// lightView / lightType [0, M) + lightMap (compact -> light index).  The rest of the chain then runs on M lights (k0_lights reads the compact records,
// the group lists hold compact indices, k1_tile_cull translates them when a list leaves).  With SAILOR_CULL_PREPARE_LIGHTS the prepared views of ALL
// lights are derived here (they outlive the band).
// Ordered compaction WITHOUT any wait between blocks (round 5): k0_band_count -- a block takes 1 024 lights, four per thread, tests them and leaves its
// sixteen ballot words and its count; a kernel boundary; k0_band_scatter -- the block of the same 1 024 lights adds up the counts of all blocks in
// front of it itself (<= 4 KB of L2 reads, as k1_pack adds up the list lengths) and writes its kept lights behind them.  Round 4 did both in one
// launch, a block polling the status words of the blocks in front of it: forward progress then rested on blocks being started in index order and on
// the whole grid being resident beside whatever else runs (it runs beside the previous frame's band shade) -- neither is promised (VERDICT / ADVICE
// r04).  A ticket per block (CUB's remedy: one relaxed atomic on one word, 1 024 of them) removes the assumption and was measured first: 12.4 -> 23.4 us
// on an eighth of C5 -- the word serialises at ~88 tickets per us and every block's loads wait for its ticket's round trip.  Two launches without any
// inter-block traffic cost a boundary (~1.7 us) and a second read of the kept lights' 20 bytes.
// ------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ uint64_t lanemask_lt();
struct SelectArgs {
    Mat4 view, invProj;
    const SailorLightShaderData* lights;
    const float4* soaPosRadius; const uint32_t* soaType;
    float4* prepPosRadius; uint32_t* prepType; float4* prepStaged;
    float4* lightView; uint32_t* lightType; uint32_t* lightMap;
    uint32_t* state;           // [0] = M (written by the last block of k0_band_scatter), [2 + b] = block b's count
    unsigned long long* keep;  // [16 b + 4 k + wave]: block b's ballot of round k (lights 1024 b + 256 k + 64 wave ..)
    int N, vpW, vpH, Tx, tileRow0, bandRows;
    int stageSelectedOnly; // SAILOR_CULL_PREPARE_SELECTED: the staged shade records only of the lights that are kept (the 20-byte cull views of all)
    float planeMargin;
};

// a light's 20-byte cull view from whichever form the call has (k0_lights' three sources), and its view-space record by k0_lights' operations: the
// records the tile tests read are the same bits
__device__ __forceinline__ void select_load(const SelectArgs& a, const int j, float& x, float& y, float& z, float& radius, uint32_t& type)
{
    if (a.soaPosRadius) {
        const float4 pr = a.soaPosRadius[j];
        x = pr.x; y = pr.y; z = pr.z; radius = pr.w;
        type = a.soaType[j];
    } else {
        const SailorLightShaderData* L = a.lights + j;
        x = L->worldPosition[0]; y = L->worldPosition[1]; z = L->worldPosition[2]; radius = L->bounds[0];
        type = L->type;
    }
}
__device__ __forceinline__ float4 select_view(const SelectArgs& a, const float x, const float y, const float z, const float radius)
{
    float4 p = glsl_mul(a.view, x, y, z, 1.0f);
    const float w = p.w;
    p.x = p.x / w; p.y = p.y / w; p.z = p.z / w;
    p.z = p.z * -1.0f; // "Reverse Z"
    return make_float4(p.x, p.y, p.z, radius);
}

__global__ __launch_bounds__(256) void k0_band_count(const SelectArgs a)
{
    __shared__ float4 sPl[2];
    __shared__ uint32_t sCnt[4][4];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const uint32_t b = blockIdx.x;
    if (threadIdx.x == 0) {
        Frustum4 f;
        frustum_from_rect(a.invProj, 0.0f, (float)(a.tileRow0 * TILE), (float)(a.Tx * TILE), (float)((a.tileRow0 + a.bandRows) * TILE), a.vpW, a.vpH, f);
        sPl[0] = make_float4(f.n[2][0], f.n[2][1], f.n[2][2], 0.0f); // top: the first row band's
        sPl[1] = make_float4(f.n[3][0], f.n[3][1], f.n[3][2], 0.0f); // bottom: the last row band's
    }
    float4 lv[4];
    uint32_t type[4];
#pragma unroll
    for (int k = 0; k < 4; k++) {
        const int j = (int)b * SEL_LIGHTS + k * 256 + (int)threadIdx.x;
        lv[k] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
        type[k] = 1u;
        if (j < a.N) {
            float x, y, z, radius;
            if (a.prepStaged) { // every light is dirty: the prepared views of all of them, as k0_lights derives them without this kernel in front of it
                const float4* L = reinterpret_cast<const float4*>(a.lights + j);
                const float4 q0 = L[0], q1 = L[1], q2 = L[2], q3 = L[3], q4 = L[4], q5 = L[5], q6 = L[6];
                x = q1.x; y = q1.y; z = q1.z; radius = q6.x;
                type[k] = __float_as_uint(q0.x);
                a.prepPosRadius[j] = make_float4(x, y, z, radius);
                a.prepType[j] = type[k];
                if (!a.stageSelectedOnly) {
                    float4 o0, o1, o2, o3, o4;
                    stage_light_record(q0, q1, q2, q3, q4, q5, q6, o0, o1, o2, o3, o4);
                    float4* o = a.prepStaged + (size_t)j * LREC;
                    o[0] = o0; o[1] = o1; o[2] = o2; o[3] = o3; o[4] = o4;
                }
            } else select_load(a, j, x, y, z, radius, type[k]);
            lv[k] = select_view(a, x, y, z, radius);
        }
    }
    __syncthreads(); // the planes are in LDS
#pragma unroll
    for (int k = 0; k < 4; k++) {
        const int j = (int)b * SEL_LIGHTS + k * 256 + (int)threadIdx.x;
        const bool valid = j < a.N;
        const float r = lv[k].w;
        const float m = a.planeMargin * ((fabsf(lv[k].x) + fabsf(lv[k].y)) + (fabsf(lv[k].z) + fabsf(r)));
        const bool inFront = (lv[k].z - r) > m; // false for NaN => never plane-culled
        const float thr = -(r + m);
        const bool keepAlways = valid && (type[k] == 0u || !inFront);
        const float4 nA = sPl[0], nB = sPl[1];
        const bool out = dot3f(nA.x, nA.y, nA.z, lv[k].x, lv[k].y, lv[k].z) < thr || dot3f(nB.x, nB.y, nB.z, lv[k].x, lv[k].y, lv[k].z) < thr;
        const unsigned long long keep = __ballot(keepAlways || (valid && !out));
        if (lane == 0) { a.keep[(size_t)b * 16u + (uint32_t)(4 * k + wave)] = keep; sCnt[k][wave] = (uint32_t)__popcll(keep); }
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        uint32_t total = 0u;
#pragma unroll
        for (int k = 0; k < 4; k++)
#pragma unroll
            for (int w = 0; w < 4; w++) total += sCnt[k][w];
        a.state[2u + b] = total;
    }
}

__global__ __launch_bounds__(256) void k0_band_scatter(const SelectArgs a)
{
    __shared__ uint32_t sPart[4];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const uint32_t b = blockIdx.x;
    // the block's sixteen ballot words (block-uniform addresses: scalar loads), requested before the counts are added up
    unsigned long long keep[4];
    uint32_t before[4], total = 0u;
#pragma unroll
    for (int k = 0; k < 4; k++) {
        before[k] = total;
#pragma unroll
        for (int w = 0; w < 4; w++) {
            const unsigned long long m = a.keep[(size_t)b * 16u + (uint32_t)(4 * k + w)];
            const uint32_t c = (uint32_t)__popcll(m);
            if (w == wave) keep[k] = m;
            if (w < wave) before[k] += c;
            total += c;
        }
    }
    // the lights of all blocks in front of this one: their counts, added up by the block itself (at most 1 024 words at a million lights)
    uint32_t acc = 0u;
    for (uint32_t i = threadIdx.x; i < b; i += 256u) acc += a.state[2u + i];
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) acc += (uint32_t)__shfl_xor((int)acc, d, 64);
    if (lane == 0) sPart[wave] = acc;
    __syncthreads();
    const uint32_t base = (sPart[0] + sPart[1]) + (sPart[2] + sPart[3]);
    if (b == gridDim.x - 1u && threadIdx.x == 0) a.state[0] = base + total;
#pragma unroll
    for (int k = 0; k < 4; k++) {
        if ((keep[k] >> lane) & 1ull) {
            const uint32_t c = base + before[k] + (uint32_t)__popcll(keep[k] & lanemask_lt());
            const int j = (int)b * SEL_LIGHTS + k * 256 + (int)threadIdx.x;
            float x, y, z, radius;
            uint32_t type;
            if (a.prepStaged) { // (k0_band_count has just written the 20-byte views of all lights)
                const float4 pr = a.prepPosRadius[j];
                x = pr.x; y = pr.y; z = pr.z; radius = pr.w;
                type = a.prepType[j];
            } else select_load(a, j, x, y, z, radius, type);
            a.lightView[c] = select_view(a, x, y, z, radius);
            a.lightType[c] = type;
            a.lightMap[c] = (uint32_t)j;
            if (a.prepStaged && a.stageSelectedOnly) { // (the kept tenth: its records are read a second time for 80 of 100 bytes per light less to write)
                const float4* L = reinterpret_cast<const float4*>(a.lights + j);
                float4 o0, o1, o2, o3, o4;
                stage_light_record(L[0], L[1], L[2], L[3], L[4], L[5], L[6], o0, o1, o2, o3, o4);
                float4* o = a.prepStaged + (size_t)j * LREC;
                o[0] = o0; o[1] = o1; o[2] = o2; o[3] = o3; o[4] = o4;
            }
        }
    }
}

#define LDS_K01_PREPARE (2 * MAX_BANDS_PER_BLOCK * 16)
// (82 scalar registers = seven blocks per CU; capped at 80 as k1_tile_cull is, the lower half band's step went 102-104 -> 107-109 us beside the previous frame's
// shade and nothing else moved: left alone -- profiles/r05/ab_sgpr_cap.txt)
#ifndef CULL_SGPR_CAP_PREPARE
#define CULL_SGPR_CAP_PREPARE 96
#endif
__global__ __launch_bounds__(256) __attribute__((amdgpu_num_sgpr(CULL_SGPR_CAP_PREPARE))) void k01_prepare(PrepareArgs a)
{
    __shared__ __attribute__((aligned(16))) unsigned char lds[LDS_K01_PREPARE];
    const int b = (int)blockIdx.x;
    if (b == 0 && threadIdx.x == 0) a.heavy[0] = 0u;
    if (b < a.lightRoleBlocks) k0_lights(b, lds, a);
    else if (b < a.lightRoleBlocks + a.frustumBlocks) k1_tile_frusta(b - a.lightRoleBlocks, a);
    else k1_tile_setup(b - a.lightRoleBlocks - a.frustumBlocks, lds, a);
}

__device__ __forceinline__ uint64_t lanemask_lt()
{
    const uint32_t lane = threadIdx.x & 63;
    return lane == 0 ? 0ull : (~0ull >> (64 - lane));
}

// LDS hand-off between the lanes of ONE wave: the wave's DS operations execute in order, so only the compiler has to be
// told not to move accesses across this point (and to wait for outstanding DS results).
#define WAVE_SYNC() __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup")

// ------------------------------------------------------------------------------------------------------------
// K1b2: candidate lists of 4x4-tile groups.  One 256-thread block per group ANDs the masks of the group's column and
// row (256 words = 16 384 lights per step) and writes the set bits -- ascending light index -- as a
// contiguous list: a block-wide prefix sum of the words' popcounts gives every word its output position, so the
// sparse-bits -> dense-list conversion is fully parallel.  Done once per 16 tiles, not per tile (profiles/r01: the
// per-tile scalar version saturated the CUs' scalar ALUs; a per-group scalar version was tail-bound by cluster groups).
// ------------------------------------------------------------------------------------------------------------
#define GL_WPT 4 // words per thread and round in k1_group_lists
__global__ __launch_bounds__(256) void k1_group_lists(const unsigned long long* __restrict__ masks, const unsigned long long* __restrict__ dirWords,
                                                       int stride, int groupsX, uint32_t* __restrict__ groupCount, uint32_t* __restrict__ groupList,
                                                       uint32_t* __restrict__ heavy, uint32_t heavyMin, const uint32_t* __restrict__ selCount)
{
    __shared__ uint32_t sW[4];
    // (2-D grid: group column, group row -- no division by the run-time groupsX)
    const int g = (int)blockIdx.y * groupsX + (int)blockIdx.x, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const unsigned long long* __restrict__ c = masks + (size_t)blockIdx.x * stride;
    const unsigned long long* __restrict__ r = masks + (size_t)(groupsX + (int)blockIdx.y) * stride;
    const int words = selCount ? min((int)(((*selCount + 63u) / 64u + 1u) & ~1u), stride) : stride; // (behind the band selection: the selected lights' words; the masks keep the capacity's stride)
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
    if (threadIdx.x == 0) {
        uint32_t word = base > CAPG ? GROUP_OVERFLOW : base;
        // a light cluster (several chunks and a nearest-128 selection per tile: ~20 us for one of its row blocks, wherever in the grid it sits):
        // listed, so that k1_tile_cull can start with it
        if (base > heavyMin) {
            const uint32_t slot = atomicAdd(&heavy[0], 1u);
            if (slot < (uint32_t)HEAVY_MAX) {
                heavy[1u + slot] = (uint32_t)g;
                word = word != GROUP_OVERFLOW ? (word | GROUP_LISTED) : GROUP_OVERFLOW_LISTED;
            }
        }
        groupCount[g] = word;
    }
}

// The same lists for LARGE light sets (>= 4 096 mask words: C5's million lights = 16 384 words per band, ~420 candidates per group, 97 % of the
// AND's words empty).  The kernel above takes 152 us there and is bound by vector-instruction issue: every round of 1 024 words pays a block-wide
// scan and, because a handful of its 256 threads do hold a candidate, the whole divergent bit loop (64-bit ctz / clear-lowest / shifts and the
// store's address arithmetic: ~50 of the ~63 instructions a 128-word row costs a wave; measured with SQ counters on three rewrites that kept
// that shape and all ran 152-167 us -- fewer bytes, prefetching and DPP scans changed nothing).  So the two halves are separated:
//   * one wave per group walks all the words of its column's mask in rows of 128 (two words per lane, one coalesced 16-byte load per row),
//     ANDs them with the row band's mask and only QUEUES the non-empty words -- word index + bits, in word order, by ballot + mbcnt -- in LDS;
//   * whenever 64 words are queued, the wave turns to them one LANE per WORD: one DPP prefix sum over the 64 popcounts gives every word its
//     output position, and the bit loop runs with all lanes busy (~8 such batches per group instead of 128 near-empty ones).
// A block takes four column-adjacent groups of a group row, one wave each: the row band's mask -- the same for the four -- is fetched once per
// block (each wave loads one row of every step of four) and handed round through LDS, 160 instead of 256 KB of L2 reads per group.
__device__ __forceinline__ uint32_t wave_incl_scan_u32(uint32_t v)
{
    // Hillis-Steele inside each row of 16 lanes (a shifted-in lane outside the row reads 0), then the row totals: lane 15 of rows 0 / 2 into rows
    // 1 / 3, lane 31 into rows 2 and 3.  (A DPP read of a VGPR needs 2 wait states after the VALU write.)
    asm volatile("s_nop 1\n\t"
                 "v_add_u32_dpp %0, %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:0\n\t"
                 "s_nop 1\n\t"
                 "v_add_u32_dpp %0, %0, %0 row_shr:2 row_mask:0xf bank_mask:0xf bound_ctrl:0\n\t"
                 "s_nop 1\n\t"
                 "v_add_u32_dpp %0, %0, %0 row_shr:4 row_mask:0xf bank_mask:0xf bound_ctrl:0\n\t"
                 "s_nop 1\n\t"
                 "v_add_u32_dpp %0, %0, %0 row_shr:8 row_mask:0xf bank_mask:0xf bound_ctrl:0\n\t"
                 "s_nop 1\n\t"
                 "v_add_u32_dpp %0, %0, %0 row_bcast:15 row_mask:0xa bank_mask:0xf\n\t"
                 "s_nop 1\n\t"
                 "v_add_u32_dpp %0, %0, %0 row_bcast:31 row_mask:0xc bank_mask:0xf\n\t"
                 "s_nop 1"
                 : "+v"(v));
    return v;
}

#define GLW_ROWS 4  // rows of 128 words per step
#define GLW_Q 256   // queued words per wave (ring; < 64 pending + <= 128 new per row)
template <bool EXACT> // EXACT: words is a multiple of 128 * GLW_ROWS, no load needs a bounds check
__global__ __launch_bounds__(256) void k1_group_lists_wide(const unsigned long long* __restrict__ masks, const unsigned long long* __restrict__ dirWords,
                                                            int stride, int groupsX, uint32_t* __restrict__ groupCount, uint32_t* __restrict__ groupList,
                                                            const uint32_t* __restrict__ dirFlag, const uint32_t* __restrict__ selCount)
{
    const int words = (!EXACT && selCount) ? min((int)(((*selCount + 63u) / 64u + 1u) & ~1u), stride) : stride; // (behind the band selection: the selected lights' words)
    __shared__ __attribute__((aligned(16))) unsigned long long sRow[2][GLW_ROWS][128];
    const bool anyDir = *dirFlag != 0u; // no directional light in the set (the usual case): the draining waves need not wait for their words of dirWords
    __shared__ unsigned long long sQBits[4][GLW_Q];
    __shared__ uint32_t sQWord[4][GLW_Q];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int blocksX = (groupsX + 3) / 4;
    const int gy = (int)blockIdx.x / blocksX, gx = ((int)blockIdx.x % blocksX) * 4 + wave;
    const bool mine = gx < groupsX; // (a wave beyond the last column still fetches its share of the row band's mask)
    const int g = gy * groupsX + (mine ? gx : groupsX - 1);
    const unsigned long long* __restrict__ c = masks + (size_t)(mine ? gx : groupsX - 1) * stride;
    const unsigned long long* __restrict__ r = masks + (size_t)(groupsX + gy) * stride;
    uint32_t* __restrict__ list = groupList + (size_t)g * CAPG;
    unsigned long long* qBits = sQBits[wave];
    uint32_t* qWord = sQWord[wave];
    const int rows = (words + 127) / 128, steps = (rows + GLW_ROWS - 1) / GLW_ROWS;
    const ulonglong2 zero2 = { 0ull, 0ull };
    // (words is even on this path: 16-byte aligned pairs)
    auto load2 = [&](const unsigned long long* __restrict__ m, int row) {
        const int w = row * 128 + lane * 2;
        if (EXACT) return *reinterpret_cast<const ulonglong2*>(m + (row < rows ? w : lane * 2)); // (the prefetch past the last step re-reads row 0; unused)
        return w < words ? *reinterpret_cast<const ulonglong2*>(m + w) : zero2;
    };
    uint32_t base = 0;             // the group's entries so far (wave-uniform)
    uint32_t qHead = 0, qTail = 0; // ring indices (wave-uniform)
    // up to 64 queued words, one lane per word: positions by one prefix sum, then every lane writes its word's set bits (ascending)
    auto drain = [&](uint32_t n) {
        const uint32_t qi = (qHead + (uint32_t)lane) & (GLW_Q - 1);
        const bool have = (uint32_t)lane < n;
        unsigned long long mm = have ? qBits[qi] : 0ull;
        const uint32_t w = have ? qWord[qi] : 0u;
        const unsigned long long dd = (have && anyDir) ? dirWords[w] : 0ull;
        const uint32_t cnt = (uint32_t)__popcll(mm);
        const uint32_t incl = wave_incl_scan_u32(cnt);
        uint32_t pos = base + incl - cnt;
        const uint32_t first = w * 64u;
        while (mm != 0ull) {
            const int bit = __builtin_ctzll(mm);
            mm &= mm - 1ull;
            if (pos < CAPG) list[pos] = (first + (uint32_t)bit) | (uint32_t)((dd >> bit) & 1ull) << 31; // bit 31 = directional
            pos++;
        }
        base += (uint32_t)__builtin_amdgcn_readlane((int)incl, 63);
        qHead += n;
    };
    ulonglong2 cNext[GLW_ROWS], rNext;
#pragma unroll
    for (int k = 0; k < GLW_ROWS; k++) cNext[k] = load2(c, k);
    rNext = load2(r, wave);
    for (int s = 0; s < steps; s++) {
        ulonglong2 cv[GLW_ROWS];
#pragma unroll
        for (int k = 0; k < GLW_ROWS; k++) cv[k] = cNext[k];
        *reinterpret_cast<ulonglong2*>(&sRow[s & 1][wave][lane * 2]) = rNext;
        // the next step's words are requested before this step's work
#pragma unroll
        for (int k = 0; k < GLW_ROWS; k++) cNext[k] = load2(c, (s + 1) * GLW_ROWS + k);
        rNext = load2(r, (s + 1) * GLW_ROWS + wave);
        __syncthreads(); // this step's four rows of the row band's mask are in LDS (the buffer of step s - 1 is free: every wave has passed its reads)
        if (!mine) continue;
#pragma unroll
        for (int k = 0; k < GLW_ROWS; k++) {
            const ulonglong2 rv = *reinterpret_cast<const ulonglong2*>(&sRow[s & 1][k][lane * 2]);
            const unsigned long long m0 = cv[k].x & rv.x, m1 = cv[k].y & rv.y;
            const unsigned long long b0 = __ballot(m0 != 0ull), b1 = __ballot(m1 != 0ull);
            if ((b0 | b1) == 0ull) continue; // (wave-uniform)
            const unsigned long long lt = lanemask_lt();
            const uint32_t i0 = qTail + (uint32_t)__popcll(b0 & lt) + (uint32_t)__popcll(b1 & lt);
            const uint32_t w = (uint32_t)((s * GLW_ROWS + k) * 128 + lane * 2);
            if (m0 != 0ull) { qBits[i0 & (GLW_Q - 1)] = m0; qWord[i0 & (GLW_Q - 1)] = w; }
            const uint32_t i1 = i0 + (m0 != 0ull ? 1u : 0u);
            if (m1 != 0ull) { qBits[i1 & (GLW_Q - 1)] = m1; qWord[i1 & (GLW_Q - 1)] = w + 1u; }
            qTail += (uint32_t)__popcll(b0) + (uint32_t)__popcll(b1);
            while (qTail - qHead >= 64u) { WAVE_SYNC(); drain(64u); WAVE_SYNC(); }
        }
    }
    if (mine) {
        if (qTail != qHead) { WAVE_SYNC(); drain(qTail - qHead); }
        // (round 5, measured and dropped: light clusters listed here too, as k1_group_lists does, their tiles a block each at the front of k1_tile_cull's grid
        // -- thresholds of 768 / 1 024 / 1 536 candidates on an eighth of the 8K frame under a million lights: 27.0 / 27.7 / 27.8 us against 27.6 without;
        // what that launch lost there was the shading hint's atomics, 37.6 -> 27.7 us)
        if (lane == 0) groupCount[g] = base > CAPG ? GROUP_OVERFLOW : base;
    }
}

// Round 6: the same lists with the masks read LESS OFTEN.  In the kernel above a block is four column-adjacent groups of one group row: every group streams its
// column's whole mask and a quarter of the row band's -- (4 + 1) / 4 = 1.25 mask rows per group and step; at a million lights on the 8K frame that is 8 160 groups
// x 128 KB x 1.25 = 1.3 GB out of the Infinity Cache (the masks are 24.6 MB: they live there, not in the XCDs' 4 MB L2s) for 84 us -- 370 MB of it HBM reads
// (profiles/r05/traffic_C5.json), fifteen times the masks' size.  Here a block is a 4 x 4 patch of groups, sixteen waves: per step the block fetches each of its
// four column masks and each of its four row-band masks ONCE (sixteen 1 KB rows -- one load per wave) into LDS, and every wave ANDs its own pair from there:
// (4 + 4) / 16 = 0.5 mask rows per group and step.  The rest is the kernel above -- queue the non-empty words, drain 64 at a time one lane per word -- with the
// row's 128 words taken as two halves of 64 (lane l: words l and 64 + l) so that a queue of 128 entries is enough (< 64 pending + <= 64 new) and sixteen of them
// fit beside the mask rows: 56 KB of LDS a block, two blocks -- 32 waves -- per CU.  Same lists, bit for bit: entries leave in word order, bits ascending.
#define GLW16_ROWS 2  // rows of 128 words per step and source
#define GLW16_Q 128   // queued words per wave
template <bool EXACT> // EXACT: words is a multiple of 128 * GLW16_ROWS, no load needs a bounds check
__global__ __launch_bounds__(1024) void k1_group_lists_wide16(const unsigned long long* __restrict__ masks, const unsigned long long* __restrict__ dirWords, int stride,
                                                              int groupsX, int groupsY, uint32_t* __restrict__ groupCount, uint32_t* __restrict__ groupList,
                                                              const uint32_t* __restrict__ dirFlag, const uint32_t* __restrict__ selCount)
{
    const int words = (!EXACT && selCount) ? min((int)(((*selCount + 63u) / 64u + 1u) & ~1u), stride) : stride;
    __shared__ __attribute__((aligned(16))) unsigned long long sMask[2][8][GLW16_ROWS][128]; // [buffer][0-3: the block's columns, 4-7: its row bands][row of the step][word]
    __shared__ unsigned long long sQBits[16][GLW16_Q];
    __shared__ uint32_t sQWord[16][GLW16_Q];
    const bool anyDir = *dirFlag != 0u;
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int wc = wave & 3, wr = wave >> 2;
    const int blocksX = (groupsX + 3) / 4;
    const int by = (int)blockIdx.x / blocksX, bx = (int)blockIdx.x % blocksX;
    const int gx = bx * 4 + wc, gy = by * 4 + wr;
    const bool mine = gx < groupsX && gy < groupsY;
    const int g = min(gy, groupsY - 1) * groupsX + min(gx, groupsX - 1);
    // this wave's share of the block's fetches: source src (a column or a row band of the patch), row k of every step
    const int src = wave >> 1, k = wave & 1;
    const unsigned long long* __restrict__ m = masks + (size_t)(src < 4 ? min(bx * 4 + src, groupsX - 1) : groupsX + min(by * 4 + (src - 4), groupsY - 1)) * stride;
    uint32_t* __restrict__ list = groupList + (size_t)g * CAPG;
    unsigned long long* qBits = sQBits[wave];
    uint32_t* qWord = sQWord[wave];
    const int rows = (words + 127) / 128, steps = (rows + GLW16_ROWS - 1) / GLW16_ROWS;
    const ulonglong2 zero2 = { 0ull, 0ull };
    auto load2 = [&](int row) {
        const int w = row * 128 + lane * 2;
        if (EXACT) return *reinterpret_cast<const ulonglong2*>(m + (row < rows ? w : lane * 2)); // (the prefetch past the last step re-reads row 0; unused)
        return w < words ? *reinterpret_cast<const ulonglong2*>(m + w) : zero2;
    };
    uint32_t base = 0, qHead = 0, qTail = 0; // the group's entries so far; ring indices (wave-uniform)
    auto drain = [&](uint32_t n) {
        const uint32_t qi = (qHead + (uint32_t)lane) & (GLW16_Q - 1);
        const bool have = (uint32_t)lane < n;
        unsigned long long mm = have ? qBits[qi] : 0ull;
        const uint32_t w = have ? qWord[qi] : 0u;
        const unsigned long long dd = (have && anyDir) ? dirWords[w] : 0ull;
        const uint32_t cnt = (uint32_t)__popcll(mm);
        const uint32_t incl = wave_incl_scan_u32(cnt);
        uint32_t pos = base + incl - cnt;
        const uint32_t first = w * 64u;
        while (mm != 0ull) {
            const int bit = __builtin_ctzll(mm);
            mm &= mm - 1ull;
            if (pos < CAPG) list[pos] = (first + (uint32_t)bit) | (uint32_t)((dd >> bit) & 1ull) << 31; // bit 31 = directional
            pos++;
        }
        base += (uint32_t)__builtin_amdgcn_readlane((int)incl, 63);
        qHead += n;
    };
    ulonglong2 next = load2(k);
    for (int s = 0; s < steps; s++) {
        *reinterpret_cast<ulonglong2*>(&sMask[s & 1][src][k][lane * 2]) = next;
        next = load2((s + 1) * GLW16_ROWS + k); // the next step's row is requested before this step's work
        __syncthreads(); // this step's sixteen rows are in LDS (the buffer of step s - 1 is free: every wave has passed its reads of it)
        if (!mine) continue;
#pragma unroll
        for (int r = 0; r < GLW16_ROWS; r++) {
#pragma unroll
            for (int h = 0; h < 2; h++) { // the row's words [64 h, 64 h + 64), one per lane
                const unsigned long long mm = sMask[s & 1][wc][r][h * 64 + lane] & sMask[s & 1][4 + wr][r][h * 64 + lane];
                const unsigned long long b = __ballot(mm != 0ull);
                if (b == 0ull) continue; // (wave-uniform)
                if (mm != 0ull) {
                    const uint32_t i = (qTail + (uint32_t)__popcll(b & lanemask_lt())) & (GLW16_Q - 1);
                    qBits[i] = mm; qWord[i] = (uint32_t)((s * GLW16_ROWS + r) * 128 + h * 64 + lane);
                }
                qTail += (uint32_t)__popcll(b);
                if (qTail - qHead >= 64u) { WAVE_SYNC(); drain(64u); WAVE_SYNC(); }
            }
        }
    }
    if (mine) {
        if (qTail != qHead) { WAVE_SYNC(); drain(qTail - qHead); }
        if (lane == 0) groupCount[g] = base > CAPG ? GROUP_OVERFLOW : base;
    }
}

// ------------------------------------------------------------------------------------------------------------
// K1c: exact per-tile cull, one wave per tile, canonical offsets by decoupled look-back, lists written in place.
// ------------------------------------------------------------------------------------------------------------
struct TileCtx {
    float n[4][3];
    float cx, cy, cz, zNear, zFar;
};

// Math.glsl:224-239 SphereFrustumOverlaps + ComputeLightCulling.shader:187 impact
// tile_test for 64 candidates at once, as the wave mask of the lanes that PASS.  Every comparison is evaluated and the six results are combined
// as scalar masks -- no early outs, no per-lane booleans (a bool lives in a VGPR as 0 / 1 and costs a v_cndmask + v_cmp per use).  With the early
// outs the test was eight exec-mask save / restore pairs and branches per 64 candidates: ~90 instructions a step, a third of them scalar
// bookkeeping, in a kernel that SQ counters show to be bound by instruction issue (264 vector + 292 scalar + 57 branch instructions per wave in
// round 2), not by its round trips.  Same comparisons on the same operands, NaN behaviour included: a comparison that is false next to a NaN
// rejects nothing, as before.
__device__ __forceinline__ unsigned long long tile_test_mask(const TileCtx& t, const float4 lv)
{
    const float r = lv.w;
    unsigned long long rej = __ballot(lv.z - r > t.zNear) | __ballot(lv.z + r < t.zFar);
#pragma unroll
    for (int k = 0; k < 4; k++) rej |= __ballot(dot3f(t.n[k][0], t.n[k][1], t.n[k][2], lv.x, lv.y, lv.z) < -r);
    return ~rej;
}
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

// the same append from a wave mask (the lanes of `mask` hold a passing candidate j)
__device__ __forceinline__ void wave_append_mask(const unsigned long long mask, uint32_t j, uint32_t& count, uint32_t* sIdx)
{
    const uint32_t pos = count + (uint32_t)__popcll(mask & lanemask_lt());
    const unsigned long long fits = mask & __ballot(pos < CAND);
    if (__builtin_amdgcn_inverse_ballot_w64(fits)) sIdx[pos] = j;
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



struct CullArgs {
    const float4* lightView; const uint32_t* lightType; const float4* tileInfo;
    const unsigned long long* masks; const uint32_t* groupCount; const uint32_t* groupList;
    uint32_t* tileNum; uint8_t* tileNum8; uint32_t* tileLists; uint32_t* dirFlag;
    int N, words, Tx, groupsX, bandRows;
    const uint32_t* lightMap; const uint32_t* selCount; // behind the band selection (kernels with SEL): compact index -> light index; the number of selected lights
    const uint32_t* heavy; int headRows; // k1_group_lists' cluster list and the grid rows in front of the tile rows that take its tiles (0: none)
};

// The <= 196 candidates of a tile (sIdx, ascending light index) -> its list at `out` (Appendix A steps 4 + 5).  One wave.
// (SEL: the candidates are compact indices of the band selection's light set -- ascending like the light indices they stand for, so every rank and tie-break
// below is the one of the full set -- and `map` turns an entry into its light index on the way out)
template <bool SEL>
__device__ __forceinline__ void emit_list(const TileCtx& t, const uint32_t n, uint32_t* sIdx, float* sImp, const float4* __restrict__ lightView,
                                          uint32_t* __restrict__ out, const uint32_t* __restrict__ map)
{
    auto light_of = [&](const uint32_t e) -> uint32_t { const uint32_t i = e & 0x7FFFFFFFu; if constexpr (SEL) return map[i]; else return i; };
    const int lane = threadIdx.x & 63;
    const uint32_t num = n < KEEP ? n : KEEP;
    WAVE_SYNC(); // orders the wave's LDS writes (the appends) with the reads below
    if (n <= KEEP) {
        // :235-238 culledLights.indices[offset + i] = candidateIndices[numCandidates - i - 1]
        for (uint32_t i = lane; i < num; i += 64) out[i] = light_of(sIdx[n - 1 - i]);
        return;
    }
    // ---- 196 -> 128 selection (ComputeLightCulling.shader:198-225)
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
                        const float x = sImp[j], y = sImp[j + 1];
                        if (x < y) {
                            sImp[j] = y; sImp[j + 1] = x;
                            const uint32_t q = sIdx[j]; sIdx[j] = sIdx[j + 1]; sIdx[j + 1] = q;
                        }
                    }
                    if (--numSorted == 0) break;
                }
            }
            WAVE_SYNC();
        }
        for (uint32_t i = lane; i < num; i += 64) out[i] = light_of(sIdx[n - 1 - i]); // :235-238
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
            const float gq[4] = { gv.x, gv.y, gv.z, gv.w };
#pragma unroll
            for (int e = 0; e < 4; e++) {
#pragma unroll
                for (int i = 0; i < 4; i++) {
                    if (jb < i) rank[i] += (gq[e] < f[i]) ? 1u : 0u;
                    else if (jb > i) rank[i] += (gq[e] <= f[i]) ? 1u : 0u;
                    else {
                        const uint32_t q = q4 * 4u + e, k = lane + 64u * i;
                        rank[i] += (gq[e] < f[i] || (gq[e] == f[i] && q > k)) ? 1u : 0u;
                    }
                }
            }
        }
    }
#pragma unroll
    for (int i = 0; i < 4; i++) {
        const uint32_t k = lane + 64u * i;
        if (k < n && rank[i] < KEEP) out[rank[i]] = light_of(sIdx[k]);
    }
}

// the `cn` candidates staged in LDS through one tile's exact test, 64 per step, ordered append
__device__ __forceinline__ void test_staged(const TileCtx& t, const uint32_t cn, const uint32_t* sE, const float4* sLV, uint32_t& count, uint32_t* sIdx)
{
    const uint32_t lane = threadIdx.x & 63;
    for (uint32_t base = 0; base < cn && count < CAND; base += 64u) {
        const uint32_t i = base + lane;
        // (unconditional reads -- the staging arrays are CHUNK entries whatever the count -- and one combined mask: see tile_test)
        const uint32_t ee = sE[i & (CHUNK - 1)];
        const float4 lv = sLV[i & (CHUNK - 1)];
        const unsigned long long mask = __ballot(i < cn) & (__ballot((int)ee < 0) | tile_test_mask(t, lv)); // directional (bit 31): always a candidate, impact 0 (:153-162)
        wave_append_mask(mask, ee, count, sIdx);
    }
}

#define LDS_K1_TILE_CULL (CHUNK * 16 + CHUNK * 4 + 4 * CAND * 4 + 4 * CAND * 4)
// make EXTRA=-DCULL_PROF + scripts/cull_prof.py: per-block phase times (s_memrealtime: the 100 MHz constant clock, one time base for the whole chip)
#ifdef CULL_PROF
__device__ unsigned long long g_cullProf[65536][4];
extern "C" __attribute__((visibility("default"))) int sailor_hip_debug_read_cull_prof(void* dst, size_t bytes)
{
    return (int)hipMemcpyFromSymbol(dst, HIP_SYMBOL(g_cullProf), bytes);
}
#define PROF_T(i) if (threadIdx.x == 0 && blockIdx.y * gridDim.x + blockIdx.x < 65536) { g_cullProf[blockIdx.y * gridDim.x + blockIdx.x][i] = __builtin_amdgcn_s_memrealtime(); \
                                                                                        if ((i) == 0) g_cullProf[blockIdx.y * gridDim.x + blockIdx.x][2] = __builtin_amdgcn_s_getreg((20 << 0) | (0 << 6) | (3 << 11)) & 0xF; } /* [2] = HW_REG_XCC_ID (id 20), bits 0..3 */
#else
#define PROF_T(i)
#endif

__device__ __forceinline__ void load_tile_ctx(const float4* __restrict__ tileInfo, const int bandTile, TileCtx& t)
{
    const float4* ti = tileInfo + (size_t)bandTile * 4;
    const float4 q0 = ti[0], q1 = ti[1], q2 = ti[2], q3 = ti[3];
    t.n[0][0] = q0.x; t.n[0][1] = q0.y; t.n[0][2] = q0.z; t.cx = q0.w;
    t.n[1][0] = q1.x; t.n[1][1] = q1.y; t.n[1][2] = q1.z; t.cy = q1.w;
    t.n[2][0] = q2.x; t.n[2][1] = q2.y; t.n[2][2] = q2.z; t.zNear = q2.w;
    t.n[3][0] = q3.x; t.n[3][1] = q3.y; t.n[3][2] = q3.z; t.zFar = q3.w;
    t.cz = (t.zFar + t.zNear) * 0.5f;
}

// A tile of an OVERFLOWED group (more than CAPG candidates: a very dense region, or lights around the eye): one wave walks the tile's own two
// masks.  sQ: QCAP words of LDS of the wave's own.
__device__ __forceinline__ void walk_tile_masks(const TileCtx& t, const CullArgs& a, const int gx, const int tyLocal, uint32_t* sQ, uint32_t& count, uint32_t* sIdx)
{
    const int lane = threadIdx.x & 63;
    const unsigned long long* __restrict__ col = a.masks + (size_t)gx * a.words;
    const unsigned long long* __restrict__ row = a.masks + (size_t)(a.groupsX + tyLocal / GROUP) * a.words;
    const int words = a.selCount ? (int)((*a.selCount + 63u) / 64u) : a.words; // (behind the band selection: the selected lights' words)
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
                test_candidates(t, a.lightView, a.lightType, true, j, count, sIdx);
                qHead += 64u;
                WAVE_SYNC();
            }
        }
    }
    WAVE_SYNC();
    if (qTail != qHead && count < CAND) { // final partial round (< 64 pending)
        const uint32_t n = qTail - qHead;
        const uint32_t j = sQ[(qHead + lane) & (QCAP - 1)];
        test_candidates(t, a.lightView, a.lightType, (uint32_t)lane < n, j, count, sIdx);
    }
}

// What a tile leaves behind besides its list: the length, as a word (the shade's grid entry, k1_pack's scan) and as a byte (k1_pack's base sums; the
// band shade's split blocks find the band's long tiles in the bytes: sailor_hip_light_cull_tile_order).  (Measured and dropped: the length also added
// to a per-tile-row total with a relaxed device-scope atomic, for k1_pack's offsets -- 32 400 atomics on 135 words = five cache lines serialise at
// the memory side: k1_tile_cull 26 -> 140 us.  Round 4's shading hint -- the long tiles appended to a list by two device-scope counters -- paid the
// same way as soon as a band had more than a few hundred long tiles: +3 us on an eighth of the 4K frame, +6 on a quarter, +12.7 on a half (17.0 ->
// 29.7 us); round 5's split blocks read the length bytes instead and nothing here is atomic.)
__device__ __forceinline__ void publish_tile(const CullArgs& a, const int bandTile, const uint32_t num)
{
    a.tileNum[bandTile] = num;
    a.tileNum8[bandTile] = (uint8_t)num;
}

// The selection's rank of candidate k (ComputeLightCulling.shader:198-225: the partial bubble sort == rank under (impact ascending, candidate position
// descending); rank < 128 stays): the number of candidates q with impact_q < impact_k, or impact_q == impact_k and q > k.  Impacts are non-negative
// floats here (a NaN impact is sent to the literal sort before this is called), so they order like their BITS, and on integers
//     bits_q < bits_k  or  (bits_q == bits_k and q > k)   <=>   bits_q < bits_k + (q > k ? 1 : 0):
// one unsigned compare and one add-with-carry per candidate, the threshold chosen once per four candidates (the four that hold k itself are compared
// with bits_k, and the at most three behind k that tie with it are added afterwards) -- 2.75 instructions per candidate where the float form took
// six: a selection by the block 4.4 -> ~2 us.  sImp: n impacts, padded with 0xFFFFFFFF (above every threshold) to a multiple of four, 16-byte aligned.
__device__ __forceinline__ uint32_t rank_among(const float* sImp, const uint32_t n, const uint32_t k, const float imp)
{
    const uint4* s4 = reinterpret_cast<const uint4*>(sImp);
    const uint32_t t0 = __float_as_uint(imp), t1 = t0 + 1u;
    const uint32_t n4 = (n + 3u) / 4u;
    uint32_t rank = 0u;
    // (measured: four or eight reads in flight per trip change nothing -- a block's lone wave per SIMD issues one dependent instruction per ~10
    // cycles whatever the LDS does; a block whose four tiles ALL select spends ~11 us here, the tail of a cluster band's launch)
    for (uint32_t q4 = 0; q4 < n4; q4++) {
        const uint4 v = s4[q4];
        const uint32_t thr = 4u * q4 > k ? t1 : t0;
        rank += (v.x < thr ? 1u : 0u) + (v.y < thr ? 1u : 0u) + (v.z < thr ? 1u : 0u) + (v.w < thr ? 1u : 0u);
    }
    const uint4 own = s4[k >> 2];
    const uint32_t c = k & 3u;
    rank += (c < 1u && own.y == t0 ? 1u : 0u) + (c < 2u && own.z == t0 ? 1u : 0u) + (c < 3u && own.w == t0 ? 1u : 0u);
    return rank;
}

// ---- a tile of a LIGHT CLUSTER, one 256-thread block for the one tile (round 4).  A group with several chunks of candidates costs every one of
// its tiles 10-25 steps of the exact test and, where more than 196 pass, a 196 -> 128 selection -- ~20 us on one wave, the tail of the whole launch
// and THE cull of a cluster band.  Here the group's candidate list is cut into four contiguous shares, one per wave: each wave tests its share (its
// list entries, then its gathers: the records go from L2 straight into the lanes that test them, no LDS staging -- every record is used once) and
// appends what passes to a list of its own; the tile's candidate sequence is the four lists one after the other (ascending light index, cut at
// 196: what a wave collects beyond its first 196 can never be among the tile's first 196), and the selection's rank -- one candidate per THREAD
// against all n -- takes ~200 comparisons per thread instead of ~800 per lane.  Same candidates in the same order, same impacts, same rank rule
// (impact ascending, position descending): the list is the one-wave form's bit for bit (tests/test_light_cull_gpu.py: default == brute force).
template <bool SEL>
__device__ __forceinline__ void cluster_tile(const CullArgs& a, unsigned char* __restrict__ lds, uint32_t* sCnt, const int gx, const int tyLocal, const int col)
{
    uint32_t (*sIdxW)[CAND] = reinterpret_cast<uint32_t (*)[CAND]>(lds + CHUNK * 20);                  // [4][CAND]: the waves' own lists
    uint32_t* sAll = reinterpret_cast<uint32_t*>(lds + CHUNK * 20 + 4 * CAND * 4);                      // [CAND + 4]: the tile's candidates
    float* sImp = reinterpret_cast<float*>(lds + CHUNK * 20 + 4 * CAND * 4 + 2 * CAND * 4);             // [CAND + 4], 16-byte aligned (2 * CAND * 4 = 1568)
    const int tx = gx * GROUP + col;
    if (tx >= a.Tx) return;
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)), lane = threadIdx.x & 63;
    const int bandTile = tyLocal * a.Tx + tx, g = (tyLocal / GROUP) * a.groupsX + gx;
    const float4* __restrict__ lightView = a.lightView;
    const uint32_t* __restrict__ list = a.groupList + (size_t)g * CAPG;
    uint32_t gn = a.groupCount[g];
    gn = gn >= GROUP_OVERFLOW_LISTED ? GROUP_OVERFLOW : (gn & ~GROUP_LISTED);
    // the wave's share: a multiple of 64 candidates, at most CL_STEPS steps; the list entries of all of it are requested at once
    const uint32_t per = gn == GROUP_OVERFLOW ? 0u : ((gn + 255u) / 256u) * 64u;
    const uint32_t lo = (uint32_t)wave * per, hi = min(gn == GROUP_OVERFLOW ? 0u : gn, lo + per);
    uint32_t e[CL_STEPS];
#pragma unroll
    for (int k = 0; k < CL_STEPS; k++) { const uint32_t i = lo + 64u * k + (uint32_t)lane; e[k] = i < hi ? list[i] : 0u; }
    TileCtx t;
    load_tile_ctx(a.tileInfo, bandTile, t);
    if (threadIdx.x == 0) sCnt[4] = 0u; // "a NaN impact" (read after the second barrier below)
    uint32_t count = 0;
    uint32_t* sIdx = sIdxW[wave];
    if (gn != GROUP_OVERFLOW) {
        // (the gathers in two batches of four: 16 registers of records in flight, not 32 -- the ordinary blocks of this kernel live on 64 registers)
#pragma unroll
        for (int h = 0; h < CL_STEPS; h += 4) {
            if (lo + 64u * h >= hi) break;
            float4 lv[4];
#pragma unroll
            for (int k = 0; k < 4; k++) lv[k] = lightView[min(e[h + k] & 0x7FFFFFFFu, (uint32_t)(a.N - 1))];
#pragma unroll
            for (int k = 0; k < 4; k++) {
                const uint32_t i = lo + 64u * (h + k) + (uint32_t)lane;
                if (lo + 64u * (h + k) < hi && count < CAND) {
                    const unsigned long long mask = __ballot(i < hi) & (__ballot((int)e[h + k] < 0) | tile_test_mask(t, lv[k])); // directional (bit 31): always a candidate
                    wave_append_mask(mask, e[h + k], count, sIdx);
                }
            }
        }
    } else if (wave == 0) {
        walk_tile_masks(t, a, gx, tyLocal, reinterpret_cast<uint32_t*>(lds), count, sIdx); // (the staging area of the ordinary path: unused here)
    }
    if (lane == 0) sCnt[wave] = count < CAND ? count : CAND;
    __syncthreads();
    const uint32_t c0 = sCnt[0], c1 = sCnt[1], c2 = sCnt[2], c3 = sCnt[3];
    const uint32_t before = (wave > 0 ? c0 : 0u) + (wave > 1 ? c1 : 0u) + (wave > 2 ? c2 : 0u);
    const uint32_t all = (c0 + c1) + (c2 + c3), n = all < CAND ? all : CAND, num = n < KEEP ? n : KEEP;
    for (uint32_t i = lane; i < sCnt[wave] && before + i < CAND; i += 64) sAll[before + i] = sIdx[i];
    __syncthreads();
    if (threadIdx.x == 0) publish_tile(a, bandTile, num);
    uint32_t* __restrict__ out = a.tileLists + (size_t)bandTile * KEEP;
    if (n <= KEEP) { // :235-238 culledLights.indices[offset + i] = candidateIndices[numCandidates - i - 1]
        if (threadIdx.x < n) { const uint32_t i = sAll[n - 1 - threadIdx.x] & 0x7FFFFFFFu; out[threadIdx.x] = SEL ? a.lightMap[i] : i; }
        return;
    }
    // ---- 196 -> 128 (ComputeLightCulling.shader:198-225): one candidate per thread
    const uint32_t k = threadIdx.x;
    uint32_t mine = 0u;
    float imp = 0.0f;
    if (k < n) {
        mine = sAll[k];
        imp = (mine & 0x80000000u) ? 0.0f : tile_impact(t, lightView[mine & 0x7FFFFFFFu]);
        sImp[k] = imp;
    } else if (k < (uint32_t)CAND + 4u) reinterpret_cast<uint32_t*>(sImp)[k] = 0xFFFFFFFFu; // (rank_among's padding)
    if (__ballot(k < n && imp != imp) != 0ull && lane == 0) sCnt[4] = 1u;
    __syncthreads();
    if (sCnt[4] != 0u) { // a NaN impact has no rank: the literal bubble sort, on one wave (emit_list)
        if (wave == 0) emit_list<SEL>(t, n, sAll, sImp, lightView, out, a.lightMap);
        return;
    }
    if (k >= n) return;
    const uint32_t rank = rank_among(sImp, n, k, imp);
    if (rank < KEEP) { const uint32_t i = mine & 0x7FFFFFFFu; out[rank] = SEL ? a.lightMap[i] : i; }
}

// The validation path (SAILOR_CULL_BRUTE_FORCE; also tiny light sets and odd projections): no pre-filter, no staging, no cooperation between
// waves -- every tile walks ALL lights by itself, 64 per step, and selects with emit_list on its own wave.  One block per run of four tiles.
__global__ __launch_bounds__(256) void k1_tile_cull_brute(const CullArgs a)
{
    __shared__ __attribute__((aligned(16))) unsigned char lds[2 * 4 * CAND * 4];
    uint32_t (*sIdxAll)[CAND] = reinterpret_cast<uint32_t (*)[CAND]>(lds);
    float (*sImpAll)[CAND] = reinterpret_cast<float (*)[CAND]>(lds + 4 * CAND * 4);
    if (blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0) *a.dirFlag = 0u;
    const int gx = (int)blockIdx.x, tyLocal = (int)blockIdx.y;
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)), lane = threadIdx.x & 63;
    const int tx = gx * GROUP + wave;
    if (tx >= a.Tx) return;
    const int bandTile = tyLocal * a.Tx + tx;
    TileCtx t;
    load_tile_ctx(a.tileInfo, bandTile, t);
    uint32_t count = 0;
    for (int base = 0; base < a.N && count < CAND; base += 64) {
        const int j = base + lane;
        test_candidates(t, a.lightView, a.lightType, j < a.N, (uint32_t)j, count, sIdxAll[wave]);
    }
    const uint32_t n = count < CAND ? count : CAND;
    if (lane == 0) publish_tile(a, bandTile, n < KEEP ? n : KEEP);
    emit_list<false>(t, n, sIdxAll[wave], sImpAll[wave], a.lightView, a.tileLists + (size_t)bandTile * KEEP, nullptr);
}

// The 196 -> 128 selections of a block's tiles by the WHOLE block (ComputeLightCulling.shader:198-225): one candidate per thread and tile.  sIdxAll[w]:
// tile w's candidates (ascending light index, bit 31 = directional), sCnt[w] their number (selection needed where > 128), sImpAll[w]: CAND floats of
// its own, 16-byte aligned.  Two phases with ONE barrier between them: the impacts of every tile that needs a selection (the gathers of up to four
// tiles in flight together), then the ranks.  A tile with a NaN impact has no rank: the literal bubble sort, on one wave (emit_list).
template <bool SEL>
__device__ __forceinline__ void block_select(const CullArgs& a, const int firstBandTile, const uint32_t* sCnt, uint32_t (*sIdxAll)[CAND], float (*sImpAll)[CAND], uint32_t* sFlags)
{
    const uint32_t k = threadIdx.x, lane = threadIdx.x & 63;
    const float4* __restrict__ lightView = a.lightView;
    uint32_t mine[4];
    float imp[4];
#pragma unroll
    for (int w = 0; w < 4; w++) {
        const uint32_t n = sCnt[w];
        mine[w] = 0u; imp[w] = 0.0f;
        if (n <= (uint32_t)KEEP) continue; // (block-uniform)
        const float4* ti = a.tileInfo + (size_t)(firstBandTile + w) * 4; // (block-uniform: scalar loads)
        const float cx = ti[0].w, cy = ti[1].w, cz = (ti[3].w + ti[2].w) * 0.5f;
        if (k < n) {
            mine[w] = sIdxAll[w][k];
            if (!(mine[w] & 0x80000000u)) { // :187 impact = distance(light, frustum centre); a directional light's is 0 (:153-162)
                const float4 lv = lightView[mine[w] & 0x7FFFFFFFu];
                const float dx = lv.x - cx, dy = lv.y - cy, dz = lv.z - cz;
                imp[w] = sqrtf(dot3f(dx, dy, dz, dx, dy, dz));
            }
            sImpAll[w][k] = imp[w];
        } else if (k < (uint32_t)CAND) reinterpret_cast<uint32_t*>(sImpAll[w])[k] = 0xFFFFFFFFu; // (rank_among's padding)
        if (__ballot(k < n && imp[w] != imp[w]) != 0ull && lane == 0) sFlags[w] = 1u;
    }
    __syncthreads();
#pragma unroll
    for (int w = 0; w < 4; w++) {
        const uint32_t n = sCnt[w];
        if (n <= (uint32_t)KEEP) continue;
        uint32_t* __restrict__ out = a.tileLists + (size_t)(firstBandTile + w) * KEEP;
        if (sFlags[w] != 0u) {
            if (threadIdx.x < 64) {
                TileCtx t;
                load_tile_ctx(a.tileInfo, firstBandTile + w, t);
                emit_list<SEL>(t, n, sIdxAll[w], sImpAll[w], lightView, out, a.lightMap);
            }
        } else if (k < n) {
            const uint32_t rank = rank_among(sImpAll[w], n, k, imp[w]);
            if (rank < KEEP) { const uint32_t i = mine[w] & 0x7FFFFFFFu; out[rank] = SEL ? a.lightMap[i] : i; }
        }
    }
}

// COOP: tiles with more than 128 candidates get their selection from the whole block (block_select: 2.8 us a tile) behind a barrier every block pays,
// instead of from their own wave (emit_list: ~11 us on a block's lone wave -- the launch's tail, profiles/r04/cull_block_timeline.txt: blocks with ONE
// full tile that start 10-20 us into the launch and end at 24-29 while 99 % of the blocks are done at 22).  On for the 4K frame and its bands (a band
// IS its longest block: k1_tile_cull 14.7 -> 9.7 us on a cluster band of an 8-way split), off on the wide path's 420-candidate lists (8K, a million
// lights: no listed clusters, seven test steps per tile, and the barrier costs the throughput phase 15 %: 99 -> 114 us).
// (The scalar registers capped at 80 -- round 5: a CU admits 256-thread blocks up to floor(800 / (ceil(sgprs / 16) 16 + 16)) of them (MI355X_MICROARCH.md), i.e.
// eight with up to 80 scalar registers, SEVEN with the 81-95 this kernel takes by itself; 25-42 of them then live in the lanes of a vector register.  Same
// box, alternating: the frame pipeline's step 166-167 -> 162-164 us, a quarter band's 62 -> 58.  Measured with it and dropped: the candidates' RECORDS written beside the group lists'
// indices by k1_group_lists, so that this kernel stages them in one round trip instead of a dependent gather -- k1_tile_cull 24.2 -> 22.7 us, but
// k1_group_lists, one round of blocks whose life IS the launch, 5.7 -> 8.8 us; a band's step +1.5 us.)
#ifndef CULL_SGPR_CAP
#define CULL_SGPR_CAP 80
#endif
template <bool COOP, bool SEL>
__global__ __launch_bounds__(256) __attribute__((amdgpu_num_sgpr(CULL_SGPR_CAP))) void k1_tile_cull(const CullArgs a)
{
    // One 256-thread block per run of four tiles (a group's four columns in one tile row), one wave per tile.  The group's candidate records are
    // staged in LDS, CHUNK at a time, by the four waves together (list entries first, then the dependent 16-byte gathers, four of each in flight
    // per thread), and every tile streams them out of LDS.
    // Measured and dropped in round 4 (scripts/cull_prof.py: the per-block timeline on the 100 MHz clock): TWO tile rows per block, the staged
    // candidates serving both -- half the blocks, half the staging traffic, 90 % of the block slots filled instead of 70 % -- but a block then lives
    // 8.9 us instead of 3.3: the waves spend their time in the tests, not in the round trips, once the CU is full; chain 57 us against 48.
    // (Rounds 2 and 3: one block per group with a wave per tile row, 38.8 us against 32.4.)
    __shared__ __attribute__((aligned(16))) unsigned char lds[LDS_K1_TILE_CULL];
    __shared__ uint32_t sCnt[8]; // [0..3] the waves' candidate counts, [4..7] "a NaN impact" per tile (cluster_tile: [4])
    float4* sLV = reinterpret_cast<float4*>(lds);                                                         // [CHUNK] candidate (view pos, radius)
    uint32_t* sE = reinterpret_cast<uint32_t*>(lds + CHUNK * 20 - CHUNK * 4);                             // [CHUNK] candidate light index | directional << 31
    uint32_t (*sIdxAll)[CAND] = reinterpret_cast<uint32_t (*)[CAND]>(lds + CHUNK * 20);                   // [4][CAND]
    float (*sImpAll)[CAND] = reinterpret_cast<float (*)[CAND]>(lds + CHUNK * 20 + 4 * CAND * 4);          // [4][CAND], 16-byte aligned (CAND * 4 = 784)
    const float4* __restrict__ lightView = a.lightView;
    const int N = a.N, Tx = a.Tx, groupsX = a.groupsX;
    // (a 2-D grid, (group column, head rows + tile rows): no division by a run-time divisor)
    const int gx = (int)blockIdx.x, tyLocal = (int)blockIdx.y - a.headRows;
    if (blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0) *a.dirFlag = 0u; // (k01_prepare of the NEXT cull sets it again if a light is directional; k1_group_lists_wide has read it)
    if (tyLocal < 0) {
        // HEAD ROWS: the tiles of the light clusters k1_group_lists listed, sixteen blocks per group, at the front of the grid (they are the
        // launch's longest blocks); the blocks at their regular positions further down leave at once.
        const uint32_t hb = blockIdx.y * (uint32_t)groupsX + blockIdx.x;
        if ((hb >> 4) >= min(a.heavy[0], (uint32_t)HEAVY_MAX)) return;
        const int hg = (int)a.heavy[1u + (hb >> 4)];
        const int row = (hg / groupsX) * GROUP + (int)((hb >> 2) & 3u);
        if (row >= a.bandRows) return;
        PROF_T(0);
        cluster_tile<SEL>(a, lds, sCnt, hg % groupsX, row, (int)(hb & 3u));
        PROF_T(3);
        return;
    }
    PROF_T(0);
    const int g = (tyLocal / GROUP) * groupsX + gx;
    // (the wave index as a scalar: a tile's frustum -- the same 64 bytes for all lanes -- then arrives by scalar loads)
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)), lane = threadIdx.x & 63;
    uint32_t* sIdx = sIdxAll[wave];

    // The kernel is bound by the latency of its dependent loads, so the first chunk's list entries are requested together with the group's count,
    // not after it: slots past the count hold stale entries of an earlier frame or nothing at all, hence the clamp to N - 1.
    const uint32_t* __restrict__ list = a.groupList + (size_t)g * CAPG;
    uint32_t e[CHUNK / 256];
    e[0] = list[threadIdx.x];
    uint32_t gn = a.groupCount[g];
    if (a.headRows > 0 && (gn == GROUP_OVERFLOW_LISTED || (gn != GROUP_OVERFLOW && (gn & GROUP_LISTED)))) return; // a listed cluster: the head rows have it
    gn = gn >= GROUP_OVERFLOW_LISTED ? GROUP_OVERFLOW : (gn & ~GROUP_LISTED);
#pragma unroll
    for (int k = 1; k < CHUNK / 256; k++) e[k] = (gn != GROUP_OVERFLOW && gn > 256u * k) ? list[threadIdx.x + 256u * k] : 0u;

    const int tx = gx * GROUP + wave;
    const bool active = tx < Tx;
    const int bandTile = tyLocal * Tx + tx;
    TileCtx t;
    if (active) load_tile_ctx(a.tileInfo, bandTile, t);
    uint32_t count = 0;
    if (gn != GROUP_OVERFLOW) {
        for (uint32_t c0 = 0; c0 < gn; c0 += CHUNK) {
            const uint32_t cn = min((uint32_t)CHUNK, gn - c0);
            if (c0) {
                __syncthreads(); // every wave is done with the previous chunk
#pragma unroll
                for (int k = 0; k < CHUNK / 256; k++) { const uint32_t i = threadIdx.x + 256u * k; e[k] = i < cn ? list[c0 + i] : 0u; }
            }
            float4 lv[CHUNK / 256];
            // (only the slots below the count: a gather through a stale entry of an earlier frame is a random 16-byte request for nothing)
#pragma unroll
            for (int k = 0; k < CHUNK / 256; k++) {
                lv[k] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
                if (threadIdx.x + 256u * k < cn) lv[k] = lightView[min(e[k] & 0x7FFFFFFFu, (uint32_t)(N - 1))];
            }
#pragma unroll
            for (int k = 0; k < CHUNK / 256; k++) { const uint32_t i = threadIdx.x + 256u * k; if (i < cn) { sE[i] = e[k]; sLV[i] = lv[k]; } }
            __syncthreads();
            if (active) test_staged(t, cn, sE, sLV, count, sIdx);
        }
    } else if (active) {
        // overflowed group (very dense region / lights around the eye): every tile walks its own two masks
        walk_tile_masks(t, a, gx, tyLocal, reinterpret_cast<uint32_t*>(sLV) + wave * QCAP, count, sIdx); // sLV is unused on this path
    }
    PROF_T(1);
    const uint32_t n = count < CAND ? count : CAND; // (0 for a wave beyond the last tile column)
    if (active && lane == 0) publish_tile(a, bandTile, n < KEEP ? n : KEEP);
    // The tile's list into the tile's own slot: k1_pack moves it to its canonical place, the shade reads it where it is.
    if (!COOP || gn == GROUP_OVERFLOW) { // (an overflowed group's staging area is the waves' queues: every wave selects for itself)
        // a tile with more than 128 candidates selects on its own wave (emit_list: ~9 us)
        if (active) emit_list<SEL>(t, n, sIdx, sImpAll[wave], lightView, a.tileLists + (size_t)bandTile * KEEP, a.lightMap);
    } else {
        // short lists leave at once (:235-238 culledLights.indices[offset + i] = candidateIndices[numCandidates - i - 1]); the others wait for the block
        if (lane == 0) sCnt[wave] = n;
        if (threadIdx.x < 4) sCnt[4 + threadIdx.x] = 0u; // block_select's "a NaN impact", one flag per tile
        if (active && n <= KEEP) {
            WAVE_SYNC();
            uint32_t* __restrict__ out = a.tileLists + (size_t)bandTile * KEEP;
            for (uint32_t i = lane; i < n; i += 64) { const uint32_t c = sIdx[n - 1 - i] & 0x7FFFFFFFu; out[i] = SEL ? a.lightMap[c] : c; }
        }
        __syncthreads();
        if (((sCnt[0] > (uint32_t)KEEP) | (sCnt[1] > (uint32_t)KEEP)) | ((sCnt[2] > (uint32_t)KEEP) | (sCnt[3] > (uint32_t)KEEP))) // (block-uniform)
            block_select<SEL>(a, tyLocal * Tx + gx * GROUP, sCnt, sIdxAll, sImpAll, sCnt + 4);
    }
    PROF_T(3);
}

// ------------------------------------------------------------------------------------------------------------
// K1d: canonical offsets (Appendix A step 6) and compaction.  Block p owns the tiles [64 p, 64 p + 64) of the band (tile-index order).  Its
// base is the sum of the list lengths of all tiles before them, read (the lengths as bytes: <= 32 KB at 4K, L2-resident) and added up by the block itself:
// no scan kernel, no look-back chain, no atomics.  One wave turns the 64 lengths into offsets (lightsGrid); the 64 lists are gathered from their slots into LDS, back to back,
// and leave for culledLights as one contiguous run.
// ------------------------------------------------------------------------------------------------------------
struct PackArgs {
    const uint32_t* tileNum; const uint8_t* tileNum8; const uint32_t* tileLists;
    SailorLightsGrid* grid; uint32_t* culled;
    int Tx, bandRows, bandTiles;
    uint32_t capacity;
};

__global__ __launch_bounds__(256) void k1_pack(const PackArgs a)
{
    __shared__ __attribute__((aligned(16))) uint32_t sBuf[PACK_TILES * KEEP]; // 32 KB: the block's lists, back to back
    __shared__ uint32_t sPart[4];
    __shared__ uint32_t sPre[PACK_TILES], sNum[PACK_TILES];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int t0 = (int)blockIdx.x * PACK_TILES;
    // this block's 64 tiles: requested now, used after the base is known
    uint32_t num = 0u;
    const int tile = t0 + lane;
    const bool have = wave == 0 && tile < a.bandTiles;
    if (have) num = a.tileNum[tile];
    // entries (and class counts) of every tile before t0 (a multiple of 64): the lengths as BYTES, sixteen tiles per 16-byte load, four loads in flight
    // per thread -- at most 32 KB at 4K, L2-resident (v_sad_u8 adds the four bytes of a word to the accumulator)
    uint32_t acc = 0u;
    {
        const uint4* __restrict__ v4 = reinterpret_cast<const uint4*>(a.tileNum8);
        const int count4 = t0 / 16;
        for (int i = threadIdx.x; i < count4; i += 1024) {
            uint4 q[4];
#pragma unroll
            for (int k = 0; k < 4; k++) q[k] = i + 256 * k < count4 ? v4[i + 256 * k] : make_uint4(0u, 0u, 0u, 0u);
#pragma unroll
            for (int k = 0; k < 4; k++) {
                const uint32_t w[4] = { q[k].x, q[k].y, q[k].z, q[k].w };
#pragma unroll
                for (int c = 0; c < 4; c++) {
                    acc = __builtin_amdgcn_sad_u8(w[c], 0u, acc);
                }
            }
        }
    }
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) acc += (uint32_t)__shfl_xor((int)acc, d, 64);
    if (lane == 0) sPart[wave] = acc;
    __syncthreads();
    const uint32_t baseSum = (sPart[0] + sPart[1]) + (sPart[2] + sPart[3]);
    if (wave == 0) {
        // the 64 tiles: offset = 1 + entries of all earlier tiles
        uint32_t tincl = num;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const uint32_t u = (uint32_t)__shfl_up((int)tincl, d);
            if (lane >= d) tincl += u;
        }
        sPre[lane] = tincl - num;
        sNum[lane] = num;
        const uint32_t offset = 1u + baseSum + tincl - num;
        if (have) {
            // A list that does not fit the caller's buffer is cut, and the grid says so: the shade never reads past `capacity`.
            const uint32_t fit = offset >= a.capacity ? 0u : min(num, a.capacity - offset);
            a.grid[tile].offset = offset;
            a.grid[tile].num = fit;
        }
        const bool last = t0 + PACK_TILES >= a.bandTiles;
        if (last && lane == 63) { // Appendix A step 6: indices[0] = sum of num
            const uint32_t tot = baseSum + tincl;
            a.culled[0] = a.capacity ? min(tot, a.capacity - 1u) : 0u;
        }
    }
    __syncthreads();
    // Gather: four threads per tile, 32 entries (eight 16-byte loads, all in flight) each, into the tile's place in the block's run.  (Reading a
    // whole uint4 whose first entry is inside the list stays inside the tile's 128-entry slot; what lies behind the list is never stored.)
    {
        const int j = threadIdx.x >> 2, q = threadIdx.x & 3;
        const uint32_t n = sNum[j], pre = sPre[j];
        const uint4* __restrict__ src = reinterpret_cast<const uint4*>(a.tileLists + (size_t)(t0 + j) * KEEP) + q * 8;
        uint4 v[8];
#pragma unroll
        for (int k = 0; k < 8; k++) if ((uint32_t)(32 * q + 4 * k) < n) v[k] = src[k];
#pragma unroll
        for (int k = 0; k < 8; k++) {
            const uint32_t i = (uint32_t)(32 * q + 4 * k);
            if (i < n) {
                uint32_t* d = sBuf + pre + i;
                d[0] = v[k].x;
                if (i + 1u < n) d[1] = v[k].y;
                if (i + 2u < n) d[2] = v[k].z;
                if (i + 3u < n) d[3] = v[k].w;
            }
        }
    }
    __syncthreads();
    const uint32_t total = sPre[PACK_TILES - 1] + sNum[PACK_TILES - 1];
    for (uint32_t i = threadIdx.x; i < total; i += 256u) {
        const uint32_t d = 1u + baseSum + i;
        if (d < a.capacity) a.culled[d] = sBuf[i];
    }
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

// The BAND FORM of the shade (shade.hip: split blocks for the long tiles in front of the one-block-per-tile grid) is what sailor_hip_light_cull_tile_order
// switches on by handing a band's per-tile list lengths to the shade.  Every band of a split frame gets it -- the split threshold rises with the band's
// size (shade_body.h: SPLIT_MIN_*) -- but a band of more than BAND_FORM_MAX_TILES tiles under a LARGE light set (from 131 072 lights on: an eighth of the
// 8K frame under a million lights, 16 320 tiles): such a set means short lists and next to no long tiles (225 of 16 320 there), so there is no tail to cut,
// and the band takes the whole frame's form -- one block per tile on the XCD-aware grid -- whose launch pipelines better beside the next frame's cull
// (step 116.5 against 126.3 us, same box; the 4K frame's halves under 65 536 lights: 103.2 / 95.7 us in the band form, 108.0 / 91.9 in the other).
// SAILOR_BAND_FORM_TILES=<n> overrides the limit (A / B; 0: never the band form).
static int band_form_max_tiles()
{
    static const int v = [] { const char* e = getenv("SAILOR_BAND_FORM_TILES"); return e ? atoi(e) : -1; }();
    return v;
}
static bool layout_has_hint(const CullLayout& L, const int lightsCapacity)
{
    if (L.bandRows >= L.Ty || L.bandTiles <= 0) return false;
    const int limit = band_form_max_tiles();
    if (limit >= 0) return L.bandTiles <= limit;
    return L.bandTiles <= BAND_FORM_MAX_TILES || lightsCapacity < 131072;
}

static int launch_pack(SailorHipContext* ctx, const CullLayout& L, char* ws, SailorLightsGrid* dLightsGrid, uint32_t* dCulledLights, size_t culledCapacity)
{
    PackArgs ka;
    ka.tileNum = (const uint32_t*)(ws + L.offTileNum); ka.tileNum8 = (const uint8_t*)(ws + L.offTileNum8); ka.tileLists = (const uint32_t*)(ws + L.offTileLists);
    ka.grid = dLightsGrid; ka.culled = dCulledLights;
    ka.Tx = L.Tx; ka.bandRows = L.bandRows; ka.bandTiles = L.bandTiles;
    ka.capacity = (uint32_t)(culledCapacity > 0xFFFFFFFFull ? 0xFFFFFFFFull : culledCapacity);
    sailor_launch(ctx, k1_pack, dim3(L.packBlocks), dim3(256), ka);
    SAILOR_CHECK_LAUNCH(ctx, "k1_pack");
    return SAILOR_HIP_OK;
}

extern "C" {

int sailor_hip_band_is_valid(int32_t width, int32_t height, const SailorBand* band)
{
    return (width > 0 && height > 0 && band_valid(width, height, band)) ? 1 : 0;
}

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
    return sailor_hip_light_cull_prepared(ctx, frame, pc, dLights, dLinearDepth, dLightsGrid, dCulledLights, culledCapacity, dWorkspace, workspaceBytes, band, flags,
                                          nullptr, 0);
}

int sailor_hip_light_cull_prepared(SailorHipContext* ctx, const SailorUboFrameData* frame, const SailorLightCullPushConstants* pc,
                                   const SailorLightShaderData* dLights, const float* dLinearDepth,
                                   SailorLightsGrid* dLightsGrid, uint32_t* dCulledLights, size_t culledCapacity,
                                   void* dWorkspace, size_t workspaceBytes, const SailorBand* band, uint32_t flags,
                                   const void* dPreparedLights, int32_t preparedCapacity)
{
    if (!ctx || !frame || !pc || !dLinearDepth || !dLightsGrid || !dCulledLights || !dWorkspace) return SAILOR_HIP_ERR_INVALID_ARGUMENT;
    const int W = pc->viewportSize[0], H = pc->viewportSize[1], N = pc->lightsNum;
    if (W <= 0 || H <= 0 || N < 0 || (N > 0 && !dLights && !dPreparedLights)) return SAILOR_HIP_ERR_INVALID_ARGUMENT;
    if (dPreparedLights && (preparedCapacity < N || ((uintptr_t)dPreparedLights & 15))) return SAILOR_HIP_ERR_INVALID_ARGUMENT;
    if ((flags & SAILOR_CULL_PREPARE_LIGHTS) && (!dPreparedLights || (N > 0 && (!dLights || ((uintptr_t)dLights & 15))))) return SAILOR_HIP_ERR_INVALID_ARGUMENT;
    if (N > 0x3FFFFFFF) return SAILOR_HIP_ERR_UNSUPPORTED; // bit 31 of a candidate entry carries the "directional" flag
    // Appendix A: the depth extent (push constants) and the window viewport (frame UBO) must agree
    if (frame->viewportSize[0] != W || frame->viewportSize[1] != H) return SAILOR_HIP_ERR_UNSUPPORTED;
    SailorBand whole;
    if (!band) { sailor_hip_band_whole_frame(W, H, &whole); band = &whole; }
    if (!band_valid(W, H, band)) return SAILOR_HIP_ERR_INVALID_ARGUMENT;
    const CullLayout L = make_layout(W, H, N, *band);
    if (pc->numTiles[0] != L.Tx || pc->numTiles[1] != L.Ty) return SAILOR_HIP_ERR_INVALID_ARGUMENT;
    if (workspaceBytes < L.total) return SAILOR_HIP_ERR_WORKSPACE_TOO_SMALL;
    // The reference allocates numTiles * 128 uints (LightCullingNode.cpp:64: one short of the worst case); anything smaller than
    // that cannot hold a band of full lists and is refused up front.  A buffer of exactly that size is accepted: a list that does
    // not fit is cut and lightsGrid[tile].num / culledLights[0] say what was written.
    if (culledCapacity < 1 || culledCapacity < (size_t)L.bandTiles * KEEP) return SAILOR_HIP_ERR_INVALID_ARGUMENT;
    if ((unsigned long long)L.bandTiles * KEEP > 0x7FFFFFFFull) return SAILOR_HIP_ERR_UNSUPPORTED;
    if (((uintptr_t)dWorkspace & 255) != 0) return SAILOR_HIP_ERR_INVALID_ARGUMENT;
    SAILOR_TRY_HIP(ctx, hipSetDevice(ctx->device));

    hipStream_t s = ctx->stream;
    char* ws = (char*)dWorkspace;

    if (L.bandTiles == 0) {
        SAILOR_TRY_HIP(ctx, hipMemsetAsync(dCulledLights, 0, 4, s));
        return SAILOR_HIP_OK;
    }

    // The side-plane margin argument needs a sane perspective; tiny light counts are cheaper brute force.
    const float p00 = fabsf(frame->projection[0]), p11 = fabsf(frame->projection[5]);
    const bool sane = p00 > 1e-2f && p11 > 1e-2f && p00 < 1e4f && p11 < 1e4f;
    const bool brute = (flags & SAILOR_CULL_BRUTE_FORCE) || !sane || N < 512;

    PrepareArgs pa;
    memcpy(pa.view.m, frame->view, 64);
    memcpy(pa.invProj.m, frame->invProjection, 64);
    pa.lights = dLights; pa.depth = dLinearDepth;
    pa.soaPosRadius = nullptr; pa.soaType = nullptr;
    pa.prepPosRadius = nullptr; pa.prepType = nullptr; pa.prepStaged = nullptr;
    if (dPreparedLights) {
        const void *pr = nullptr, *ty = nullptr, *st = nullptr;
        sailor_hip_prepared_lights_views(preparedCapacity, dPreparedLights, &pr, &ty, &st);
        if (flags & SAILOR_CULL_PREPARE_LIGHTS) { // the views are this call's OUTPUT (for the shade and for later culls); the light role reads the records
            pa.prepPosRadius = (float4*)pr; pa.prepType = (uint32_t*)ty; pa.prepStaged = (float4*)st;
        } else { pa.soaPosRadius = (const float4*)pr; pa.soaType = (const uint32_t*)ty; }
    }
    pa.lightView = (float4*)(ws + L.offLightView); pa.lightType = (uint32_t*)(ws + L.offLightType); pa.tileInfo = (float4*)(ws + L.offTileInfo);
    // A band of a split frame with a large light set: the lights that can reach the band are selected first (k0_band_count + k0_band_scatter) and the chain runs on them.
    // From 131 072 lights on (below, the light role and the group lists of a band sit at their launch floors whatever the count); SAILOR_CULL_BAND_SELECT
    // forces it for any set the pre-filter runs on, SAILOR_CULL_NO_BAND_SELECT switches it off.  (Two launches, no block waits for another: any
    // number of blocks.)
    uint32_t* selState = (uint32_t*)(ws + L.offSelState);
    const bool select = !brute && L.bandRows < L.Ty && !(flags & SAILOR_CULL_NO_BAND_SELECT) && ((flags & SAILOR_CULL_BAND_SELECT) || N >= 131072);
    pa.selCount = nullptr;
    if (select) {
        SelectArgs sa;
        memcpy(sa.view.m, frame->view, 64);
        memcpy(sa.invProj.m, frame->invProjection, 64);
        sa.lights = dLights; sa.soaPosRadius = pa.soaPosRadius; sa.soaType = pa.soaType;
        sa.prepPosRadius = pa.prepPosRadius; sa.prepType = pa.prepType; sa.prepStaged = pa.prepStaged;
        sa.lightView = pa.lightView; sa.lightType = pa.lightType; sa.lightMap = (uint32_t*)(ws + L.offLightMap); sa.state = selState;
        sa.keep = (unsigned long long*)(ws + L.offSelKeep);
        sa.N = N; sa.vpW = frame->viewportSize[0]; sa.vpH = frame->viewportSize[1]; sa.Tx = L.Tx; sa.tileRow0 = band->tileRowBegin; sa.bandRows = L.bandRows;
        sa.planeMargin = 1e-3f;
        sa.stageSelectedOnly = (flags & SAILOR_CULL_PREPARE_SELECTED) ? 1 : 0;
        sailor_launch(ctx, k0_band_count, dim3((unsigned)L.selBlocks), dim3(256), sa);
        SAILOR_CHECK_LAUNCH(ctx, "k0_band_count");
        sailor_launch(ctx, k0_band_scatter, dim3((unsigned)L.selBlocks), dim3(256), sa);
        SAILOR_CHECK_LAUNCH(ctx, "k0_band_scatter");
        pa.selCount = selState;
        pa.prepPosRadius = nullptr; pa.prepType = nullptr; pa.prepStaged = nullptr; // (derived by k0_band_count; the light role reads the compact records)
    }
    pa.masks = (unsigned long long*)(ws + L.offMasks); pa.dirWords = (unsigned long long*)(ws + L.offDirWords);
    pa.N = N; pa.words = L.words;
    pa.lightBlocks = (N + 255) / 256;
    pa.numBands = brute ? 0 : L.numBands;
    // Few lights: the bands are spread over several blocks per 256 lights (parallelism; every block repeats the cheap transform).  Many lights
    // (the 1 M of configs[4]): the light blocks fill the chip by themselves, and repeating the 112-byte record reads per split is what costs.
    const int perBlock = (pa.lightBlocks >= 1024 || (flags & SAILOR_CULL_INTERVAL_MASKS)) ? MAX_BANDS_PER_BLOCK : BANDS_PER_BLOCK;
    const int splits = brute ? 1 : (L.numBands + perBlock - 1) / perBlock;
    pa.bandsPerBlock = brute ? 0 : (L.numBands + splits - 1) / splits; // the bands spread evenly over the splits
    pa.lightRoleBlocks = pa.lightBlocks * splits;
    // the masks by bisection for each light's band intervals instead of 2 x numBands plane tests: needs all bands in one block
    pa.intervals = (!brute && splits == 1 && (pa.lightBlocks >= 1024 || (flags & SAILOR_CULL_INTERVAL_MASKS))) ? 1 : 0;
    pa.stripsPerRow = (L.Tx + 16 * SETUP_STRIPS - 1) / (16 * SETUP_STRIPS);
    pa.setupBlocks = pa.stripsPerRow * L.bandRows;
    pa.frustumBlocks = (L.bandTiles + 255) / 256;
    pa.vpW = frame->viewportSize[0]; pa.vpH = frame->viewportSize[1]; pa.W = W; pa.H = H; pa.Tx = L.Tx; pa.Ty = L.Ty;
    pa.tileRow0 = band->tileRowBegin; pa.bandRow0 = band->fbRowBegin; pa.bandRows = L.bandRows; pa.groupsX = L.groupsX;
    pa.vecOK = (((uintptr_t)dLinearDepth & 15) == 0 && (W & 3) == 0) ? 1 : 0;
    pa.rawDepth = (flags & SAILOR_CULL_RAW_DEPTH) ? 1 : 0;
    pa.zNearCam = frame->cameraZNearZFar[0];
    pa.planeMargin = 1e-3f;
    pa.dirFlag = (uint32_t*)(ws + L.offDirFlag);
    pa.heavy = (uint32_t*)(ws + L.offHeavy);
    sailor_launch(ctx, k01_prepare, dim3(pa.lightRoleBlocks + pa.frustumBlocks + pa.setupBlocks), dim3(256), pa);
    SAILOR_CHECK_LAUNCH(ctx, "k01_prepare");

    CullArgs ca;
    ca.lightView = pa.lightView; ca.lightType = pa.lightType; ca.tileInfo = pa.tileInfo; ca.masks = pa.masks;
    ca.groupCount = (const uint32_t*)(ws + L.offGroupCount); ca.groupList = (const uint32_t*)(ws + L.offGroupList);
    ca.tileNum = (uint32_t*)(ws + L.offTileNum); ca.tileNum8 = (uint8_t*)(ws + L.offTileNum8); ca.tileLists = (uint32_t*)(ws + L.offTileLists);
    ca.dirFlag = pa.dirFlag;
    ca.N = N; ca.words = L.words; ca.Tx = L.Tx; ca.groupsX = L.groupsX; ca.bandRows = L.bandRows;
    ca.heavy = (const uint32_t*)(ws + L.offHeavy); ca.headRows = 0;
    ca.lightMap = select ? (const uint32_t*)(ws + L.offLightMap) : nullptr; ca.selCount = pa.selCount;
    if (brute) {
        sailor_launch(ctx, k1_tile_cull_brute, dim3(L.groupsX, L.bandRows), dim3(256), ca);
        SAILOR_CHECK_LAUNCH(ctx, "k1_tile_cull<brute>");
    } else {
        const bool wide = L.words >= 4096 && (L.words & 1) == 0;
        if (wide)
        {
            // (round 6: a 4 x 4 patch of groups per block -- every mask row crosses into the block once for sixteen groups.  SAILOR_CULL_WIDE16=1; OFF by default:
            // on C5 it halves the kernel's traffic, 370 -> 194 MB, leaves its duration where it was, 84.1 -> 85.3 us -- the kernel is its waves' 128-step chains,
            // not bandwidth -- and its 1 024-thread, 56 KB blocks find room beside the previous frame's shade later than 256-thread ones do: the serial step
            // 762 -> 756 us, the frame pipeline's step 746 -> 784 us, same box, two runs each: profiles/r06/README.md section 3)
            static const bool wide16 = [] { const char* e = getenv("SAILOR_CULL_WIDE16"); return e && atoi(e) != 0; }();
            const dim3 wideGrid((unsigned)(((L.groupsX + 3) / 4) * L.groupsY));
            const dim3 wide16Grid((unsigned)(((L.groupsX + 3) / 4) * ((L.groupsY + 3) / 4)));
            if (wide16 && L.words % (128 * GLW16_ROWS) == 0 && !select)
                sailor_launch(ctx, k1_group_lists_wide16<true>, wide16Grid, dim3(1024), pa.masks, pa.dirWords, L.words, L.groupsX, L.groupsY, (uint32_t*)(ws + L.offGroupCount),
                                   (uint32_t*)(ws + L.offGroupList), (const uint32_t*)(ws + L.offDirFlag), (const uint32_t*)nullptr);
            else if (wide16)
                sailor_launch(ctx, k1_group_lists_wide16<false>, wide16Grid, dim3(1024), pa.masks, pa.dirWords, L.words, L.groupsX, L.groupsY, (uint32_t*)(ws + L.offGroupCount),
                                   (uint32_t*)(ws + L.offGroupList), (const uint32_t*)(ws + L.offDirFlag), pa.selCount);
            else if (L.words % (128 * GLW_ROWS) == 0 && !select)
                sailor_launch(ctx, k1_group_lists_wide<true>, wideGrid, dim3(256), pa.masks, pa.dirWords, L.words, L.groupsX, (uint32_t*)(ws + L.offGroupCount),
                                   (uint32_t*)(ws + L.offGroupList), (const uint32_t*)(ws + L.offDirFlag), (const uint32_t*)nullptr);
            else
                sailor_launch(ctx, k1_group_lists_wide<false>, wideGrid, dim3(256), pa.masks, pa.dirWords, L.words, L.groupsX, (uint32_t*)(ws + L.offGroupCount),
                                   (uint32_t*)(ws + L.offGroupList), (const uint32_t*)(ws + L.offDirFlag), pa.selCount);
        }
        else {
            sailor_launch(ctx, k1_group_lists, dim3(L.groupsX, L.groupsY), dim3(256), pa.masks, pa.dirWords, L.words, L.groupsX, (uint32_t*)(ws + L.offGroupCount),
                               (uint32_t*)(ws + L.offGroupList), (uint32_t*)(ws + L.offHeavy), (uint32_t)(L.bandRows * 2 > L.Ty ? HEAVY_MIN_FRAME : HEAVY_MIN_BAND), pa.selCount);
            ca.headRows = (16 * HEAVY_MAX + L.groupsX - 1) / L.groupsX; // grid rows for the listed clusters' tiles (a block each), in front of the tile rows
        }
        SAILOR_CHECK_LAUNCH(ctx, wide ? "k1_group_lists_wide" : "k1_group_lists");
        // (the block-wide selection everywhere but on the long lists of the wide path: see k1_tile_cull)
        const bool coop = !wide;
        const dim3 cgrid(L.groupsX, ca.headRows + L.bandRows);
        // (COOP: the block-wide selection; SEL: behind the band selection -- compact indices, translated when a list leaves)
        if (coop) { if (select) sailor_launch(ctx, k1_tile_cull<true, true>, cgrid, dim3(256), ca); else sailor_launch(ctx, k1_tile_cull<true, false>, cgrid, dim3(256), ca); }
        else { if (select) sailor_launch(ctx, k1_tile_cull<false, true>, cgrid, dim3(256), ca); else sailor_launch(ctx, k1_tile_cull<false, false>, cgrid, dim3(256), ca); }
        SAILOR_CHECK_LAUNCH(ctx, "k1_tile_cull");
    }
    if (flags & SAILOR_CULL_DEFER_PACK) return SAILOR_HIP_OK; // the caller records sailor_hip_light_cull_pack where it wants it (another stream, beside the shade)
    return launch_pack(ctx, L, ws, dLightsGrid, dCulledLights, culledCapacity);
}

int sailor_hip_light_cull_pack(SailorHipContext* ctx, int32_t width, int32_t height, int32_t lightsCapacity, const SailorBand* band, const void* dWorkspace,
                               SailorLightsGrid* dLightsGrid, uint32_t* dCulledLights, size_t culledCapacity)
{
    if (!ctx || !dWorkspace || !dLightsGrid || !dCulledLights || width <= 0 || height <= 0 || lightsCapacity < 0) return SAILOR_HIP_ERR_INVALID_ARGUMENT;
    SailorBand whole;
    if (!band) { sailor_hip_band_whole_frame(width, height, &whole); band = &whole; }
    if (!band_valid(width, height, band)) return SAILOR_HIP_ERR_INVALID_ARGUMENT;
    // (everything the pack reads lies in the part of the workspace whose layout depends on the geometry alone)
    const CullLayout L = make_layout(width, height, lightsCapacity, *band);
    if (culledCapacity < 1 || culledCapacity < (size_t)L.bandTiles * KEEP) return SAILOR_HIP_ERR_INVALID_ARGUMENT;
    if (((uintptr_t)dWorkspace & 255) != 0) return SAILOR_HIP_ERR_INVALID_ARGUMENT;
    SAILOR_TRY_HIP(ctx, hipSetDevice(ctx->device));
    if (L.bandTiles == 0) {
        SAILOR_TRY_HIP(ctx, hipMemsetAsync(dCulledLights, 0, 4, ctx->stream));
        return SAILOR_HIP_OK;
    }
    return launch_pack(ctx, L, (char*)dWorkspace, dLightsGrid, dCulledLights, culledCapacity);
}

int sailor_hip_light_cull_tile_lists(int32_t width, int32_t height, int32_t lightsCapacity, const SailorBand* band, const void* dWorkspace,
                                     const uint32_t** outTileNum, const uint32_t** outTileLists)
{
    if (!dWorkspace || width <= 0 || height <= 0 || lightsCapacity < 0) return SAILOR_HIP_ERR_INVALID_ARGUMENT;
    SailorBand whole;
    if (!band) { sailor_hip_band_whole_frame(width, height, &whole); band = &whole; }
    if (!band_valid(width, height, band)) return SAILOR_HIP_ERR_INVALID_ARGUMENT;
    const CullLayout L = make_layout(width, height, lightsCapacity, *band);
    if (outTileNum) *outTileNum = (const uint32_t*)((const char*)dWorkspace + L.offTileNum);
    if (outTileLists) *outTileLists = (const uint32_t*)((const char*)dWorkspace + L.offTileLists);
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
    SAILOR_TRY_HIP(ctx, hipSetDevice(ctx->device));
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
    const CullLayout L = make_layout(width, height, lightsCapacity, *band);
    if (!layout_has_hint(L, lightsCapacity)) return nullptr; // whole frame: the one-block-per-tile form in raster order
    // (the band's list lengths as bytes, one per tile in tile order -- every cull writes them: the band shade's split blocks find the long tiles there)
    return (const uint32_t*)((const char*)dWorkspace + L.offTileNum8);
}

// The band's own light set as the last cull of this geometry AND light count left it (k0_band_count + k0_band_scatter): the number of selected lights and
// lightMap (compact index -> light index, ascending).  Both lie in the part of the workspace whose place depends on the light count, hence
// lightsNum = that cull's pc->lightsNum.  Meaningful only if that cull ran the selection (sailor_hip_context_launch_log names its kernels).
int sailor_hip_light_cull_band_selection(int32_t width, int32_t height, int32_t lightsNum, const SailorBand* band, const void* dWorkspace,
                                         const uint32_t** outSelectedCount, const uint32_t** outLightMap)
{
    if (!dWorkspace || width <= 0 || height <= 0 || lightsNum < 0) return SAILOR_HIP_ERR_INVALID_ARGUMENT;
    SailorBand whole;
    if (!band) { sailor_hip_band_whole_frame(width, height, &whole); band = &whole; }
    if (!band_valid(width, height, band)) return SAILOR_HIP_ERR_INVALID_ARGUMENT;
    const CullLayout L = make_layout(width, height, lightsNum, *band);
    if (outSelectedCount) *outSelectedCount = (const uint32_t*)((const char*)dWorkspace + L.offSelState);
    if (outLightMap) *outLightMap = (const uint32_t*)((const char*)dWorkspace + L.offLightMap);
    return SAILOR_HIP_OK;
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
        if (c >= GROUP_OVERFLOW_LISTED) { over++; continue; }
        c &= ~GROUP_LISTED; // (a count may carry k1_group_lists' "listed cluster" flag)
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
    SAILOR_TRY_HIP(ctx, hipSetDevice(ctx->device));
    hipLaunchKernelGGL(k_grid_rebase, dim3((numTiles + 255) / 256), dim3(256), 0, ctx->stream, dLightsGrid, numTiles, globalBase);
    SAILOR_CHECK_LAUNCH(ctx, "k_grid_rebase");
    return SAILOR_HIP_OK;
}

} // extern "C"
