// K4 -- ECS transform + bounds + frustum-cull sweep, and the sphere-vs-view-frustum instance cull, for gfx950.
//
// Replaces:
//   TransformECS::Tick full-sweep branch + CalculateMatrices   ECS/TransformECS.cpp:144-212
//   Transform::Matrix                                          Math/Transform.cpp:39-42
//   AABB::Apply                                                Math/Bounds.cpp:479-492 (called at ECS/StaticMeshRendererECS.cpp:56)
//   Frustum::OverlapsAABB                                      Math/Bounds.cpp:245-260 (called through RHI/SceneView.cpp:56)
//   FrustumCulling                                             Content/Shaders/ComputeMeshCulling.shader:96-110
//
// The reference walks a pointer-linked hierarchy recursively on one CPU thread and fans the bounds update out in
// 1024-entity tasks.  Here the entities are flat level-sorted SoA records and one launch per hierarchy level does
// the whole per-entity chain (TRS -> relative -> world -> 8-corner AABB -> 6-plane test) in registers, so every
// input byte is read once and every output byte written once (164.125 B/entity, HBM-bound).  Visibility leaves as
// one wave ballot = one 64-bit word per 64 entities.
//
// Arithmetic follows glm's own evaluation order (mat4*mat4 left-to-right column sums, mat4*vec4 as
// (c0 x + c1 y) + (c2 z + c3 w)) with no FMA contraction, so results match the CPU restatement bit for bit,
// including the reference's FLT_MIN seed for AABB::Apply's max (Bounds.cpp:484).
#include "common.h"
#include <float.h>

// glm operator*(mat4, mat4): Result[c] = A0*B[c][0] + A1*B[c][1] + A2*B[c][2] + A3*B[c][3]
__device__ __forceinline__ void mat_mul(const float* a, const float* b, float* o)
{
#pragma unroll
    for (int c = 0; c < 4; c++)
#pragma unroll
        for (int i = 0; i < 4; i++)
            o[c * 4 + i] = ((a[0 + i] * b[c * 4 + 0] + a[4 + i] * b[c * 4 + 1]) + a[8 + i] * b[c * 4 + 2]) + a[12 + i] * b[c * 4 + 3];
}

// Math/Transform.cpp:41: glm::translate(mat4(1), pos) * glm::toMat4(rot) * glm::scale(mat4(1), scale), literally
__device__ __forceinline__ void transform_matrix(const float4 pos, const float4 q, const float4 sc, float* out)
{
    float T[16] = { 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1 };
    T[12] = ((1.0f * pos.x + 0.0f * pos.y) + 0.0f * pos.z) + 0.0f;
    T[13] = ((0.0f * pos.x + 1.0f * pos.y) + 0.0f * pos.z) + 0.0f;
    T[14] = ((0.0f * pos.x + 0.0f * pos.y) + 1.0f * pos.z) + 0.0f;
    T[15] = ((0.0f * pos.x + 0.0f * pos.y) + 0.0f * pos.z) + 1.0f;
    const float qxx = q.x * q.x, qyy = q.y * q.y, qzz = q.z * q.z, qxz = q.x * q.z, qxy = q.x * q.y, qyz = q.y * q.z;
    const float qwx = q.w * q.x, qwy = q.w * q.y, qwz = q.w * q.z;
    float R[16];
    R[0] = 1.0f - 2.0f * (qyy + qzz); R[1] = 2.0f * (qxy + qwz); R[2] = 2.0f * (qxz - qwy); R[3] = 0.0f;
    R[4] = 2.0f * (qxy - qwz); R[5] = 1.0f - 2.0f * (qxx + qzz); R[6] = 2.0f * (qyz + qwx); R[7] = 0.0f;
    R[8] = 2.0f * (qxz + qwy); R[9] = 2.0f * (qyz - qwx); R[10] = 1.0f - 2.0f * (qxx + qyy); R[11] = 0.0f;
    R[12] = 0.0f; R[13] = 0.0f; R[14] = 0.0f; R[15] = 1.0f;
    float S[16];
#pragma unroll
    for (int i = 0; i < 4; i++) {
        S[0 + i] = (i == 0 ? 1.0f : 0.0f) * sc.x;
        S[4 + i] = (i == 1 ? 1.0f : 0.0f) * sc.y;
        S[8 + i] = (i == 2 ? 1.0f : 0.0f) * sc.z;
        S[12 + i] = (i == 3 ? 1.0f : 0.0f);
    }
    float TR[16];
    mat_mul(T, R, TR);
    mat_mul(TR, S, out);
}

struct Planes6 { float p[24]; };

// AABB::Apply (Bounds.cpp:479-492, corner order of Bounds.h:119-130) + Frustum::OverlapsAABB (Bounds.cpp:245-260)
__device__ __forceinline__ bool ecs_bounds_vis(const float* W, const float* la, const Planes6& planes, float* omin, float* omax)
{
    const float mnx = la[0], mny = la[1], mnz = la[2], mxx = la[3], mxy = la[4], mxz = la[5];
    const float px[8] = { mnx, mxx, mnx, mxx, mxx, mxx, mnx, mnx };
    const float py[8] = { mny, mxy, mxy, mny, mxy, mny, mxy, mny };
    const float pz[8] = { mnz, mxz, mxz, mxz, mnz, mnz, mnz, mxz };
    omin[0] = omin[1] = omin[2] = FLT_MAX;
    omax[0] = omax[1] = omax[2] = FLT_MIN;
#pragma unroll
    for (int k = 0; k < 8; k++) {
#pragma unroll
        for (int c = 0; c < 3; c++) {
            const float t = (W[0 + c] * px[k] + W[4 + c] * py[k]) + (W[8 + c] * pz[k] + W[12 + c] * 1.0f);
            omin[c] = (omin[c] < t) ? omin[c] : t;
            omax[c] = (t < omax[c]) ? omax[c] : t;
        }
    }
    bool vis = true;
#pragma unroll
    for (int p = 0; p < 6; p++) {
        const float ax = omin[0] * planes.p[4 * p + 0], bx = omax[0] * planes.p[4 * p + 0];
        const float ay = omin[1] * planes.p[4 * p + 1], by = omax[1] * planes.p[4 * p + 1];
        const float az = omin[2] * planes.p[4 * p + 2], bz = omax[2] * planes.p[4 * p + 2];
        const float d = (ax < bx ? bx : ax) + (ay < by ? by : ay) + (az < bz ? bz : az) + planes.p[4 * p + 3];
        vis = vis && (d > 0.0f);
    }
    return vis;
}

// world = parentWorld * relative (TransformECS.cpp:201) or relative for a root (:192-195)
__device__ __forceinline__ void ecs_world(const float4 pos, const float4 rot, const float4 scl, bool hasParent, const float4* __restrict__ parentWorld, float* W)
{
    float rel[16];
    transform_matrix(pos, rot, scl, rel);
    if (!hasParent) {
#pragma unroll
        for (int k = 0; k < 16; k++) W[k] = rel[k];
    } else {
        const float4 c0 = parentWorld[0], c1 = parentWorld[1], c2 = parentWorld[2], c3 = parentWorld[3];
        const float P[16] = { c0.x, c0.y, c0.z, c0.w, c1.x, c1.y, c1.z, c1.w, c2.x, c2.y, c2.z, c2.w, c3.x, c3.y, c3.z, c3.w };
        mat_mul(P, rel, W);
    }
}

// FUSED levels: the world matrix of an entity of level d is the left-to-right product rel(root) * rel(..) * rel(self), every
// factor recomputed from that ancestor's 48-byte TRS exactly as its own lane computes it -- so the bits are those of the
// level-by-level sweep, but no lane waits for another launch.  Up to ECS_FUSED_LEVELS levels (the hierarchy of the 1 M-entity
// workload has 3): one launch instead of one per level; the short dependent launches of the child levels ran at a third of the
// root level's bandwidth.
#define ECS_FUSED_LEVELS 4
__device__ __forceinline__ void ecs_world_fused(const float4* __restrict__ trs, const uint32_t* __restrict__ parent, const float4 pos, const float4 rot,
                                                const float4 scl, uint32_t par, float* W)
{
    uint32_t anc[ECS_FUSED_LEVELS - 1];
    int depth = 0;
#pragma unroll
    for (int k = 0; k < ECS_FUSED_LEVELS - 1; k++) {
        anc[k] = par;
        if (par != 0xFFFFFFFFu) { depth = k + 1; par = parent[par]; }
    }
    float rel[16];
    bool started = false;
#pragma unroll
    for (int k = ECS_FUSED_LEVELS - 2; k >= 0; k--) { // from the root down
        if (k < depth) {
            const float4* t = trs + (size_t)anc[k] * 3;
            transform_matrix(t[0], t[1], t[2], rel);
            if (!started) {
#pragma unroll
                for (int q = 0; q < 16; q++) W[q] = rel[q];
                started = true;
            } else {
                float T[16];
                mat_mul(W, rel, T);
#pragma unroll
                for (int q = 0; q < 16; q++) W[q] = T[q];
            }
        }
    }
    transform_matrix(pos, rot, scl, rel);
    if (!started) {
#pragma unroll
        for (int q = 0; q < 16; q++) W[q] = rel[q];
    } else {
        float T[16];
        mat_mul(W, rel, T);
#pragma unroll
        for (int q = 0; q < 16; q++) W[q] = T[q];
    }
}

#define ECS_WAVE_F4 352 // float4 slots of LDS per wave: max(TRS 192 + box 96, world 256 + box 96)

template <bool FUSED>
__global__ __launch_bounds__(256) void k4_ecs_level(uint32_t lo, uint32_t hi, const float4* __restrict__ trs, const uint32_t* __restrict__ parent,
                                                     const float* __restrict__ localAabb, Planes6 planes, int boxVec,
                                                     float4* __restrict__ world, float* __restrict__ worldAabb, unsigned long long* __restrict__ visibility)
{
    // One wave owns one 64-entity visibility word.  The records are AoS (48 B TRS, 24 B box in; 64 B matrix, 24 B box out), so a
    // lane-per-entity access is a 16-byte (or 4-byte) request every 48 / 64 / 24 bytes -- four to six partial-line requests
    // per line.  A wave whose 64 entities all belong to this level therefore moves its records through LDS: every global
    // request is a full 1 KiB of consecutive float4s, the stride is paid in LDS.  Words that straddle a level boundary
    // (at most two per launch) and unaligned box arrays take the direct path.
    __shared__ float4 sStage[4][ECS_WAVE_F4];
    const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    // fused: deepest level first -- its waves are latency-bound (dependent ancestor gathers), the roots behind them are
    // bandwidth-bound, so the two overlap instead of the slow waves forming the tail of the launch
    const uint32_t blk = FUSED ? (gridDim.x - 1u - blockIdx.x) : blockIdx.x;
    const uint32_t word = (lo >> 6) + blk * 4 + wave;
    const uint32_t base = word * 64, i = base + lane;
    const bool active = i >= lo && i < hi;
    const unsigned long long amask = __ballot(active);
    bool vis = false;
    float W[16], omin[3], omax[3];
    if (amask == ~0ull && boxVec) {
        float4* S = sStage[wave];
        const float4* gT = trs + (size_t)base * 3;
        const float4* gB = reinterpret_cast<const float4*>(localAabb + (size_t)base * 6);
        const uint32_t par = parent[i];
        S[lane] = gT[lane]; S[64 + lane] = gT[64 + lane]; S[128 + lane] = gT[128 + lane];
        S[192 + lane] = gB[lane];
        if (lane < 32) S[256 + lane] = gB[64 + lane];
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        const float4 pos = S[lane * 3 + 0], rot = S[lane * 3 + 1], scl = S[lane * 3 + 2];
        const float* sb = reinterpret_cast<const float*>(S + 192) + lane * 6;
        const float la[6] = { sb[0], sb[1], sb[2], sb[3], sb[4], sb[5] };
        if (FUSED) ecs_world_fused(trs, parent, pos, rot, scl, par, W);
        else ecs_world(pos, rot, scl, par != 0xFFFFFFFFu, world + (size_t)(par != 0xFFFFFFFFu ? par : 0u) * 4, W);
        vis = ecs_bounds_vis(W, la, planes, omin, omax);
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); // every lane has read its inputs: the slots are reused for the outputs
        S[lane * 4 + 0] = make_float4(W[0], W[1], W[2], W[3]);
        S[lane * 4 + 1] = make_float4(W[4], W[5], W[6], W[7]);
        S[lane * 4 + 2] = make_float4(W[8], W[9], W[10], W[11]);
        S[lane * 4 + 3] = make_float4(W[12], W[13], W[14], W[15]);
        float* so = reinterpret_cast<float*>(S + 256) + lane * 6;
        so[0] = omin[0]; so[1] = omin[1]; so[2] = omin[2]; so[3] = omax[0]; so[4] = omax[1]; so[5] = omax[2];
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        float4* oW = world + (size_t)base * 4;
        oW[lane] = S[lane]; oW[64 + lane] = S[64 + lane]; oW[128 + lane] = S[128 + lane]; oW[192 + lane] = S[192 + lane];
        float4* oB = reinterpret_cast<float4*>(worldAabb + (size_t)base * 6);
        oB[lane] = S[256 + lane];
        if (lane < 32) oB[64 + lane] = S[320 + lane];
    } else if (active) {
        const float4 pos = trs[(size_t)i * 3 + 0], rot = trs[(size_t)i * 3 + 1], scl = trs[(size_t)i * 3 + 2];
        const uint32_t par = parent[i];
        const float* gl = localAabb + (size_t)i * 6;
        const float la[6] = { gl[0], gl[1], gl[2], gl[3], gl[4], gl[5] };
        if (FUSED) ecs_world_fused(trs, parent, pos, rot, scl, par, W);
        else ecs_world(pos, rot, scl, par != 0xFFFFFFFFu, world + (size_t)(par != 0xFFFFFFFFu ? par : 0u) * 4, W);
        vis = ecs_bounds_vis(W, la, planes, omin, omax);
        float4* ow = world + (size_t)i * 4;
        ow[0] = make_float4(W[0], W[1], W[2], W[3]);
        ow[1] = make_float4(W[4], W[5], W[6], W[7]);
        ow[2] = make_float4(W[8], W[9], W[10], W[11]);
        ow[3] = make_float4(W[12], W[13], W[14], W[15]);
        float* oa = worldAabb + (size_t)i * 6;
        oa[0] = omin[0]; oa[1] = omin[1]; oa[2] = omin[2]; oa[3] = omax[0]; oa[4] = omax[1]; oa[5] = omax[2];
    }
    const unsigned long long vmask = __ballot(vis);
    if (lane == 0 && amask) {
        // a word straddling two levels is completed by two stream-ordered launches: keep the other launch's bits
        const unsigned long long old = (amask == ~0ull) ? 0ull : visibility[word];
        visibility[word] = (old & ~amask) | (vmask & amask);
    }
}

// ---- ComputeMeshCulling.shader:96-110 FrustumCulling -------------------------------------------------------------
__device__ __forceinline__ void msc_screen_to_view(const Mat4& invProj, float sx, float sy, float vpW, float vpH, float* o)
{
    const float tx = sx / vpW, ty = sy / vpH;
    const float4 v = glsl_mul(invProj, tx * 2.0f - 1.0f, ty * 2.0f - 1.0f, -1.0f, 1.0f);
    const float w = v.w;
    o[0] = v.x / w; o[1] = v.y / w; o[2] = (v.z / w) * -1.0f;
}
__device__ __forceinline__ void msc_plane(const float* p1, const float* p2, float* n)
{
    const float cx = p1[1] * p2[2] - p2[1] * p1[2];
    const float cy = p1[2] * p2[0] - p2[2] * p1[0];
    const float cz = p1[0] * p2[1] - p2[0] * p1[1];
    const float len = sqrtf(dot3f(cx, cy, cz, cx, cy, cz));
    n[0] = cx / len; n[1] = cy / len; n[2] = cz / len;
}

// ---- cascade caster sets (ECS/LightingECS.cpp:287-296): which entities does each shadow cascade have to draw? ----------------------
// One lane per entity, its world AABB (K4's output) against the six planes of up to four cascade frusta (Frustum::OverlapsAABB,
// Math/Bounds.cpp:245-260, the same expression as the camera test above); one ballot word per cascade and 64 entities.
struct CascadePlanes { float p[SAILOR_NUM_CSM_CASCADES][24]; };

__global__ __launch_bounds__(256) void k4_csm_caster_masks(uint32_t n, const float* __restrict__ worldAabb, CascadePlanes P, int numCascades,
                                                            unsigned long long* __restrict__ masks, uint32_t words)
{
    __shared__ float sBox[4][384]; // a wave's 64 boxes = 1 536 contiguous bytes: moved as 96 float4s, then read back one box per lane
    const uint32_t i = blockIdx.x * 256 + threadIdx.x;
    const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const uint32_t first = blockIdx.x * 256 + wave * 64;      // the wave's first entity
    float mn[3] = { 0.0f, 0.0f, 0.0f }, mx[3] = { 0.0f, 0.0f, 0.0f };
    if (first + 64 <= n && (((uintptr_t)worldAabb) & 15) == 0) {
        const float4* src = reinterpret_cast<const float4*>(worldAabb + 6 * (size_t)first);
        float4* dst = reinterpret_cast<float4*>(sBox[wave]);
        dst[lane] = src[lane];
        if (lane < 32) dst[64 + lane] = src[64 + lane];
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        const float* b = sBox[wave] + 6 * lane;
        mn[0] = b[0]; mn[1] = b[1]; mn[2] = b[2]; mx[0] = b[3]; mx[1] = b[4]; mx[2] = b[5];
    } else if (i < n) {
        const float* b = worldAabb + 6 * (size_t)i;
        mn[0] = b[0]; mn[1] = b[1]; mn[2] = b[2]; mx[0] = b[3]; mx[1] = b[4]; mx[2] = b[5];
    }
#pragma unroll
    for (int k = 0; k < SAILOR_NUM_CSM_CASCADES; k++) {
        if (k >= numCascades) break;
        bool inside = i < n;
#pragma unroll
        for (int p = 0; p < 6; p++) {
            const float ax = mn[0] * P.p[k][4 * p + 0], bx = mx[0] * P.p[k][4 * p + 0];
            const float ay = mn[1] * P.p[k][4 * p + 1], by = mx[1] * P.p[k][4 * p + 1];
            const float az = mn[2] * P.p[k][4 * p + 2], bz = mx[2] * P.p[k][4 * p + 2];
            const float d = (ax < bx ? bx : ax) + (ay < by ? by : ay) + (az < bz ? bz : az) + P.p[k][4 * p + 3];
            inside = inside && (d > 0.0f);
        }
        const unsigned long long m = __ballot(inside);
        if ((threadIdx.x & 63) == 0 && (i >> 6) < words) masks[(size_t)k * words + (i >> 6)] = m;
    }
}

// ---- Hi-Z pyramid (DepthHighZNode.cpp:74-96, ComputeDepthHighZ.shader) and OcclusionCulling (ComputeMeshCulling.shader:62-94) ----
// The pyramid's sampler has `reduction: Min` (DefaultRenderer.renderer:51-57).  Canonical fetch == oracle/sailor_oracle.c hiz_fetch_min:
// bilinear footprint (x = u W - 0.5, clamp-to-edge), minimum over the texels whose weight is not zero.
struct HiZArgs { const float* pyramid; int width, height, levels; };

__device__ __forceinline__ int hiz_coord(float x, int size)
{
    if (!(x == x)) return 0;
    if (x < -1.0f) x = -1.0f;
    if (x > (float)size) x = (float)size;
    const int i = (int)floorf(x);
    return i < 0 ? 0 : (i > size - 1 ? size - 1 : i);
}

__device__ __forceinline__ float hiz_fetch_min(const float* __restrict__ tex, int W, int H, float u, float v)
{
    const float x = u * (float)W - 0.5f, y = v * (float)H - 0.5f;
    const float fx = x - floorf(x), fy = y - floorf(y);
    const int x0 = hiz_coord(x, W), x1 = hiz_coord(x + 1.0f, W), y0 = hiz_coord(y, H), y1 = hiz_coord(y + 1.0f, H);
    const bool useX1 = !(fx == 0.0f), useY1 = !(fy == 0.0f);
    float m = tex[(size_t)y0 * W + x0];
    if (useX1) { const float t = tex[(size_t)y0 * W + x1]; m = t < m ? t : m; }
    if (useY1) {
        const float t = tex[(size_t)y1 * W + x0]; m = t < m ? t : m;
        if (useX1) { const float t2 = tex[(size_t)y1 * W + x1]; m = t2 < m ? t2 : m; }
    }
    return m;
}

// ComputeDepthHighZ.shader:22-30, one thread per output texel
__global__ __launch_bounds__(256) void k_hiz_downscale(const float* __restrict__ src, int srcW, int srcH, float* __restrict__ dst, int dstW, int dstH)
{
    const int x = blockIdx.x * 32 + (threadIdx.x & 31), y = blockIdx.y * 8 + (threadIdx.x >> 5);
    if (x >= dstW || y >= dstH) return;
    dst[(size_t)y * dstW + x] = hiz_fetch_min(src, srcW, srcH, ((float)x + 0.5f) / (float)dstW, ((float)y + 0.5f) / (float)dstH);
}

// The tail of the pyramid in ONE launch: from a level of at most 64 x 64 texels down to 1 x 1, one 256-thread block, a barrier between levels
// (the levels are a dependent chain of launches otherwise, ~3 us each for a handful of texels).  Same fetches, same values.
__global__ __launch_bounds__(256) void k_hiz_tail(float* __restrict__ pyramid, int srcW, int srcH, size_t srcOffset, int levels)
{
    const float* src = pyramid + srcOffset;
    float* dst = pyramid + srcOffset + (size_t)srcW * srcH;
    int sw = srcW, sh = srcH;
    for (int l = 0; l < levels; l++) {
        const int w = max(sw >> 1, 1), h = max(sh >> 1, 1);
        for (int i = threadIdx.x; i < w * h; i += 256) {
            const int x = i % w, y = i / w;
            dst[i] = hiz_fetch_min(src, sw, sh, ((float)x + 0.5f) / (float)w, ((float)y + 0.5f) / (float)h);
        }
        __threadfence_block();
        __syncthreads();
        src = dst; sw = w; sh = h;
        dst += (size_t)w * h;
    }
}

// Math.glsl:296-315 ProjectSphere, op for op as the oracle's project_sphere
__device__ __forceinline__ bool msc_project_sphere(float Cx, float Cy, float Cz, float r, float znear, float P00, float P11, float* aabb)
{
    if (Cz < r + znear) return false;
    const float cx0 = -Cx, cx1 = -Cz;
    const float vx0 = sqrtf((cx0 * cx0 + cx1 * cx1) - r * r), vx1 = r;
    const float minx0 = vx0 * cx0 + (-vx1) * cx1, minx1 = vx1 * cx0 + vx0 * cx1;
    const float maxx0 = vx0 * cx0 + vx1 * cx1, maxx1 = (-vx1) * cx0 + vx0 * cx1;
    const float cy0 = -Cy, cy1 = -Cz;
    const float vy0 = sqrtf((cy0 * cy0 + cy1 * cy1) - r * r), vy1 = r;
    const float miny0 = vy0 * cy0 + (-vy1) * cy1, miny1 = vy1 * cy0 + vy0 * cy1;
    const float maxy0 = vy0 * cy0 + vy1 * cy1, maxy1 = (-vy1) * cy0 + vy0 * cy1;
    const float a0 = minx0 / minx1 * P00, a1 = miny0 / miny1 * P11, a2 = maxx0 / maxx1 * P00, a3 = maxy0 / maxy1 * P11;
    aabb[0] = a0 * 0.5f + 0.5f; aabb[1] = a3 * -0.5f + 0.5f; aabb[2] = a2 * 0.5f + 0.5f; aabb[3] = a1 * -0.5f + 0.5f;
    return true;
}

__device__ __forceinline__ bool msc_occluded(const HiZArgs& hz, float cx, float cy, float cz, float radius, float znear, float P00, float P11)
{
    float aabb[4];
    if (!msc_project_sphere(cx, cy, cz, radius, znear, P00, P11, aabb)) return false;
    const float width = (aabb[2] - aabb[0]) * (float)hz.width, height = (aabb[3] - aabb[1]) * (float)hz.height;
    const float m = width < height ? height : width;
    int level = 0; // floor(log2(m)) = the exponent field; textureLod clamps it to the pyramid
    if (m > 0.0f) {
        const int e = (int)((__float_as_uint(m) >> 23) & 0xFFu) - 127;
        level = e < 0 ? 0 : (e > hz.levels - 1 ? hz.levels - 1 : e);
    }
    const float u = (aabb[0] + aabb[2]) * 0.5f, v = (aabb[1] + aabb[3]) * 0.5f;
    const float* tex = hz.pyramid;
    for (int l = 0; l < level; l++) tex += (size_t)max(hz.width >> l, 1) * max(hz.height >> l, 1);
    const float depth = hiz_fetch_min(tex, max(hz.width >> level, 1), max(hz.height >> level, 1), u, v);
    const float depthSphere = znear / (cz - radius);
    return depthSphere < depth;
}

#define MSC_PER_BLOCK 1024 // instances per 256-thread block: the frustum set-up below is paid once per 1024 instances

template <bool OCCLUSION>
__global__ __launch_bounds__(256) void k4_mesh_frustum_cull(Mat4 view, Mat4 invProj, int vpW, int vpH, float zNearArg, float zFarArg,
                                                             SailorPerInstanceData* __restrict__ inst, uint32_t first, uint32_t count, HiZArgs hz, float P00, float P11)
{
    __shared__ float sV[4][3];
    __shared__ float sN[4][3];
    // Math.glsl:185-222 CreateViewFrustum(frame.viewportSize, frame.invProjection): one lane per corner, then one per plane
    if (threadIdx.x < 4) {
        const float fw = (float)vpW, fh = (float)vpH;
        msc_screen_to_view(invProj, (threadIdx.x & 1) ? fw : 0.0f, (threadIdx.x & 2) ? fh : 0.0f, fw, fh, sV[threadIdx.x]);
    }
    __syncthreads();
    if (threadIdx.x < 4) {
        // planes (2,0) (1,3) (0,1) (3,2): left, right, top, bottom
        const int pa = threadIdx.x == 0 ? 2 : threadIdx.x == 1 ? 1 : threadIdx.x == 2 ? 0 : 3;
        const int pb = threadIdx.x == 0 ? 0 : threadIdx.x == 1 ? 3 : threadIdx.x == 2 ? 1 : 2;
        float p1[3] = { sV[pa][0], sV[pa][1], sV[pa][2] }, p2[3] = { sV[pb][0], sV[pb][1], sV[pb][2] }, n[3];
        msc_plane(p1, p2, n);
        sN[threadIdx.x][0] = n[0]; sN[threadIdx.x][1] = n[1]; sN[threadIdx.x][2] = n[2];
    }
    __syncthreads();
#pragma unroll
    for (int it = 0; it < MSC_PER_BLOCK / 256; it++) {
        const uint32_t k = blockIdx.x * MSC_PER_BLOCK + it * 256 + threadIdx.x;
        if (k >= count) break;
        SailorPerInstanceData* I = inst + first + k;
        const float4* m4 = reinterpret_cast<const float4*>(I);
        Mat4 model;
        const float4 c0 = m4[0], c1 = m4[1], c2 = m4[2], c3 = m4[3], sb = m4[4];
        model.m[0] = c0.x; model.m[1] = c0.y; model.m[2] = c0.z; model.m[3] = c0.w; model.m[4] = c1.x; model.m[5] = c1.y; model.m[6] = c1.z; model.m[7] = c1.w;
        model.m[8] = c2.x; model.m[9] = c2.y; model.m[10] = c2.z; model.m[11] = c2.w; model.m[12] = c3.x; model.m[13] = c3.y; model.m[14] = c3.z; model.m[15] = c3.w;
        const float4 wc = glsl_mul(model, sb.x, sb.y, sb.z, 1.0f);
        const float4 vc = glsl_mul(view, wc.x, wc.y, wc.z, wc.w);
        const float cx = vc.x / vc.w, cy = vc.y / vc.w, cz = (vc.z / vc.w) * -1.0f;
        const float lossyScale = sqrtf(dot3f(c0.x, c0.y, c0.z, c0.x, c0.y, c0.z));
        const float radius = sb.w * lossyScale;
        // SphereFrustumOverlaps(center, radius, frustum, zNear = frame.cameraZNearZFar.y, zFar = frame.cameraZNearZFar.x) (:107)
        bool overlaps = !(cz - radius > zNearArg || cz + radius < zFarArg);
#pragma unroll
        for (int p = 0; p < 4; p++)
            if (dot3f(sN[p][0], sN[p][1], sN[p][2], cx, cy, cz) < -radius) overlaps = false;
        bool culled = !overlaps;
        if (OCCLUSION && !culled) culled = msc_occluded(hz, cx, cy, cz, radius, zFarArg /* = frame.cameraZNearZFar.x */, P00, P11); // FrustumCulling || OcclusionCulling (:139)
        I->isCulled = culled ? 1u : 0u;
    }
}

extern "C" {

// [entityBegin, entityEnd): the slice of the entity array this call sweeps (the split of K4 across the ranks of a node, SURVEY.md 8e: contiguous index
// ranges, one all-gather of the visibility words behind it).  Hierarchies of up to ECS_FUSED_LEVELS levels -- the one-launch form, in which an entity
// rebuilds its ancestors' relative matrices from their TRS records and so needs nothing another rank computes -- take any slice; a deeper hierarchy
// reads its parents' WORLD matrices, which only a sweep of the whole set has: a proper slice of one is refused (SAILOR_HIP_ERR_UNSUPPORTED: sweep it whole
// on every rank).  Only the slice's entries of dWorld / dWorldAabb and the slice's bits of dVisibility are written.
int sailor_hip_ecs_sweep_range(SailorHipContext* ctx, uint32_t numEntities, const SailorTransform* dTransforms, const uint32_t* dParent,
                               const uint32_t* levelOffsets, uint32_t numLevels, const SailorAABB* dLocalAabb, const float* planes,
                               float* dWorld, SailorAABB* dWorldAabb, uint64_t* dVisibility, uint32_t entityBegin, uint32_t entityEnd)
{
    if (!ctx || !levelOffsets || !planes) return SAILOR_HIP_ERR_INVALID_ARGUMENT;
    SAILOR_TRY_HIP(ctx, hipSetDevice(ctx->device)); // a host thread may drive several contexts
    if (entityBegin > entityEnd || entityEnd > numEntities) return SAILOR_HIP_ERR_INVALID_ARGUMENT;
    if (numEntities == 0 || entityBegin == entityEnd) return SAILOR_HIP_OK;
    if (!dTransforms || !dParent || !dLocalAabb || !dWorld || !dWorldAabb || !dVisibility) return SAILOR_HIP_ERR_INVALID_ARGUMENT;
    if (numLevels == 0 || levelOffsets[0] != 0 || levelOffsets[numLevels] != numEntities) return SAILOR_HIP_ERR_INVALID_ARGUMENT;
    if (((uintptr_t)dTransforms & 15) || ((uintptr_t)dWorld & 15)) return SAILOR_HIP_ERR_INVALID_ARGUMENT;
    Planes6 P;
    memcpy(P.p, planes, sizeof P.p);
    const int boxVec = ((((uintptr_t)dLocalAabb | (uintptr_t)dWorldAabb) & 15) == 0) ? 1 : 0; // 64 boxes = 1536 B: float4-addressable per word
    for (uint32_t l = 0; l < numLevels; l++)
        if (levelOffsets[l + 1] < levelOffsets[l] || levelOffsets[l + 1] > numEntities) return SAILOR_HIP_ERR_INVALID_ARGUMENT;
    if (numLevels <= ECS_FUSED_LEVELS) { // shallow hierarchy: every level in one launch (ancestors' relative matrices are recomputed)
        const uint32_t words = ((entityEnd + 63) >> 6) - (entityBegin >> 6);
        hipLaunchKernelGGL(k4_ecs_level<true>, dim3((words + 3) / 4), dim3(256), 0, ctx->stream, entityBegin, entityEnd, (const float4*)dTransforms, dParent,
                           (const float*)dLocalAabb, P, boxVec, (float4*)dWorld, (float*)dWorldAabb, (unsigned long long*)dVisibility);
        SAILOR_CHECK_LAUNCH(ctx, "k4_ecs_level<fused>");
        return SAILOR_HIP_OK;
    }
    if (entityBegin != 0 || entityEnd != numEntities) return SAILOR_HIP_ERR_UNSUPPORTED;
    for (uint32_t l = 0; l < numLevels; l++) {
        const uint32_t lo = levelOffsets[l], hi = levelOffsets[l + 1];
        if (hi == lo) continue;
        const uint32_t words = ((hi + 63) >> 6) - (lo >> 6);
        hipLaunchKernelGGL(k4_ecs_level<false>, dim3((words + 3) / 4), dim3(256), 0, ctx->stream, lo, hi, (const float4*)dTransforms, dParent,
                           (const float*)dLocalAabb, P, boxVec, (float4*)dWorld, (float*)dWorldAabb, (unsigned long long*)dVisibility);
        SAILOR_CHECK_LAUNCH(ctx, "k4_ecs_level");
    }
    return SAILOR_HIP_OK;
}

int sailor_hip_ecs_sweep(SailorHipContext* ctx, uint32_t numEntities, const SailorTransform* dTransforms, const uint32_t* dParent,
                         const uint32_t* levelOffsets, uint32_t numLevels, const SailorAABB* dLocalAabb, const float* planes,
                         float* dWorld, SailorAABB* dWorldAabb, uint64_t* dVisibility)
{
    return sailor_hip_ecs_sweep_range(ctx, numEntities, dTransforms, dParent, levelOffsets, numLevels, dLocalAabb, planes, dWorld, dWorldAabb, dVisibility, 0u, numEntities);
}

// The slice of rank `rank` of `worldSize` for sailor_hip_ecs_sweep_range: whole 64-entity visibility words, ceil(words / worldSize) of them per rank
// (the last ranks may get fewer, or none), so that the ranks' visibility words are disjoint runs of equal length: ONE in-place ncclAllGather of
// *outWordsPerRank uint64 per rank over a buffer of worldSize * *outWordsPerRank words rebuilds the whole bitmask on every rank.
int sailor_hip_ecs_range_for_rank(uint32_t numEntities, int32_t rank, int32_t worldSize, uint32_t* outBegin, uint32_t* outEnd, uint32_t* outWordsPerRank)
{
    if (worldSize <= 0 || rank < 0 || rank >= worldSize || !outBegin || !outEnd) return SAILOR_HIP_ERR_INVALID_ARGUMENT;
    const uint64_t words = ((uint64_t)numEntities + 63) >> 6;
    const uint64_t per = (words + (uint64_t)worldSize - 1) / (uint64_t)worldSize;
    uint64_t b = per * (uint64_t)rank * 64, e = per * (uint64_t)(rank + 1) * 64;
    if (b > numEntities) b = numEntities;
    if (e > numEntities) e = numEntities;
    *outBegin = (uint32_t)b; *outEnd = (uint32_t)e;
    if (outWordsPerRank) *outWordsPerRank = (uint32_t)per;
    return SAILOR_HIP_OK;
}

int sailor_hip_mesh_cull_flags(SailorHipContext* ctx, const SailorUboFrameData* frame, SailorPerInstanceData* dInstances, uint32_t numInstances,
                               uint32_t firstInstanceIndex, const SailorHiZDesc* hiz)
{
    if (!ctx || !frame) return SAILOR_HIP_ERR_INVALID_ARGUMENT;
    SAILOR_TRY_HIP(ctx, hipSetDevice(ctx->device)); // a host thread may drive several contexts
    if (hiz && (!hiz->pyramid || hiz->width <= 0 || hiz->height <= 0 || hiz->levels <= 0 || hiz->levels > 16)) return SAILOR_HIP_ERR_INVALID_ARGUMENT;
    if (numInstances == 0) return SAILOR_HIP_OK;
    if (!dInstances || ((uintptr_t)dInstances & 15)) return SAILOR_HIP_ERR_INVALID_ARGUMENT;
    Mat4 view, invProj;
    memcpy(view.m, frame->view, 64);
    memcpy(invProj.m, frame->invProjection, 64);
    HiZArgs hz = { nullptr, 1, 1, 1 };
    if (hiz) { hz.pyramid = hiz->pyramid; hz.width = hiz->width; hz.height = hiz->height; hz.levels = hiz->levels; }
    const dim3 grid((numInstances + MSC_PER_BLOCK - 1) / MSC_PER_BLOCK);
    if (hiz)
        hipLaunchKernelGGL(k4_mesh_frustum_cull<true>, grid, dim3(256), 0, ctx->stream, view, invProj, frame->viewportSize[0], frame->viewportSize[1],
                           frame->cameraZNearZFar[1], frame->cameraZNearZFar[0], dInstances, firstInstanceIndex, numInstances, hz, frame->projection[0],
                           frame->projection[5]);
    else
        hipLaunchKernelGGL(k4_mesh_frustum_cull<false>, grid, dim3(256), 0, ctx->stream, view, invProj, frame->viewportSize[0], frame->viewportSize[1],
                           frame->cameraZNearZFar[1], frame->cameraZNearZFar[0], dInstances, firstInstanceIndex, numInstances, hz, 0.0f, 0.0f);
    SAILOR_CHECK_LAUNCH(ctx, "k4_mesh_frustum_cull");
    return SAILOR_HIP_OK;
}

int sailor_hip_mesh_frustum_cull(SailorHipContext* ctx, const SailorUboFrameData* frame, SailorPerInstanceData* dInstances,
                                 uint32_t numInstances, uint32_t firstInstanceIndex)
{
    return sailor_hip_mesh_cull_flags(ctx, frame, dInstances, numInstances, firstInstanceIndex, nullptr);
}

int sailor_hip_csm_caster_masks(SailorHipContext* ctx, uint32_t numEntities, const SailorAABB* dWorldAabb, const float* cascadePlanes, uint32_t numCascades,
                                uint64_t* dMasks)
{
    if (!ctx || !cascadePlanes || numCascades == 0 || numCascades > SAILOR_NUM_CSM_CASCADES) return SAILOR_HIP_ERR_INVALID_ARGUMENT;
    SAILOR_TRY_HIP(ctx, hipSetDevice(ctx->device)); // a host thread may drive several contexts
    if (numEntities == 0) return SAILOR_HIP_OK;
    if (!dWorldAabb || !dMasks) return SAILOR_HIP_ERR_INVALID_ARGUMENT;
    CascadePlanes P;
    memset(&P, 0, sizeof P);
    memcpy(P.p, cascadePlanes, (size_t)numCascades * 24 * sizeof(float));
    const uint32_t words = (numEntities + 63) / 64;
    hipLaunchKernelGGL(k4_csm_caster_masks, dim3((numEntities + 255) / 256), dim3(256), 0, ctx->stream, numEntities, (const float*)dWorldAabb, P, (int)numCascades,
                       (unsigned long long*)dMasks, words);
    SAILOR_CHECK_LAUNCH(ctx, "k4_csm_caster_masks");
    return SAILOR_HIP_OK;
}

int sailor_hip_hiz_downscale(SailorHipContext* ctx, const float* dSrc, int32_t srcWidth, int32_t srcHeight, float* dDst, int32_t dstWidth, int32_t dstHeight)
{
    if (!ctx || !dSrc || !dDst || srcWidth <= 0 || srcHeight <= 0 || dstWidth <= 0 || dstHeight <= 0) return SAILOR_HIP_ERR_INVALID_ARGUMENT;
    SAILOR_TRY_HIP(ctx, hipSetDevice(ctx->device)); // a host thread may drive several contexts
    hipLaunchKernelGGL(k_hiz_downscale, dim3((dstWidth + 31) / 32, (dstHeight + 7) / 8), dim3(256), 0, ctx->stream, dSrc, srcWidth, srcHeight, dDst, dstWidth,
                       dstHeight);
    SAILOR_CHECK_LAUNCH(ctx, "k_hiz_downscale");
    return SAILOR_HIP_OK;
}

int sailor_hip_hiz_build(SailorHipContext* ctx, const float* dDepth, int32_t depthWidth, int32_t depthHeight, float* dPyramid, int32_t width, int32_t height,
                         int32_t levels)
{
    if (!ctx || levels <= 0 || levels > 16 || width <= 0 || height <= 0) return SAILOR_HIP_ERR_INVALID_ARGUMENT;
    const float* src = dDepth;
    int sw = depthWidth, sh = depthHeight;
    float* dst = dPyramid;
    if (!dDepth || !dPyramid) return SAILOR_HIP_ERR_INVALID_ARGUMENT;
    for (int l = 0; l < levels; l++) { // DepthHighZNode.cpp:78-95: mip 0 from the depth attachment, mip i + 1 from mip i
        const int w = (width >> l) > 1 ? (width >> l) : 1, h = (height >> l) > 1 ? (height >> l) : 1;
        // mip sizes follow max(size >> level, 1), and (size >> l) >> 1 == size >> (l + 1): once a level fits one block, the rest is one launch
        if (l > 0 && sw <= 64 && sh <= 64 && src != dDepth) {
            hipLaunchKernelGGL(k_hiz_tail, dim3(1), dim3(256), 0, ctx->stream, dPyramid, sw, sh, (size_t)(src - dPyramid), levels - l);
            SAILOR_CHECK_LAUNCH(ctx, "k_hiz_tail");
            return SAILOR_HIP_OK;
        }
        const int rc = sailor_hip_hiz_downscale(ctx, src, sw, sh, dst, w, h);
        if (rc != SAILOR_HIP_OK) return rc;
        src = dst; sw = w; sh = h;
        dst += (size_t)w * h;
    }
    return SAILOR_HIP_OK;
}

} // extern "C"
