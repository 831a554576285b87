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

__global__ __launch_bounds__(256) void k4_ecs_level(uint32_t lo, uint32_t hi, const float4* __restrict__ trs, const uint32_t* __restrict__ parent,
                                                     const float* __restrict__ localAabb, Planes6 planes,
                                                     float4* __restrict__ world, float* __restrict__ worldAabb, unsigned long long* __restrict__ visibility)
{
    // one wave owns one 64-entity visibility word; entities outside [lo, hi) of the word are masked
    const uint32_t word = (lo >> 6) + blockIdx.x * 4 + (threadIdx.x >> 6);
    const uint32_t i = word * 64 + (threadIdx.x & 63);
    const bool active = i >= lo && i < hi;
    bool vis = false;
    if (active) {
        const float4 pos = trs[(size_t)i * 3 + 0], rot = trs[(size_t)i * 3 + 1], scl = trs[(size_t)i * 3 + 2];
        float rel[16], W[16];
        transform_matrix(pos, rot, scl, rel);
        const uint32_t par = parent[i];
        if (par == 0xFFFFFFFFu) {
#pragma unroll
            for (int k = 0; k < 16; k++) W[k] = rel[k]; // TransformECS.cpp:192-195
        } else {
            float P[16];
            const float4* pw = world + (size_t)par * 4;
            const float4 c0 = pw[0], c1 = pw[1], c2 = pw[2], c3 = pw[3];
            P[0] = c0.x; P[1] = c0.y; P[2] = c0.z; P[3] = c0.w; P[4] = c1.x; P[5] = c1.y; P[6] = c1.z; P[7] = c1.w;
            P[8] = c2.x; P[9] = c2.y; P[10] = c2.z; P[11] = c2.w; P[12] = c3.x; P[13] = c3.y; P[14] = c3.z; P[15] = c3.w;
            mat_mul(P, rel, W); // TransformECS.cpp:201
        }
        float4* ow = world + (size_t)i * 4;
        ow[0] = make_float4(W[0], W[1], W[2], W[3]);
        ow[1] = make_float4(W[4], W[5], W[6], W[7]);
        ow[2] = make_float4(W[8], W[9], W[10], W[11]);
        ow[3] = make_float4(W[12], W[13], W[14], W[15]);

        // AABB::Apply (Bounds.cpp:479-492), corner order of Bounds.h:119-130
        const float* la = localAabb + (size_t)i * 6;
        const float mnx = la[0], mny = la[1], mnz = la[2], mxx = la[3], mxy = la[4], mxz = la[5];
        const float px[8] = { mnx, mxx, mnx, mxx, mxx, mxx, mnx, mnx };
        const float py[8] = { mny, mxy, mxy, mny, mxy, mny, mxy, mny };
        const float pz[8] = { mnz, mxz, mxz, mxz, mnz, mnz, mnz, mxz };
        float omin[3] = { FLT_MAX, FLT_MAX, FLT_MAX }, omax[3] = { FLT_MIN, FLT_MIN, FLT_MIN };
#pragma unroll
        for (int k = 0; k < 8; k++) {
#pragma unroll
            for (int c = 0; c < 3; c++) {
                const float t = (W[0 + c] * px[k] + W[4 + c] * py[k]) + (W[8 + c] * pz[k] + W[12 + c] * 1.0f);
                omin[c] = (omin[c] < t) ? omin[c] : t;
                omax[c] = (t < omax[c]) ? omax[c] : t;
            }
        }
        float* oa = worldAabb + (size_t)i * 6;
        oa[0] = omin[0]; oa[1] = omin[1]; oa[2] = omin[2]; oa[3] = omax[0]; oa[4] = omax[1]; oa[5] = omax[2];

        // Frustum::OverlapsAABB (Bounds.cpp:245-260)
        vis = true;
#pragma unroll
        for (int p = 0; p < 6; p++) {
            const float ax = omin[0] * planes.p[4 * p + 0], bx = omax[0] * planes.p[4 * p + 0];
            const float ay = omin[1] * planes.p[4 * p + 1], by = omax[1] * planes.p[4 * p + 1];
            const float az = omin[2] * planes.p[4 * p + 2], bz = omax[2] * planes.p[4 * p + 2];
            const float d = (ax < bx ? bx : ax) + (ay < by ? by : ay) + (az < bz ? bz : az) + planes.p[4 * p + 3];
            vis = vis && (d > 0.0f);
        }
    }
    const unsigned long long vmask = __ballot(vis), amask = __ballot(active);
    if ((threadIdx.x & 63) == 0 && amask) {
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

__global__ __launch_bounds__(256) void k4_mesh_frustum_cull(Mat4 view, Mat4 invProj, int vpW, int vpH, float zNearArg, float zFarArg,
                                                             SailorPerInstanceData* __restrict__ inst, uint32_t first, uint32_t count)
{
    __shared__ float sN[4][3];
    if (threadIdx.x == 0) { // Math.glsl:185-222 CreateViewFrustum(frame.viewportSize, frame.invProjection)
        float vs[4][3];
        const float fw = (float)vpW, fh = (float)vpH;
        msc_screen_to_view(invProj, 0.0f, 0.0f, fw, fh, vs[0]);
        msc_screen_to_view(invProj, fw, 0.0f, fw, fh, vs[1]);
        msc_screen_to_view(invProj, 0.0f, fh, fw, fh, vs[2]);
        msc_screen_to_view(invProj, fw, fh, fw, fh, vs[3]);
        msc_plane(vs[2], vs[0], sN[0]);
        msc_plane(vs[1], vs[3], sN[1]);
        msc_plane(vs[0], vs[1], sN[2]);
        msc_plane(vs[3], vs[2], sN[3]);
    }
    __syncthreads();
    const uint32_t k = blockIdx.x * 256 + threadIdx.x;
    if (k >= count) return;
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
    I->isCulled = overlaps ? 0u : 1u;
}

extern "C" {

int sailor_hip_ecs_sweep(SailorHipContext* ctx, uint32_t numEntities, const SailorTransform* dTransforms, const uint32_t* dParent,
                         const uint32_t* levelOffsets, uint32_t numLevels, const SailorAABB* dLocalAabb, const float* planes,
                         float* dWorld, SailorAABB* dWorldAabb, uint64_t* dVisibility)
{
    if (!ctx || !levelOffsets || !planes) return SAILOR_HIP_ERR_INVALID_ARGUMENT;
    if (numEntities == 0) return SAILOR_HIP_OK;
    if (!dTransforms || !dParent || !dLocalAabb || !dWorld || !dWorldAabb || !dVisibility) return SAILOR_HIP_ERR_INVALID_ARGUMENT;
    if (numLevels == 0 || levelOffsets[0] != 0 || levelOffsets[numLevels] != numEntities) return SAILOR_HIP_ERR_INVALID_ARGUMENT;
    if (((uintptr_t)dTransforms & 15) || ((uintptr_t)dWorld & 15)) return SAILOR_HIP_ERR_INVALID_ARGUMENT;
    Planes6 P;
    memcpy(P.p, planes, sizeof P.p);
    for (uint32_t l = 0; l < numLevels; l++) {
        const uint32_t lo = levelOffsets[l], hi = levelOffsets[l + 1];
        if (hi < lo || hi > numEntities) return SAILOR_HIP_ERR_INVALID_ARGUMENT;
        if (hi == lo) continue;
        const uint32_t words = ((hi + 63) >> 6) - (lo >> 6);
        hipLaunchKernelGGL(k4_ecs_level, dim3((words + 3) / 4), dim3(256), 0, ctx->stream, lo, hi, (const float4*)dTransforms, dParent,
                           (const float*)dLocalAabb, P, (float4*)dWorld, (float*)dWorldAabb, (unsigned long long*)dVisibility);
        SAILOR_CHECK_LAUNCH(ctx, "k4_ecs_level");
    }
    return SAILOR_HIP_OK;
}

int sailor_hip_mesh_frustum_cull(SailorHipContext* ctx, const SailorUboFrameData* frame, SailorPerInstanceData* dInstances,
                                 uint32_t numInstances, uint32_t firstInstanceIndex)
{
    if (!ctx || !frame) return SAILOR_HIP_ERR_INVALID_ARGUMENT;
    if (numInstances == 0) return SAILOR_HIP_OK;
    if (!dInstances || ((uintptr_t)dInstances & 15)) return SAILOR_HIP_ERR_INVALID_ARGUMENT;
    Mat4 view, invProj;
    memcpy(view.m, frame->view, 64);
    memcpy(invProj.m, frame->invProjection, 64);
    hipLaunchKernelGGL(k4_mesh_frustum_cull, dim3((numInstances + 255) / 256), dim3(256), 0, ctx->stream, view, invProj,
                       frame->viewportSize[0], frame->viewportSize[1], frame->cameraZNearZFar[1], frame->cameraZNearZFar[0],
                       dInstances, firstInstanceIndex, numInstances);
    SAILOR_CHECK_LAUNCH(ctx, "k4_mesh_frustum_cull");
    return SAILOR_HIP_OK;
}

} // extern "C"
