// Host-side math of the path (pure CPU): the glm-based camera / frustum / CSM set-up the reference does on the
// main and render threads before any GPU work is recorded.  glm itself is an un-vendored submodule of the reference
// (External/glm, .gitmodules:4-6); the few routines the path needs are written here against glm's published
// evaluation order so that frame constants are reproducible bit for bit.
//
//   PerspectiveRH (reversed Z)            Math/Math.cpp:18-21
//   glm::inverse                          ECS/CameraECS.cpp:20,33 ; ECS/LightingECS.cpp:227
//   Transform::Matrix                     Math/Transform.cpp:39-42
//   FillFrameData                         FrameGraph/RHIFrameGraph.cpp:60-67
//   Frustum::ExtractFrustumPlanes         Math/Bounds.cpp:142-193
//   CalculateOrthoMatrixByView            Math/Bounds.cpp:78-109
//   CalculateLightProjectionForCascades   FrameGraph/ShadowPrepassNode.cpp:387-404 (+ ECS/LightingECS.cpp:292)
//   LightShaderData packing               ECS/LightingECS.cpp:163-172
#define SAILOR_HIP_BUILD 1
#include "../../include/sailor_hip.h"
#include <cmath>
#include <cstring>
#include <vector>
#include <new>
#include <algorithm>
#include <limits>

namespace {

struct V3 {
    float x, y, z;
};
struct V4 {
    float x, y, z, w;
    float& operator[](int i) { return (&x)[i]; }
    float operator[](int i) const { return (&x)[i]; }
};
inline V4 operator*(const V4& a, float s) { return { a.x * s, a.y * s, a.z * s, a.w * s }; }
inline V4 operator+(const V4& a, const V4& b) { return { a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w }; }
inline V3 operator*(const V3& a, float s) { return { a.x * s, a.y * s, a.z * s }; }
inline V3 operator*(float s, const V3& a) { return { s * a.x, s * a.y, s * a.z }; }
inline V3 operator+(const V3& a, const V3& b) { return { a.x + b.x, a.y + b.y, a.z + b.z }; }
inline V3 operator-(const V3& a, const V3& b) { return { a.x - b.x, a.y - b.y, a.z - b.z }; }
inline V3 operator-(const V3& a) { return { -a.x, -a.y, -a.z }; }

struct M4 {
    V4 c[4]; // columns, as glm::mat4
    V4& operator[](int i) { return c[i]; }
    const V4& operator[](int i) const { return c[i]; }
};

inline M4 load(const float* p) { M4 m; std::memcpy(&m, p, 64); return m; }
inline void store(const M4& m, float* p) { std::memcpy(p, &m, 64); }
inline M4 identity() { return { { { 1, 0, 0, 0 }, { 0, 1, 0, 0 }, { 0, 0, 1, 0 }, { 0, 0, 0, 1 } } }; }

// glm: operator*(mat4, mat4) -- column c = A0*B[c].x + A1*B[c].y + A2*B[c].z + A3*B[c].w, summed left to right
inline M4 mul(const M4& a, const M4& b)
{
    M4 r;
    for (int c = 0; c < 4; c++) r[c] = ((a[0] * b[c].x + a[1] * b[c].y) + a[2] * b[c].z) + a[3] * b[c].w;
    return r;
}
// glm: operator*(mat4, vec4) -- (m0*v.x + m1*v.y) + (m2*v.z + m3*v.w)
inline V4 mul(const M4& m, const V4& v) { return (m[0] * v.x + m[1] * v.y) + (m[2] * v.z + m[3] * v.w); }

inline float dot(const V3& a, const V3& b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
inline V3 cross(const V3& x, const V3& y) { return { x.y * y.z - y.y * x.z, x.z * y.x - y.z * x.x, x.x * y.y - y.x * x.y }; }
inline V3 normalize(const V3& v) { return v * (1.0f / std::sqrt(dot(v, v))); } // v * inversesqrt(dot(v, v))

// glm compute_inverse<4, 4>
M4 inverse(const M4& m)
{
    const float c00 = m[2][2] * m[3][3] - m[3][2] * m[2][3], c02 = m[1][2] * m[3][3] - m[3][2] * m[1][3], c03 = m[1][2] * m[2][3] - m[2][2] * m[1][3];
    const float c04 = m[2][1] * m[3][3] - m[3][1] * m[2][3], c06 = m[1][1] * m[3][3] - m[3][1] * m[1][3], c07 = m[1][1] * m[2][3] - m[2][1] * m[1][3];
    const float c08 = m[2][1] * m[3][2] - m[3][1] * m[2][2], c10 = m[1][1] * m[3][2] - m[3][1] * m[1][2], c11 = m[1][1] * m[2][2] - m[2][1] * m[1][2];
    const float c12 = m[2][0] * m[3][3] - m[3][0] * m[2][3], c14 = m[1][0] * m[3][3] - m[3][0] * m[1][3], c15 = m[1][0] * m[2][3] - m[2][0] * m[1][3];
    const float c16 = m[2][0] * m[3][2] - m[3][0] * m[2][2], c18 = m[1][0] * m[3][2] - m[3][0] * m[1][2], c19 = m[1][0] * m[2][2] - m[2][0] * m[1][2];
    const float c20 = m[2][0] * m[3][1] - m[3][0] * m[2][1], c22 = m[1][0] * m[3][1] - m[3][0] * m[1][1], c23 = m[1][0] * m[2][1] - m[2][0] * m[1][1];
    const V4 f0 { c00, c00, c02, c03 }, f1 { c04, c04, c06, c07 }, f2 { c08, c08, c10, c11 };
    const V4 f3 { c12, c12, c14, c15 }, f4 { c16, c16, c18, c19 }, f5 { c20, c20, c22, c23 };
    const V4 v0 { m[1][0], m[0][0], m[0][0], m[0][0] }, v1 { m[1][1], m[0][1], m[0][1], m[0][1] };
    const V4 v2 { m[1][2], m[0][2], m[0][2], m[0][2] }, v3 { m[1][3], m[0][3], m[0][3], m[0][3] };
    auto had = [](const V4& a, const V4& b) { return V4 { a.x * b.x, a.y * b.y, a.z * b.z, a.w * b.w }; };
    auto sub = [](const V4& a, const V4& b) { return V4 { a.x - b.x, a.y - b.y, a.z - b.z, a.w - b.w }; };
    const V4 i0 = sub(had(v1, f0), had(v2, f1)) + had(v3, f2);
    const V4 i1 = sub(had(v0, f0), had(v2, f3)) + had(v3, f4);
    const V4 i2 = sub(had(v0, f1), had(v1, f3)) + had(v3, f5);
    const V4 i3 = sub(had(v0, f2), had(v1, f4)) + had(v2, f5);
    const V4 sa { +1, -1, +1, -1 }, sb { -1, +1, -1, +1 };
    M4 inv { { had(i0, sa), had(i1, sb), had(i2, sa), had(i3, sb) } };
    const V4 row0 { inv[0][0], inv[1][0], inv[2][0], inv[3][0] };
    const V4 d0 = had(m[0], row0);
    const float d1 = (d0.x + d0.y) + (d0.z + d0.w);
    const float ood = 1.0f / d1;
    for (int c = 0; c < 4; c++) inv[c] = inv[c] * ood;
    return inv;
}

// glm::perspectiveRH_ZO
M4 perspective_rh_zo(float fovy, float aspect, float zNear, float zFar)
{
    const float t = std::tan(fovy / 2.0f);
    M4 r {};
    r[0][0] = 1.0f / (aspect * t);
    r[1][1] = 1.0f / t;
    r[2][2] = zFar / (zNear - zFar);
    r[2][3] = -1.0f;
    r[3][2] = -(zFar * zNear) / (zFar - zNear);
    return r;
}

// glm::orthoRH_NO
M4 ortho_rh_no(float left, float right, float bottom, float top, float zNear, float zFar)
{
    M4 r = identity();
    r[0][0] = 2.0f / (right - left);
    r[1][1] = 2.0f / (top - bottom);
    r[2][2] = -2.0f / (zFar - zNear);
    r[3][0] = -(right + left) / (right - left);
    r[3][1] = -(top + bottom) / (top - bottom);
    r[3][2] = -(zFar + zNear) / (zFar - zNear);
    return r;
}

M4 transform_matrix(const SailorTransform& t)
{
    const M4 I = identity();
    // glm::translate(m, v): Result[3] = m[0]*v[0] + m[1]*v[1] + m[2]*v[2] + m[3]
    M4 T = I;
    T[3] = ((I[0] * t.position[0] + I[1] * t.position[1]) + I[2] * t.position[2]) + I[3];
    // glm::mat4_cast(quat), quat memory order x,y,z,w
    const float x = t.rotation[0], y = t.rotation[1], z = t.rotation[2], w = t.rotation[3];
    const float qxx = x * x, qyy = y * y, qzz = z * z, qxz = x * z, qxy = x * y, qyz = y * z, qwx = w * x, qwy = w * y, qwz = w * z;
    M4 R = I;
    R[0][0] = 1.0f - 2.0f * (qyy + qzz); R[0][1] = 2.0f * (qxy + qwz); R[0][2] = 2.0f * (qxz - qwy);
    R[1][0] = 2.0f * (qxy - qwz); R[1][1] = 1.0f - 2.0f * (qxx + qzz); R[1][2] = 2.0f * (qyz + qwx);
    R[2][0] = 2.0f * (qxz + qwy); R[2][1] = 2.0f * (qyz - qwx); R[2][2] = 1.0f - 2.0f * (qxx + qyy);
    // glm::scale(m, v): Result[i] = m[i] * v[i]; Result[3] = m[3]
    M4 S;
    S[0] = I[0] * t.scale[0]; S[1] = I[1] * t.scale[1]; S[2] = I[2] * t.scale[2]; S[3] = I[3];
    return mul(mul(T, R), S);
}

struct Plane {
    V4 abcd;
    Plane() : abcd { 0, 0, 0, 0 } {}
    Plane(const V3& n, const V3& p) : abcd { n.x, n.y, n.z, -dot(p, n) } {} // Bounds.h:80-87
    void normalize()
    { // Bounds.cpp:9-13
        const float mag = std::sqrt(dot(V3 { abcd.x, abcd.y, abcd.z }, V3 { abcd.x, abcd.y, abcd.z }));
        abcd.x /= mag; abcd.y /= mag; abcd.z /= mag; abcd.w /= mag;
    }
};

struct Frustum {
    Plane planes[6]; // L R T B N F
    V3 corners[8];
    // Bounds.cpp:142-193
    void extract(const M4& world, float aspect, float fovY, float zNear, float zFar)
    {
        const float radians = fovY * 0.01745329251994329576923690768489f;
        const float halfVSide = zFar * std::tan(radians * .5f);
        const float halfHSide = halfVSide * aspect;
        const V3 right { world[0].x, world[0].y, world[0].z };
        const V3 up { world[1].x, world[1].y, world[1].z };
        const V3 forward = -V3 { world[2].x, world[2].y, world[2].z };
        const V3 pos { world[3].x, world[3].y, world[3].z };
        const V3 frontMultFar = zFar * forward;
        planes[4] = Plane(forward, pos + forward * zNear);
        planes[5] = Plane(-forward, pos + forward * zFar);
        planes[0] = Plane(normalize(cross(frontMultFar - right * halfHSide, up)), pos);
        planes[1] = Plane(normalize(cross(up, frontMultFar + right * halfHSide)), pos);
        const V3 bottomNormal = normalize(cross(right, frontMultFar - up * halfVSide));
        const V3 topNormal = normalize(cross(frontMultFar + up * halfVSide, right));
        planes[2] = Plane(topNormal, pos);
        planes[3] = Plane(bottomNormal, pos);
        for (auto& p : planes) p.normalize();

        const V3 farEnd { 0, 0, -zFar }, eh { halfHSide, 0, 0 }, ev { 0, halfVSide, 0 };
        auto xf = [&](const V3& v) { const V4 r = mul(world, V4 { v.x, v.y, v.z, 1.0f }); return V3 { r.x, r.y, r.z }; };
        corners[0] = xf(farEnd + eh + ev);
        corners[1] = xf(farEnd - eh + ev);
        corners[2] = xf(farEnd - eh - ev);
        corners[3] = xf(farEnd + eh - ev);
        const float halfVSideNear = zNear * std::tan(radians * .5f);
        const float halfHSideNear = halfVSideNear * aspect;
        const V3 start { 0, 0, -zNear }, sx { halfHSideNear, 0, 0 }, sy { 0, halfVSideNear, 0 };
        corners[4] = xf(start + sx + sy);
        corners[5] = xf(start - sx + sy);
        corners[6] = xf(start - sx - sy);
        corners[7] = xf(start + sx - sy);
    }
    // Bounds.cpp:20-67 ExtractFrustumPlanes(projectionViewMatrix) with :110-140 CalculateCorners(matrix, bReverseZ = true)
    void extract(const M4& projectionView)
    {
        const M4 inv = inverse(projectionView);
        const float sgn[4][2] = { { 1.0f, 1.0f }, { -1.0f, 1.0f }, { -1.0f, -1.0f }, { 1.0f, -1.0f } };
        for (int k = 0; k < 8; k++) {
            const float reverseZ = -1.0f;
            const V4 pt = mul(inv, V4 { sgn[k & 3][0], sgn[k & 3][1], k < 4 ? reverseZ * 1.0f : reverseZ * -1.0f, 1.0f });
            corners[k] = V3 { pt.x / pt.w, pt.y / pt.w, pt.z / pt.w };
        }
        const V3 right = normalize(corners[0] - corners[1]), up = normalize(corners[0] - corners[3]), forward = normalize(corners[0] - corners[4]);
        const V3 centerFar = 0.5f * (corners[0] + corners[2]), centerNear = 0.5f * (corners[4] + corners[6]);
        const V3 centerBottom = 0.5f * (corners[2] + corners[7]), centerTop = 0.5f * (corners[0] + corners[5]);
        const V3 centerLeft = 0.5f * (corners[1] + corners[6]), centerRight = 0.5f * (corners[0] + corners[7]);
        planes[4] = Plane(forward, centerNear);
        planes[5] = Plane(-forward, centerFar);
        planes[0] = Plane(normalize(cross(forward, up)), centerLeft);
        planes[1] = Plane(normalize(cross(up, forward)), centerRight);
        planes[2] = Plane(normalize(cross(forward, right)), centerTop);
        planes[3] = Plane(normalize(cross(right, forward)), centerBottom);
        for (auto& p : planes) p.normalize();
    }
    // Bounds.cpp:78-109
    M4 ortho_by_view(const M4& view, float zMult) const
    {
        float minX = std::numeric_limits<float>::max(), maxX = std::numeric_limits<float>::lowest();
        float minY = minX, maxY = maxX, minZ = minX, maxZ = maxX;
        for (const V3& v : corners) {
            const V4 t = mul(view, V4 { v.x, v.y, v.z, 1 });
            minX = std::min(minX, t.x); maxX = std::max(maxX, t.x);
            minY = std::min(minY, t.y); maxY = std::max(maxY, t.y);
            minZ = std::min(minZ, t.z); maxZ = std::max(maxZ, t.z);
        }
        minZ = minZ < 0 ? minZ * zMult : minZ / zMult;
        maxZ = maxZ < 0 ? maxZ / zMult : maxZ * zMult;
        const float zFar = -minZ, zNear = -maxZ;
        return ortho_rh_no(minX, maxX, minY, maxY, zFar, zNear); // reversed Z for shadows
    }
};

constexpr float kShadowCascadeLevels[SAILOR_NUM_CSM_CASCADES] = { 1.0f / 20.0f, 1.0f / 10.0f, 1.0f / 3.0f, 1.0f / 2.0f }; // ECS/LightingECS.h:66

} // namespace

extern "C" {

int sailor_host_perspective_rh(float fovRadians, float aspect, float zNear, float zFar, float* outMat4)
{
    if (!outMat4) return SAILOR_HIP_ERR_INVALID_ARGUMENT;
    store(perspective_rh_zo(fovRadians, aspect, zFar, zNear), outMat4); // Math.cpp:20: near/far swapped
    return SAILOR_HIP_OK;
}

int sailor_host_mat4_inverse(const float* m, float* outMat4)
{
    if (!m || !outMat4) return SAILOR_HIP_ERR_INVALID_ARGUMENT;
    store(inverse(load(m)), outMat4);
    return SAILOR_HIP_OK;
}

int sailor_host_mat4_mul(const float* a, const float* b, float* outMat4)
{
    if (!a || !b || !outMat4) return SAILOR_HIP_ERR_INVALID_ARGUMENT;
    store(mul(load(a), load(b)), outMat4);
    return SAILOR_HIP_OK;
}

int sailor_host_transform_matrix(const SailorTransform* t, float* outMat4)
{
    if (!t || !outMat4) return SAILOR_HIP_ERR_INVALID_ARGUMENT;
    store(transform_matrix(*t), outMat4);
    return SAILOR_HIP_OK;
}

int sailor_host_fill_frame_data(const float* cameraWorld, float fovDegrees, float aspect, float zNear, float zFar,
                                int32_t viewportWidth, int32_t viewportHeight, float currentTime, float deltaTime, SailorUboFrameData* outFrame)
{
    if (!cameraWorld || !outFrame || viewportWidth <= 0 || viewportHeight <= 0) return SAILOR_HIP_ERR_INVALID_ARGUMENT;
    const M4 world = load(cameraWorld);
    const M4 view = mul(identity(), inverse(world)); // CameraECS.cpp:19-20: origin * inverse(world)
    const M4 proj = perspective_rh_zo(fovDegrees * 0.01745329251994329576923690768489f, aspect, zFar, zNear); // CameraComponent: radians(fov)
    const M4 invProj = inverse(proj);                                                                          // CameraECS.cpp:31-34
    store(view, outFrame->view);
    store(proj, outFrame->projection);
    store(invProj, outFrame->invProjection);
    outFrame->cameraPosition[0] = world[3].x; outFrame->cameraPosition[1] = world[3].y;
    outFrame->cameraPosition[2] = world[3].z; outFrame->cameraPosition[3] = world[3].w;
    outFrame->viewportSize[0] = viewportWidth; outFrame->viewportSize[1] = viewportHeight;
    outFrame->cameraZNearZFar[0] = zNear; outFrame->cameraZNearZFar[1] = zFar;
    outFrame->currentTime = currentTime; outFrame->deltaTime = deltaTime;
    return SAILOR_HIP_OK;
}

int sailor_host_extract_frustum_planes(const float* worldMatrix, float aspect, float fovYDegrees, float zNear, float zFar, float* outPlanes24, float* outCorners24)
{
    if (!worldMatrix || !outPlanes24) return SAILOR_HIP_ERR_INVALID_ARGUMENT;
    Frustum f;
    f.extract(load(worldMatrix), aspect, fovYDegrees, zNear, zFar);
    for (int i = 0; i < 6; i++) std::memcpy(outPlanes24 + 4 * i, &f.planes[i].abcd, 16);
    if (outCorners24) std::memcpy(outCorners24, f.corners, sizeof f.corners);
    return SAILOR_HIP_OK;
}

int sailor_host_extract_frustum_planes_matrix(const float* projectionViewMatrix, float* outPlanes24, float* outCorners24)
{
    if (!projectionViewMatrix || !outPlanes24) return SAILOR_HIP_ERR_INVALID_ARGUMENT;
    Frustum f;
    f.extract(load(projectionViewMatrix));
    for (int i = 0; i < 6; i++) std::memcpy(outPlanes24 + 4 * i, &f.planes[i].abcd, 16);
    if (outCorners24) std::memcpy(outCorners24, f.corners, sizeof f.corners);
    return SAILOR_HIP_OK;
}

int sailor_host_csm_matrices(const float* lightView, const float* cameraWorld, float aspect, float fovYDegrees, float cameraNear, float cameraFar, float* outMatrices64)
{
    if (!lightView || !cameraWorld || !outMatrices64) return SAILOR_HIP_ERR_INVALID_ARGUMENT;
    const M4 lv = load(lightView), cw = load(cameraWorld);
    for (int k = 0; k < SAILOR_NUM_CSM_CASCADES; k++) {
        // ShadowPrepassNode.cpp:392-400: cascade 0 spans [near, far*L0], cascade k spans [far*L(k-1), far*Lk]; zMult 10
        const float n = k == 0 ? cameraNear : cameraFar * kShadowCascadeLevels[k - 1];
        const float f = cameraFar * kShadowCascadeLevels[k];
        Frustum fr;
        fr.extract(cw, aspect, fovYDegrees, n, f);
        const M4 ortho = fr.ortho_by_view(lv, 10.0f);
        store(mul(ortho, lv), outMatrices64 + 16 * k); // LightingECS.cpp:292
    }
    return SAILOR_HIP_OK;
}

int sailor_host_pack_light(uint32_t type, uint32_t shadowType, const float* worldPosition, const float* direction, const float* intensity,
                           const float* attenuation, const float* cutOffDegrees, const float* bounds, SailorLightShaderData* outLight)
{
    if (!worldPosition || !direction || !intensity || !attenuation || !cutOffDegrees || !bounds || !outLight) return SAILOR_HIP_ERR_INVALID_ARGUMENT;
    std::memset(outLight, 0, sizeof *outLight);
    outLight->type = type;
    outLight->shadowType = shadowType;
    for (int i = 0; i < 3; i++) {
        outLight->worldPosition[i] = worldPosition[i];
        outLight->direction[i] = direction[i];
        outLight->intensity[i] = intensity[i];
        outLight->attenuation[i] = attenuation[i];
        outLight->bounds[i] = bounds[i];
    }
    // LightingECS.cpp:171: vec2(cos(radians(cutOff.x)), cos(radians(cutOff.y)))
    outLight->cutOff[0] = std::cos(cutOffDegrees[0] * 0.01745329251994329576923690768489f);
    outLight->cutOff[1] = std::cos(cutOffDegrees[1] * 0.01745329251994329576923690768489f);
    return SAILOR_HIP_OK;
}

// ---- Math/Bounds.cpp:211-243: the scalar sphere tests; their only caller is LightingECS::GetLightsInFrustum (ContainsSphere, :237) ----
int sailor_host_overlaps_sphere(const float* planes24, const float* sphere4)
{
    if (!planes24 || !sphere4) return SAILOR_HIP_ERR_INVALID_ARGUMENT;
    int res = 1;
    for (int p = 0; p < 6; p++)
        if (planes24[4 * p + 0] * sphere4[0] + planes24[4 * p + 1] * sphere4[1] + planes24[4 * p + 2] * sphere4[2] + planes24[4 * p + 3] < -sphere4[3]) res = 0;
    return res;
}

int sailor_host_contains_sphere(const float* planes24, const float* sphere4)
{
    if (!planes24 || !sphere4) return SAILOR_HIP_ERR_INVALID_ARGUMENT;
    int res = 1;
    for (int p = 0; p < 6; p++)
        if (planes24[4 * p + 0] * sphere4[0] + planes24[4 * p + 1] * sphere4[1] + planes24[4 * p + 2] * sphere4[2] + planes24[4 * p + 3] < sphere4[3]) res = 0;
    return res;
}

// ---- ECS/LightingECS.cpp:209-260 GetLightsInFrustum: the shadow-casting lights of a view.  Directional lights in component order; point
// and spot lights whose bounding sphere (radius = the largest bound) lies INSIDE the frustum, each list sorted by distance to the camera
// through the reference's own insertion: position = std::lower_bound(list, proxy) under operator< on the distance, i.e. a light goes
// in FRONT of the lights already there at the same distance.
int sailor_host_lights_in_frustum(const float* planes24, const float* cameraPosition3, uint32_t numLights, const uint32_t* types, const uint32_t* shadowTypes,
                                  const uint8_t* active, const float* positions3, const float* bounds3,
                                  uint32_t* outDirectional, uint32_t* outNumDirectional,
                                  uint32_t* outPoint, float* outPointDistance, uint32_t* outNumPoint,
                                  uint32_t* outSpot, float* outSpotDistance, uint32_t* outNumSpot)
{
    if (!planes24 || !cameraPosition3 || (numLights && (!types || !shadowTypes || !positions3 || !bounds3)) || !outDirectional || !outNumDirectional || !outPoint ||
        !outPointDistance || !outNumPoint || !outSpot || !outSpotDistance || !outNumSpot)
        return SAILOR_HIP_ERR_INVALID_ARGUMENT;
    uint32_t nd = 0, np = 0, ns = 0;
    auto insert = [](uint32_t* idx, float* dist, uint32_t& n, uint32_t index, float d) {
        uint32_t lo = 0, hi = n; // lower_bound: first element that is not < d
        while (lo < hi) { const uint32_t mid = (lo + hi) / 2; if (dist[mid] < d) lo = mid + 1; else hi = mid; }
        for (uint32_t k = n; k > lo; k--) { idx[k] = idx[k - 1]; dist[k] = dist[k - 1]; }
        idx[lo] = index; dist[lo] = d; n++;
    };
    for (uint32_t i = 0; i < numLights; i++) {
        if (shadowTypes[i] == 0u /* EShadowType::None */ || (active && !active[i])) continue; // :222
        if (types[i] == 0u /* Directional */) { outDirectional[nd++] = i; continue; }           // :255
        const float* b = bounds3 + 3 * (size_t)i;
        const float* p = positions3 + 3 * (size_t)i;
        const float sphere[4] = { p[0], p[1], p[2], std::max(std::max(b[0], b[1]), b[2]) };     // :235
        if (!sailor_host_contains_sphere(planes24, sphere)) continue;                            // :237
        const float dx = p[0] - cameraPosition3[0], dy = p[1] - cameraPosition3[1], dz = p[2] - cameraPosition3[2];
        const float d = std::sqrt((dx * dx + dy * dy) + dz * dz);                                // :240 glm::length
        if (types[i] == 2u /* Spot */) insert(outSpot, outSpotDistance, ns, i, d);
        else insert(outPoint, outPointDistance, np, i, d);
    }
    *outNumDirectional = nd; *outNumPoint = np; *outNumSpot = ns;
    return SAILOR_HIP_OK;
}

} // extern "C"

// ---- ECS/LightingECS.cpp:14-38 CSMLightState + :299-366, the change tracking of LightingECS::PrepareCSMPasses -------------------------
namespace {
struct CsmLightState {
    bool hasView = false;
    SailorCsmView view {};
    std::vector<uint32_t> meshes;  // m_snapshot: (static mesh index, frame the mesh last changed)
    std::vector<uint64_t> frames;
};
static void quat_rotate(const float* q, const float* v, float* o) // glm: v + ((cross(q.xyz, v) * q.w) + cross(q.xyz, cross(q.xyz, v))) * 2 (Transform::GetForward)
{
    const float uv[3] = { q[1] * v[2] - v[1] * q[2], q[2] * v[0] - v[2] * q[0], q[0] * v[1] - v[0] * q[1] };
    const float uuv[3] = { q[1] * uv[2] - uv[1] * q[2], q[2] * uv[0] - uv[2] * q[0], q[0] * uv[1] - uv[0] * q[1] };
    for (int i = 0; i < 3; i++) o[i] = v[i] + ((uv[i] * q[3]) + uuv[i]) * 2.0f;
}
// the transform half of CSMLightState::Equals (:19-24)
static bool view_unchanged(bool hasA, const SailorCsmView& a, bool hasB, const SailorCsmView& b)
{
    if (!hasA || !hasB) return !hasA && !hasB;
    if (a.componentIndex != b.componentIndex) return false;
    float d2 = 0.0f;
    for (int i = 0; i < 4; i++) { const float d = a.cameraPosition[i] - b.cameraPosition[i]; d2 += d * d; }
    if (std::sqrt(d2) > 15.0f) return false;                         // CameraPosDelta
    const float fwd[3] = { 0.0f, 0.0f, -1.0f };
    float fa[3], fb[3];
    quat_rotate(a.cameraRotation, fwd, fa); quat_rotate(b.cameraRotation, fwd, fb);
    if ((fa[0] * fb[0] + fa[1] * fb[1]) + fa[2] * fb[2] < 0.9995f) return false; // CameraRotationDelta
    for (int i = 0; i < 4; i++)
        if (a.lightPosition[i] != b.lightPosition[i] || a.lightRotation[i] != b.lightRotation[i]) return false;
    return true;
}
} // namespace

struct SailorCsmSnapshots { std::vector<CsmLightState> states; }; // LightingECS::m_csmSnapshots

extern "C" {

SailorCsmSnapshots* sailor_host_csm_snapshots_create(void) { return new (std::nothrow) SailorCsmSnapshots(); }
SailorCsmSnapshots* sailor_host_csm_snapshots_clone(const SailorCsmSnapshots* s) { return s ? new (std::nothrow) SailorCsmSnapshots(*s) : nullptr; }
void sailor_host_csm_snapshots_destroy(SailorCsmSnapshots* s) { delete s; }
uint32_t sailor_host_csm_snapshots_count(const SailorCsmSnapshots* s) { return s ? (uint32_t)s->states.size() : 0u; }

int sailor_host_csm_snapshot_get(const SailorCsmSnapshots* s, uint32_t k, uint32_t capacity, uint32_t* outCount, uint32_t* outMeshes, uint64_t* outFrames,
                                 int32_t* outHasView, SailorCsmView* outView)
{
    if (!s || k >= s->states.size() || !outCount) return SAILOR_HIP_ERR_INVALID_ARGUMENT;
    const CsmLightState& st = s->states[k];
    *outCount = (uint32_t)st.meshes.size();
    for (uint32_t i = 0; i < st.meshes.size() && i < capacity; i++) {
        if (outMeshes) outMeshes[i] = st.meshes[i];
        if (outFrames) outFrames[i] = st.frames[i];
    }
    if (outHasView) *outHasView = st.hasView ? 1 : 0;
    if (outView) *outView = st.view;
    return SAILOR_HIP_OK;
}

int sailor_host_csm_plan_passes(SailorCsmSnapshots* state, uint32_t firstSnapshot, uint32_t numCascades, uint32_t numEntities, const uint64_t* overlapMasks,
                                const uint32_t* shadowTypes, const uint64_t* lastChangedFrame, const SailorCsmView* view, uint32_t* outRender, uint64_t* outMasks)
{
    if (!state || !overlapMasks || !shadowTypes || (numEntities && !lastChangedFrame) || !outRender || !outMasks || numCascades > 16) return SAILOR_HIP_ERR_INVALID_ARGUMENT;
    const size_t words = ((size_t)numEntities + 63) / 64;
    uint32_t added[16];                       // bCascadeAdded[z]: the shadow type of a cascade that is rendered this frame, 0 = None
    uint32_t snapshotIndex = firstSnapshot;
    for (uint32_t k = 0; k < numCascades; k++) {
        added[k] = 0u;
        uint64_t* m = outMasks + (size_t)k * words;
        std::memcpy(m, overlapMasks + (size_t)k * words, words * 8);
        for (uint32_t z = 0; z < k; z++)       // :310-327 "don't duplicate data for higher cascades"
            if (added[z] != 0u && added[z] == shadowTypes[k])
                for (size_t w = 0; w < words; w++) m[w] &= ~overlapMasks[(size_t)z * words + w];
        CsmLightState snap;                    // :334-347
        snap.hasView = view != nullptr;
        if (view) snap.view = *view;
        for (uint32_t i = 0; i < numEntities; i++)
            if ((m[i >> 6] >> (i & 63)) & 1ull) { snap.meshes.push_back(i); snap.frames.push_back(lastChangedFrame[i]); }
        bool same = false;
        if (snapshotIndex < state->states.size()) {
            const CsmLightState& old = state->states[snapshotIndex];
            same = old.meshes == snap.meshes && old.frames == snap.frames;
            if (same && (snap.hasView || old.hasView)) same = view_unchanged(old.hasView, old.view, snap.hasView, snap.view);
            if (!same) state->states[snapshotIndex] = std::move(snap); // :358
        } else {
            state->states.resize(snapshotIndex);  // (a gap can only come from a caller that skips indices: filled with empty states)
            state->states.push_back(std::move(snap)); // :363
        }
        outRender[k] = same ? 0u : 1u;         // an equal snapshot is kept as it is, camera included (:353-357)
        if (!same) added[k] = shadowTypes[k];
        snapshotIndex++;
    }
    return SAILOR_HIP_OK;
}

} // extern "C"
