/*
 * sailor_oracle.c -- CPU restatement of the reference's Forward+ lighting path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under oracle/ is part of the product: only tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this library, and there only
 * as the checker / the CPU baseline -- never as the thing shipped or measured as the GPU path.
 *
 * PARITY UNPINNED: aantropov/Sailor ships no tests, golden vectors or fixtures for this path
 *  and cannot be built here (Win32 + MSVC + Vulkan + un-vendored glm).
 * This file therefore restates the algorithm from the reference's shader / C++ text, function by
 * function, with the reference file:line each function follows.  The only reference-owned facts
 * it can be checked against are the constants and layouts of SURVEY.md Appendix B (tests do so).
 *
 * Third-party arithmetic not present under /root/reference: glm (git submodule External/glm,
 * .gitmodules:4-6, commit unrecorded).  The glm routines used by the path (mat4*mat4, mat4*vec4,
 * inverse, perspectiveRH_ZO, orthoRH_NO, mat4_cast, translate, scale, normalize, cross, dot)
 * are restated from glm's published algorithms (0.9.9 / 1.0 series headers) below.
 *
 * Canonical numeric rules: IEEE-754 binary32, round-to-nearest-even, NO FMA
 * contraction (build with -ffp-contract=off -fno-fast-math), evaluation in the order written.
 *   GLSL  mat4*vec4 :  ((c0*x + c1*y) + c2*z) + c3*w          (shader-side code)
 *   glm   mat4*vec4 :  (c0*x + c1*y) + (c2*z + c3*w)          (host-side code, glm's own order)
 *   dot3            :  (a.x*b.x + a.y*b.y) + a.z*b.z
 *   length          :  sqrtf(dot)          normalize (GLSL) : v / length(v)
 *   glm::normalize  :  v * (1.0f / sqrtf(dot(v,v)))
 */
#include <stdint.h>
#include <stddef.h>
#include <string.h>
#include <stdlib.h>
#include <math.h>
#include <float.h>
#include <xmmintrin.h>
#include <emmintrin.h>

#define ORACLE_API __attribute__((visibility("default")))

/* Content/Shaders/Constants.glsl:13-15,23-24 ; FrameGraph/LightCullingNode.h:15-16 */
#define TILE 16
#define CAND 196
#define KEEP 128
#define NUM_CASCADES 4

ORACLE_API int oracle_const_tile_size(void) { return TILE; }
ORACLE_API int oracle_const_candidates_per_tile(void) { return CAND; }
ORACLE_API int oracle_const_lights_per_tile(void) { return KEEP; }
ORACLE_API int oracle_const_num_cascades(void) { return NUM_CASCADES; }
/* Constants.glsl:24 (GLSL literal; the C++ side uses 1/20,1/10,1/3,1/2 -- ECS/LightingECS.h:66) */
static const float kShadowCascadeLevelsGlsl[4] = { 0.05f, 0.1f, 0.333333f, 0.5f };
static const float kShadowCascadeLevelsCpp[4] = { 1.0f / 20.0f, 1.0f / 10.0f, 1.0f / 3.0f, 1.0f / 2.0f };
ORACLE_API float oracle_const_cascade_level_glsl(int i) { return kShadowCascadeLevelsGlsl[i]; }
ORACLE_API float oracle_const_cascade_level_cpp(int i) { return kShadowCascadeLevelsCpp[i]; }

/* ------------------------------------------------------------------------------------------- */
/* Layouts                                                               */
/* ------------------------------------------------------------------------------------------- */

/* RHI/Types.h:751-761 -- 232 bytes */
typedef struct {
    float view[16];
    float projection[16];
    float invProjection[16];
    float cameraPosition[4];
    int32_t viewportSize[2];
    float cameraZNearZFar[2];
    float currentTime;
    float deltaTime;
} UboFrameData;

/* Lighting.glsl:4-15 == ECS/LightingECS.h:71-81 -- std430, stride 112 */
typedef struct {
    uint32_t type;          /* @0  */
    uint32_t shadowType;    /* @4  */
    uint32_t _pad0[2];
    float worldPosition[3]; /* @16 */
    float _pad1;
    float direction[3];     /* @32 */
    float _pad2;
    float intensity[3];     /* @48 */
    float _pad3;
    float attenuation[3];   /* @64 */
    float _pad4;
    float cutOff[2];        /* @80 */
    float _pad5[2];
    float bounds[3];        /* @96 */
    float _pad6;
} LightData;

ORACLE_API int oracle_sizeof_ubo(void) { return (int)sizeof(UboFrameData); }
ORACLE_API int oracle_sizeof_light(void) { return (int)sizeof(LightData); }
ORACLE_API int oracle_offsetof_light(int field)
{
    switch (field) {
    case 0: return (int)offsetof(LightData, type);
    case 1: return (int)offsetof(LightData, shadowType);
    case 2: return (int)offsetof(LightData, worldPosition);
    case 3: return (int)offsetof(LightData, direction);
    case 4: return (int)offsetof(LightData, intensity);
    case 5: return (int)offsetof(LightData, attenuation);
    case 6: return (int)offsetof(LightData, cutOff);
    case 7: return (int)offsetof(LightData, bounds);
    }
    return -1;
}

/* ------------------------------------------------------------------------------------------- */
/* Small vector helpers                                                                         */
/* ------------------------------------------------------------------------------------------- */

static inline float dot3(const float* a, const float* b) { return (a[0] * b[0] + a[1] * b[1]) + a[2] * b[2]; }
static inline float length3(const float* a) { return sqrtf(dot3(a, a)); }
static inline void cross3(const float* a, const float* b, float* o)
{
    /* GLSL / glm cross: (a.y*b.z - b.y*a.z, a.z*b.x - b.z*a.x, a.x*b.y - b.x*a.y) */
    float x = a[1] * b[2] - b[1] * a[2];
    float y = a[2] * b[0] - b[2] * a[0];
    float z = a[0] * b[1] - b[0] * a[1];
    o[0] = x; o[1] = y; o[2] = z;
}
static inline void normalize3_glsl(const float* a, float* o)
{
    float l = length3(a);
    o[0] = a[0] / l; o[1] = a[1] / l; o[2] = a[2] / l;
}
static inline void normalize3_glm(const float* a, float* o)
{
    /* glm::normalize(v) = v * inversesqrt(dot(v, v)); inversesqrt(x) = 1 / sqrt(x) */
    float inv = 1.0f / sqrtf(dot3(a, a));
    o[0] = a[0] * inv; o[1] = a[1] * inv; o[2] = a[2] * inv;
}
/* normalize() inside the SHADE path (K2).  GLSL leaves normalize's rounding to the implementation; the Vulkan spec's
 * "Precision of GLSL.std.450 instructions" table defines it as inherited from x * inversesqrt(dot(x, x)).  K2 is checked
 * to a tolerance, but NdfGGX's denominator (cosLh^2 (a^2 - 1) + 1) cancels catastrophically at the specular peak of
 * smooth surfaces, amplifying a 1-ulp difference in Lh ~10^4 times -- so the half-vector chain is specified exactly:
 * inversesqrt(x) = 1.0f / sqrtf(x), both correctly rounded, then three multiplies.  (K1 keeps SURVEY 8c's v / length(v).) */
static inline void normalize3_vk(const float* a, float* o)
{
    const float inv = 1.0f / sqrtf(dot3(a, a));
    o[0] = a[0] * inv; o[1] = a[1] * inv; o[2] = a[2] * inv;
}
static inline float clampf(float x, float lo, float hi) { return fminf(fmaxf(x, lo), hi); }

/* column-major mat4: element (col c, row r) = m[c*4 + r] */
static inline void glsl_mat4_mul_vec4(const float* m, const float* v, float* o)
{
    float r[4];
    for (int i = 0; i < 4; i++)
        r[i] = ((m[0 * 4 + i] * v[0] + m[1 * 4 + i] * v[1]) + m[2 * 4 + i] * v[2]) + m[3 * 4 + i] * v[3];
    o[0] = r[0]; o[1] = r[1]; o[2] = r[2]; o[3] = r[3];
}
static inline void glm_mat4_mul_vec4(const float* m, const float* v, float* o)
{
    /* glm type_mat4x4.inl operator*(mat4, vec4): Add0 = m0*v0 + m1*v1; Add1 = m2*v2 + m3*v3; Add0 + Add1 */
    float r[4];
    for (int i = 0; i < 4; i++)
        r[i] = (m[0 * 4 + i] * v[0] + m[1 * 4 + i] * v[1]) + (m[2 * 4 + i] * v[2] + m[3 * 4 + i] * v[3]);
    o[0] = r[0]; o[1] = r[1]; o[2] = r[2]; o[3] = r[3];
}
static inline void glm_mat4_mul_mat4(const float* a, const float* b, float* o)
{
    /* glm operator*(mat4, mat4): Result[c] = A0*B[c][0] + A1*B[c][1] + A2*B[c][2] + A3*B[c][3], left to right */
    float r[16];
    for (int c = 0; c < 4; c++)
        for (int i = 0; i < 4; i++)
            r[c * 4 + i] = ((a[0 * 4 + i] * b[c * 4 + 0] + a[1 * 4 + i] * b[c * 4 + 1]) + a[2 * 4 + i] * b[c * 4 + 2]) + a[3 * 4 + i] * b[c * 4 + 3];
    memcpy(o, r, sizeof r);
}

ORACLE_API void oracle_mat4_mul_mat4(const float* a, const float* b, float* o) { glm_mat4_mul_mat4(a, b, o); }

/* glm::inverse(mat4) -- glm/detail/func_matrix.inl compute_inverse<4,4> (cofactor form).
 * Call sites: ECS/CameraECS.cpp:20,33,38 ; Math/Bounds.cpp:113 ; ECS/LightingECS.cpp:227 */
ORACLE_API void oracle_mat4_inverse(const float* m_, float* out)
{
#define M(c, r) m_[(c) * 4 + (r)]
    float Coef00 = M(2, 2) * M(3, 3) - M(3, 2) * M(2, 3);
    float Coef02 = M(1, 2) * M(3, 3) - M(3, 2) * M(1, 3);
    float Coef03 = M(1, 2) * M(2, 3) - M(2, 2) * M(1, 3);
    float Coef04 = M(2, 1) * M(3, 3) - M(3, 1) * M(2, 3);
    float Coef06 = M(1, 1) * M(3, 3) - M(3, 1) * M(1, 3);
    float Coef07 = M(1, 1) * M(2, 3) - M(2, 1) * M(1, 3);
    float Coef08 = M(2, 1) * M(3, 2) - M(3, 1) * M(2, 2);
    float Coef10 = M(1, 1) * M(3, 2) - M(3, 1) * M(1, 2);
    float Coef11 = M(1, 1) * M(2, 2) - M(2, 1) * M(1, 2);
    float Coef12 = M(2, 0) * M(3, 3) - M(3, 0) * M(2, 3);
    float Coef14 = M(1, 0) * M(3, 3) - M(3, 0) * M(1, 3);
    float Coef15 = M(1, 0) * M(2, 3) - M(2, 0) * M(1, 3);
    float Coef16 = M(2, 0) * M(3, 2) - M(3, 0) * M(2, 2);
    float Coef18 = M(1, 0) * M(3, 2) - M(3, 0) * M(1, 2);
    float Coef19 = M(1, 0) * M(2, 2) - M(2, 0) * M(1, 2);
    float Coef20 = M(2, 0) * M(3, 1) - M(3, 0) * M(2, 1);
    float Coef22 = M(1, 0) * M(3, 1) - M(3, 0) * M(1, 1);
    float Coef23 = M(1, 0) * M(2, 1) - M(2, 0) * M(1, 1);

    float Fac0[4] = { Coef00, Coef00, Coef02, Coef03 };
    float Fac1[4] = { Coef04, Coef04, Coef06, Coef07 };
    float Fac2[4] = { Coef08, Coef08, Coef10, Coef11 };
    float Fac3[4] = { Coef12, Coef12, Coef14, Coef15 };
    float Fac4[4] = { Coef16, Coef16, Coef18, Coef19 };
    float Fac5[4] = { Coef20, Coef20, Coef22, Coef23 };

    float Vec0[4] = { M(1, 0), M(0, 0), M(0, 0), M(0, 0) };
    float Vec1[4] = { M(1, 1), M(0, 1), M(0, 1), M(0, 1) };
    float Vec2[4] = { M(1, 2), M(0, 2), M(0, 2), M(0, 2) };
    float Vec3[4] = { M(1, 3), M(0, 3), M(0, 3), M(0, 3) };

    static const float SignA[4] = { +1, -1, +1, -1 };
    static const float SignB[4] = { -1, +1, -1, +1 };
    float Inv[16];
    for (int i = 0; i < 4; i++) {
        float Inv0 = (Vec1[i] * Fac0[i] - Vec2[i] * Fac1[i]) + Vec3[i] * Fac2[i];
        float Inv1 = (Vec0[i] * Fac0[i] - Vec2[i] * Fac3[i]) + Vec3[i] * Fac4[i];
        float Inv2 = (Vec0[i] * Fac1[i] - Vec1[i] * Fac3[i]) + Vec3[i] * Fac5[i];
        float Inv3 = (Vec0[i] * Fac2[i] - Vec1[i] * Fac4[i]) + Vec2[i] * Fac5[i];
        Inv[0 * 4 + i] = Inv0 * SignA[i];
        Inv[1 * 4 + i] = Inv1 * SignB[i];
        Inv[2 * 4 + i] = Inv2 * SignA[i];
        Inv[3 * 4 + i] = Inv3 * SignB[i];
    }
    float Row0[4] = { Inv[0], Inv[4], Inv[8], Inv[12] };
    float Dot0[4] = { M(0, 0) * Row0[0], M(0, 1) * Row0[1], M(0, 2) * Row0[2], M(0, 3) * Row0[3] };
    float Dot1 = (Dot0[0] + Dot0[1]) + (Dot0[2] + Dot0[3]);
    float OneOverDeterminant = 1.0f / Dot1;
    for (int i = 0; i < 16; i++) out[i] = Inv[i] * OneOverDeterminant;
#undef M
}

/* Math/Math.cpp:18-21: PerspectiveRH(fov, aspect, zNear, zFar) = glm::perspectiveRH(fov, aspect, zFar, zNear)
 * under GLM_FORCE_DEPTH_ZERO_TO_ONE + GLM_FORCE_RIGHT_HANDED (Core/Defines.h:17-26) = perspectiveRH_ZO
 * with near/far swapped (reversed Z). */
ORACLE_API void oracle_perspective_rh_reversed_z(float fovRadians, float aspect, float zNear, float zFar, float* out)
{
    /* glm::perspectiveRH_ZO(fovy, aspect, zNear', zFar') with zNear' = zFar, zFar' = zNear */
    const float n = zFar, f = zNear;
    const float tanHalfFovy = tanf(fovRadians / 2.0f);
    memset(out, 0, 16 * sizeof(float));
    out[0 * 4 + 0] = 1.0f / (aspect * tanHalfFovy);
    out[1 * 4 + 1] = 1.0f / (tanHalfFovy);
    out[2 * 4 + 2] = f / (n - f);
    out[2 * 4 + 3] = -1.0f;
    out[3 * 4 + 2] = -(f * n) / (f - n);
}

/* glm::orthoRH_NO(left, right, bottom, top, zNear, zFar) -- glm/ext/matrix_clip_space.inl */
static void ortho_rh_no(float left, float right, float bottom, float top, float zNear, float zFar, float* out)
{
    memset(out, 0, 16 * sizeof(float));
    out[0] = 1.0f; out[5] = 1.0f; out[10] = 1.0f; out[15] = 1.0f;
    out[0 * 4 + 0] = 2.0f / (right - left);
    out[1 * 4 + 1] = 2.0f / (top - bottom);
    out[2 * 4 + 2] = -2.0f / (zFar - zNear);
    out[3 * 4 + 0] = -(right + left) / (right - left);
    out[3 * 4 + 1] = -(top + bottom) / (top - bottom);
    out[3 * 4 + 2] = -(zFar + zNear) / (zFar - zNear);
}

/* ------------------------------------------------------------------------------------------- */
/* K1 -- tile light cull: Content/Shaders/ComputeLightCulling.shader:49-240 under the canonical  */
/* sequential semantics of SURVEY.md Appendix A                                                  */
/* ------------------------------------------------------------------------------------------- */

typedef struct {
    float planes[4][4];
    float center[2];
} ViewFrustum; /* Math.glsl:116-120 */

/* Math.glsl:143-154 ClipSpaceToViewSpace + :164-173 ScreenSpaceToViewSpace(vec4, vec2, mat4) */
static void screen_space_to_view_space(const float* screen, int vpW, int vpH, const float* invProjection, float* out)
{
    float tx = screen[0] / (float)vpW;
    float ty = screen[1] / (float)vpH;
    float clip[4] = { tx * 2.0f - 1.0f, ty * 2.0f - 1.0f, screen[2], screen[3] };
    float v[4];
    glsl_mat4_mul_vec4(invProjection, clip, v);
    float w = v[3];
    v[0] = v[0] / w; v[1] = v[1] / w; v[2] = v[2] / w; v[3] = v[3] / w;
    v[2] = v[2] * -1.0f;
    out[0] = v[0]; out[1] = v[1]; out[2] = v[2]; out[3] = v[3];
}

/* Math.glsl:122-134 ComputePlane */
static void compute_plane(const float* p0, const float* p1, const float* p2, float* plane)
{
    float v0[3] = { p1[0] - p0[0], p1[1] - p0[1], p1[2] - p0[2] };
    float v2[3] = { p2[0] - p0[0], p2[1] - p0[1], p2[2] - p0[2] };
    float c[3];
    cross3(v0, v2, c);
    normalize3_glsl(c, plane);
    plane[3] = dot3(plane, p0);
}

/* ComputeLightCulling.shader:57-95 CreateFrustum(tileId); also Math.glsl:185-222 CreateViewFrustum when
 * called with the whole viewport as one "tile" */
static void create_frustum_rect(float x0, float y0, float x1, float y1, int vpW, int vpH, const float* invProjection, ViewFrustum* f)
{
    const float eye[3] = { 0.0f, 0.0f, 0.0f };
    float ss[5][4] = {
        { x0, y0, -1.0f, 1.0f }, /* "top left"     */
        { x1, y0, -1.0f, 1.0f }, /* "top right"    */
        { x0, y1, -1.0f, 1.0f }, /* "bottom left"  */
        { x1, y1, -1.0f, 1.0f }, /* "bottom right" */
        { 0, 0, 0, 0 }
    };
    for (int k = 0; k < 4; k++) ss[4][k] = (ss[0][k] + ss[3][k]) * 0.5f;
    float vs[5][4];
    for (int i = 0; i < 5; i++) screen_space_to_view_space(ss[i], vpW, vpH, invProjection, vs[i]);
    compute_plane(eye, vs[2], vs[0], f->planes[0]); /* left   */
    compute_plane(eye, vs[1], vs[3], f->planes[1]); /* right  */
    compute_plane(eye, vs[0], vs[1], f->planes[2]); /* top    */
    compute_plane(eye, vs[3], vs[2], f->planes[3]); /* bottom */
    f->center[0] = vs[4][0];
    f->center[1] = vs[4][1];
}

static void create_tile_frustum(int tx, int ty, int vpW, int vpH, const float* invProjection, ViewFrustum* f)
{
    /* ivec2 * int -> float conversions are exact for any sane viewport */
    create_frustum_rect((float)(tx * TILE), (float)(ty * TILE), (float)((tx + 1) * TILE), (float)((ty + 1) * TILE),
                        vpW, vpH, invProjection, f);
}

/* Math.glsl:224-239 */
static int sphere_frustum_overlaps(const float* p, float radius, const ViewFrustum* f, float zNear, float zFar)
{
    if (p[2] - radius > zNear || p[2] + radius < zFar) return 0;
    for (int i = 0; i < 4; i++)
        if (dot3(f->planes[i], p) - f->planes[i][3] < -radius) return 0;
    return 1;
}

ORACLE_API void oracle_tile_frustum(const void* ubo_, int tx, int ty, float* outPlanes16, float* outCenter2)
{
    const UboFrameData* ubo = (const UboFrameData*)ubo_;
    ViewFrustum f;
    create_tile_frustum(tx, ty, ubo->viewportSize[0], ubo->viewportSize[1], ubo->invProjection, &f);
    memcpy(outPlanes16, f.planes, sizeof f.planes);
    outCenter2[0] = f.center[0]; outCenter2[1] = f.center[1];
}

/* Appendix A step 1: depth bounds of one tile (ComputeLightCulling.shader:119-128).  The shader compares
 * float BITS as uints (atomicMin/atomicMax on floatBitsToUint); reproduced literally. */
static void tile_depth_bounds(const float* depth, int W, int H, int tx, int ty, float* outMin, float* outMax)
{
    uint32_t mn = 0xFFFFFFFFu, mx = 0u;
    for (int ly = 0; ly < TILE; ly++) {
        int gy = TILE * ty + ly;
        int row = H - 1 - gy;
        if (row < 0) row = 0;
        if (row > H - 1) row = H - 1;
        for (int lx = 0; lx < TILE; lx++) {
            int gx = TILE * tx + lx;
            int col = gx < W - 1 ? gx : W - 1;
            uint32_t bits;
            memcpy(&bits, &depth[(size_t)row * W + col], 4);
            if (bits > mx) mx = bits;
            if (bits < mn) mn = bits;
        }
    }
    memcpy(outMin, &mn, 4);
    memcpy(outMax, &mx, 4);
}

ORACLE_API void oracle_tile_depth_bounds(const float* depth, int W, int H, int tx, int ty, float* outMinMax)
{
    tile_depth_bounds(depth, W, H, tx, ty, &outMinMax[0], &outMinMax[1]);
}

/* Appendix A step 4, literal form: ComputeLightCulling.shader:198-225 */
static void select_literal_bubble(uint32_t* idx, float* imp, uint32_t n)
{
    uint32_t numSorted = KEEP;
    for (uint32_t i = 0; i + 1 < n; i++) {
        for (uint32_t j = 0; j < n - i - 1; j++) {
            if (imp[j] < imp[j + 1]) {
                float v = imp[j]; imp[j] = imp[j + 1]; imp[j + 1] = v;
                uint32_t t = idx[j]; idx[j] = idx[j + 1]; idx[j + 1] = t;
            }
        }
        --numSorted;
        if (numSorted == 0) break;
    }
}

/* Appendix A step 4, closed form: emitted list[i] (i < 128) = candidate of rank i under the total order
 * (impact ascending, candidate position descending). */
static void select_closed_form(const uint32_t* idx, const float* imp, uint32_t n, uint32_t* list)
{
    for (uint32_t k = 0; k < n; k++) {
        uint32_t rank = 0;
        for (uint32_t q = 0; q < n; q++)
            if (imp[q] < imp[k] || (imp[q] == imp[k] && q > k)) rank++;
        if (rank < KEEP) list[rank] = idx[k];
    }
}

ORACLE_API void oracle_select_emit(const uint32_t* candIdx, const float* candImpact, uint32_t n, int literal, uint32_t* outList, uint32_t* outNum)
{
    uint32_t idx[CAND]; float imp[CAND];
    memcpy(idx, candIdx, n * 4); memcpy(imp, candImpact, n * 4);
    uint32_t num = n < KEEP ? n : KEEP;
    int anyNaN = 0;
    for (uint32_t k = 0; k < n; k++) anyNaN |= (imp[k] != imp[k]);
    /* The closed form presumes a total order.  A NaN impact (a tile with nothing drawn: depth +inf -> frustum centre NaN;
     * or a non-finite light) has none -- the shader's compare at :207 is false next to it -- so such tiles always take
     * the literal sort. */
    if (n > KEEP && !literal && !anyNaN) {
        select_closed_form(idx, imp, n, outList);
    } else {
        if (n > KEEP) select_literal_bubble(idx, imp, n);
        for (uint32_t i = 0; i < num; i++) outList[i] = idx[n - i - 1]; /* :235-238 */
    }
    *outNum = num;
}

/*
 * Full light cull over tile rows [tileRowBegin, tileRowEnd).
 *   outGrid      : 2 uints per tile of the band {offset, num}; offset = 1 + sum of num over earlier tiles of the band
 *   outIndices   : [0] = sum of num over the band; [offset + i] = list[i]
 *   outCandCount : optional, per tile: number of lights that PASS the test, uncapped (statistics only)
 *   literalSelect: 1 = run the shader's partial bubble sort literally, 0 = closed form (must agree)
 */
/* ComputeLightCulling.shader:164-169: the per-light view transform does not depend on the tile */
static float* light_view_positions(const UboFrameData* ubo, const LightData* lights, int lightsNum)
{
    float* pv = (float*)malloc((size_t)(lightsNum > 0 ? lightsNum : 1) * 4 * sizeof(float));
    for (int j = 0; j < lightsNum; j++) {
        float wp[4] = { lights[j].worldPosition[0], lights[j].worldPosition[1], lights[j].worldPosition[2], 1.0f };
        float p[4];
        glsl_mat4_mul_vec4(ubo->view, wp, p);
        float w = p[3];
        p[0] = p[0] / w; p[1] = p[1] / w; p[2] = p[2] / w; p[3] = p[3] / w;
        p[2] = p[2] * -1.0f;
        memcpy(&pv[j * 4], p, 16);
    }
    return pv;
}

/* one tile of the cull (ComputeLightCulling.shader:97-240): its list (<= KEEP indices) and, optionally, the uncapped number of passing lights */
static void cull_one_tile(const UboFrameData* ubo, const LightData* lights, const float* pv, int lightsNum, const float* depth, int W, int H,
                          int tx, int ty, int literalSelect, uint32_t* list, uint32_t* num, uint32_t* outPassing)
{
    const int vpW = ubo->viewportSize[0], vpH = ubo->viewportSize[1];
    float minD, maxD;
    tile_depth_bounds(depth, W, H, tx, ty, &minD, &maxD);
    ViewFrustum fr;
    create_tile_frustum(tx, ty, vpW, vpH, ubo->invProjection, &fr);

    /* :171-177 "Add extra bounds" -- swaps near and far, in fp32 */
    float zFar = maxD, zNear = minD;
    const float diff = zFar - zNear;
    zFar -= diff;
    zNear += diff;

    uint32_t candIdx[CAND]; float candImp[CAND];
    uint32_t count = 0, passing = 0;
    for (int j = 0; j < lightsNum; j++) {
        int pass; float impact;
        if (lights[j].type == 0) { pass = 1; impact = 0.0f; } /* :153-162 */
        else {
            const float radius = lights[j].bounds[0];
            const float* p = &pv[j * 4];
            pass = sphere_frustum_overlaps(p, radius, &fr, zNear, zFar);
            if (pass) {
                float c[3] = { fr.center[0], fr.center[1], (zFar + zNear) * 0.5f };
                float d[3] = { p[0] - c[0], p[1] - c[1], p[2] - c[2] };
                impact = length3(d); /* :187 */
            } else impact = 0.0f;
        }
        if (pass) {
            passing++;
            if (count < CAND) { candIdx[count] = (uint32_t)j; candImp[count] = impact; count++; }
            if (count == CAND && !outPassing) break; /* canonical early-out (:147) */
        }
    }
    oracle_select_emit(candIdx, candImp, count, literalSelect, list, num);
    if (outPassing) *outPassing = passing;
}

ORACLE_API void oracle_light_cull(const void* ubo_, int W, int H, int lightsNum, const void* lights_, const float* depth,
                                  uint32_t* outGrid, uint32_t* outIndices, uint32_t* outCandCount,
                                  int tileRowBegin, int tileRowEnd, int literalSelect)
{
    const UboFrameData* ubo = (const UboFrameData*)ubo_;
    const LightData* lights = (const LightData*)lights_;
    const int Tx = (W - 1) / TILE + 1; /* FrameGraph/LightCullingNode.cpp:56-57 */
    float* pv = light_view_positions(ubo, lights, lightsNum);
    uint32_t running = 0;
    for (int ty = tileRowBegin; ty < tileRowEnd; ty++) {
        for (int tx = 0; tx < Tx; tx++) {
            const int bandTile = (ty - tileRowBegin) * Tx + tx;
            uint32_t list[KEEP], num;
            cull_one_tile(ubo, lights, pv, lightsNum, depth, W, H, tx, ty, literalSelect, list, &num, outCandCount ? &outCandCount[bandTile] : NULL);
            const uint32_t offset = running + 1; /* :229 canonicalised: prefix sum in tile order */
            outGrid[2 * bandTile + 0] = offset;
            outGrid[2 * bandTile + 1] = num;
            for (uint32_t i = 0; i < num; i++) outIndices[offset + i] = list[i];
            running += num;
        }
    }
    outIndices[0] = running;
    free(pv);
}

/* ------------------------------------------------------------------------------------------- */
/* K2 + K3 -- shade over per-tile lists with CSM: Standard.shader:253-341,377-439 ;             */
/* Lighting.glsl:39-76,168-284                                                                  */
/* ------------------------------------------------------------------------------------------- */

enum { MAP_R16F = 0, MAP_RGBA32F = 1, MAP_R32F = 2 };

typedef struct {
    float lightsMatrices[NUM_CASCADES][16]; /* Standard.shader:223-226 (binding 6) */
    const void* maps[NUM_CASCADES];         /* Standard.shader:233 shadowMaps[cascade] */
    int32_t width[NUM_CASCADES];
    int32_t height[NUM_CASCADES];
    int32_t format[NUM_CASCADES];
} OracleCsm;

static inline float half_to_float(uint16_t h)
{
    uint32_t sign = (uint32_t)(h & 0x8000u) << 16;
    uint32_t exp = (h >> 10) & 0x1Fu;
    uint32_t man = h & 0x3FFu;
    uint32_t bits;
    if (exp == 0) {
        if (man == 0) bits = sign;
        else { /* subnormal */
            int e = -1;
            do { e++; man <<= 1; } while ((man & 0x400u) == 0);
            bits = sign | (uint32_t)(127 - 15 - e) << 23 | ((man & 0x3FFu) << 13);
        }
    } else if (exp == 31) bits = sign | 0x7F800000u | (man << 13);
    else bits = sign | ((exp + 112u) << 23) | (man << 13);
    float f; memcpy(&f, &bits, 4); return f;
}

static inline void fetch_texel(const void* map, int fmt, int W, int x, int y, float* rgba)
{
    size_t i = (size_t)y * W + x;
    if (fmt == MAP_R16F) { rgba[0] = half_to_float(((const uint16_t*)map)[i]); rgba[1] = 0; rgba[2] = 0; rgba[3] = 1; }
    else if (fmt == MAP_R32F) { rgba[0] = ((const float*)map)[i]; rgba[1] = 0; rgba[2] = 0; rgba[3] = 1; }
    else { const float* p = &((const float*)map)[i * 4]; rgba[0] = p[0]; rgba[1] = p[1]; rgba[2] = p[2]; rgba[3] = p[3]; }
}

/* texture(sampler2D, uv): bilinear, clamp-to-edge (ECS/LightingECS.cpp:58-60: Linear + Clamp), fp32 weights.
 * Canonical order: top = t00*(1-ax) + t10*ax ; bot = t01*(1-ax) + t11*ax ; top*(1-ay) + bot*ay */
static void sample_bilinear(const void* map, int fmt, int W, int H, float u, float v, float* rgba)
{
    float x = u * (float)W - 0.5f, y = v * (float)H - 0.5f;
    float fx = floorf(x), fy = floorf(y);
    float ax = x - fx, ay = y - fy;
    int x0 = (int)fx, y0 = (int)fy, x1 = x0 + 1, y1 = y0 + 1;
    x0 = x0 < 0 ? 0 : (x0 > W - 1 ? W - 1 : x0); x1 = x1 < 0 ? 0 : (x1 > W - 1 ? W - 1 : x1);
    y0 = y0 < 0 ? 0 : (y0 > H - 1 ? H - 1 : y0); y1 = y1 < 0 ? 0 : (y1 > H - 1 ? H - 1 : y1);
    float t00[4], t10[4], t01[4], t11[4];
    fetch_texel(map, fmt, W, x0, y0, t00); fetch_texel(map, fmt, W, x1, y0, t10);
    fetch_texel(map, fmt, W, x0, y1, t01); fetch_texel(map, fmt, W, x1, y1, t11);
    const int nc = fmt == MAP_RGBA32F ? 4 : 1;
    for (int c = 0; c < nc; c++) {
        float top = t00[c] * (1.0f - ax) + t10[c] * ax;
        float bot = t01[c] * (1.0f - ax) + t11[c] * ax;
        rgba[c] = top * (1.0f - ay) + bot * ay;
    }
    for (int c = nc; c < 4; c++) rgba[c] = c == 3 ? 1.0f : 0.0f;
}

/* exp() for the EVSM warp (Lighting.glsl:277-278).  GLSL exp precision is implementation-defined; the compared
 * quantity d = exp(40 z) - moment is ill-conditioned, so the path is specified with ONE fixed fp32 algorithm
 * (Cephes-style range reduction + degree-5 polynomial, no FMA) that both sides evaluate bit-identically. */
static float canonical_expf(float x)
{
    if (x > 88.0f) x = 88.0f;
    if (x < -87.0f) x = -87.0f;
    float n = floorf(x * 1.44269504088896341f + 0.5f);
    float r = x - n * 0.693359375f;
    r = r - n * -2.12194440e-4f;
    float z = r * r;
    float p = 1.9875691500e-4f;
    p = p * r + 1.3981999507e-3f;
    p = p * r + 8.3334519073e-3f;
    p = p * r + 4.1665795894e-2f;
    p = p * r + 1.6666665459e-1f;
    p = p * r + 5.0000001201e-1f;
    float y = (p * z + r) + 1.0f;
    return ldexpf(y, (int)n);
}
ORACLE_API float oracle_canonical_expf(float x) { return canonical_expf(x); }

/* Lighting.glsl:168-197 */
static const float kPoissonDisk[16][2] = {
    { -0.94201624f, -0.39906216f }, { 0.94558609f, -0.76890725f },
    { -0.094184101f, -0.92938870f }, { 0.34495938f, 0.29387760f },
    { -0.91588581f, 0.45771432f }, { -0.81544232f, -0.87912464f },
    { -0.38277543f, 0.27676845f }, { 0.97484398f, 0.75648379f },
    { 0.44323325f, -0.97511554f }, { 0.53742981f, -0.47373420f },
    { -0.26496911f, -0.41893023f }, { 0.79197514f, 0.19090188f },
    { -0.24188840f, 0.99706507f }, { -0.81409955f, 0.91437590f },
    { 0.19984126f, 0.78641367f }, { 0.14383161f, -0.14100790f }
};
ORACLE_API float oracle_const_poisson(int i, int c) { return kPoissonDisk[i][c]; }

static float manual_pcf(const void* map, int fmt, int W, int H, const float* projCoords, float currentDepth, float bias)
{
    float shadow = 0.0f;
    float texel[2] = { 1.0f / (float)W, 1.0f / (float)H };
    const float radius = 2.0f;
    for (int i = 0; i < 16; i++) {
        float ox = kPoissonDisk[i][0] * radius * texel[0];
        float oy = kPoissonDisk[i][1] * radius * texel[1];
        float t[4];
        sample_bilinear(map, fmt, W, H, projCoords[0] + ox, projCoords[1] + oy, t);
        float pcfDepth = t[0] * 0.5f + 0.5f;
        shadow += (currentDepth + bias > pcfDepth) ? 1.0f : 0.0f;
    }
    shadow /= 16.0f;
    return shadow;
}

/* Lighting.glsl:242-261 */
static float shadow_pcf(const void* map, int fmt, int W, int H, const float* fragPosLightSpace, float bias)
{
    float pc[3] = { fragPosLightSpace[0] / fragPosLightSpace[3], fragPosLightSpace[1] / fragPosLightSpace[3], fragPosLightSpace[2] / fragPosLightSpace[3] };
    pc[0] = pc[0] * 0.5f + 0.5f; pc[1] = pc[1] * 0.5f + 0.5f; pc[2] = pc[2] * 0.5f + 0.5f;
    pc[1] = 1.0f - pc[1];
    if (pc[0] > 1.0f || pc[1] > 1.0f || pc[0] < 0.0f || pc[1] < 0.0f || pc[2] < 0.5f) return 1.0f;
    return manual_pcf(map, fmt, W, H, pc, pc[2], bias);
}

/* Lighting.glsl:218-240 */
static float chebyshev(float m0, float m1, float currentDepth, float minVariance, float lin)
{
    float d = currentDepth - m0;
    if (d < 0) return 1.0f;
    float variance = fmaxf(minVariance, m1 - m0 * m0);
    float pmax = variance / (variance + d * d);
    return clampf((pmax - lin) / (1.0f - lin), 0.0f, 1.0f);
}

/* Lighting.glsl:263-284 */
static float shadow_evsm(const void* map, int fmt, int W, int H, const float* fragPosLightSpace, float bias, int cascadeLayer)
{
    float pc[3] = { fragPosLightSpace[0] / fragPosLightSpace[3], fragPosLightSpace[1] / fragPosLightSpace[3], fragPosLightSpace[2] / fragPosLightSpace[3] };
    pc[0] = pc[0] * 0.5f + 0.5f; pc[1] = pc[1] * 0.5f + 0.5f;
    pc[1] = 1.0f - pc[1];
    if (pc[0] > 1.0f || pc[1] > 1.0f || pc[0] < 0.0f || pc[1] < 0.0f || pc[2] < 0.0f) return 1.0f;
    float s[4];
    sample_bilinear(map, fmt, W, H, pc[0], pc[1], s);
    float p05 = 1.0f; /* pow(0.5, cascadeLayer) -- exact powers of two */
    for (int i = 0; i < cascadeLayer; i++) p05 = p05 * 0.5f;
    const float currentDepth = canonical_expf(40.0f * (pc[2] + 0.003f * bias * p05));
    const float negCurrentDepth = -canonical_expf(-40.0f * (pc[2] + 0.0001f * bias));
    float posValue = chebyshev(s[0], s[1], currentDepth, 0.01f, 0.0f);
    float negValue = chebyshev(s[2], s[3], negCurrentDepth, 0.0f, 0.0f) * (cascadeLayer > 2 ? 0.0f : 1.0f);
    return clampf(1.0f - fmaxf(posValue, negValue), 0.0f, 1.0f);
}

/* Lighting.glsl:200-216 */
static int select_cascade(const float* view, const float* worldPos, const float* zNearZFar)
{
    float wp[4] = { worldPos[0], worldPos[1], worldPos[2], 1.0f }, p[4];
    glsl_mat4_mul_vec4(view, wp, p);
    float depthValue = fabsf(p[2] / p[3]);
    int layer = NUM_CASCADES;
    for (int i = 0; i < NUM_CASCADES; i++)
        if (depthValue < zNearZFar[1] * kShadowCascadeLevelsGlsl[i]) { layer = i; break; }
    return layer;
}

/* Shadow factor of a directional light (Standard.shader:266-283).  Returns the cascade in *outCascade. */
static float directional_shadow(const UboFrameData* ubo, const LightData* L, const OracleCsm* csm, const float* normal, const float* worldPos, int* outCascade)
{
    int cascade = select_cascade(ubo->view, worldPos, ubo->cameraZNearZFar);
    if (cascade > NUM_CASCADES - 1) cascade = NUM_CASCADES - 1;
    if (outCascade) *outCascade = cascade;
    if (!csm || !csm->maps[cascade]) return 1.0f; /* no shadow maps bound: synthetic frames without CSM */
    float wp[4] = { worldPos[0], worldPos[1], worldPos[2], 1.0f }, lp[4];
    glsl_mat4_mul_vec4(csm->lightsMatrices[cascade], wp, lp);
    const float ndl = dot3(normal, L->direction);
    if (L->shadowType == 2 && cascade == 0) {
        const float bias = (1.0f - ndl) * (float)(1 + cascade);
        return shadow_evsm(csm->maps[cascade], csm->format[cascade], csm->width[cascade], csm->height[cascade], lp, bias, cascade);
    }
    const float bias = fmaxf(0.000075f * (1.0f - ndl), 0.000005f);
    return shadow_pcf(csm->maps[cascade], csm->format[cascade], csm->width[cascade], csm->height[cascade], lp, bias);
}

ORACLE_API float oracle_directional_shadow(const void* ubo, const void* light, const void* csm, const float* normal, const float* worldPos, int* outCascade)
{
    return directional_shadow((const UboFrameData*)ubo, (const LightData*)light, (const OracleCsm*)csm, normal, worldPos, outCascade);
}

/* Lighting.glsl:41-76 */
static float ndf_ggx(float cosLh, float roughness)
{
    float alpha = roughness * roughness;
    float alphaSq = alpha * alpha;
    float denom = (cosLh * cosLh) * (alphaSq - 1.0f) + 1.0f;
    return alphaSq / (3.14159265359f * denom * denom);
}
static float geometry_schlick_g1(float cosTheta, float k) { return cosTheta / (cosTheta * (1.0f - k) + k); }
static float geometry_schlick_ggx(float cosLi, float cosLo, float roughness)
{
    float r = roughness + 1.0f;
    float k = (r * r) / 8.0f;
    return geometry_schlick_g1(cosLi, k) * geometry_schlick_g1(cosLo, k);
}

/* Standard.shader:259-341 */
static void calculate_lighting(const UboFrameData* ubo, const LightData* L, const OracleCsm* csm,
                               const float* albedo, float metallic, float roughness,
                               const float* F0, const float* Lo, float cosLo, const float* normal, const float* worldPos, float* out)
{
    float falloff = 1.0f, shadow = 1.0f;
    if (L->type == 0) {
        shadow = directional_shadow(ubo, L, csm, normal, worldPos, NULL);
    } else if (L->type == 1) {
        float d[3] = { L->worldPosition[0] - worldPos[0], L->worldPosition[1] - worldPos[1], L->worldPosition[2] - worldPos[2] };
        const float distance = length3(d);
        const float attenuation = 1.0f / (L->attenuation[0] + L->attenuation[1] * distance + L->attenuation[2] * (distance * distance));
        falloff = attenuation * (1.0f - powf(clampf(distance / L->bounds[0], 0.0f, 1.0f), 2.0f));
    } else if (L->type == 2) {
        float d[3] = { L->worldPosition[0] - worldPos[0], L->worldPosition[1] - worldPos[1], L->worldPosition[2] - worldPos[2] };
        float lightDir[3]; normalize3_vk(d, lightDir);
        float epsilon = L->cutOff[0] - L->cutOff[1];
        float nd[3] = { -L->direction[0], -L->direction[1], -L->direction[2] }, ndn[3];
        normalize3_vk(nd, ndn);
        float theta = dot3(lightDir, ndn);
        const float distance = length3(d);
        const float attenuation = 1.0f / (L->attenuation[0] + L->attenuation[1] * distance + L->attenuation[2] * (distance * distance));
        falloff = attenuation * clampf((theta - L->cutOff[1]) / epsilon, 0.0f, 1.0f);
        if (theta < L->cutOff[1]) falloff = 0.0f;
    }
    float Li[3] = { -L->direction[0], -L->direction[1], -L->direction[2] };
    float s[3] = { Li[0] + Lo[0], Li[1] + Lo[1], Li[2] + Lo[2] }, Lh[3];
    normalize3_vk(s, Lh);
    float cosLi = fmaxf(0.0f, dot3(normal, Li));
    float cosLh = fmaxf(0.0f, dot3(normal, Lh));
    float f5 = powf(1.0f - fmaxf(0.0f, dot3(Lh, Lo)), 5.0f);
    float F[3];
    for (int c = 0; c < 3; c++) F[c] = F0[c] + (1.0f - F0[c]) * f5;
    float D = ndf_ggx(cosLh, roughness);
    float G = geometry_schlick_ggx(cosLi, cosLo, roughness);
    float denom = fmaxf(0.00001f, 4.0f * cosLi * cosLo);
    for (int c = 0; c < 3; c++) {
        float kd = (1.0f - F[c]) * (1.0f - metallic) + 0.0f * metallic; /* mix(1-F, 0, metallic) */
        float diffuse = kd * albedo[c];
        float specular = (F[c] * D * G) / denom;
        out[c] = shadow * ((diffuse + specular) * L->intensity[c] * cosLi) * falloff;
    }
}

/* ------------------------------------------------------------------------------------------- */
/* EVSM shadow-map blur: Lighting.glsl:83-127 GaussianBlur_Evsm as drawn by  */
/* Blur.shader:66-98 (defines EVSM + HORIZONTAL | VERTICAL) from ShadowPrepassNode.cpp:283-356     */
/* ------------------------------------------------------------------------------------------- */
static const float kEvsmBlurWeights[12][12] = { /* Lighting.glsl:87-99 */
    { 0.5f, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0 },
    { 0.281088f, 0.218912f, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0 },
    { 0.197159f, 0.176426f, 0.126415f, 0, 0, 0, 0, 0, 0, 0, 0, 0 },
    { 0.152068f, 0.142855f, 0.118431f, 0.0866459f, 0, 0, 0, 0, 0, 0, 0, 0 },
    { 0.123827f, 0.118971f, 0.105518f, 0.0863909f, 0.0652929f, 0, 0, 0, 0, 0, 0, 0 },
    { 0.104454f, 0.101593f, 0.0934699f, 0.0813492f, 0.0669741f, 0.0521595f, 0, 0, 0, 0, 0, 0 },
    { 0.0903332f, 0.0885083f, 0.083252f, 0.0751759f, 0.0651684f, 0.0542336f, 0.0433285f, 0, 0, 0, 0, 0 },
    { 0.07958f, 0.0783462f, 0.0747585f, 0.0691403f, 0.061977f, 0.0538465f, 0.0453433f, 0.0370081f, 0, 0, 0, 0 },
    { 0.0711171f, 0.0702445f, 0.0676904f, 0.0636383f, 0.0583697f, 0.0522315f, 0.0455989f, 0.0388376f, 0.0322721f, 0, 0, 0 },
    { 0.0642825f, 0.0636429f, 0.0617619f, 0.0587498f, 0.0547779f, 0.0500633f, 0.0448484f, 0.0393811f, 0.0338957f, 0.0285966f, 0, 0 },
    { 0.0586472f, 0.0581645f, 0.0567402f, 0.0544433f, 0.0513831f, 0.0476999f, 0.0435548f, 0.039118f, 0.0345572f, 0.0300277f, 0.0256641f, 0 },
    { 0.0539209f, 0.0535478f, 0.0524437f, 0.050654f, 0.0482506f, 0.0453272f, 0.0419936f, 0.0383686f, 0.034573f, 0.0307232f, 0.0269255f, 0.0232718f } };
ORACLE_API float oracle_const_evsm_blur_weight(int row, int i) { return kEvsmBlurWeights[row][i]; }

/* One pass over a W x H RGBA32F image.  radius = ivec2(data.blurRadius.xy) = (umbra -> .zw, penumbra -> .xy)
 * (Blur.shader:94, RHI/SceneView.h:60).  The taps uv +- i * texelSize sit on texel centres: canonical = the texel itself,
 * clamp-to-edge.  vertical != 0: texelSize.x = 0 (Blur.shader:68-70), else texelSize.y = 0 (:72-74). */
ORACLE_API void oracle_evsm_blur_pass(const float* src, float* dst, int W, int H, int radiusX, int radiusY, int vertical)
{
    const int stepCount = 12;
    const int mx = radiusX > radiusY ? radiusX : radiusY;
    const int blurRadius = mx < stepCount ? mx : stepCount;
    const int blurRadius1 = radiusX < stepCount ? radiusX : stepCount, blurRadius2 = radiusY < stepCount ? radiusY : stepCount;
    for (int y = 0; y < H; y++) {
        for (int x = 0; x < W; x++) {
            float sum[4] = { 0.0f, 0.0f, 0.0f, 0.0f };
            for (int i = 0; i < blurRadius; i++) {
                int xa = x, xb = x, ya = y, yb = y;
                if (vertical) { ya = y + i > H - 1 ? H - 1 : y + i; yb = y - i < 0 ? 0 : y - i; }
                else          { xa = x + i > W - 1 ? W - 1 : x + i; xb = x - i < 0 ? 0 : x - i; }
                const float* a = src + ((size_t)ya * W + xa) * 4;
                const float* b = src + ((size_t)yb * W + xb) * 4;
                if (i < radiusX) { /* :113-117 umbra.zw; the .xy of the vec4 sum receive + 0 * w */
                    const float w = kEvsmBlurWeights[blurRadius1 - 1][i];
                    sum[0] += 0.0f * w; sum[1] += 0.0f * w;
                    sum[2] += (a[2] + b[2]) * w; sum[3] += (a[3] + b[3]) * w;
                }
                if (i < radiusY) { /* :119-123 penumbra.xy */
                    const float w = kEvsmBlurWeights[blurRadius2 - 1][i];
                    sum[0] += (a[0] + b[0]) * w; sum[1] += (a[1] + b[1]) * w;
                    sum[2] += 0.0f * w; sum[3] += 0.0f * w;
                }
            }
            float* o = dst + ((size_t)y * W + x) * 4;
            o[0] = sum[0]; o[1] = sum[1]; o[2] = sum[2]; o[3] = sum[3];
        }
    }
}

/* ------------------------------------------------------------------------------------------- */
/* Ambient / image-based lighting: Standard.shader:343-372 AmbientLighting  */
/* ------------------------------------------------------------------------------------------- */
/* The reference samples three textures through Vulkan samplers (linear filtering, driver-defined precision).  The
 * canonical sampler restated here -- and implemented identically on the GPU -- is:
 *   cube face + (s, t) by the Vulkan spec's major-axis table (ties: z over y over x), bilinear INSIDE the face with
 *   clamp-to-edge (no filtering across face seams), level-of-detail by linear interpolation between the two nearest
 *   mip levels, lod clamped to [0, levels - 1]; 2-D textures: bilinear, clamp-to-edge, texel centres at (i + 0.5) / size.
 * Texels are float4 (cubes) / float2 (LUT) in fp32; the reference stores RGBA16F / RG16F images. */
typedef struct {
    const float* irradiance; int irrSize;         /* 6 faces (+X,-X,+Y,-Y,+Z,-Z) x irrSize^2 float4, face-major, row t = 0 first */
    const float* env; int envSize; int envLevels; /* mip chain, level-major: level l = 6 faces x max(1, envSize >> l)^2 float4 */
    const float* brdfLut; int lutW, lutH;         /* float2 (DFG1, DFG2), row-major: u = cosLo, v = roughness */
    const float* ao;                              /* optional W*H floats (the g_aoSampler target), NULL = 1.0 */
} OracleIbl;

static void cube_face_st(const float* r, int* face, float* s, float* t)
{
    const float ax = fabsf(r[0]), ay = fabsf(r[1]), az = fabsf(r[2]);
    float sc, tc, ma;
    if (az >= ax && az >= ay) { *face = r[2] < 0.0f ? 5 : 4; sc = r[2] < 0.0f ? -r[0] : r[0]; tc = -r[1]; ma = az; }
    else if (ay >= ax)        { *face = r[1] < 0.0f ? 3 : 2; sc = r[0]; tc = r[1] < 0.0f ? -r[2] : r[2]; ma = ay; }
    else                      { *face = r[0] < 0.0f ? 1 : 0; sc = r[0] < 0.0f ? r[2] : -r[2]; tc = -r[1]; ma = ax; }
    *s = 0.5f * (sc / ma + 1.0f);
    *t = 0.5f * (tc / ma + 1.0f);
}

static void bilinear4(const float* tex, int size, float s, float t, float* out) /* tex = one face, float4 texels */
{
    const float x = s * (float)size - 0.5f, y = t * (float)size - 0.5f;
    const float fx = floorf(x), fy = floorf(y);
    const float ax = x - fx, ay = y - fy;
    int x0 = (int)fx, y0 = (int)fy, x1 = x0 + 1, y1 = y0 + 1;
    x0 = x0 < 0 ? 0 : (x0 > size - 1 ? size - 1 : x0); x1 = x1 < 0 ? 0 : (x1 > size - 1 ? size - 1 : x1);
    y0 = y0 < 0 ? 0 : (y0 > size - 1 ? size - 1 : y0); y1 = y1 < 0 ? 0 : (y1 > size - 1 ? size - 1 : y1);
    for (int c = 0; c < 4; c++) {
        const float t00 = tex[((size_t)y0 * size + x0) * 4 + c], t10 = tex[((size_t)y0 * size + x1) * 4 + c];
        const float t01 = tex[((size_t)y1 * size + x0) * 4 + c], t11 = tex[((size_t)y1 * size + x1) * 4 + c];
        const float top = t00 * (1.0f - ax) + t10 * ax, bot = t01 * (1.0f - ax) + t11 * ax;
        out[c] = top * (1.0f - ay) + bot * ay;
    }
}

static void cube_sample_level(const float* cube, int size0, int level, const float* dir, float* out)
{
    size_t off = 0;
    for (int l = 0; l < level; l++) { const int sz = (size0 >> l) > 1 ? (size0 >> l) : 1; off += (size_t)6 * sz * sz * 4; }
    const int size = (size0 >> level) > 1 ? (size0 >> level) : 1;
    int face; float s, t;
    cube_face_st(dir, &face, &s, &t);
    bilinear4(cube + off + (size_t)face * size * size * 4, size, s, t, out);
}

static void cube_sample_lod(const float* cube, int size0, int levels, const float* dir, float lod, float* out)
{
    const float maxLod = (float)(levels - 1);
    lod = lod < 0.0f ? 0.0f : (lod > maxLod ? maxLod : lod);
    const float fl = floorf(lod);
    const int l0 = (int)fl, l1 = l0 + 1 > levels - 1 ? levels - 1 : l0 + 1;
    const float f = lod - fl;
    float a[4], b[4];
    cube_sample_level(cube, size0, l0, dir, a);
    cube_sample_level(cube, size0, l1, dir, b);
    for (int c = 0; c < 4; c++) out[c] = a[c] * (1.0f - f) + b[c] * f;
}

static void lut_sample(const float* lut, int W, int H, float u, float v, float* outRG)
{
    const float x = u * (float)W - 0.5f, y = v * (float)H - 0.5f;
    const float fx = floorf(x), fy = floorf(y);
    const float ax = x - fx, ay = y - fy;
    int x0 = (int)fx, y0 = (int)fy, x1 = x0 + 1, y1 = y0 + 1;
    x0 = x0 < 0 ? 0 : (x0 > W - 1 ? W - 1 : x0); x1 = x1 < 0 ? 0 : (x1 > W - 1 ? W - 1 : x1);
    y0 = y0 < 0 ? 0 : (y0 > H - 1 ? H - 1 : y0); y1 = y1 < 0 ? 0 : (y1 > H - 1 ? H - 1 : y1);
    for (int c = 0; c < 2; c++) {
        const float t00 = lut[((size_t)y0 * W + x0) * 2 + c], t10 = lut[((size_t)y0 * W + x1) * 2 + c];
        const float t01 = lut[((size_t)y1 * W + x0) * 2 + c], t11 = lut[((size_t)y1 * W + x1) * 2 + c];
        const float top = t00 * (1.0f - ax) + t10 * ax, bot = t01 * (1.0f - ax) + t11 * ax;
        outRG[c] = top * (1.0f - ay) + bot * ay;
    }
}

ORACLE_API void oracle_cube_sample_lod(const float* cube, int size0, int levels, const float* dir, float lod, float* out4)
{
    cube_sample_lod(cube, size0, levels, dir, lod, out4);
}
ORACLE_API void oracle_cube_face_st(const float* dir, int* face, float* st) { cube_face_st(dir, face, &st[0], &st[1]); }

/* Standard.shader:343-372 */
static void ambient_lighting(const OracleIbl* ibl, const float* albedo, float metallic, float roughness, float ao,
                             const float* F0, const float* Lr, const float* normal, float cosLo, float* out)
{
    float irr[4], spec[4], brdf[2];
    cube_sample_lod(ibl->irradiance, ibl->irrSize, 1, normal, 0.0f, irr);              /* :346 */
    const float x = 1.0f - cosLo, x2 = x * x, f5 = x2 * x2 * x;                         /* :352 pow(1 - cosLo, 5) */
    cube_sample_lod(ibl->env, ibl->envSize, ibl->envLevels, Lr, roughness * (float)ibl->envLevels, spec); /* :361-362 */
    lut_sample(ibl->brdfLut, ibl->lutW, ibl->lutH, cosLo, roughness, brdf);             /* :365 */
    for (int c = 0; c < 3; c++) {
        const float F = F0[c] + (1.0f - F0[c]) * f5;
        const float kd = (1.0f - F) * (1.0f - metallic) + 0.0f * metallic;             /* :355 mix(1 - F, 0, metallic) */
        const float diffuseIBL = kd * albedo[c] * irr[c];                              /* :358 */
        const float specularIBL = (F0[c] * brdf[0] + brdf[1]) * spec[c];               /* :368 */
        out[c] = ao * (diffuseIBL + specularIBL);                                      /* :371 */
    }
}


/* ---- the cubemap pre-filters of the ambient term (one-off work of FrameGraph/EnvironmentNode.cpp:196-273) ------------------
 * Content/Shaders/ComputeIrradianceMap.shader and ComputeEnvMap_IBL.shader share these helpers (:22-77 / :31-73).  Cube layout as
 * everywhere here: level-major, then face, then size x size RGBA32F texels; textureLod = the canonical sampler above. */
static float radical_inverse_vdc(uint32_t bits) /* Math.glsl:285-293 */
{
    bits = (bits << 16) | (bits >> 16);
    bits = ((bits & 0x55555555u) << 1) | ((bits & 0xAAAAAAAAu) >> 1);
    bits = ((bits & 0x33333333u) << 2) | ((bits & 0xCCCCCCCCu) >> 2);
    bits = ((bits & 0x0F0F0F0Fu) << 4) | ((bits & 0xF0F0F0F0u) >> 4);
    bits = ((bits & 0x00FF00FFu) << 8) | ((bits & 0xFF00FF00u) >> 8);
    return (float)bits * 2.3283064365386963e-10f;
}

static void prefilter_normalize(float* v)
{
    const float inv = 1.0f / sqrtf((v[0] * v[0] + v[1] * v[1]) + v[2] * v[2]);
    v[0] *= inv; v[1] *= inv; v[2] *= inv;
}

/* GetSamplingVector (ComputeIrradianceMap.shader:43-58): the direction of texel (x, y) of face z, from its CORNER (no + 0.5) */
static void prefilter_sampling_vector(int x, int y, int face, int size, float* N)
{
    const float stx = (float)x / (float)size, sty = (float)y / (float)size;
    const float ux = 2.0f * stx - 1.0f, uy = 2.0f * (1.0f - sty) - 1.0f;
    switch (face) {
    case 0: N[0] = 1.0f; N[1] = uy; N[2] = -ux; break;
    case 1: N[0] = -1.0f; N[1] = uy; N[2] = ux; break;
    case 2: N[0] = ux; N[1] = 1.0f; N[2] = -uy; break;
    case 3: N[0] = ux; N[1] = -1.0f; N[2] = uy; break;
    case 4: N[0] = ux; N[1] = uy; N[2] = 1.0f; break;
    default: N[0] = -ux; N[1] = uy; N[2] = -1.0f; break;
    }
    prefilter_normalize(N);
}

/* ComputeBasisVectors (:61-69): T = cross(N, +Y), or cross(N, +X) when that is degenerate; S = normalize(cross(N, T)) */
static void prefilter_basis(const float* N, float* S, float* T)
{
    T[0] = N[1] * 0.0f - N[2] * 1.0f; T[1] = N[2] * 0.0f - N[0] * 0.0f; T[2] = N[0] * 1.0f - N[1] * 0.0f;
    if ((T[0] * T[0] + T[1] * T[1]) + T[2] * T[2] < 0.00001f) { /* step(Epsilon, dot(T, T)) == 0 */
        T[0] = N[1] * 0.0f - N[2] * 0.0f; T[1] = N[2] * 1.0f - N[0] * 0.0f; T[2] = N[0] * 0.0f - N[1] * 1.0f;
    }
    prefilter_normalize(T);
    S[0] = N[1] * T[2] - N[2] * T[1]; S[1] = N[2] * T[0] - N[0] * T[2]; S[2] = N[0] * T[1] - N[1] * T[0];
    prefilter_normalize(S);
}

static void tangent_to_world(const float* v, const float* N, const float* S, const float* T, float* out) /* (:72-75) */
{
    for (int c = 0; c < 3; c++) out[c] = (S[c] * v[0] + T[c] * v[1]) + N[c] * v[2];
}

/* ComputeIrradianceMap.shader:78-101: 65 536 uniformly distributed hemisphere samples per texel, summed in sample order.
 * env: the (pre-filtered) environment cube, sampled at lod 0; out: 6 x size x size RGBA32F. */
ORACLE_API void oracle_compute_irradiance_map(const float* env, int envSize, int envLevels, float* out, int size)
{
    const uint32_t NumSamples = 64u * 1024u;
    const float InvNumSamples = 1.0f / (float)NumSamples, TwoPI = 6.283185307179586f;
    for (int face = 0; face < 6; face++)
        for (int y = 0; y < size; y++)
            for (int x = 0; x < size; x++) {
                float N[3], S[3], T[3], irr[3] = { 0.0f, 0.0f, 0.0f };
                prefilter_sampling_vector(x, y, face, size, N);
                prefilter_basis(N, S, T);
                for (uint32_t i = 0; i < NumSamples; i++) {
                    const float u1 = (float)i * InvNumSamples, u2 = radical_inverse_vdc(i);
                    const float u1p = sqrtf(fmaxf(0.0f, 1.0f - u1 * u1));
                    const float h[3] = { cosf(TwoPI * u2) * u1p, sinf(TwoPI * u2) * u1p, u1 }; /* SampleHemisphere (:32-36) */
                    float Li[3], texel[4];
                    tangent_to_world(h, N, S, T, Li);
                    const float cosTheta = fmaxf(0.0f, (Li[0] * N[0] + Li[1] * N[1]) + Li[2] * N[2]);
                    cube_sample_lod(env, envSize, envLevels, Li, 0.0f, texel);
                    for (int c = 0; c < 3; c++) irr[c] += (2.0f * texel[c]) * cosTheta;
                }
                float* o = out + (((size_t)face * size + y) * size + x) * 4;
                for (int c = 0; c < 3; c++) o[c] = irr[c] / (float)NumSamples;
                o[3] = 1.0f;
            }
}

/* ComputeEnvMap_IBL.shader:76-136 for one output mip `level` (>= 1) of the pre-filtered cube: 1 024 GGX importance samples per
 * texel with mip-filtered lookups into the raw cube (all its levels).  EnvironmentNode.cpp:219-233 runs it for level 1..L-1 with
 * roughness = level / (L - 1); level 0 is a copy of the raw level 0 (:200-203).  out = the same mip-chain layout as raw. */
ORACLE_API void oracle_prefilter_env_level(const float* raw, int size0, int levels, float* out, int level, float roughness)
{
    const uint32_t NumSamples = 1024u;
    const float InvNumSamples = 1.0f / (float)NumSamples, TwoPI = 6.283185307179586f, PI = 3.14159265359f;
    size_t off = 0;
    for (int l = 0; l < level; l++) { const int sz = (size0 >> l) > 1 ? (size0 >> l) : 1; off += (size_t)6 * sz * sz * 4; }
    const int size = (size0 >> level) > 1 ? (size0 >> level) : 1;
    const float wt = 4.0f * PI / (6.0f * (float)size0 * (float)size0);
    const float alpha = roughness * roughness, alphaSq = alpha * alpha;
    for (int face = 0; face < 6; face++)
        for (int y = 0; y < size; y++)
            for (int x = 0; x < size; x++) {
                float N[3], S[3], T[3], color[3] = { 0.0f, 0.0f, 0.0f }, weight = 0.0f;
                prefilter_sampling_vector(x, y, face, size, N);
                prefilter_basis(N, S, T);
                for (uint32_t i = 0; i < NumSamples; i++) {
                    const float u1 = (float)i * InvNumSamples, u2 = radical_inverse_vdc(i);
                    /* SampleGGX (Lighting.glsl:27-37) */
                    const float cosT = sqrtf((1.0f - u2) / (1.0f + (alpha * alpha - 1.0f) * u2));
                    const float sinT = sqrtf(1.0f - cosT * cosT), phi = TwoPI * u1;
                    const float h[3] = { sinT * cosf(phi), sinT * sinf(phi), cosT };
                    float Lh[3], Li[3];
                    tangent_to_world(h, N, S, T, Lh);
                    const float d = (N[0] * Lh[0] + N[1] * Lh[1]) + N[2] * Lh[2]; /* Lo = N */
                    for (int c = 0; c < 3; c++) Li[c] = (2.0f * d) * Lh[c] - N[c];
                    const float cosLi = (N[0] * Li[0] + N[1] * Li[1]) + N[2] * Li[2];
                    if (cosLi > 0.0f) {
                        const float cosLh = fmaxf(d, 0.0f);
                        const float denom = (cosLh * cosLh) * (alphaSq - 1.0f) + 1.0f;
                        const float pdf = (alphaSq / (PI * denom * denom)) * 0.25f;  /* NdfGGX * 0.25 */
                        const float ws = 1.0f / ((float)NumSamples * pdf);
                        const float mip = fmaxf(0.5f * log2f(ws / wt) + 1.0f, 0.0f);
                        float texel[4];
                        cube_sample_lod(raw, size0, levels, Li, mip, texel);
                        for (int c = 0; c < 3; c++) color[c] += texel[c] * cosLi;
                        weight += cosLi;
                    }
                }
                float* o = out + off + (((size_t)face * size + y) * size + x) * 4;
                for (int c = 0; c < 3; c++) o[c] = color[c] / weight;
                o[3] = 1.0f;
            }
}

/* ---- the raw environment cube (FrameGraph/EnvironmentNode.cpp:116-140): equirect -> cube level 0, then the mip chain ------------
 * Content/Shaders/ComputeEquirect2Cube.shader:20-58.  One invocation per (x, y, face): st = xy / size (NO half-texel offset, as the
 * shader), the direction by the face table at :27-33, normalize = v / length(v), phi = atan(v.z, v.x), theta = acos(v.y),
 * texture(src, (phi / TwoPI, theta / PI)) with the shader's own PI = 3.141592.  A compute shader has no derivatives: level 0 of
 * the equirect texture, bilinear; its sampler wraps (TextureAssetInfo.h:30 m_clamping = Repeat) or clamps, per `repeat`.
 * VulkanGraphicsDriver.cpp:1680-1683 dispatches equirectExtent / 32 groups of 32 x 32, not cubeSize / 32: texels at
 * x >= coverW or y >= coverH are never written (imageStore outside the image is discarded) -- they keep their old value.
 * atan2f / acosf are the platform's; GPU and CPU differ in the last ulp, the parity test carries the tolerance.
 * Texels are float4 in fp32 (the reference stores RGBA16F). */
static void equirect_sample(const float* tex, int W, int H, int repeat, float u, float v, float* out)
{
    const float x = u * (float)W - 0.5f, y = v * (float)H - 0.5f;
    const float fx = floorf(x), fy = floorf(y);
    const float ax = x - fx, ay = y - fy;
    int x0 = (int)fx, y0 = (int)fy, x1 = x0 + 1, y1 = y0 + 1;
    if (repeat) {
        x0 = ((x0 % W) + W) % W; x1 = ((x1 % W) + W) % W;
        y0 = ((y0 % H) + H) % H; y1 = ((y1 % H) + H) % H;
    } else {
        x0 = x0 < 0 ? 0 : (x0 > W - 1 ? W - 1 : x0); x1 = x1 < 0 ? 0 : (x1 > W - 1 ? W - 1 : x1);
        y0 = y0 < 0 ? 0 : (y0 > H - 1 ? H - 1 : y0); y1 = y1 < 0 ? 0 : (y1 > H - 1 ? H - 1 : y1);
    }
    for (int c = 0; c < 4; c++) {
        const float t00 = tex[((size_t)y0 * W + x0) * 4 + c], t10 = tex[((size_t)y0 * W + x1) * 4 + c];
        const float t01 = tex[((size_t)y1 * W + x0) * 4 + c], t11 = tex[((size_t)y1 * W + x1) * 4 + c];
        const float top = t00 * (1.0f - ax) + t10 * ax, bot = t01 * (1.0f - ax) + t11 * ax;
        out[c] = top * (1.0f - ay) + bot * ay;
    }
}

ORACLE_API void oracle_equirect_to_cube(const float* equirect, int eqW, int eqH, int repeat, float* cube, int size, int coverW, int coverH)
{
    const float PI = 3.141592f, TwoPI = 2.0f * PI; /* ComputeEquirect2Cube.shader:11-12 */
    for (int face = 0; face < 6; face++)
        for (int y = 0; y < size && y < coverH; y++)
            for (int x = 0; x < size && x < coverW; x++) {
                const float stx = (float)x / (float)size, sty = (float)y / (float)size;
                const float ux = 2.0f * stx - 1.0f, uy = 2.0f * (1.0f - sty) - 1.0f; /* :22-23 */
                float r[3];
                switch (face) { /* :27-33 */
                case 0: r[0] = 1.0f;  r[1] = uy;    r[2] = -ux;  break;
                case 1: r[0] = -1.0f; r[1] = uy;    r[2] = ux;   break;
                case 2: r[0] = ux;    r[1] = 1.0f;  r[2] = -uy;  break;
                case 3: r[0] = ux;    r[1] = -1.0f; r[2] = uy;   break;
                case 4: r[0] = ux;    r[1] = uy;    r[2] = 1.0f; break;
                default: r[0] = -ux;  r[1] = uy;    r[2] = -1.0f; break;
                }
                float v[3];
                normalize3_glsl(r, v);
                const float phi = atan2f(v[2], v[0]), theta = acosf(v[1]); /* :44-45 */
                equirect_sample(equirect, eqW, eqH, repeat, phi / TwoPI, theta / PI, cube + (((size_t)face * size + y) * size + x) * 4);
            }
}

/* VulkanCommandBuffer.cpp:814-907 GenerateMipMaps: level i = vkCmdBlitImage(level i - 1, VK_FILTER_LINEAR) at exactly half the
 * extent (or 1), every array layer (cube face) on its own.  A 2:1 linear blit puts each destination texel centre on the corner
 * shared by four source texels: weights 1/4 each -- ((a + b) + (c + d)) * 0.25; along an axis that is already 1 wide the two
 * taps coincide.  `cube` is the level-major chain; level 0 is the input. */
ORACLE_API void oracle_generate_mipmaps_cube(float* cube, int size0, int levels)
{
    size_t srcOff = 0;
    for (int l = 1; l < levels; l++) {
        const int ss = (size0 >> (l - 1)) > 1 ? (size0 >> (l - 1)) : 1, ds = ss > 1 ? ss / 2 : 1;
        const size_t dstOff = srcOff + (size_t)6 * ss * ss * 4;
        for (int face = 0; face < 6; face++) {
            const float* src = cube + srcOff + (size_t)face * ss * ss * 4;
            float* dst = cube + dstOff + (size_t)face * ds * ds * 4;
            for (int y = 0; y < ds; y++)
                for (int x = 0; x < ds; x++) {
                    const int x0 = ss > 1 ? 2 * x : 0, x1 = ss > 1 ? 2 * x + 1 : 0, y0 = ss > 1 ? 2 * y : 0, y1 = ss > 1 ? 2 * y + 1 : 0;
                    for (int c = 0; c < 4; c++) {
                        const float a = src[((size_t)y0 * ss + x0) * 4 + c], b = src[((size_t)y0 * ss + x1) * 4 + c];
                        const float d = src[((size_t)y1 * ss + x0) * 4 + c], e = src[((size_t)y1 * ss + x1) * 4 + c];
                        dst[((size_t)y * ds + x) * 4 + c] = ((a + b) + (d + e)) * 0.25f;
                    }
                }
        }
        srcOff = dstOff;
    }
}

/* ComputeBrdfLut.shader:26-71 (Lighting.glsl:27-37 SampleGGX, :65-70 GeometrySchlickGGX_IBL, Math.glsl:285-293).
 * outRG = w*h float2; pow(x, 5) as x^2 * x^2 * x (x in [0, 1]); the reference stores RG16F. */
ORACLE_API void oracle_compute_brdf_lut(int w, int h, float* outRG)
{
    const float TwoPI = 6.283185307179586f;
    const uint32_t NumSamples = 1024;
    const float InvNumSamples = 1.0f / (float)NumSamples;
    for (int gy = 0; gy < h; gy++) {
        for (int gx = 0; gx < w; gx++) {
            float cosLo = (float)gx / (float)w;
            const float roughness = (float)gy / (float)h;
            cosLo = fmaxf(cosLo, 0.001f);
            const float Lo[3] = { sqrtf(1.0f - cosLo * cosLo), 0.0f, cosLo };
            float DFG1 = 0.0f, DFG2 = 0.0f;
            for (uint32_t i = 0; i < NumSamples; i++) {
                const float u1 = (float)i * InvNumSamples, u2 = radical_inverse_vdc(i);
                const float alpha = roughness * roughness;
                const float cosTheta = sqrtf((1.0f - u2) / (1.0f + (alpha * alpha - 1.0f) * u2));
                const float sinTheta = sqrtf(1.0f - cosTheta * cosTheta);
                const float phi = TwoPI * u1;
                const float Lh[3] = { sinTheta * cosf(phi), sinTheta * sinf(phi), cosTheta };
                const float d = dot3(Lo, Lh);
                const float Li[3] = { 2.0f * d * Lh[0] - Lo[0], 2.0f * d * Lh[1] - Lo[1], 2.0f * d * Lh[2] - Lo[2] };
                const float cosLi = Li[2], cosLh = Lh[2], cosLoLh = fmaxf(d, 0.0f);
                if (cosLi > 0.0f) {
                    const float k = (roughness * roughness) / 2.0f;
                    const float G = geometry_schlick_g1(cosLi, k) * geometry_schlick_g1(cosLo, k);
                    const float Gv = G * cosLoLh / (cosLh * cosLo);
                    const float x = 1.0f - cosLoLh, x2 = x * x, Fc = x2 * x2 * x;
                    DFG1 += (1.0f - Fc) * Gv;
                    DFG2 += Fc * Gv;
                }
            }
            outRG[((size_t)gy * w + gx) * 2 + 0] = DFG1 * InvNumSamples;
            outRG[((size_t)gy * w + gx) * 2 + 1] = DFG2 * InvNumSamples;
        }
    }
}

/*
 * Shade framebuffer rows [fbRowBegin, fbRowEnd).  surface = 3 planes of W*H float4 (plane-major):
 *   P0 = (worldPos.xyz, albedo.a)  P1 = (normal.xyz, roughness)  P2 = (albedo.rgb, metallic)   
 * out = W*H float4 (rgb = sum of lights, ambient == 0; a = albedo.a -- Standard.shader:425-438).
 * grid/indices are in the GLOBAL canonical layout (tile index = ty*Tx + tx).
 */
static void shade_impl(const void* ubo_, int W, int H, const float* surface, const void* lights_,
                       const uint32_t* grid, const uint32_t* indices, const void* csm_, const OracleIbl* ibl, float* out,
                       int fbRowBegin, int fbRowEnd)
{
    const UboFrameData* ubo = (const UboFrameData*)ubo_;
    const LightData* lights = (const LightData*)lights_;
    const OracleCsm* csm = (const OracleCsm*)csm_;
    const size_t plane = (size_t)W * H * 4;
    const int vpW = ubo->viewportSize[0], vpH = ubo->viewportSize[1];
    /* Standard.shader:413-420 */
    const int numTilesX = vpW / TILE + ((vpW % TILE) < 1 ? (vpW % TILE) : 1);
    for (int py = fbRowBegin; py < fbRowEnd; py++) {
        for (int px = 0; px < W; px++) {
            const size_t pix = ((size_t)py * W + px) * 4;
            const float* P0 = &surface[pix];
            const float* P1 = &surface[plane + pix];
            const float* P2 = &surface[2 * plane + pix];
            const float worldPos[3] = { P0[0], P0[1], P0[2] };
            const float normal[3] = { P1[0], P1[1], P1[2] };
            const float roughness = P1[3], metallic = P2[3];
            const float albedo[3] = { P2[0], P2[1], P2[2] };
            float vd[3] = { worldPos[0] - ubo->cameraPosition[0], worldPos[1] - ubo->cameraPosition[1], worldPos[2] - ubo->cameraPosition[2] };
            float viewDir[3]; normalize3_vk(vd, viewDir);
            float Lo[3] = { -viewDir[0], -viewDir[1], -viewDir[2] };
            float cosLo = fmaxf(0.0f, dot3(normal, Lo));
            float F0[3];
            for (int c = 0; c < 3; c++) F0[c] = 0.04f * (1.0f - metallic) + albedo[c] * metallic; /* mix(Fdielectric, albedo, metallic) */
            /* gl_FragCoord = (px + 0.5, py + 0.5), origin upper-left */
            const float fragX = (float)px + 0.5f, fragY = (float)py + 0.5f;
            const int sx = (int)fragX, sy = (int)((float)vpH - fragY);
            const int tileX = sx / TILE, tileY = sy / TILE;
            const uint32_t tileIndex = (uint32_t)(tileY * numTilesX + tileX);
            const uint32_t offset = grid[2 * tileIndex + 0];
            const uint32_t numLights = grid[2 * tileIndex + 1];
            float acc[3] = { 0.0f, 0.0f, 0.0f };
            if (ibl) { /* Standard.shader:396 Lr = 2 cosLo n + viewDirection; :425 outColor = AmbientLighting(...) */
                const float Lr[3] = { 2.0f * cosLo * normal[0] + viewDir[0], 2.0f * cosLo * normal[1] + viewDir[1], 2.0f * cosLo * normal[2] + viewDir[2] };
                const float ao = ibl->ao ? ibl->ao[(size_t)py * W + px] : 1.0f; /* :386 texel (px, py) of the AO target */
                ambient_lighting(ibl, albedo, metallic, roughness, ao, F0, Lr, normal, cosLo, acc);
            }
            for (uint32_t i = 0; i < numLights; i++) {
                uint32_t index = indices[offset + i];
                if (index == 0xFFFFFFFFu) break;
                float c[3];
                calculate_lighting(ubo, &lights[index], csm, albedo, metallic, roughness, F0, Lo, cosLo, normal, worldPos, c);
                acc[0] += c[0]; acc[1] += c[1]; acc[2] += c[2];
            }
            out[pix + 0] = acc[0]; out[pix + 1] = acc[1]; out[pix + 2] = acc[2]; out[pix + 3] = P0[3];
        }
    }
}

ORACLE_API void oracle_shade(const void* ubo_, int W, int H, const float* surface, const void* lights_,
                             const uint32_t* grid, const uint32_t* indices, const void* csm_, float* out,
                             int fbRowBegin, int fbRowEnd)
{
    shade_impl(ubo_, W, H, surface, lights_, grid, indices, csm_, NULL, out, fbRowBegin, fbRowEnd);
}

/* as oracle_shade, with the ambient term of Standard.shader:425 (ibl = OracleIbl*) */
ORACLE_API void oracle_shade_ibl(const void* ubo_, int W, int H, const float* surface, const void* lights_,
                                 const uint32_t* grid, const uint32_t* indices, const void* csm_, const void* ibl_, float* out,
                                 int fbRowBegin, int fbRowEnd)
{
    shade_impl(ubo_, W, H, surface, lights_, grid, indices, csm_, (const OracleIbl*)ibl_, out, fbRowBegin, fbRowEnd);
}

/* ------------------------------------------------------------------------------------------- */
/* K4 + CPU baseline -- ECS transform sweep, bounds update and frustum cull                      */
/* ------------------------------------------------------------------------------------------- */

/* Math/Transform.cpp:39-42: translate(I, pos) * toMat4(rot) * scale(I, scale).
 * trs = { vec4 position, quat rotation (memory x,y,z,w), vec4 scale } = 48 B (Math/Transform.h) */
static void transform_matrix(const float* trs, float* out)
{
    const float* pos = trs; const float* q = trs + 4; const float* sc = trs + 8;
    /* glm::translate(mat4(1), v): Result[3] = m[0]*v[0] + m[1]*v[1] + m[2]*v[2] + m[3] */
    float T[16] = { 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1 };
    for (int i = 0; i < 4; i++) {
        float I0 = (i == 0), I1 = (i == 1), I2 = (i == 2), I3 = (i == 3);
        T[12 + i] = ((I0 * pos[0] + I1 * pos[1]) + I2 * pos[2]) + I3;
    }
    /* glm::mat4_cast(quat) -- gtc/quaternion.inl mat3_cast */
    const float qx = q[0], qy = q[1], qz = q[2], qw = q[3];
    float qxx = qx * qx, qyy = qy * qy, qzz = qz * qz, qxz = qx * qz, qxy = qx * qy, qyz = qy * qz, qwx = qw * qx, qwy = qw * qy, qwz = qw * qz;
    float R[16] = { 0 };
    R[0] = 1.0f - 2.0f * (qyy + qzz); R[1] = 2.0f * (qxy + qwz); R[2] = 2.0f * (qxz - qwy);
    R[4] = 2.0f * (qxy - qwz); R[5] = 1.0f - 2.0f * (qxx + qzz); R[6] = 2.0f * (qyz + qwx);
    R[8] = 2.0f * (qxz + qwy); R[9] = 2.0f * (qyz - qwx); R[10] = 1.0f - 2.0f * (qxx + qyy);
    R[15] = 1.0f;
    /* glm::scale(mat4(1), v): Result[i] = m[i] * v[i], Result[3] = m[3] */
    float S[16] = { 0 };
    for (int i = 0; i < 4; i++) {
        S[0 + i] = (float)(i == 0) * sc[0];
        S[4 + i] = (float)(i == 1) * sc[1];
        S[8 + i] = (float)(i == 2) * sc[2];
    }
    S[15] = 1.0f;
    float TR[16];
    glm_mat4_mul_mat4(T, R, TR);
    glm_mat4_mul_mat4(TR, S, out);
}
ORACLE_API void oracle_transform_matrix(const float* trs, float* out) { transform_matrix(trs, out); }

/* Math/Bounds.cpp:479-492 AABB::Apply + Bounds.h:119-130 GetPoints.  aabb = {min.xyz, max.xyz}.
 * Quirk reproduced: m_max is seeded with numeric_limits<float>::min() (smallest POSITIVE float). */
static void aabb_apply(const float* aabb, const float* M, float* out)
{
    const float* mn = aabb; const float* mx = aabb + 3;
    const float pts[8][3] = {
        { mn[0], mn[1], mn[2] }, { mx[0], mx[1], mx[2] }, { mn[0], mx[1], mx[2] }, { mx[0], mn[1], mx[2] },
        { mx[0], mx[1], mn[2] }, { mx[0], mn[1], mn[2] }, { mn[0], mx[1], mn[2] }, { mn[0], mn[1], mx[2] }
    };
    float omax[3] = { FLT_MIN, FLT_MIN, FLT_MIN };
    float omin[3] = { FLT_MAX, FLT_MAX, FLT_MAX };
    for (int k = 0; k < 8; k++) {
        float v[4] = { pts[k][0], pts[k][1], pts[k][2], 1.0f }, t[4];
        glm_mat4_mul_vec4(M, v, t);
        for (int c = 0; c < 3; c++) {
            /* Extend: m_min = glm::min(inner, m_min) = (m_min < inner) ? m_min : inner ; m_max = glm::max(inner, m_max) = (inner < m_max) ? m_max : inner */
            omin[c] = (omin[c] < t[c]) ? omin[c] : t[c];
            omax[c] = (t[c] < omax[c]) ? omax[c] : t[c];
        }
    }
    out[0] = omin[0]; out[1] = omin[1]; out[2] = omin[2];
    out[3] = omax[0]; out[4] = omax[1]; out[5] = omax[2];
}
ORACLE_API void oracle_aabb_apply(const float* aabb, const float* M, float* out) { aabb_apply(aabb, M, out); }

static inline float glm_dot3(const float* a, const float* b) { return a[0] * b[0] + a[1] * b[1] + a[2] * b[2]; }

static void plane_from_normal_point(const float* n, const float* p, float* plane)
{
    /* Bounds.h:80-87 Plane(normal, point): w = -dot(point, normal) */
    plane[0] = n[0]; plane[1] = n[1]; plane[2] = n[2];
    plane[3] = -glm_dot3(p, n);
}
static void plane_normalize(float* plane)
{
    /* Bounds.cpp:9-13 */
    const float mag = sqrtf(glm_dot3(plane, plane));
    plane[0] /= mag; plane[1] /= mag; plane[2] /= mag; plane[3] /= mag;
}

/* Math/Bounds.cpp:20-67 Frustum::ExtractFrustumPlanes(projectionViewMatrix, bNormalizePlanes = true) with :110-140 CalculateCorners(matrix,
 * bReverseZ = true): the eight corners are inverse(matrix) * (+-1, +-1, -+1, 1) / w, the planes are built from corner differences.  Used for the
 * cascade frusta of the shadow passes (ECS/LightingECS.cpp:287-292: frustums[k].ExtractFrustumPlanes(lightCascadesMatrices[k] * lightMatrix)). */
ORACLE_API void oracle_extract_frustum_planes_matrix(const float* matrix, float* outPlanes, float* outCorners)
{
    float inv[16], corners[8][3];
    oracle_mat4_inverse(matrix, inv);
    const float sgn[4][2] = { { 1.0f, 1.0f }, { -1.0f, 1.0f }, { -1.0f, -1.0f }, { 1.0f, -1.0f } };
    for (int k = 0; k < 8; k++) {
        const float reverseZ = -1.0f;
        const float v[4] = { sgn[k & 3][0], sgn[k & 3][1], k < 4 ? reverseZ * 1.0f : reverseZ * -1.0f, 1.0f };
        float pt[4];
        glm_mat4_mul_vec4(inv, v, pt);
        for (int c = 0; c < 3; c++) corners[k][c] = pt[c] / pt[3];
    }
    float right[3], up[3], forward[3], t[3], c3[3], n[3];
    for (int c = 0; c < 3; c++) t[c] = corners[0][c] - corners[1][c];
    normalize3_glm(t, right);
    for (int c = 0; c < 3; c++) t[c] = corners[0][c] - corners[3][c];
    normalize3_glm(t, up);
    for (int c = 0; c < 3; c++) t[c] = corners[0][c] - corners[4][c];
    normalize3_glm(t, forward);
    float centerFar[3], centerNear[3], centerBottom[3], centerTop[3], centerLeft[3], centerRight[3], negForward[3];
    for (int c = 0; c < 3; c++) {
        centerFar[c] = 0.5f * (corners[0][c] + corners[2][c]);
        centerNear[c] = 0.5f * (corners[4][c] + corners[6][c]);
        centerBottom[c] = 0.5f * (corners[2][c] + corners[7][c]);
        centerTop[c] = 0.5f * (corners[0][c] + corners[5][c]);
        centerLeft[c] = 0.5f * (corners[1][c] + corners[6][c]);
        centerRight[c] = 0.5f * (corners[0][c] + corners[7][c]);
        negForward[c] = -forward[c];
    }
    plane_from_normal_point(forward, centerNear, outPlanes + 4 * 4);
    plane_from_normal_point(negForward, centerFar, outPlanes + 5 * 4);
    cross3(forward, up, c3); normalize3_glm(c3, n); plane_from_normal_point(n, centerLeft, outPlanes + 0 * 4);
    cross3(up, forward, c3); normalize3_glm(c3, n); plane_from_normal_point(n, centerRight, outPlanes + 1 * 4);
    cross3(forward, right, c3); normalize3_glm(c3, n); plane_from_normal_point(n, centerTop, outPlanes + 2 * 4);
    cross3(right, forward, c3); normalize3_glm(c3, n); plane_from_normal_point(n, centerBottom, outPlanes + 3 * 4);
    for (int i = 0; i < 6; i++) plane_normalize(outPlanes + 4 * i);
    if (outCorners) memcpy(outCorners, corners, sizeof corners);
}

/* Math/Bounds.cpp:142-193 Frustum::ExtractFrustumPlanes(worldMatrix, aspect, fovY[deg], zNear, zFar).
 * outPlanes: 6 x vec4 (L,R,T,B,N,F), outCorners: 8 x vec3 (may be NULL). */
ORACLE_API void oracle_extract_frustum_planes(const float* worldMatrix, float aspect, float fovY, float zNear, float zFar, float* outPlanes, float* outCorners)
{
    const float radians = fovY * 0.01745329251994329576923690768489f; /* glm::radians */
    const float halfVSide = zFar * tanf(radians * .5f);
    const float halfHSide = halfVSide * aspect;
    const float right[3] = { worldMatrix[0], worldMatrix[1], worldMatrix[2] };
    const float up[3] = { worldMatrix[4], worldMatrix[5], worldMatrix[6] };
    const float forward[3] = { -worldMatrix[8], -worldMatrix[9], -worldMatrix[10] };
    const float pos[3] = { worldMatrix[12], worldMatrix[13], worldMatrix[14] };
    float frontMultFar[3], t[3], u[3], c[3], n[3];
    for (int i = 0; i < 3; i++) frontMultFar[i] = zFar * forward[i];

    for (int i = 0; i < 3; i++) t[i] = pos[i] + forward[i] * zNear;
    plane_from_normal_point(forward, t, outPlanes + 4 * 4);
    for (int i = 0; i < 3; i++) { t[i] = pos[i] + forward[i] * zFar; u[i] = -forward[i]; }
    plane_from_normal_point(u, t, outPlanes + 5 * 4);

    for (int i = 0; i < 3; i++) t[i] = frontMultFar[i] - right[i] * halfHSide;
    cross3(t, up, c); normalize3_glm(c, n); plane_from_normal_point(n, pos, outPlanes + 0 * 4);     /* left  */
    for (int i = 0; i < 3; i++) t[i] = frontMultFar[i] + right[i] * halfHSide;
    cross3(up, t, c); normalize3_glm(c, n); plane_from_normal_point(n, pos, outPlanes + 1 * 4);     /* right */
    for (int i = 0; i < 3; i++) t[i] = frontMultFar[i] - up[i] * halfVSide;
    cross3(right, t, c); normalize3_glm(c, n); plane_from_normal_point(n, pos, outPlanes + 3 * 4);  /* bottom */
    for (int i = 0; i < 3; i++) t[i] = frontMultFar[i] + up[i] * halfVSide;
    cross3(t, right, c); normalize3_glm(c, n); plane_from_normal_point(n, pos, outPlanes + 2 * 4);  /* top   */

    for (int i = 0; i < 6; i++) plane_normalize(outPlanes + 4 * i);

    if (outCorners) {
        const float halfVSideNear = zNear * tanf(radians * .5f);
        const float halfHSideNear = halfVSideNear * aspect;
        const float sgn[4][2] = { { +1, +1 }, { -1, +1 }, { -1, -1 }, { +1, -1 } };
        for (int k = 0; k < 8; k++) {
            const int far_ = k < 4;
            const float hx = far_ ? halfHSide : halfHSideNear, hy = far_ ? halfVSide : halfVSideNear;
            const float z = far_ ? -zFar : -zNear;
            /* farEnd +- endSizeHorizontal +- endSizeVertical, componentwise, left to right */
            float v[4] = { (0.0f + sgn[k & 3][0] * hx) + 0.0f, (0.0f + 0.0f) + sgn[k & 3][1] * hy, (z + 0.0f) + 0.0f, 1.0f }, r[4];
            glm_mat4_mul_vec4(worldMatrix, v, r);
            outCorners[3 * k + 0] = r[0]; outCorners[3 * k + 1] = r[1]; outCorners[3 * k + 2] = r[2];
        }
    }
}

/* Math/Bounds.cpp:245-260 Frustum::OverlapsAABB(const AABB&) */
static inline int overlaps_aabb(const float* planes, const float* aabb)
{
    int inside = 1;
    for (int i = 0; i < 6; i++) {
        const float* p = planes + 4 * i;
        float ax = aabb[0] * p[0], bx = aabb[3] * p[0];
        float ay = aabb[1] * p[1], by = aabb[4] * p[1];
        float az = aabb[2] * p[2], bz = aabb[5] * p[2];
        const float d = (ax < bx ? bx : ax) + (ay < by ? by : ay) + (az < bz ? bz : az) + p[3];
        inside &= d > 0;
    }
    return inside;
}

/* The cascade mesh lists of LightingECS::PrepareCSMPasses (ECS/LightingECS.cpp:287-296) as bitmasks: bit i of mask k = the world AABB of entity i
 * overlaps the frustum of cascade k (Frustum::OverlapsAABB, Math/Bounds.cpp:245-260).  planes = numCascades x 6 x vec4; masks = numCascades x
 * ceil(n / 64) words, LSB first.  (The octree walk of RHISceneView::TraceScene visits the same elements; the removal of meshes already covered
 * by an earlier cascade (:310-327) and the change tracking (:334-366) work on these sets on the host.) */
ORACLE_API void oracle_csm_caster_masks(uint32_t numEntities, const float* worldAabb, const float* planes, uint32_t numCascades, uint64_t* masks)
{
    const uint32_t words = (numEntities + 63) / 64;
    memset(masks, 0, (size_t)numCascades * words * 8);
    for (uint32_t k = 0; k < numCascades; k++)
        for (uint32_t i = 0; i < numEntities; i++)
            if (overlaps_aabb(planes + 24 * k, worldAabb + 6 * (size_t)i)) masks[(size_t)k * words + (i >> 6)] |= 1ull << (i & 63);
}
ORACLE_API int oracle_overlaps_aabb(const float* planes, const float* aabb) { return overlaps_aabb(planes, aabb); }

/* Content/Shaders/LinearizeDepth.shader:61-73 with REVERSE_Z_INF_FAR_PLANE (:6), drawn by
 * FrameGraph/LinearizeDepthNode.cpp:22-109 over the whole target: the texcoord flip of the quad (:49) and the flipped
 * viewport (GraphicsDriver/Vulkan/VulkanDevice.cpp:681) cancel, so texel (x, r) of the target reads texel (x, r) of
 * the depth attachment.  `invVss` / `zvs` (:65-67) are dead code. */
ORACLE_API void oracle_linearize_depth(float zNear, const float* raw, size_t count, float* out)
{
    for (size_t i = 0; i < count; i++) {
        const float depth = raw[i];
        const float linearDepth = -zNear / depth; /* :70 */
        out[i] = -linearDepth;                    /* :74 outColor = vec4(-linearDepth) */
    }
}

/* Math/Bounds.cpp:211-243 Frustum::OverlapsSphere / ContainsSphere (scalar) */
ORACLE_API int oracle_overlaps_sphere(const float* planes, const float* sphere)
{
    int res = 1;
    for (int p = 0; p < 6; p++)
        if (planes[4 * p + 0] * sphere[0] + planes[4 * p + 1] * sphere[1] + planes[4 * p + 2] * sphere[2] + planes[4 * p + 3] < -sphere[3]) res = 0;
    return res;
}
ORACLE_API int oracle_contains_sphere(const float* planes, const float* sphere)
{
    int res = 1;
    for (int p = 0; p < 6; p++)
        if (planes[4 * p + 0] * sphere[0] + planes[4 * p + 1] * sphere[1] + planes[4 * p + 2] * sphere[2] + planes[4 * p + 3] < sphere[3]) res = 0;
    return res;
}

/* Math/Bounds.cpp:264-325 Frustum::OverlapsAABB(AABB*, n, int32*) -- the SSE batch form, restated LITERALLY.
 * NOTE (reference behaviour, reproduced): the code loads six __m128 rows at float offsets 0,4,8 / 12,16,20, advances by 24 floats
 * and transposes (row0, row1, row2, zero).  With the reference's own 24-byte AABB {vec3 min, vec3 max} the rows straddle boxes, so
 * the lane results do not correspond to boxes i..i+3, and _MM_TRANSPOSE4_PS overwrites the `zero` register that is later the
 * comparison operand with the rows' fourth components.  The arithmetic is only meaningful for input laid out as THREE boxes per
 * 24 floats -- vec4 min0, min1, min2, vec4 max0, max1, max2 with zero .w padding: lanes 0..2 are then those boxes (lane 3 a box
 * of zeros) and `zero` stays zero.  tests/test_oracle_cpu.py checks exactly that against the scalar form; no caller in the
 * reference uses the function.  It is kept as the reference's "most optimized" cost proxy for the CPU baseline.
 * Outputs: 0x80000000 where any plane compare (distance <= 0) was true, i.e. CULLED, 0 otherwise (inverted w.r.t. the scalar form).
 * Requires 16-byte aligned input and numObjects % 4 == 0. */
ORACLE_API void oracle_overlaps_aabb_sse(const float* planes, const float* aabbs, uint32_t numObjects, int32_t* outResults)
{
    const float* pAabbData = aabbs;
    __m128 planesX[6], planesY[6], planesZ[6], planesD[6];
    for (int i = 0; i < 6; i++) {
        planesX[i] = _mm_set1_ps(planes[4 * i + 0]);
        planesY[i] = _mm_set1_ps(planes[4 * i + 1]);
        planesZ[i] = _mm_set1_ps(planes[4 * i + 2]);
        planesD[i] = _mm_set1_ps(planes[4 * i + 3]);
    }
    __m128 zero = _mm_setzero_ps();
    for (uint32_t i = 0; i < numObjects; i += 4) {
        __m128 aabbMinX = _mm_load_ps(pAabbData);
        __m128 aabbMinY = _mm_load_ps(pAabbData + 4);
        __m128 aabbMinZ = _mm_load_ps(pAabbData + 8);
        __m128 aabbMaxX = _mm_load_ps(pAabbData + 12);
        __m128 aabbMaxY = _mm_load_ps(pAabbData + 16);
        __m128 aabbMaxZ = _mm_load_ps(pAabbData + 20);
        pAabbData += 24;
        _MM_TRANSPOSE4_PS(aabbMinX, aabbMinY, aabbMinZ, zero);
        _MM_TRANSPOSE4_PS(aabbMaxX, aabbMaxY, aabbMaxZ, zero);
        __m128 intersectionRes = _mm_setzero_ps();
        for (int j = 0; j < 6; j++) {
            __m128 resX = _mm_max_ps(_mm_mul_ps(aabbMinX, planesX[j]), _mm_mul_ps(aabbMaxX, planesX[j]));
            __m128 resY = _mm_max_ps(_mm_mul_ps(aabbMinY, planesY[j]), _mm_mul_ps(aabbMaxY, planesY[j]));
            __m128 resZ = _mm_max_ps(_mm_mul_ps(aabbMinZ, planesZ[j]), _mm_mul_ps(aabbMaxZ, planesZ[j]));
            __m128 sumXy = _mm_add_ps(resX, resY);
            __m128 sumZw = _mm_add_ps(resZ, planesD[j]);
            __m128 distanceToPlane = _mm_add_ps(sumXy, sumZw);
            __m128 planeRes = _mm_cmple_ps(distanceToPlane, zero);
            intersectionRes = _mm_or_ps(intersectionRes, planeRes);
        }
        __m128i intersectionResI = _mm_cvtps_epi32(intersectionRes);
        _mm_storeu_si128((__m128i*)&outResults[i], intersectionResI);
    }
}

/*
 * The ECS sweep over level-sorted entities (ECS/TransformECS.cpp:144-212 full-sweep branch with every component
 * dirty + ECS/StaticMeshRendererECS.cpp:40-58 + RHI/SceneView.cpp:56 cull with flat float AABBs):
 *   relative = Transform::Matrix(trs[i]);  world = parent == 0xFFFFFFFF ? relative : world[parent] * relative
 *   worldAabb = AABB::Apply(localAabb, world);  visible = Frustum::OverlapsAABB(worldAabb)
 * parent[i] < i is required (level-sorted).  visibility: 1 bit per entity, LSB-first in uint64 words.
 * Entities [begin, end) are processed; world[] of parents outside the range must already be filled.
 */
ORACLE_API void oracle_ecs_sweep(uint32_t begin, uint32_t end, const float* trs, const uint32_t* parent, const float* localAabb,
                                 const float* planes, float* world, float* worldAabb, uint64_t* visibility)
{
    for (uint32_t i = begin; i < end; i++) {
        float rel[16];
        transform_matrix(trs + 12 * (size_t)i, rel);
        if (parent[i] == 0xFFFFFFFFu) memcpy(world + 16 * (size_t)i, rel, sizeof rel);
        else glm_mat4_mul_mat4(world + 16 * (size_t)parent[i], rel, world + 16 * (size_t)i);
        aabb_apply(localAabb + 6 * (size_t)i, world + 16 * (size_t)i, worldAabb + 6 * (size_t)i);
        const int vis = overlaps_aabb(planes, worldAabb + 6 * (size_t)i);
        uint64_t bit = 1ull << (i & 63);
        if (vis) __atomic_fetch_or(&visibility[i >> 6], bit, __ATOMIC_RELAXED);
        else __atomic_fetch_and(&visibility[i >> 6], ~bit, __ATOMIC_RELAXED);
    }
}

/*
 * The same sweep on `numThreads` host threads, the way the reference parallelises its per-entity loops: chunks of 1 024 entities
 * handed to worker threads (ECS/StaticMeshRendererECS.cpp:19 `const size_t numThreads = ...; 1024 per task`, Tasks/Scheduler.cpp:150-152),
 * one hierarchy level after the other (a child needs its parent's world matrix).  Workers pull chunks from a shared counter; a barrier
 * separates the levels.  Returns the seconds spent between the start barrier and the last level's barrier (thread creation excluded,
 * as the reference's scheduler threads exist before the frame starts).  CPU baseline only (bench.py).
 */
#include <pthread.h>
#include <time.h>
typedef struct {
    const float* trs; const uint32_t* parent; const float* localAabb; const float* planes;
    float* world; float* worldAabb; uint64_t* visibility;
    const uint32_t* levelOffsets; uint32_t numLevels;
    uint32_t* nextChunk; /* one counter per level */
    pthread_barrier_t* bar;
    pthread_mutex_t mu; pthread_cond_t cv; int go; /* 0: wait for the barrier to be sized, 1: run, -1: leave */
} SweepJob;

static void* sweep_worker(void* arg)
{
    SweepJob* j = (SweepJob*)arg;
    /* the barrier is sized only once the number of threads that could be created is known: a worker parks on the condition variable until then
     * (a thread parked in pthread_barrier_wait cannot be called back: the wait is no cancellation point) */
    pthread_mutex_lock(&j->mu);
    while (j->go == 0) pthread_cond_wait(&j->cv, &j->mu);
    const int go = j->go;
    pthread_mutex_unlock(&j->mu);
    if (go < 0) return NULL;
    pthread_barrier_wait(j->bar); /* start */
    for (uint32_t l = 0; l < j->numLevels; l++) {
        const uint32_t lo = j->levelOffsets[l], hi = j->levelOffsets[l + 1];
        for (;;) {
            const uint32_t c = __atomic_fetch_add(&j->nextChunk[l], 1u, __ATOMIC_RELAXED);
            const uint64_t b = (uint64_t)lo + (uint64_t)c * 1024u;
            if (b >= hi) break;
            const uint32_t e = (uint32_t)(b + 1024u < hi ? b + 1024u : hi);
            oracle_ecs_sweep((uint32_t)b, e, j->trs, j->parent, j->localAabb, j->planes, j->world, j->worldAabb, j->visibility);
        }
        pthread_barrier_wait(j->bar); /* the level is complete */
    }
    return NULL;
}

ORACLE_API double oracle_ecs_sweep_threads(uint32_t numLevels, const uint32_t* levelOffsets, const float* trs, const uint32_t* parent,
                                           const float* localAabb, const float* planes, float* world, float* worldAabb, uint64_t* visibility,
                                           uint32_t numThreads)
{
    if (numThreads < 1) numThreads = 1;
    if (numThreads > 1024) numThreads = 1024;
    uint32_t counters[64];
    if (numLevels > 64) return -1.0;
    memset(counters, 0, sizeof counters);
    pthread_t* tid = (pthread_t*)malloc(sizeof(pthread_t) * numThreads);
    if (!tid) return -1.0;
    pthread_barrier_t bar;
    SweepJob job = { trs, parent, localAabb, planes, world, worldAabb, visibility, levelOffsets, numLevels, counters, &bar,
                     PTHREAD_MUTEX_INITIALIZER, PTHREAD_COND_INITIALIZER, 0 };
    uint32_t started = 0;
    for (; started < numThreads; started++)
        if (pthread_create(&tid[started], NULL, sweep_worker, &job) != 0) break;
    /* fewer threads than asked for is fine (the chunks are handed out dynamically); none, or no barrier, is a failure -- the parked workers are told to leave */
    const int ok = started > 0 && pthread_barrier_init(&bar, NULL, started + 1) == 0;
    pthread_mutex_lock(&job.mu);
    job.go = ok ? 1 : -1;
    pthread_cond_broadcast(&job.cv);
    pthread_mutex_unlock(&job.mu);
    double seconds = -1.0;
    if (ok) {
        struct timespec t0, t1;
        pthread_barrier_wait(&bar);
        clock_gettime(CLOCK_MONOTONIC, &t0);
        for (uint32_t l = 0; l < numLevels; l++) pthread_barrier_wait(&bar);
        clock_gettime(CLOCK_MONOTONIC, &t1);
        seconds = (double)(t1.tv_sec - t0.tv_sec) + 1e-9 * (double)(t1.tv_nsec - t0.tv_nsec);
    }
    for (uint32_t i = 0; i < started; i++) pthread_join(tid[i], NULL);
    if (ok) pthread_barrier_destroy(&bar);
    free(tid);
    return seconds;
}

/* ---- whole-frame checkers on all host cores: the cull and the shade over tile rows / framebuffer rows handed out dynamically -----------------
 * Same per-tile / per-pixel code as oracle_light_cull / oracle_shade (so: the same bits), only the loop over rows is spread over threads; the
 * cull's prefix sum over the tiles stays sequential.  They exist so that the GPU tests can compare ENTIRE 4K frames (32 400 tiles x 65 536 lights
 * = 2.1 G sphere tests) in seconds on the GPU box's host instead of a few sampled tile rows.  A worker that could not be created is simply absent.
 */
typedef struct {
    const UboFrameData* ubo; const LightData* lights; const float* pv; int lightsNum; const float* depth; int W, H, Tx;
    int rowBegin, rowEnd, literalSelect;
    uint32_t* nums;   /* per band tile */
    uint32_t* lists;  /* per band tile: KEEP slots */
    uint32_t* passing;
    int next;
} CullRowsJob;

static void* cull_rows_worker(void* arg)
{
    CullRowsJob* j = (CullRowsJob*)arg;
    for (;;) {
        const int ty = j->rowBegin + __atomic_fetch_add(&j->next, 1, __ATOMIC_RELAXED);
        if (ty >= j->rowEnd) break;
        for (int tx = 0; tx < j->Tx; tx++) {
            const size_t t = (size_t)(ty - j->rowBegin) * j->Tx + tx;
            cull_one_tile(j->ubo, j->lights, j->pv, j->lightsNum, j->depth, j->W, j->H, tx, ty, j->literalSelect, &j->lists[t * KEEP], &j->nums[t],
                          j->passing ? &j->passing[t] : NULL);
        }
    }
    return NULL;
}

static uint32_t run_workers(void* (*fn)(void*), void* job, uint32_t numThreads)
{
    if (numThreads < 1) numThreads = 1;
    if (numThreads > 1024) numThreads = 1024;
    pthread_t* tid = (pthread_t*)malloc(sizeof(pthread_t) * numThreads);
    uint32_t started = 0;
    if (tid)
        for (; started < numThreads - 1; started++)
            if (pthread_create(&tid[started], NULL, fn, job) != 0) break;
    fn(job); /* the calling thread works too: the job completes whatever could be created */
    for (uint32_t i = 0; i < started; i++) pthread_join(tid[i], NULL);
    free(tid);
    return started + 1;
}

/* as oracle_light_cull; returns the number of threads that worked, 0 on an allocation failure */
ORACLE_API uint32_t oracle_light_cull_threads(const void* ubo_, int W, int H, int lightsNum, const void* lights_, const float* depth,
                                              uint32_t* outGrid, uint32_t* outIndices, uint32_t* outCandCount,
                                              int tileRowBegin, int tileRowEnd, int literalSelect, uint32_t numThreads)
{
    const UboFrameData* ubo = (const UboFrameData*)ubo_;
    const LightData* lights = (const LightData*)lights_;
    const int Tx = (W - 1) / TILE + 1;
    const size_t tiles = (size_t)(tileRowEnd > tileRowBegin ? tileRowEnd - tileRowBegin : 0) * Tx;
    uint32_t* nums = (uint32_t*)malloc((tiles ? tiles : 1) * sizeof(uint32_t));
    uint32_t* lists = (uint32_t*)malloc((tiles ? tiles : 1) * KEEP * sizeof(uint32_t));
    if (!nums || !lists) { free(nums); free(lists); return 0; }
    float* pv = light_view_positions(ubo, lights, lightsNum);
    CullRowsJob job = { ubo, lights, pv, lightsNum, depth, W, H, Tx, tileRowBegin, tileRowEnd, literalSelect, nums, lists, outCandCount, 0 };
    const uint32_t used = run_workers(cull_rows_worker, &job, numThreads);
    uint32_t running = 0;
    for (size_t t = 0; t < tiles; t++) { /* :229 canonicalised: prefix sum in tile order */
        const uint32_t offset = running + 1;
        outGrid[2 * t + 0] = offset;
        outGrid[2 * t + 1] = nums[t];
        memcpy(&outIndices[offset], &lists[t * KEEP], nums[t] * sizeof(uint32_t));
        running += nums[t];
    }
    outIndices[0] = running;
    free(pv); free(nums); free(lists);
    return used;
}

typedef struct {
    const void* ubo; int W, H; const float* surface; const void* lights; const uint32_t* grid; const uint32_t* indices; const void* csm; const OracleIbl* ibl;
    float* out; int rowBegin, rowEnd, next;
} ShadeRowsJob;

static void* shade_rows_worker(void* arg)
{
    ShadeRowsJob* j = (ShadeRowsJob*)arg;
    for (;;) {
        const int r0 = j->rowBegin + 4 * __atomic_fetch_add(&j->next, 1, __ATOMIC_RELAXED);
        if (r0 >= j->rowEnd) break;
        shade_impl(j->ubo, j->W, j->H, j->surface, j->lights, j->grid, j->indices, j->csm, j->ibl, j->out, r0, r0 + 4 < j->rowEnd ? r0 + 4 : j->rowEnd);
    }
    return NULL;
}

/* as oracle_shade / oracle_shade_ibl (ibl_ may be NULL); returns the number of threads that worked */
ORACLE_API uint32_t oracle_shade_threads(const void* ubo_, int W, int H, const float* surface, const void* lights_,
                                         const uint32_t* grid, const uint32_t* indices, const void* csm_, const void* ibl_, float* out,
                                         int fbRowBegin, int fbRowEnd, uint32_t numThreads)
{
    ShadeRowsJob job = { ubo_, W, H, surface, lights_, grid, indices, csm_, (const OracleIbl*)ibl_, out, fbRowBegin, fbRowEnd, 0 };
    return run_workers(shade_rows_worker, &job, numThreads);
}

/* ---- Hi-Z pyramid (FrameGraph/DepthHighZNode.cpp:74-96, Content/Shaders/ComputeDepthHighZ.shader) ----------------------------
 * The pyramid's sampler is created with `reduction: Min` (Content/DefaultRenderer.renderer:51-57): a texture() fetch returns the
 * minimum over the bilinear footprint instead of the weighted mean.  Canonical form (Vulkan: "the texels of the footprint with
 * non-zero weights"): x = u * W - 0.5, i0 = floor(x), i1 = i0 + 1, weights (1 - frac, frac), clamp-to-edge; likewise y; the
 * minimum over the up to four texels whose weight is not zero.  Pyramid layout: level-major, level l = max(H >> l, 1) rows of
 * max(W >> l, 1) floats (R32_SFLOAT, raw reversed-Z depth: the minimum is the FARTHEST depth of the footprint). */
static int hiz_coord(float x, int size) /* floor to a texel index, clamp-to-edge; NaN -> 0 */
{
    if (!(x == x)) return 0;
    if (x < -1.0f) x = -1.0f;
    if (x > (float)size) x = (float)size;
    const int i = (int)floorf(x);
    return i < 0 ? 0 : (i > size - 1 ? size - 1 : i);
}

static float hiz_fetch_min(const float* tex, int W, int H, float u, float v)
{
    const float x = u * (float)W - 0.5f, y = v * (float)H - 0.5f;
    const float fx = x - floorf(x), fy = y - floorf(y);
    const int x0 = hiz_coord(x, W), x1 = hiz_coord(x + 1.0f, W), y0 = hiz_coord(y, H), y1 = hiz_coord(y + 1.0f, H);
    /* weights (1-fx)(1-fy), fx(1-fy), (1-fx)fy, fx fy: 1 - frac is never 0 (frac < 1); frac may be 0 (or NaN: then both columns count) */
    const int useX1 = !(fx == 0.0f), useY1 = !(fy == 0.0f);
    float m = tex[(size_t)y0 * W + x0];
    if (useX1) { const float t = tex[(size_t)y0 * W + x1]; m = t < m ? t : m; }
    if (useY1) {
        const float t = tex[(size_t)y1 * W + x0]; m = t < m ? t : m;
        if (useX1) { const float t2 = tex[(size_t)y1 * W + x1]; m = t2 < m ? t2 : m; }
    }
    return m;
}

/* ComputeDepthHighZ.shader:22-30: out(pos) = texture(inputDepth, (pos + 0.5) / outputSize).x with the min-reduction sampler */
ORACLE_API void oracle_hiz_downscale(const float* src, int srcW, int srcH, float* dst, int dstW, int dstH)
{
    for (int y = 0; y < dstH; y++)
        for (int x = 0; x < dstW; x++)
            dst[(size_t)y * dstW + x] = hiz_fetch_min(src, srcW, srcH, ((float)x + 0.5f) / (float)dstW, ((float)y + 0.5f) / (float)dstH);
}

/* DepthHighZNode.cpp:78-95: mip 0 from the depth attachment ("src"), mip i + 1 from mip i */
ORACLE_API void oracle_hiz_build(const float* depth, int depthW, int depthH, float* pyramid, int W, int H, int levels)
{
    const float* src = depth;
    int sw = depthW, sh = depthH;
    float* dst = pyramid;
    for (int l = 0; l < levels; l++) {
        const int w = (W >> l) > 1 ? (W >> l) : 1, h = (H >> l) > 1 ? (H >> l) : 1;
        oracle_hiz_downscale(src, sw, sh, dst, w, h);
        src = dst; sw = w; sh = h;
        dst += (size_t)w * h;
    }
}

/* Math.glsl:296-315 ProjectSphere (2D Polyhedral Bounds of a Clipped, Perspective-Projected 3D Sphere, Mara & McGuire 2013) */
static int project_sphere(const float* C, float r, float znear, float P00, float P11, float* aabb)
{
    if (C[2] < r + znear) return 0;
    const float cx[2] = { -C[0], -C[2] };
    const float vx[2] = { sqrtf((cx[0] * cx[0] + cx[1] * cx[1]) - r * r), r };
    const float minx[2] = { vx[0] * cx[0] + (-vx[1]) * cx[1], vx[1] * cx[0] + vx[0] * cx[1] }; /* mat2(vx.x, vx.y, -vx.y, vx.x) * cx */
    const float maxx[2] = { vx[0] * cx[0] + vx[1] * cx[1], (-vx[1]) * cx[0] + vx[0] * cx[1] }; /* mat2(vx.x, -vx.y, vx.y, vx.x) * cx */
    const float cy[2] = { -C[1], -C[2] };
    const float vy[2] = { sqrtf((cy[0] * cy[0] + cy[1] * cy[1]) - r * r), r };
    const float miny[2] = { vy[0] * cy[0] + (-vy[1]) * cy[1], vy[1] * cy[0] + vy[0] * cy[1] };
    const float maxy[2] = { vy[0] * cy[0] + vy[1] * cy[1], (-vy[1]) * cy[0] + vy[0] * cy[1] };
    const float a[4] = { minx[0] / minx[1] * P00, miny[0] / miny[1] * P11, maxx[0] / maxx[1] * P00, maxy[0] / maxy[1] * P11 };
    /* aabb = aabb.xwzy * vec4(0.5, -0.5, 0.5, -0.5) + vec4(0.5) */
    aabb[0] = a[0] * 0.5f + 0.5f; aabb[1] = a[3] * -0.5f + 0.5f; aabb[2] = a[2] * 0.5f + 0.5f; aabb[3] = a[1] * -0.5f + 0.5f;
    return 1;
}

/* floor(log2(m)) of the shader (:82), as the exponent field -- exact for every positive float; textureLod clamps the level to the
 * pyramid (m <= 0, NaN -> level 0) */
static int hiz_level(float m, int levels)
{
    if (!(m > 0.0f)) return 0;
    uint32_t bits;
    memcpy(&bits, &m, 4);
    const int e = (int)((bits >> 23) & 0xFFu) - 127;
    return e < 0 ? 0 : (e > levels - 1 ? levels - 1 : e);
}

typedef struct { const float* pyramid; int32_t width, height, levels; } OracleHiZ;

/* ComputeMeshCulling.shader:62-94 OcclusionCulling: 1 = occluded */
static int occlusion_culling(const UboFrameData* ubo, const float* center, float radius, const OracleHiZ* hz)
{
    float aabb[4];
    if (!project_sphere(center, radius, ubo->cameraZNearZFar[0], ubo->projection[0], ubo->projection[5], aabb)) return 0;
    const float width = (aabb[2] - aabb[0]) * (float)hz->width, height = (aabb[3] - aabb[1]) * (float)hz->height;
    const int level = hiz_level(width < height ? height : width, hz->levels); /* max(width, height) (:82) */
    const float u = (aabb[0] + aabb[2]) * 0.5f, v = (aabb[1] + aabb[3]) * 0.5f;
    const float* tex = hz->pyramid;
    for (int l = 0; l < level; l++) tex += (size_t)((hz->width >> l) > 1 ? (hz->width >> l) : 1) * ((hz->height >> l) > 1 ? (hz->height >> l) : 1);
    const int w = (hz->width >> level) > 1 ? (hz->width >> level) : 1, h = (hz->height >> level) > 1 ? (hz->height >> level) : 1;
    const float depth = hiz_fetch_min(tex, w, h, u, v);
    const float depthSphere = ubo->cameraZNearZFar[0] / (center[2] - radius);
    return depthSphere < depth;
}

/* Content/Shaders/ComputeMeshCulling.shader:96-110 FrustumCulling over PerInstanceData (96 B: mat4 model@0,
 * vec4 sphereBounds@64, u32 materialInstance@80, u32 isCulled@84 -- FrameGraph/RenderSceneNode.h:16-33), and with `hiz`
 * (OracleHiZ*, the OCCLUSION_CULLING build) `|| OcclusionCulling` (:139).  Writes isCulled in place (1 = culled). */
static void mesh_cull_flags(const UboFrameData* ubo, void* instances, uint32_t numInstances, const OracleHiZ* hz)
{
    ViewFrustum fr;
    /* Math.glsl:185-222 CreateViewFrustum(viewportSize, invProjection) */
    create_frustum_rect(0.0f, 0.0f, (float)ubo->viewportSize[0], (float)ubo->viewportSize[1], ubo->viewportSize[0], ubo->viewportSize[1], ubo->invProjection, &fr);
    for (uint32_t i = 0; i < numInstances; i++) {
        uint8_t* inst = (uint8_t*)instances + 96 * (size_t)i;
        const float* model = (const float*)inst;
        const float* sb = (const float*)(inst + 64);
        float c[4] = { sb[0], sb[1], sb[2], 1.0f }, wc[4], vc[4];
        glsl_mat4_mul_vec4(model, c, wc);
        glsl_mat4_mul_vec4(ubo->view, wc, vc);
        float center[3] = { vc[0] / vc[3], vc[1] / vc[3], vc[2] / vc[3] };
        center[2] = center[2] * -1.0f;
        const float lossyScale = length3(model);
        const float radius = sb[3] * lossyScale;
        const int overlaps = sphere_frustum_overlaps(center, radius, &fr, ubo->cameraZNearZFar[1], ubo->cameraZNearZFar[0]);
        uint32_t culled = overlaps ? 0u : 1u;
        if (!culled && hz && hz->pyramid) culled = occlusion_culling(ubo, center, radius, hz) ? 1u : 0u;
        memcpy(inst + 84, &culled, 4);
    }
}

ORACLE_API void oracle_mesh_frustum_cull(const void* ubo_, void* instances, uint32_t numInstances)
{
    mesh_cull_flags((const UboFrameData*)ubo_, instances, numInstances, NULL);
}

ORACLE_API void oracle_mesh_cull_occlusion(const void* ubo_, void* instances, uint32_t numInstances, const float* pyramid, int32_t width, int32_t height,
                                           int32_t levels)
{
    const OracleHiZ hz = { pyramid, width, height, levels };
    mesh_cull_flags((const UboFrameData*)ubo_, instances, numInstances, &hz);
}

/* Content/Shaders/ComputeMeshCulling.shader:119-177 main() without the OCCLUSION_CULLING define: step 2 = FrustumCulling over
 * the instance window (as above), barrier, step 3 = "remove empty draw calls": per DrawIndexedIndirectData (20 B: indexCount,
 * instanceCount, firstIndex, vertexOffset, firstInstance -- :28-35) a stable in-place compaction of the batch's instance records
 * (:154-174) and instanceCount = number kept (:176).  Canonical reading of the shader's cross-workgroup race (SURVEY.md
 * Appendix C): every flag of step 2 is written before any batch of step 3 is compacted; batches own disjoint instance ranges. */
static void mesh_cull_compact_impl(const void* ubo_, void* instances, uint32_t numInstances, uint32_t firstInstanceIndex, void* batches, uint32_t numBatches,
                                   const OracleHiZ* hz)
{
    uint8_t* inst = (uint8_t*)instances;
    mesh_cull_flags((const UboFrameData*)ubo_, inst + 96 * (size_t)firstInstanceIndex, numInstances, hz);
    for (uint32_t b = 0; b < numBatches; b++) {
        uint32_t* batch = (uint32_t*)((uint8_t*)batches + 20 * (size_t)b);
        const uint32_t first = batch[4], count = batch[1];
        uint32_t readIndex = first, writeIndex = first;
        for (uint32_t i = 0; i < count; i++) {
            uint32_t culled;
            memcpy(&culled, inst + 96 * (size_t)readIndex + 84, 4);
            if (culled == 0) {
                if (readIndex != writeIndex) memcpy(inst + 96 * (size_t)writeIndex, inst + 96 * (size_t)readIndex, 96);
                writeIndex++;
            }
            readIndex++;
        }
        batch[1] = writeIndex - first;
    }
}

ORACLE_API void oracle_mesh_cull_compact(const void* ubo_, void* instances, uint32_t numInstances, uint32_t firstInstanceIndex,
                                         void* batches, uint32_t numBatches)
{
    mesh_cull_compact_impl(ubo_, instances, numInstances, firstInstanceIndex, batches, numBatches, NULL);
}

/* the shader as shipped (defines: OCCLUSION_CULLING): frustum || occlusion against the Hi-Z pyramid, then the compaction */
ORACLE_API void oracle_mesh_cull_compact_hiz(const void* ubo_, void* instances, uint32_t numInstances, uint32_t firstInstanceIndex, void* batches,
                                             uint32_t numBatches, const float* pyramid, int32_t width, int32_t height, int32_t levels)
{
    const OracleHiZ hz = { pyramid, width, height, levels };
    mesh_cull_compact_impl(ubo_, instances, numInstances, firstInstanceIndex, batches, numBatches, &hz);
}

/* ------------------------------------------------------------------------------------------- */
/* CSM matrix set-up (S9): FrameGraph/ShadowPrepassNode.cpp:378-404 ; Math/Bounds.cpp:78-109 ;  */
/* ECS/LightingECS.cpp:276-298                                                                  */
/* ------------------------------------------------------------------------------------------- */

static void calculate_ortho_matrix_by_view(const float* corners, const float* view, float zMult, float* out)
{
    float minX = FLT_MAX, maxX = -FLT_MAX, minY = FLT_MAX, maxY = -FLT_MAX, minZ = FLT_MAX, maxZ = -FLT_MAX;
    for (int k = 0; k < 8; k++) {
        float v[4] = { corners[3 * k], corners[3 * k + 1], corners[3 * k + 2], 1.0f }, t[4];
        glm_mat4_mul_vec4(view, v, t);
        minX = t[0] < minX ? t[0] : minX; maxX = maxX < t[0] ? t[0] : maxX; /* std::min(a,b) = (b<a)?b:a ; std::max(a,b) = (a<b)?b:a */
        minY = t[1] < minY ? t[1] : minY; maxY = maxY < t[1] ? t[1] : maxY;
        minZ = t[2] < minZ ? t[2] : minZ; maxZ = maxZ < t[2] ? t[2] : maxZ;
    }
    minZ = minZ < 0 ? minZ * zMult : minZ / zMult;
    maxZ = maxZ < 0 ? maxZ / zMult : maxZ * zMult;
    const float zFar = -minZ, zNear = -maxZ;
    ortho_rh_no(minX, maxX, minY, maxY, zFar, zNear, out);
}

/* out = 4 mat4: lightsMatrices[k] = ortho_k * lightView (LightingECS.cpp:292), lightView = inverse(lightWorld) (:227).
 * outOrtho (optional) = the 4 bare ortho matrices. */
ORACLE_API void oracle_csm_matrices(const float* lightView, const float* cameraWorld, float aspect, float fovY, float cameraNear, float cameraFar, float* out, float* outOrtho)
{
    for (int k = 0; k < NUM_CASCADES; k++) {
        const float n = k == 0 ? cameraNear : cameraFar * kShadowCascadeLevelsCpp[k - 1];
        const float f = cameraFar * kShadowCascadeLevelsCpp[k];
        float planes[24], corners[24], ortho[16];
        oracle_extract_frustum_planes(cameraWorld, aspect, fovY, n, f, planes, corners);
        calculate_ortho_matrix_by_view(corners, lightView, 10.0f, ortho);
        if (outOrtho) memcpy(outOrtho + 16 * k, ortho, sizeof ortho);
        glm_mat4_mul_mat4(ortho, lightView, out + 16 * k);
    }
}


/* ------------------------------------------------------------------------------------------- */
/* Depth rasterisation: the shadow-map producer (FrameGraph/ShadowPrepassNode.cpp:219-262,     */
/* Content/Shaders/ShadowCaster.shader:46-79).                                                 */
/* ------------------------------------------------------------------------------------------- */
/* The reference draws instanced meshes with the light matrix as a push constant, depth attachment cleared to 0, reversed Z (the
 * ortho matrices of Bounds.cpp:78-109 swap near and far), viewport (0, H, W, -H).  What a rasteriser does between the vertex
 * shader and gl_FragCoord is fixed here as follows (Vulkan's rules with the freedoms pinned):
 *   clip   = (lightMatrix * model) * vec4(position, 1)            GLSL order, mat * mat column by column, ((c0 x + c1 y) + c2 z) + c3 w
 *   clip   : triangles with a vertex beyond the near plane (z_clip > w_clip) are cut against it first -- raster_near_clip below
 *   window = x: (ndc.x + 1) * (W / 2), y: (ndc.y + 1) * (-H / 2) + H, z: ndc.z   (beyond +-2^22 pixels: dropped)
 *   snap   = x, y to 1/256 pixel (round to nearest even), 64-bit integer edge functions, both windings, top-left fill rule
 *   z      = (z0 + (z1 - z0) * w1) + (z2 - z0) * w2, w_k = float(edge_k) / float(2 * area), no fused operations
 *   test   = fragments with z outside [0, 1] are clipped; GREATER against the stored depth (0 = cleared).  (The reference's materials compare
 *            GreaterOrEqual, RHI/Types.h:537: the stored depth is the same maximum; the one difference -- a fragment at exactly z = 0 would pass against
 *            the cleared 0 and write its colour -- is not reproduced: depth 0 means "nothing drawn" here.)
 *   cull   = optional: back faces (Vulkan's signed area with frontFace COUNTER_CLOCKWISE) are discarded, as ECullMode::Back does */
static void glsl_mat4_mul_mat4(const float* a, const float* b, float* o)
{
    for (int j = 0; j < 4; j++) glsl_mat4_mul_vec4(a, b + 4 * j, o + 4 * j);
}

typedef struct { int64_t x, y; float z; } RasterVertex;

/* view == NULL: clip = LM * position with LM = lightMatrix * model (ShadowCaster.shader:58);
 * view != NULL: clip = projection * (view * (model * position)) (DepthOnly.shader:51), LM = projection here */
static void raster_clip_vertices(const float* LM, const float* view, const float* model, const float* positions, const uint32_t* tri, float clip[3][4])
{
    for (int k = 0; k < 3; k++) {
        const float* p = positions + 3 * (size_t)tri[k];
        const float pos[4] = { p[0], p[1], p[2], 1.0f };
        if (view) {
            float a[4], b[4];
            glsl_mat4_mul_vec4(model, pos, a);
            glsl_mat4_mul_vec4(view, a, b);
            glsl_mat4_mul_vec4(LM, b, clip[k]);
        } else glsl_mat4_mul_vec4(LM, pos, clip[k]);
    }
}

static void raster_cut(const float* I, float dI, const float* O, float dO, float* P)
{
    const float t = dI / (dI - dO);
    for (int c = 0; c < 4; c++) P[c] = I[c] + (O[c] - I[c]) * t;
}

/* Near-plane clipping.  With the reversed depth range the near plane is z_clip = w_clip (ndc z = 1); a vertex beyond it -- which includes
 * every vertex behind the eye -- cannot be projected.  Vulkan clips primitives against the plane geometrically; the arithmetic is pinned as:
 * d = w - z per vertex, inside = d >= 0.  All inside: the triangle as it is (fragments beyond the plane fail z <= 1).  None: dropped.
 * Otherwise the new vertex on an edge is computed from its INSIDE end, P = I + (O - I) * (dI / (dI - dO)); with one vertex inside (A; B, C
 * follow it in the triangle's own order) the result is (A, AB, AC); with two inside (A, B; C outside follows them) the quad A, B, BC, AC is
 * drawn as (A, B, BC) and (A, BC, AC).  Returns the number of triangles written to out (0, 1 or 2). */
static int raster_near_clip(float clip[3][4], float out[2][3][4])
{
    float d[3];
    int mask = 0;
    for (int k = 0; k < 3; k++) { d[k] = clip[k][3] - clip[k][2]; if (d[k] >= 0.0f) mask |= 1 << k; }
    if (mask == 0) return 0;
    if (mask == 7) { memcpy(out[0], clip, sizeof(float) * 12); return 1; }
    const int one = (mask & (mask - 1)) == 0;
    int r;
    if (one) r = mask == 1 ? 0 : (mask == 2 ? 1 : 2);
    else r = mask == 6 ? 1 : (mask == 5 ? 2 : 0); /* the vertex after the outside one */
    const float *A = clip[r], *B = clip[(r + 1) % 3], *C = clip[(r + 2) % 3];
    const float dA = d[r], dB = d[(r + 1) % 3], dC = d[(r + 2) % 3];
    if (one) {
        memcpy(out[0][0], A, 16);
        raster_cut(A, dA, B, dB, out[0][1]);
        raster_cut(A, dA, C, dC, out[0][2]);
        return 1;
    }
    float BC[4], AC[4];
    raster_cut(B, dB, C, dC, BC);
    raster_cut(A, dA, C, dC, AC);
    memcpy(out[0][0], A, 16); memcpy(out[0][1], B, 16); memcpy(out[0][2], BC, 16);
    memcpy(out[1][0], A, 16); memcpy(out[1][1], BC, 16); memcpy(out[1][2], AC, 16);
    return 2;
}

static int raster_setup(float clipTri[3][4], int W, int H, int cullBack, RasterVertex* v)
{
    for (int k = 0; k < 3; k++) {
        const float* clip = clipTri[k];
        if (!(clip[3] > 0.0f)) return 0;
        const float nx = clip[0] / clip[3], ny = clip[1] / clip[3], nz = clip[2] / clip[3];
        const float xf = (nx + 1.0f) * ((float)W * 0.5f);
        const float yf = (ny + 1.0f) * ((float)H * -0.5f) + (float)H;
        const float sx = xf * 256.0f, sy = yf * 256.0f;
        if (!(fabsf(sx) < 1.0e9f) || !(fabsf(sy) < 1.0e9f)) return 0; /* far outside the guard band (or NaN) */
        v[k].x = (int64_t)rintf(sx); v[k].y = (int64_t)rintf(sy); v[k].z = nz;
    }
    int64_t area2 = (v[1].x - v[0].x) * (v[2].y - v[0].y) - (v[2].x - v[0].x) * (v[1].y - v[0].y);
    if (area2 == 0) return 0;
    /* Vulkan: a = -1/2 sum(x_i y_j - x_j y_i) in framebuffer coordinates = -area2 / 2; frontFace is COUNTER_CLOCKWISE (VulkanPipileneStates.cpp:128), i.e.
     * a > 0 is front-facing; ECullMode::Back (the shadow and depth materials, ShadowPrepassNode.cpp:39) discards the others */
    if (cullBack && area2 > 0) return 0;
    if (area2 < 0) { const RasterVertex t = v[1]; v[1] = v[2]; v[2] = t; }
    return 1;
}

static inline int64_t raster_edge(const RasterVertex* a, const RasterVertex* b, int64_t px, int64_t py)
{
    return (b->x - a->x) * (py - a->y) - (b->y - a->y) * (px - a->x);
}
static inline int raster_top_left(const RasterVertex* a, const RasterVertex* b)
{
    const int64_t dx = b->x - a->x, dy = b->y - a->y;
    return (dy == 0 && dx > 0) || dy < 0;
}
static inline int64_t floor_div256(int64_t a) { return a >= 0 ? a / 256 : -((-a + 255) / 256); }

/* depth: W x H floats, read and written (GREATER); instanceIds == NULL draws instances 0 .. numDrawn-1 */
static void raster_depth_impl(const float* lightMatrix, const float* view, const float* positions, const uint32_t* indices, uint32_t numTriangles,
                              const float* models, const uint32_t* instanceIds, uint32_t numDrawn, int32_t W, int32_t H, float* depth, int cullBack)
{
    for (uint32_t d = 0; d < numDrawn; d++) {
        const uint32_t inst = instanceIds ? instanceIds[d] : d;
        float LM[16];
        if (view) memcpy(LM, lightMatrix, sizeof LM);
        else glsl_mat4_mul_mat4(lightMatrix, models + 16 * (size_t)inst, LM);
        for (uint32_t t = 0; t < 2 * numTriangles; t++) { /* (triangle, part): a triangle cut by the near plane can leave two */
            RasterVertex v[3];
            float clip[3][4], parts[2][3][4];
            raster_clip_vertices(LM, view, models + 16 * (size_t)inst, positions, indices + 3 * (size_t)(t >> 1), clip);
            if ((int)(t & 1) >= raster_near_clip(clip, parts)) continue;
            if (!raster_setup(parts[t & 1], W, H, cullBack, v)) continue;
            int64_t minx = v[0].x, maxx = v[0].x, miny = v[0].y, maxy = v[0].y;
            for (int k = 1; k < 3; k++) {
                minx = v[k].x < minx ? v[k].x : minx; maxx = v[k].x > maxx ? v[k].x : maxx;
                miny = v[k].y < miny ? v[k].y : miny; maxy = v[k].y > maxy ? v[k].y : maxy;
            }
            /* pixel (i, j) has its centre at (256 i + 128, 256 j + 128) */
            int64_t i0 = floor_div256(minx - 128 + 255), i1 = floor_div256(maxx - 128), j0 = floor_div256(miny - 128 + 255), j1 = floor_div256(maxy - 128);
            if (i0 < 0) i0 = 0;
            if (j0 < 0) j0 = 0;
            if (i1 > W - 1) i1 = W - 1;
            if (j1 > H - 1) j1 = H - 1;
            const float area = (float)raster_edge(&v[0], &v[1], v[2].x, v[2].y);
            const int tl0 = raster_top_left(&v[1], &v[2]), tl1 = raster_top_left(&v[2], &v[0]), tl2 = raster_top_left(&v[0], &v[1]);
            for (int64_t j = j0; j <= j1; j++)
                for (int64_t i = i0; i <= i1; i++) {
                    const int64_t px = 256 * i + 128, py = 256 * j + 128;
                    const int64_t e0 = raster_edge(&v[1], &v[2], px, py), e1 = raster_edge(&v[2], &v[0], px, py), e2 = raster_edge(&v[0], &v[1], px, py);
                    if (e0 < 0 || e1 < 0 || e2 < 0) continue;
                    if ((e0 == 0 && !tl0) || (e1 == 0 && !tl1) || (e2 == 0 && !tl2)) continue;
                    const float w1 = (float)e1 / area, w2 = (float)e2 / area;
                    const float z = (v[0].z + (v[1].z - v[0].z) * w1) + (v[2].z - v[0].z) * w2;
                    if (!(z >= 0.0f && z <= 1.0f)) continue;
                    float* o = depth + (size_t)j * W + i;
                    if (z > *o) *o = z;
                }
        }
    }
}

ORACLE_API void oracle_raster_depth(const float* lightMatrix, const float* positions, const uint32_t* indices, uint32_t numTriangles, const float* models,
                                    const uint32_t* instanceIds, uint32_t numDrawn, int32_t W, int32_t H, float* depth, int32_t cullBack)
{
    raster_depth_impl(lightMatrix, NULL, positions, indices, numTriangles, models, instanceIds, numDrawn, W, H, depth, cullBack);
}

/* The depth prepass (FrameGraph/DepthPrepassNode.cpp:283-297, Content/Shaders/DepthOnly.shader:51): the same rasteriser with the camera's matrices,
 * gl_Position = projection * (view * (model * position)); reversed-Z projection, so GREATER against the cleared 0 again. */
ORACLE_API void oracle_raster_depth_camera(const float* projection, const float* view, const float* positions, const uint32_t* indices, uint32_t numTriangles,
                                           const float* models, const uint32_t* instanceIds, uint32_t numDrawn, int32_t W, int32_t H, float* depth, int32_t cullBack)
{
    raster_depth_impl(projection, view, positions, indices, numTriangles, models, instanceIds, numDrawn, W, H, depth, cullBack);
}

/* ShadowCaster.shader:66-78 on the winning fragment of every texel: EVSM moments (RGBA32F) or the depth itself; texels nothing was drawn
 * to keep the cleared colour 0 (ShadowPrepassNode.cpp:239-248: clear colour vec4(0), clear depth 0). */
ORACLE_API void oracle_shadow_resolve_evsm(const float* depth, int32_t W, int32_t H, float* outRGBA)
{
    for (size_t i = 0; i < (size_t)W * H; i++) {
        const float z = depth[i];
        float* o = outRGBA + 4 * i;
        if (z > 0.0f) {
            o[0] = canonical_expf(40.0f * z); o[1] = o[0] * o[0];
            o[2] = -canonical_expf(-40.0f * z); o[3] = o[2] * o[2];
        } else { o[0] = o[1] = o[2] = o[3] = 0.0f; }
    }
}

/* ==== E4 as the reference runs it: RHISceneView::TraceScene over TOctree ==========================================
 * The product sweeps FLAT float boxes ("octree stores integer-truncated boxes -- N").  This restates the reference's own
 * structure so that the size of that divergence can be COUNTED (tests/test_oracle_cpu.py: which entities' visibility flips):
 *   ECS/StaticMeshRendererECS.cpp:81,96,132  octree.Update(glm::vec4(aabb.GetCenter(), 1), aabb.GetExtents(), proxy) -- the parameters are
 *                                            `const glm::ivec3&`: centre and extents are TRUNCATED towards zero, component by component
 *                                            (Math/Bounds.cpp:455-463 GetCenter = (min + max) * 0.5f, GetExtents = (max - min) * 0.5f)
 *   RHI/SceneView.h:91-92                    TOctree{ ivec3(0), 16536 * 16, 4 }: root size 264 576, minimum node size 4
 *   Containers/Octree.h:44-53,397-437        TNode::Contains (strict, integers), Insert_Internal; :22 NumElementsInNode = 8
 *   Containers/Octree.h:43                   GetIndex(x, y, z) = (z < 0 ? 0 : 1) + (x < 0 ? 1 : 0) * 2 + (y < 0 ? 0 : 1) * 4
 *   Containers/Octree.h:439-459              Subdivide: quarter = size / 4, child size = 2 quarter, centres = offset[i] * quarter + centre
 *   Containers/Octree.h:239-274              Trace / Trace_Internal: Frustum::OverlapsAABB(AABB(ivec3 position, ivec3 extents)) per stored element,
 *                                            AABB(centre, size * 0.5f) per child node (Math/Bounds.cpp:473-477: min = c - e, max = c + e)
 * An element the root does not strictly contain is NOT inserted (Insert returns false): it is never drawn.  The reference walks a node's elements in
 * its hash map's order; here in insertion order -- the traced SET cannot depend on it (OverlapsAABB is monotone in the box, every element lies inside
 * every node on its path: the hierarchical walk visits exactly the elements whose own integer box passes; the test holds this against the flat form). */
typedef struct OctNode {
    uint32_t size; int32_t c[3];
    int32_t child;              /* index of the first of eight children, -1 = leaf */
    uint32_t num, cap; uint32_t* el;
} OctNode;
typedef struct Octree { OctNode* nodes; size_t numNodes, capNodes; uint32_t minSize; const int32_t* pos; const int32_t* ext; } Octree;

static int oct_contains(const OctNode* n, const int32_t* p, const int32_t* e)
{
    const int32_t h = (int32_t)(n->size / 2);
    return n->c[0] - h < p[0] - e[0] && n->c[1] - h < p[1] - e[1] && n->c[2] - h < p[2] - e[2] &&
           n->c[0] + h > p[0] + e[0] && n->c[1] + h > p[1] + e[1] && n->c[2] + h > p[2] + e[2];
}
static void oct_add(OctNode* n, uint32_t el)
{
    if (n->num == n->cap) { n->cap = n->cap ? n->cap * 2 : 8; n->el = (uint32_t*)realloc(n->el, n->cap * sizeof(uint32_t)); }
    n->el[n->num++] = el;
}
static void oct_subdivide(Octree* t, size_t ni)
{
    static const int32_t off[8][3] = { { 1, -1, -1 }, { 1, -1, 1 }, { -1, -1, -1 }, { -1, -1, 1 }, { 1, 1, -1 }, { 1, 1, 1 }, { -1, 1, -1 }, { -1, 1, 1 } };
    if (t->numNodes + 8 > t->capNodes) { t->capNodes = t->capNodes * 2 + 8; t->nodes = (OctNode*)realloc(t->nodes, t->capNodes * sizeof(OctNode)); }
    OctNode* n = &t->nodes[ni];
    const int32_t quarter = (int32_t)(n->size / 4);
    n->child = (int32_t)t->numNodes;
    for (int i = 0; i < 8; i++) {
        OctNode* k = &t->nodes[t->numNodes + i];
        memset(k, 0, sizeof *k);
        k->size = (uint32_t)(quarter * 2); k->child = -1;
        for (int a = 0; a < 3; a++) k->c[a] = off[i][a] * quarter + n->c[a];
    }
    t->numNodes += 8;
}
static int oct_insert(Octree* t, size_t ni, uint32_t el)
{
    const int32_t* p = t->pos + 3 * (size_t)el; const int32_t* e = t->ext + 3 * (size_t)el;
    const int leaf = t->nodes[ni].child < 0;
    if (!oct_contains(&t->nodes[ni], p, e)) return 0;
    if (leaf) {
        oct_add(&t->nodes[ni], el);
        if (t->nodes[ni].num == 8u && t->nodes[ni].size > t->minSize) {
            uint32_t moved[8];
            memcpy(moved, t->nodes[ni].el, sizeof moved);
            t->nodes[ni].num = 0;
            oct_subdivide(t, ni);                                   /* (may move t->nodes) */
            for (int i = 0; i < 8; i++) (void)oct_insert(t, ni, moved[i]);
        }
        return 1;
    }
    const int32_t dx = p[0] - t->nodes[ni].c[0], dy = p[1] - t->nodes[ni].c[1], dz = p[2] - t->nodes[ni].c[2];
    const int idx = (dz < 0 ? 0 : 1) + (dx < 0 ? 1 : 0) * 2 + (dy < 0 ? 0 : 1) * 4;
    if (oct_insert(t, (size_t)t->nodes[ni].child + (size_t)idx, el)) return 1;
    oct_add(&t->nodes[ni], el);
    return 1;
}
static int oct_overlaps_int_box(const float* planes, const int32_t* p, const int32_t* e)
{
    float box[6]; /* AABB(glm::vec3(ivec3 position), glm::vec3(ivec3 extents)): min = centre - extents, max = centre + extents (floats) */
    for (int a = 0; a < 3; a++) { box[a] = (float)p[a] - (float)e[a]; box[3 + a] = (float)p[a] + (float)e[a]; }
    return overlaps_aabb(planes, box);
}
static void oct_trace(const Octree* t, size_t ni, const float* planes, uint64_t* visible, uint64_t* visited)
{
    const OctNode* n = &t->nodes[ni];
    for (uint32_t i = 0; i < n->num; i++) {
        const uint32_t el = n->el[i];
        (*visited)++;
        if (oct_overlaps_int_box(planes, t->pos + 3 * (size_t)el, t->ext + 3 * (size_t)el)) visible[el >> 6] |= 1ull << (el & 63);
    }
    if (n->child < 0) return;
    for (int i = 0; i < 8; i++) {
        const OctNode* k = &t->nodes[(size_t)n->child + (size_t)i];
        const float h = (float)k->size * 0.5f;
        const float box[6] = { (float)k->c[0] - h, (float)k->c[1] - h, (float)k->c[2] - h, (float)k->c[0] + h, (float)k->c[1] + h, (float)k->c[2] + h };
        if (overlaps_aabb(planes, box)) oct_trace(t, (size_t)n->child + (size_t)i, planes, visible, visited);
    }
}
/* worldAabb: n x {min.xyz, max.xyz} (what K4 / oracle_ecs_sweep produce).  outVisible / outInserted: ceil(n / 64) words.  outIntBoxes (or NULL): n x
 * {position.xyz, extents.xyz} as int32 -- the truncated boxes, for the flat cross-check.  stats[4]: nodes, elements visited by the trace, elements not
 * inserted, deepest node size.  Returns 0, or -1 when out of memory. */
ORACLE_API int oracle_trace_scene_octree_boxes(uint32_t n, const float* worldAabb, const float* planes, uint32_t rootSize, uint32_t minSize,
                                               uint64_t* outVisible, uint64_t* outInserted, int32_t* outIntBoxes, uint64_t* stats)
{
    const uint32_t words = (n + 63) / 64;
    memset(outVisible, 0, (size_t)words * 8);
    memset(outInserted, 0, (size_t)words * 8);
    int32_t* pos = (int32_t*)malloc((size_t)(n ? n : 1) * 3 * sizeof(int32_t));
    int32_t* ext = (int32_t*)malloc((size_t)(n ? n : 1) * 3 * sizeof(int32_t));
    Octree t; memset(&t, 0, sizeof t);
    t.capNodes = 1024; t.nodes = (OctNode*)malloc(t.capNodes * sizeof(OctNode));
    if (!pos || !ext || !t.nodes) { free(pos); free(ext); free(t.nodes); return -1; }
    memset(&t.nodes[0], 0, sizeof(OctNode));
    t.nodes[0].size = rootSize; t.nodes[0].child = -1; t.numNodes = 1; t.minSize = minSize; t.pos = pos; t.ext = ext;
    uint64_t notInserted = 0;
    for (uint32_t i = 0; i < n; i++) {
        const float* b = worldAabb + 6 * (size_t)i;
        for (int a = 0; a < 3; a++) {
            const float c = (b[a] + b[3 + a]) * 0.5f, e = (b[3 + a] - b[a]) * 0.5f;   /* GetCenter / GetExtents */
            pos[3 * (size_t)i + a] = (int32_t)c;                                        /* glm::ivec3(glm::vec4 / glm::vec3): truncation */
            ext[3 * (size_t)i + a] = (int32_t)e;
        }
        if (outIntBoxes) { memcpy(outIntBoxes + 6 * (size_t)i, pos + 3 * (size_t)i, 12); memcpy(outIntBoxes + 6 * (size_t)i + 3, ext + 3 * (size_t)i, 12); }
        if (oct_insert(&t, 0, i)) outInserted[i >> 6] |= 1ull << (i & 63);
        else notInserted++;
    }
    uint64_t visited = 0;
    {   /* Trace (:239-247): the root's own box first */
        const float h = (float)t.nodes[0].size * 0.5f;
        const float box[6] = { (float)t.nodes[0].c[0] - h, (float)t.nodes[0].c[1] - h, (float)t.nodes[0].c[2] - h,
                               (float)t.nodes[0].c[0] + h, (float)t.nodes[0].c[1] + h, (float)t.nodes[0].c[2] + h };
        if (overlaps_aabb(planes, box)) oct_trace(&t, 0, planes, outVisible, &visited);
    }
    uint32_t smallest = rootSize;
    for (size_t k = 0; k < t.numNodes; k++) { if (t.nodes[k].size < smallest) smallest = t.nodes[k].size; free(t.nodes[k].el); }
    if (stats) { stats[0] = t.numNodes; stats[1] = visited; stats[2] = notInserted; stats[3] = smallest; }
    free(t.nodes); free(pos); free(ext);
    return 0;
}
