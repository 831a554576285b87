// The two cubemap pre-filters behind the ambient term (FrameGraph/EnvironmentNode.cpp:196-273), for gfx950:
//   ComputeIrradianceMap.shader  65 536 uniform hemisphere samples per texel of the 32 x 32 x 6 irradiance cube,
//   ComputeEnvMap_IBL.shader     1 024 GGX importance samples per texel of every mip of the 512 x 512 x 6 environment cube,
//                                with mip-filtered lookups into the raw cube.
// The shaders run one invocation per output texel (32 x 32 groups) and loop over the samples.  The irradiance cube has only
// 6 144 texels -- 96 wavefronts for 400 M samples -- so here the SAMPLES are spread over lanes: one 256-thread block per
// irradiance texel (lane j takes samples j, j + 256, ... in order), one wavefront per environment texel (16 samples per lane),
// and the partial sums are added up in a fixed tree.  The sums therefore differ from the shader's sequential order in the last
// bits (the oracle follows the shader); everything else -- Hammersley points, basis vectors, GGX sampling, pdf -> mip level,
// the canonical cube sampler -- is the oracle's arithmetic.  Tolerance-checked like the rest of the shading path.
#include "common.h"
#include "sampling.h"

#define PF_TWO_PI 6.283185307179586f
#define PF_PI 3.14159265359f

__device__ __forceinline__ float pf_radical_inverse(uint32_t i) { return (float)__brev(i) * 2.3283064365386963e-10f; } // Math.glsl:285-293

__device__ __forceinline__ void pf_normalize(float& x, float& y, float& z)
{
    const float inv = 1.0f / sqrtf((x * x + y * y) + z * z);
    x *= inv; y *= inv; z *= inv;
}

struct PfFrame { float Nx, Ny, Nz, Sx, Sy, Sz, Tx, Ty, Tz; };

// GetSamplingVector + ComputeBasisVectors (ComputeIrradianceMap.shader:43-69): texel (x, y) of `face`, from the texel's corner
__device__ __forceinline__ PfFrame pf_frame(int x, int y, int face, int size)
{
    const float stx = (float)x / (float)size, sty = (float)y / (float)size;
    const float ux = 2.0f * stx - 1.0f, uy = 2.0f * (1.0f - sty) - 1.0f;
    PfFrame f;
    switch (face) {
    case 0: f.Nx = 1.0f; f.Ny = uy; f.Nz = -ux; break;
    case 1: f.Nx = -1.0f; f.Ny = uy; f.Nz = ux; break;
    case 2: f.Nx = ux; f.Ny = 1.0f; f.Nz = -uy; break;
    case 3: f.Nx = ux; f.Ny = -1.0f; f.Nz = uy; break;
    case 4: f.Nx = ux; f.Ny = uy; f.Nz = 1.0f; break;
    default: f.Nx = -ux; f.Ny = uy; f.Nz = -1.0f; break;
    }
    pf_normalize(f.Nx, f.Ny, f.Nz);
    f.Tx = f.Ny * 0.0f - f.Nz * 1.0f; f.Ty = f.Nz * 0.0f - f.Nx * 0.0f; f.Tz = f.Nx * 1.0f - f.Ny * 0.0f; // cross(N, +Y)
    if ((f.Tx * f.Tx + f.Ty * f.Ty) + f.Tz * f.Tz < 0.00001f) {                                             // degenerate: cross(N, +X)
        f.Tx = f.Ny * 0.0f - f.Nz * 0.0f; f.Ty = f.Nz * 1.0f - f.Nx * 0.0f; f.Tz = f.Nx * 0.0f - f.Ny * 1.0f;
    }
    pf_normalize(f.Tx, f.Ty, f.Tz);
    f.Sx = f.Ny * f.Tz - f.Nz * f.Ty; f.Sy = f.Nz * f.Tx - f.Nx * f.Tz; f.Sz = f.Nx * f.Ty - f.Ny * f.Tx;
    pf_normalize(f.Sx, f.Sy, f.Sz);
    return f;
}

__device__ __forceinline__ void pf_to_world(const PfFrame& f, float hx, float hy, float hz, float& ox, float& oy, float& oz) // (:72-75)
{
    ox = (f.Sx * hx + f.Tx * hy) + f.Nx * hz;
    oy = (f.Sy * hx + f.Ty * hy) + f.Ny * hz;
    oz = (f.Sz * hx + f.Tz * hy) + f.Nz * hz;
}

__device__ __forceinline__ float pf_wave_sum(float v)
{
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) v += __shfl_xor(v, d, 64);
    return v;
}

// ---- ComputeIrradianceMap.shader:78-101 ----
#define IRR_SAMPLES (64u * 1024u)
__global__ __launch_bounds__(256) void k_irradiance_map(const float4* __restrict__ env, int envSize, int envLevels, float4* __restrict__ out, int size)
{
    __shared__ float sPart[3][4];
    const int texel = blockIdx.x; // (face, y, x)
    const int x = texel % size, y = (texel / size) % size, face = texel / (size * size);
    const PfFrame f = pf_frame(x, y, face, size);
    const float InvNumSamples = 1.0f / (float)IRR_SAMPLES;
    float r = 0.0f, g = 0.0f, b = 0.0f;
    for (uint32_t i = threadIdx.x; i < IRR_SAMPLES; i += 256u) {
        const float u1 = (float)i * InvNumSamples, u2 = pf_radical_inverse(i);
        const float u1p = sqrtf(fmaxf(0.0f, 1.0f - u1 * u1));
        float sn, cs;
        sincosf(PF_TWO_PI * u2, &sn, &cs);
        float lx, ly, lz;
        pf_to_world(f, cs * u1p, sn * u1p, u1, lx, ly, lz); // SampleHemisphere (:32-36)
        const float cosTheta = fmaxf(0.0f, (lx * f.Nx + ly * f.Ny) + lz * f.Nz);
        // textureLod(envMap, Li, 0): level 0 only (the canonical sampler's second level has weight exactly 0 there)
        int sface; float ss, st;
        cube_face_st(lx, ly, lz, sface, ss, st);
        const float4 t = cube_sample_level(env, envSize, 0, sface, ss, st);
        r += (2.0f * t.x) * cosTheta; g += (2.0f * t.y) * cosTheta; b += (2.0f * t.z) * cosTheta;
    }
    r = pf_wave_sum(r); g = pf_wave_sum(g); b = pf_wave_sum(b);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (lane == 0) { sPart[0][wave] = r; sPart[1][wave] = g; sPart[2][wave] = b; }
    __syncthreads();
    if (threadIdx.x == 0) {
        const float n = (float)IRR_SAMPLES;
        out[texel] = make_float4((((sPart[0][0] + sPart[0][1]) + sPart[0][2]) + sPart[0][3]) / n, (((sPart[1][0] + sPart[1][1]) + sPart[1][2]) + sPart[1][3]) / n,
                                 (((sPart[2][0] + sPart[2][1]) + sPart[2][2]) + sPart[2][3]) / n, 1.0f);
    }
}

// ---- ComputeEnvMap_IBL.shader:76-136, one output mip level per launch ----
#define ENV_SAMPLES 1024u
__global__ __launch_bounds__(256) void k_prefilter_env(const float4* __restrict__ raw, int size0, int levels, float4* __restrict__ outLevel, int size, float roughness)
{
    const int texel = blockIdx.x * 4 + (threadIdx.x >> 6); // one wavefront per (face, y, x)
    if (texel >= 6 * size * size) return;
    const int lane = threadIdx.x & 63;
    const int x = texel % size, y = (texel / size) % size, face = texel / (size * size);
    const PfFrame f = pf_frame(x, y, face, size);
    const float InvNumSamples = 1.0f / (float)ENV_SAMPLES;
    const float wt = 4.0f * PF_PI / (6.0f * (float)size0 * (float)size0); // solid angle of a texel of the raw cube's level 0 (:86-87)
    const float alpha = roughness * roughness, alphaSq = alpha * alpha;
    float r = 0.0f, g = 0.0f, b = 0.0f, weight = 0.0f;
    for (uint32_t i = lane; i < ENV_SAMPLES; i += 64u) {
        const float u1 = (float)i * InvNumSamples, u2 = pf_radical_inverse(i);
        // SampleGGX (Lighting.glsl:27-37)
        const float cosT = sqrtf((1.0f - u2) / (1.0f + (alpha * alpha - 1.0f) * u2));
        const float sinT = sqrtf(1.0f - cosT * cosT);
        float sn, cs;
        sincosf(PF_TWO_PI * u1, &sn, &cs);
        float hx, hy, hz;
        pf_to_world(f, sinT * cs, sinT * sn, cosT, hx, hy, hz);
        const float d = (f.Nx * hx + f.Ny * hy) + f.Nz * hz; // dot(Lo, Lh), Lo = N (:92)
        const float lx = (2.0f * d) * hx - f.Nx, ly = (2.0f * d) * hy - f.Ny, lz = (2.0f * d) * hz - f.Nz;
        const float cosLi = (f.Nx * lx + f.Ny * ly) + f.Nz * lz;
        if (cosLi > 0.0f) {
            const float cosLh = fmaxf(d, 0.0f);
            const float denom = (cosLh * cosLh) * (alphaSq - 1.0f) + 1.0f;
            const float pdf = (alphaSq / (PF_PI * denom * denom)) * 0.25f; // NdfGGX * 0.25 (:121)
            const float ws = 1.0f / ((float)ENV_SAMPLES * pdf);
            const float mip = fmaxf(0.5f * log2f(ws / wt) + 1.0f, 0.0f);    // (:127)
            const float4 t = cube_sample_lod(raw, size0, levels, lx, ly, lz, mip);
            r += t.x * cosLi; g += t.y * cosLi; b += t.z * cosLi;
            weight += cosLi;
        }
    }
    r = pf_wave_sum(r); g = pf_wave_sum(g); b = pf_wave_sum(b); weight = pf_wave_sum(weight);
    if (lane == 0) outLevel[texel] = make_float4(r / weight, g / weight, b / weight, 1.0f);
}

// ---- the raw environment cube: ComputeEquirect2Cube.shader + GenerateMipMaps (EnvironmentNode.cpp:116-140) ----------------------
// One lane per cube texel, x fastest: a wavefront walks 64 neighbouring directions, so its four-tap footprints in the equirect
// image are neighbouring float4s.  HBM-bound: 16 B written per texel, the equirect image read about once.
__global__ __launch_bounds__(256) void k_equirect_to_cube(const float4* __restrict__ equirect, int eqW, int eqH, int repeat,
                                                          float4* __restrict__ cube, int size, int coverW, int coverH)
{
    const int x = blockIdx.x * 64 + (threadIdx.x & 63), y = blockIdx.y * 4 + (threadIdx.x >> 6), face = blockIdx.z;
    if (x >= size || y >= size || x >= coverW || y >= coverH) return;
    const float PI = 3.141592f, TwoPI = 2.0f * PI;
    const PfFrame f = pf_frame(x, y, face, size); // the same face table (ComputeEquirect2Cube.shader:27-33), before normalisation
    const float len = sqrtf(f.Nx * f.Nx + f.Ny * f.Ny + f.Nz * f.Nz);
    const float vx = f.Nx / len, vy = f.Ny / len, vz = f.Nz / len;
    const float u = atan2f(vz, vx) / TwoPI, v = acosf(vy) / PI;
    const float px = u * (float)eqW - 0.5f, py = v * (float)eqH - 0.5f;
    const float fx = floorf(px), fy = floorf(py);
    const float ax = px - fx, ay = py - fy;
    int x0 = (int)fx, y0 = (int)fy, x1 = x0 + 1, y1 = y0 + 1;
    if (repeat) {
        x0 = ((x0 % eqW) + eqW) % eqW; x1 = ((x1 % eqW) + eqW) % eqW;
        y0 = ((y0 % eqH) + eqH) % eqH; y1 = ((y1 % eqH) + eqH) % eqH;
    } else {
        x0 = min(max(x0, 0), eqW - 1); x1 = min(max(x1, 0), eqW - 1);
        y0 = min(max(y0, 0), eqH - 1); y1 = min(max(y1, 0), eqH - 1);
    }
    const float4 a = equirect[(size_t)y0 * eqW + x0], b = equirect[(size_t)y0 * eqW + x1];
    const float4 c = equirect[(size_t)y1 * eqW + x0], d = equirect[(size_t)y1 * eqW + x1];
    cube[((size_t)face * size + y) * size + x] = make_float4(lerp2(a.x, b.x, c.x, d.x, ax, ay), lerp2(a.y, b.y, c.y, d.y, ax, ay),
                                                             lerp2(a.z, b.z, c.z, d.z, ax, ay), lerp2(a.w, b.w, c.w, d.w, ax, ay));
}

// One level of the 2:1 linear blit chain, all six faces: a lane per destination texel, two float4 pairs in.
__global__ __launch_bounds__(256) void k_cube_mip_level(const float4* __restrict__ src, int ss, float4* __restrict__ dst, int ds)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= 6 * ds * ds) return;
    const int face = i / (ds * ds), r = i - face * ds * ds, y = r / ds, x = r - y * ds;
    const int x0 = ss > 1 ? 2 * x : 0, x1 = ss > 1 ? 2 * x + 1 : 0, y0 = ss > 1 ? 2 * y : 0, y1 = ss > 1 ? 2 * y + 1 : 0;
    const float4* s = src + (size_t)face * ss * ss;
    const float4 a = s[(size_t)y0 * ss + x0], b = s[(size_t)y0 * ss + x1], c = s[(size_t)y1 * ss + x0], d = s[(size_t)y1 * ss + x1];
    dst[i] = make_float4(((a.x + b.x) + (c.x + d.x)) * 0.25f, ((a.y + b.y) + (c.y + d.y)) * 0.25f,
                         ((a.z + b.z) + (c.z + d.z)) * 0.25f, ((a.w + b.w) + (c.w + d.w)) * 0.25f);
}

extern "C" {

int sailor_hip_equirect_to_cube(SailorHipContext* ctx, const float* dEquirect, int32_t eqWidth, int32_t eqHeight, int32_t repeat,
                                float* dCube, int32_t size, int32_t coverWidth, int32_t coverHeight)
{
    if (!ctx || !dEquirect || !dCube || eqWidth <= 0 || eqHeight <= 0 || eqWidth > 32768 || eqHeight > 32768 || size <= 0 || size > 8192)
        return SAILOR_HIP_ERR_INVALID_ARGUMENT;
    if (((uintptr_t)dEquirect & 15) || ((uintptr_t)dCube & 15) || coverWidth < 0 || coverHeight < 0) return SAILOR_HIP_ERR_INVALID_ARGUMENT;
    SAILOR_TRY_HIP(ctx, hipSetDevice(ctx->device)); // a host thread may drive several contexts
    const int cw = coverWidth < size ? coverWidth : size, ch = coverHeight < size ? coverHeight : size;
    if (cw == 0 || ch == 0) return SAILOR_HIP_OK;
    hipLaunchKernelGGL(k_equirect_to_cube, dim3((cw + 63) / 64, (ch + 3) / 4, 6), dim3(256), 0, ctx->stream,
                       (const float4*)dEquirect, eqWidth, eqHeight, repeat ? 1 : 0, (float4*)dCube, size, cw, ch);
    SAILOR_CHECK_LAUNCH(ctx, "k_equirect_to_cube");
    return SAILOR_HIP_OK;
}

int sailor_hip_generate_mipmaps_cube(SailorHipContext* ctx, float* dCube, int32_t size, int32_t levels)
{
    if (!ctx || !dCube || size <= 0 || size > 8192 || levels <= 0 || levels > 16 || ((uintptr_t)dCube & 15)) return SAILOR_HIP_ERR_INVALID_ARGUMENT;
    SAILOR_TRY_HIP(ctx, hipSetDevice(ctx->device)); // a host thread may drive several contexts
    float4* src = (float4*)dCube;
    for (int l = 1; l < levels; l++) {
        const int ss = (size >> (l - 1)) > 1 ? (size >> (l - 1)) : 1, ds = ss > 1 ? ss / 2 : 1;
        float4* dst = src + (size_t)6 * ss * ss;
        hipLaunchKernelGGL(k_cube_mip_level, dim3((6 * ds * ds + 255) / 256), dim3(256), 0, ctx->stream, (const float4*)src, ss, dst, ds);
        SAILOR_CHECK_LAUNCH(ctx, "k_cube_mip_level");
        src = dst;
    }
    return SAILOR_HIP_OK;
}


int sailor_hip_compute_irradiance_map(SailorHipContext* ctx, const float* dEnv, int32_t envSize, int32_t envLevels, float* dIrradiance, int32_t size)
{
    if (!ctx || !dEnv || !dIrradiance || envSize <= 0 || envLevels <= 0 || envLevels > 16 || size <= 0 || size > 4096) return SAILOR_HIP_ERR_INVALID_ARGUMENT;
    SAILOR_TRY_HIP(ctx, hipSetDevice(ctx->device)); // a host thread may drive several contexts
    if (((uintptr_t)dEnv & 15) || ((uintptr_t)dIrradiance & 15)) return SAILOR_HIP_ERR_INVALID_ARGUMENT;
    hipLaunchKernelGGL(k_irradiance_map, dim3(6u * (unsigned)size * (unsigned)size), dim3(256), 0, ctx->stream, (const float4*)dEnv, envSize, envLevels,
                       (float4*)dIrradiance, size);
    SAILOR_CHECK_LAUNCH(ctx, "k_irradiance_map");
    return SAILOR_HIP_OK;
}

static size_t pf_level_offset(int size, int level) // float4 texels in front of `level`
{
    size_t off = 0;
    for (int l = 0; l < level; l++) { const int sz = (size >> l) > 1 ? (size >> l) : 1; off += (size_t)6 * sz * sz; }
    return off;
}

int sailor_hip_prefilter_env_level(SailorHipContext* ctx, const float* dRawEnv, float* dEnv, int32_t size, int32_t levels, int32_t level, float roughness)
{
    if (!ctx || !dRawEnv || !dEnv || size <= 0 || size > 8192 || levels <= 0 || levels > 16 || level < 0 || level >= levels) return SAILOR_HIP_ERR_INVALID_ARGUMENT;
    SAILOR_TRY_HIP(ctx, hipSetDevice(ctx->device)); // a host thread may drive several contexts
    if (((uintptr_t)dRawEnv & 15) || ((uintptr_t)dEnv & 15) || dRawEnv == dEnv) return SAILOR_HIP_ERR_INVALID_ARGUMENT;
    const int sz = (size >> level) > 1 ? (size >> level) : 1;
    const unsigned texels = 6u * (unsigned)sz * (unsigned)sz;
    hipLaunchKernelGGL(k_prefilter_env, dim3((texels + 3) / 4), dim3(256), 0, ctx->stream, (const float4*)dRawEnv, size, levels,
                       (float4*)dEnv + pf_level_offset(size, level), sz, roughness);
    SAILOR_CHECK_LAUNCH(ctx, "k_prefilter_env");
    return SAILOR_HIP_OK;
}

int sailor_hip_prefilter_env_map(SailorHipContext* ctx, const float* dRawEnv, float* dEnv, int32_t size, int32_t levels)
{
    if (!ctx || !dRawEnv || !dEnv || size <= 0 || size > 8192 || levels <= 0 || levels > 16) return SAILOR_HIP_ERR_INVALID_ARGUMENT;
    if (((uintptr_t)dRawEnv & 15) || ((uintptr_t)dEnv & 15) || dRawEnv == dEnv) return SAILOR_HIP_ERR_INVALID_ARGUMENT;
    // level 0 is copied (the BlitImage at EnvironmentNode.cpp:200-203), the mip tail is pre-filtered level by level (:219-233)
    SAILOR_TRY_HIP(ctx, hipMemcpyAsync(dEnv, dRawEnv, (size_t)6 * size * size * 16, hipMemcpyDeviceToDevice, ctx->stream));
    const float deltaRoughness = 1.0f / (levels - 1 > 1 ? (float)(levels - 1) : 1.0f);
    for (int level = 1; level < levels; level++) {
        const int rc = sailor_hip_prefilter_env_level(ctx, dRawEnv, dEnv, size, levels, level, (float)level * deltaRoughness);
        if (rc != SAILOR_HIP_OK) return rc;
    }
    return SAILOR_HIP_OK;
}

} // extern "C"
