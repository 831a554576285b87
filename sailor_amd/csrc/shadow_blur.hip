// EVSM shadow-map blur for gfx950.
//
// Replaces the two full-screen draws "Blur Horizontal" / "Blur Vertical" of ShadowPrepassNode::Process
// (FrameGraph/ShadowPrepassNode.cpp:283-356): Content/Shaders/Blur.shader:66-98 (defines EVSM + HORIZONTAL | VERTICAL) ->
// GaussianBlur_Evsm (Lighting.glsl:83-127) over the cascade-0 moments map (RGBA32F, 4096^2 = 256 MiB).
//
// A separable stencil of at most 12 + 11 taps per channel pair; every output reads its taps straight from global memory -- the
// horizontal pass's neighbours are the neighbouring lanes' addresses (one line serves a whole wave), the vertical pass's are the
// same columns of adjacent rows (each its own fully coalesced 1 KiB request), so the vector L1 / L2 absorb the reuse and HBM sees
// each texel once per pass: 32 bytes per texel and pass.  The .xy and .zw halves have different radii (penumbra / umbra) and are
// fetched as separate 8-byte halves, so no byte is requested that the shader would not use.  Sums in the shader's order, no FMA
// contraction (-ffp-contract=off): bit-exact against oracle_evsm_blur_pass.
#include "common.h"

struct BlurWeights { float w1[12], w2[12]; }; // rows blurRadius1-1 / blurRadius2-1 of Lighting.glsl:87-99

#define EVSM_WEIGHT_ROWS { /* Lighting.glsl:87-99 */ \
    { 0.5f, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0 }, \
    { 0.281088f, 0.218912f, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0 }, \
    { 0.197159f, 0.176426f, 0.126415f, 0, 0, 0, 0, 0, 0, 0, 0, 0 }, \
    { 0.152068f, 0.142855f, 0.118431f, 0.0866459f, 0, 0, 0, 0, 0, 0, 0, 0 }, \
    { 0.123827f, 0.118971f, 0.105518f, 0.0863909f, 0.0652929f, 0, 0, 0, 0, 0, 0, 0 }, \
    { 0.104454f, 0.101593f, 0.0934699f, 0.0813492f, 0.0669741f, 0.0521595f, 0, 0, 0, 0, 0, 0 }, \
    { 0.0903332f, 0.0885083f, 0.083252f, 0.0751759f, 0.0651684f, 0.0542336f, 0.0433285f, 0, 0, 0, 0, 0 }, \
    { 0.07958f, 0.0783462f, 0.0747585f, 0.0691403f, 0.061977f, 0.0538465f, 0.0453433f, 0.0370081f, 0, 0, 0, 0 }, \
    { 0.0711171f, 0.0702445f, 0.0676904f, 0.0636383f, 0.0583697f, 0.0522315f, 0.0455989f, 0.0388376f, 0.0322721f, 0, 0, 0 }, \
    { 0.0642825f, 0.0636429f, 0.0617619f, 0.0587498f, 0.0547779f, 0.0500633f, 0.0448484f, 0.0393811f, 0.0338957f, 0.0285966f, 0, 0 }, \
    { 0.0586472f, 0.0581645f, 0.0567402f, 0.0544433f, 0.0513831f, 0.0476999f, 0.0435548f, 0.039118f, 0.0345572f, 0.0300277f, 0.0256641f, 0 }, \
    { 0.0539209f, 0.0535478f, 0.0524437f, 0.050654f, 0.0482506f, 0.0453272f, 0.0419936f, 0.0383686f, 0.034573f, 0.0307232f, 0.0269255f, 0.0232718f } }
static const float kEvsmBlurWeights[12][12] = EVSM_WEIGHT_ROWS;                       // host: rows handed to the generic kernel
__device__ static constexpr float kEvsmBlurWeightsDev[12][12] = EVSM_WEIGHT_ROWS;  // device: literals of the unrolled kernels

// RX / RY > 0: radii known at compile time (the loop unrolls, the weights become literals, the "which half does this tap need" tests
// disappear); 0: taken from the arguments.  Texel indices are 32-bit (the host checks W * H < 2^30).
template <bool VERTICAL, int RX, int RY>
__global__ __launch_bounds__(256) void k_evsm_blur(const float2* __restrict__ src, float2* __restrict__ dst, int W, int H, int radiusXarg, int radiusYarg,
                                                   int blurRadiusArg, BlurWeights K)
{
    // horizontal: 256 x 1 texels per block (the taps are the neighbouring lanes' addresses).  vertical: 32 x 8 -- the 8 rows of a block
    // share their tap rows through the CU's vector L1 (16 rows x 512 B = 8 KB footprint)
    const int x = VERTICAL ? (int)(blockIdx.x * 32 + (threadIdx.x & 31)) : (int)(blockIdx.x * 256 + threadIdx.x);
    const int y = VERTICAL ? (int)(blockIdx.y * 8 + (threadIdx.x >> 5)) : (int)blockIdx.y;
    if (x >= W || y >= H) return;
    constexpr bool STATIC = RX > 0 || RY > 0;
    const int radiusX = STATIC ? RX : radiusXarg, radiusY = STATIC ? RY : radiusYarg;
    const int blurRadius = STATIC ? (RX > RY ? RX : RY) : blurRadiusArg;
    float sx = 0.0f, sy = 0.0f, sz = 0.0f, sw = 0.0f;
    const float4* __restrict__ src4 = reinterpret_cast<const float4*>(src);
    const uint32_t row = (uint32_t)y * (uint32_t)W;
#pragma unroll
    for (int i = 0; i < (STATIC ? (RX > RY ? RX : RY) : 12); i++) {
        if (!STATIC && i >= blurRadius) break;
        uint32_t a, b; // texel index; as float2: [2t] = .xy, [2t + 1] = .zw
        if (VERTICAL) { a = (uint32_t)min(y + i, H - 1) * (uint32_t)W + (uint32_t)x; b = (uint32_t)max(y - i, 0) * (uint32_t)W + (uint32_t)x; }
        else          { a = row + (uint32_t)min(x + i, W - 1); b = row + (uint32_t)max(x - i, 0); }
        const bool both = i < radiusX && i < radiusY;
        // one request per distinct texel and needed half: the centre tap (i == 0) is the same texel twice, and where both radii cover
        // the tap the whole float4 is fetched at once
        float2 axy = make_float2(0, 0), azw = axy, bxy = axy, bzw = axy;
        if (both) {
            const float4 pa = src4[a];
            axy = make_float2(pa.x, pa.y); azw = make_float2(pa.z, pa.w);
            if (i == 0) { bxy = axy; bzw = azw; }
            else { const float4 pb = src4[b]; bxy = make_float2(pb.x, pb.y); bzw = make_float2(pb.z, pb.w); }
        } else if (i < radiusX) {
            azw = src[2u * a + 1u];
            bzw = i == 0 ? azw : src[2u * b + 1u];
        } else if (i < radiusY) {
            axy = src[2u * a];
            bxy = i == 0 ? axy : src[2u * b];
        }
        if (i < radiusX) { // Lighting.glsl:113-117 umbra.zw (the vec4 sum's .xy receive + 0 * w)
            const float w = STATIC ? kEvsmBlurWeightsDev[(RX > 0 ? RX : 1) - 1][i] : K.w1[i];
            sx += 0.0f * w; sy += 0.0f * w;
            sz += (azw.x + bzw.x) * w; sw += (azw.y + bzw.y) * w;
        }
        if (i < radiusY) { // :119-123 penumbra.xy
            const float w = STATIC ? kEvsmBlurWeightsDev[(RY > 0 ? RY : 1) - 1][i] : K.w2[i];
            sx += (axy.x + bxy.x) * w; sy += (axy.y + bxy.y) * w;
            sz += 0.0f * w; sw += 0.0f * w;
        }
    }
    float4* o = reinterpret_cast<float4*>(dst) + (row + (uint32_t)x);
    *o = make_float4(sx, sy, sz, sw);
}

template <int RX, int RY>
static void launch_blur(hipStream_t st, bool vertical, const float* dSrc, float* dDst, int width, int height, int rx, int ry, int blurRadius, const BlurWeights& K)
{
    const dim3 grid((unsigned)((width + 255) / 256), (unsigned)height), gridV((unsigned)((width + 31) / 32), (unsigned)((height + 7) / 8));
    if (vertical)
        hipLaunchKernelGGL((k_evsm_blur<true, RX, RY>), gridV, dim3(256), 0, st, (const float2*)dSrc, (float2*)dDst, width, height, rx, ry, blurRadius, K);
    else
        hipLaunchKernelGGL((k_evsm_blur<false, RX, RY>), grid, dim3(256), 0, st, (const float2*)dSrc, (float2*)dDst, width, height, rx, ry, blurRadius, K);
}

extern "C" int sailor_hip_evsm_blur_pass(SailorHipContext* ctx, const float* dSrc, float* dDst, int32_t width, int32_t height, int32_t radiusUmbra,
                                         int32_t radiusPenumbra, int32_t vertical)
{
    if (!ctx || !dSrc || !dDst || width <= 0 || height <= 0 || radiusUmbra < 0 || radiusPenumbra < 0) return SAILOR_HIP_ERR_INVALID_ARGUMENT;
    SAILOR_TRY_HIP(ctx, hipSetDevice(ctx->device)); // a host thread may drive several contexts
    if (((uintptr_t)dSrc & 15) || ((uintptr_t)dDst & 15) || dSrc == dDst) return SAILOR_HIP_ERR_INVALID_ARGUMENT;
    const int stepCount = 12;
    const int mx = radiusUmbra > radiusPenumbra ? radiusUmbra : radiusPenumbra;
    const int blurRadius = mx < stepCount ? mx : stepCount;
    const int r1 = radiusUmbra < stepCount ? radiusUmbra : stepCount, r2 = radiusPenumbra < stepCount ? radiusPenumbra : stepCount;
    BlurWeights K;
    memset(&K, 0, sizeof K);
    if (r1 > 0) memcpy(K.w1, kEvsmBlurWeights[r1 - 1], sizeof K.w1);
    if (r2 > 0) memcpy(K.w2, kEvsmBlurWeights[r2 - 1], sizeof K.w2);
    if ((size_t)width * (size_t)height >= ((size_t)1 << 30)) return SAILOR_HIP_ERR_UNSUPPORTED; // 32-bit texel indices
    // the reference's own radii (ECS/LightingECS.h:68 ShadowCascadeBlur) get unrolled kernels
    const bool v = vertical != 0;
    if (radiusUmbra == 2 && radiusPenumbra == 5) launch_blur<2, 5>(ctx->stream, v, dSrc, dDst, width, height, 2, 5, blurRadius, K);
    else if (radiusUmbra == 1 && radiusPenumbra == 4) launch_blur<1, 4>(ctx->stream, v, dSrc, dDst, width, height, 1, 4, blurRadius, K);
    else if (radiusUmbra == 1 && radiusPenumbra == 3) launch_blur<1, 3>(ctx->stream, v, dSrc, dDst, width, height, 1, 3, blurRadius, K);
    else if (radiusUmbra == 1 && radiusPenumbra == 2) launch_blur<1, 2>(ctx->stream, v, dSrc, dDst, width, height, 1, 2, blurRadius, K);
    else launch_blur<0, 0>(ctx->stream, v, dSrc, dDst, width, height, radiusUmbra, radiusPenumbra, blurRadius, K);
    SAILOR_CHECK_LAUNCH(ctx, "k_evsm_blur");
    return SAILOR_HIP_OK;
}

extern "C" int sailor_hip_evsm_blur(SailorHipContext* ctx, float* dMap, float* dTemp, int32_t width, int32_t height, int32_t radiusUmbra, int32_t radiusPenumbra)
{
    const int st = sailor_hip_evsm_blur_pass(ctx, dMap, dTemp, width, height, radiusUmbra, radiusPenumbra, 0); // "Blur Horizontal" (ShadowPrepassNode.cpp:288-324)
    if (st != SAILOR_HIP_OK) return st;
    return sailor_hip_evsm_blur_pass(ctx, dTemp, dMap, width, height, radiusUmbra, radiusPenumbra, 1);          // "Blur Vertical" (:326-356)
}
