// Internal helpers shared by the HIP translation units of libsailor_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>
#include <string>
#include <vector>

#define SAILOR_HIP_BUILD 1
#include "../../include/sailor_hip.h"

#define TILE SAILOR_LIGHTS_CULLING_TILE_SIZE
#define CAND SAILOR_LIGHTS_CANDIDATES_PER_TILE
#define KEEP SAILOR_LIGHTS_PER_TILE

struct SailorHipContext {
    int device = 0;
    hipStream_t stream = nullptr;
    bool ownsStream = false;
    int numCUs = 256;
    std::string lastError;
    // sailor_hip_context_time_launches: event pairs that ride on the dispatch packets of the next launches (slots [timeNext, timeEnd))
    std::vector<hipEvent_t> timeStart, timeStop;
    int timeNext = 0, timeEnd = 0;
    hipEvent_t orderEvent = nullptr; // sailor_hip_context_wait_for: "everything recorded on this context so far"
    // sailor_hip_context_launch_log: how many kernels the path's entry points have launched through this context, and the names of the last few
    uint64_t launchCount = 0, launchNoted = 0;
    const char* launchNames[16] = {};
    // the split frame's exchange (exchange.hip): the slot size of its second gather is kept from one exchange to the next, so that the call itself never reads
    // anything back.  exchangeStatus = three words of pinned host memory the stitch kernel writes: [0] the number of the exchange it belongs to, [1] the largest
    // band total that exchange gathered, [2] 1 if a band's segment was clipped to its slot.  sailor_hip_exchange_adapt reads them (behind exchangeEvent).
    uint32_t* exchangeStatus = nullptr;
    uint32_t exchangeSeq = 0;          // exchanges recorded through this context
    size_t exchangeLastSegCount = 0;   // the slot size the last recorded exchange used
    size_t exchangeSegHint = 0;        // slot size (uint32) of the next exchange's second gather; 0 = the worst case (tiles of the largest band x 128)
    hipEvent_t exchangeEvent = nullptr;
    bool exchangeEventValid = false;   // (an exchange recorded under stream capture leaves no event to wait on: adapt then waits for the stream)
};

static inline int sailor_map_hip_error(SailorHipContext* ctx, hipError_t e, const char* what)
{
    if (e == hipSuccess) return SAILOR_HIP_OK;
    if (ctx) {
        char buf[256];
        snprintf(buf, sizeof buf, "%s: %s", what, hipGetErrorString(e));
        ctx->lastError = buf;
    }
    switch (e) {
    case hipErrorOutOfMemory: return SAILOR_HIP_ERR_OUT_OF_MEMORY;
    case hipErrorNoDevice:
    case hipErrorInvalidDevice:
    case hipErrorInsufficientDriver:
    case hipErrorNotInitialized: return SAILOR_HIP_ERR_NO_DEVICE;
    case hipErrorInvalidValue: return SAILOR_HIP_ERR_INVALID_ARGUMENT;
    default: return SAILOR_HIP_ERR_LAUNCH;
    }
}

#define SAILOR_TRY_HIP(ctx, expr)                                                   \
    do {                                                                            \
        hipError_t _e = (expr);                                                     \
        if (_e != hipSuccess) return sailor_map_hip_error((ctx), _e, #expr);        \
    } while (0)

// (also names the launch in the context's log: the last sailor_launch, if nothing has named it yet)
#define SAILOR_CHECK_LAUNCH(ctx, name)                                              \
    do {                                                                            \
        if ((ctx)->launchNoted != (ctx)->launchCount) {                             \
            (ctx)->launchNames[((ctx)->launchCount - 1) & 15] = name;               \
            (ctx)->launchNoted = (ctx)->launchCount;                                \
        }                                                                           \
        hipError_t _e = hipGetLastError();                                          \
        if (_e != hipSuccess) return sailor_map_hip_error((ctx), _e, name);         \
    } while (0)

static inline size_t align_up(size_t x, size_t a) { return (x + a - 1) / a * a; }

// Every kernel of the path is launched through this: an ordinary launch on the context's stream, or -- while timing slots are armed
// (sailor_hip_context_time_launches) -- the same launch with a start / stop event pair attached to ITS dispatch packet (hipExtLaunchKernel): the
// command processor's own timestamps of that kernel, i.e. what rocprofv3's kernel trace reports, read live and without draining the stream
// around the kernel as an event recorded in front of and behind it does.
template <typename K, typename... Args>
static inline void sailor_launch_lds(SailorHipContext* ctx, K kernel, const dim3 grid, const dim3 block, const unsigned dynamicLdsBytes, Args... args)
{
    ctx->launchNames[ctx->launchCount & 15] = "?";
    ctx->launchCount++;
    if (ctx->timeNext < ctx->timeEnd) {
        const int i = ctx->timeNext++;
        hipExtLaunchKernelGGL(kernel, grid, block, dynamicLdsBytes, ctx->stream, ctx->timeStart[i], ctx->timeStop[i], 0, args...);
    } else hipLaunchKernelGGL(kernel, grid, block, dynamicLdsBytes, ctx->stream, args...);
}
template <typename K, typename... Args>
static inline void sailor_launch(SailorHipContext* ctx, K kernel, const dim3 grid, const dim3 block, Args... args)
{
    sailor_launch_lds(ctx, kernel, grid, block, 0u, args...);
}

// ---- canonical fp32 helpers: this library is compiled with -ffp-contract=off, so the
// expressions below evaluate exactly as written, one IEEE rounding per operation. -----------------------
struct Mat4 { float m[16]; }; // column-major: (col c, row r) = m[c*4 + r]

__device__ __forceinline__ float dot3f(float ax, float ay, float az, float bx, float by, float bz)
{
    return (ax * bx + ay * by) + az * bz;
}
// GLSL mat4 * vec4:  ((c0*x + c1*y) + c2*z) + c3*w
__device__ __forceinline__ float4 glsl_mul(const Mat4& M, float x, float y, float z, float w)
{
    float4 r;
    r.x = ((M.m[0] * x + M.m[4] * y) + M.m[8] * z) + M.m[12] * w;
    r.y = ((M.m[1] * x + M.m[5] * y) + M.m[9] * z) + M.m[13] * w;
    r.z = ((M.m[2] * x + M.m[6] * y) + M.m[10] * z) + M.m[14] * w;
    r.w = ((M.m[3] * x + M.m[7] * y) + M.m[11] * z) + M.m[15] * w;
    return r;
}
