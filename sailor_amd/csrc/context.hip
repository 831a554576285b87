// Context, buffers and band helpers of the C-ABI (include/sailor_hip.h).
#include "common.h"
#include <vector>
#include <new>

extern "C" {

int sailor_hip_version(void) { return 1000 * 0 + 1; }

const char* sailor_hip_status_string(int status)
{
    switch (status) {
    case SAILOR_HIP_OK: return "ok";
    case SAILOR_HIP_ERR_INVALID_ARGUMENT: return "invalid argument";
    case SAILOR_HIP_ERR_NO_DEVICE: return "no usable HIP device";
    case SAILOR_HIP_ERR_OUT_OF_MEMORY: return "out of device memory";
    case SAILOR_HIP_ERR_LAUNCH: return "kernel launch / stream operation failed";
    case SAILOR_HIP_ERR_WORKSPACE_TOO_SMALL: return "workspace too small";
    case SAILOR_HIP_ERR_RCCL: return "RCCL error";
    case SAILOR_HIP_ERR_UNSUPPORTED: return "unsupported configuration";
    }
    return "unknown status";
}

int sailor_hip_device_count(int* outCount)
{
    if (!outCount) return SAILOR_HIP_ERR_INVALID_ARGUMENT;
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess) { *outCount = 0; return SAILOR_HIP_ERR_NO_DEVICE; }
    *outCount = n;
    return SAILOR_HIP_OK;
}

int sailor_hip_context_create(int deviceOrdinal, void* stream, uint32_t flags, SailorHipContext** outContext)
{
    if (!outContext) return SAILOR_HIP_ERR_INVALID_ARGUMENT;
    *outContext = nullptr;
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) return SAILOR_HIP_ERR_NO_DEVICE;
    if (deviceOrdinal < 0 || deviceOrdinal >= n) return SAILOR_HIP_ERR_NO_DEVICE;
    SailorHipContext* ctx = new (std::nothrow) SailorHipContext();
    if (!ctx) return SAILOR_HIP_ERR_OUT_OF_MEMORY;
    ctx->device = deviceOrdinal;
    if (hipSetDevice(deviceOrdinal) != hipSuccess) { delete ctx; return SAILOR_HIP_ERR_NO_DEVICE; }
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, deviceOrdinal) == hipSuccess) ctx->numCUs = prop.multiProcessorCount;
    if (!(flags & SAILOR_CTX_OWN_STREAM)) { ctx->stream = (hipStream_t)stream; ctx->ownsStream = false; }
    else {
        if (hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking) != hipSuccess) { delete ctx; return SAILOR_HIP_ERR_LAUNCH; }
        ctx->ownsStream = true;
    }
    *outContext = ctx;
    return SAILOR_HIP_OK;
}

int sailor_hip_context_destroy(SailorHipContext* ctx)
{
    if (!ctx) return SAILOR_HIP_ERR_INVALID_ARGUMENT;
    if (ctx->ownsStream && ctx->stream) { (void)hipStreamSynchronize(ctx->stream); (void)hipStreamDestroy(ctx->stream); }
    if (ctx->orderEvent) (void)hipEventDestroy(ctx->orderEvent);
    if (ctx->exchangeEvent) (void)hipEventDestroy(ctx->exchangeEvent);
    if (ctx->exchangeStatus) (void)hipHostFree(ctx->exchangeStatus);
    for (hipEvent_t e : ctx->timeStart) (void)hipEventDestroy(e);
    for (hipEvent_t e : ctx->timeStop) (void)hipEventDestroy(e);
    delete ctx;
    return SAILOR_HIP_OK;
}

int sailor_hip_context_synchronize(SailorHipContext* ctx)
{
    if (!ctx) return SAILOR_HIP_ERR_INVALID_ARGUMENT;
    SAILOR_TRY_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return SAILOR_HIP_OK;
}

int sailor_hip_context_wait_for(SailorHipContext* waiter, SailorHipContext* signaller)
{
    if (!waiter || !signaller || waiter->device != signaller->device) return SAILOR_HIP_ERR_INVALID_ARGUMENT;
    if (waiter->stream == signaller->stream) return SAILOR_HIP_OK; // one in-order stream: already ordered
    SAILOR_TRY_HIP(waiter, hipSetDevice(waiter->device));
    if (!signaller->orderEvent) SAILOR_TRY_HIP(signaller, hipEventCreateWithFlags(&signaller->orderEvent, hipEventDisableTiming));
    SAILOR_TRY_HIP(signaller, hipEventRecord(signaller->orderEvent, signaller->stream));
    SAILOR_TRY_HIP(waiter, hipStreamWaitEvent(waiter->stream, signaller->orderEvent, 0));
    return SAILOR_HIP_OK;
}

#define SAILOR_MAX_TIMING_SLOTS 4096
int sailor_hip_context_time_launches(SailorHipContext* ctx, int32_t firstSlot, int32_t count)
{
    if (!ctx || firstSlot < 0 || count < 0 || firstSlot + count > SAILOR_MAX_TIMING_SLOTS) return SAILOR_HIP_ERR_INVALID_ARGUMENT;
    SAILOR_TRY_HIP(ctx, hipSetDevice(ctx->device));
    // (a launch with events on its dispatch packet cannot be part of a hipGraph: refused while the stream is being captured)
    hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing(ctx->stream, &cap) == hipSuccess && cap != hipStreamCaptureStatusNone) {
        ctx->lastError = "sailor_hip_context_time_launches: the context's stream is being captured";
        return SAILOR_HIP_ERR_UNSUPPORTED;
    }
    while ((int)ctx->timeStart.size() < firstSlot + count) {
        // both events of a slot exist before either is kept: timeStart and timeStop always have the same length
        hipEvent_t a = nullptr, b = nullptr;
        SAILOR_TRY_HIP(ctx, hipEventCreate(&a));
        const hipError_t eb = hipEventCreate(&b);
        if (eb != hipSuccess) { (void)hipEventDestroy(a); return sailor_map_hip_error(ctx, eb, "hipEventCreate"); }
        ctx->timeStart.push_back(a);
        ctx->timeStop.push_back(b);
    }
    ctx->timeNext = firstSlot;
    ctx->timeEnd = firstSlot + count;
    return SAILOR_HIP_OK;
}

int sailor_hip_context_launch_log(SailorHipContext* ctx, uint64_t* outCount, const char** outNames, int32_t maxNames)
{
    if (!ctx || !outCount || maxNames < 0 || (maxNames > 0 && !outNames)) return SAILOR_HIP_ERR_INVALID_ARGUMENT;
    *outCount = ctx->launchCount;
    uint64_t n = ctx->launchCount < 16 ? ctx->launchCount : 16;
    if (n > (uint64_t)maxNames) n = (uint64_t)maxNames;
    for (uint64_t k = 0; k < n; k++) outNames[k] = ctx->launchNames[(ctx->launchCount - n + k) & 15]; // oldest first
    for (int32_t k = (int32_t)n; k < maxNames; k++) outNames[k] = nullptr;
    return SAILOR_HIP_OK;
}

int sailor_hip_context_timed_launch_ms(SailorHipContext* ctx, int32_t slot, float* outMs)
{
    if (!ctx || !outMs || slot < 0 || slot >= (int)ctx->timeStart.size() || slot >= (int)ctx->timeStop.size()) return SAILOR_HIP_ERR_INVALID_ARGUMENT;
    SAILOR_TRY_HIP(ctx, hipEventSynchronize(ctx->timeStop[slot]));
    SAILOR_TRY_HIP(ctx, hipEventElapsedTime(outMs, ctx->timeStart[slot], ctx->timeStop[slot]));
    return SAILOR_HIP_OK;
}

int sailor_hip_context_stream(SailorHipContext* ctx, void** outStream)
{
    if (!ctx || !outStream) return SAILOR_HIP_ERR_INVALID_ARGUMENT;
    *outStream = (void*)ctx->stream;
    return SAILOR_HIP_OK;
}

const char* sailor_hip_context_last_error(SailorHipContext* ctx)
{
    return ctx ? ctx->lastError.c_str() : "null context";
}

int sailor_hip_buffer_create(SailorHipContext* ctx, size_t bytes, void** outDevicePtr)
{
    if (!ctx || !outDevicePtr) return SAILOR_HIP_ERR_INVALID_ARGUMENT;
    *outDevicePtr = nullptr;
    SAILOR_TRY_HIP(ctx, hipSetDevice(ctx->device));
    SAILOR_TRY_HIP(ctx, hipMalloc(outDevicePtr, bytes ? bytes : 16));
    return SAILOR_HIP_OK;
}

int sailor_hip_buffer_free(SailorHipContext* ctx, void* devicePtr)
{
    if (!ctx) return SAILOR_HIP_ERR_INVALID_ARGUMENT;
    if (!devicePtr) return SAILOR_HIP_OK;
    SAILOR_TRY_HIP(ctx, hipStreamSynchronize(ctx->stream));
    SAILOR_TRY_HIP(ctx, hipFree(devicePtr));
    return SAILOR_HIP_OK;
}

int sailor_hip_buffer_upload(SailorHipContext* ctx, void* dstDevice, size_t dstOffset, const void* src, size_t bytes)
{
    if (!ctx || !dstDevice || (!src && bytes)) return SAILOR_HIP_ERR_INVALID_ARGUMENT;
    SAILOR_TRY_HIP(ctx, hipSetDevice(ctx->device)); // a host thread may drive several contexts
    if (!bytes) return SAILOR_HIP_OK;
    // Pageable-source hipMemcpyAsync stages the bytes before returning, i.e. the payload is captured at record time.
    SAILOR_TRY_HIP(ctx, hipMemcpyAsync((char*)dstDevice + dstOffset, src, bytes, hipMemcpyHostToDevice, ctx->stream));
    return SAILOR_HIP_OK;
}

int sailor_hip_buffer_download(SailorHipContext* ctx, void* dstHost, const void* srcDevice, size_t srcOffset, size_t bytes)
{
    if (!ctx || !srcDevice || (!dstHost && bytes)) return SAILOR_HIP_ERR_INVALID_ARGUMENT;
    if (!bytes) return SAILOR_HIP_OK;
    SAILOR_TRY_HIP(ctx, hipMemcpyAsync(dstHost, (const char*)srcDevice + srcOffset, bytes, hipMemcpyDeviceToHost, ctx->stream));
    SAILOR_TRY_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return SAILOR_HIP_OK;
}

__global__ void k_fill_u32(uint32_t* dst, uint32_t value, size_t count)
{
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (; i < count; i += stride) dst[i] = value;
}

// sailor_hip_copy_probe: the box's yardstick -- a float4-per-lane streaming copy, the access pattern MI355X_MICROARCH.md's 6.29 TB/s figure was taken with
__global__ __launch_bounds__(256) void k_copy_probe(const float4* __restrict__ src, float4* __restrict__ dst, size_t count4)
{
    const size_t stride = (size_t)gridDim.x * 256;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < count4; i += stride) dst[i] = src[i];
}

int sailor_hip_copy_probe(SailorHipContext* ctx, const void* dSrc, void* dDst, size_t bytes)
{
    if (!ctx || !dSrc || !dDst || (bytes & 15) || (((uintptr_t)dSrc | (uintptr_t)dDst) & 15)) return SAILOR_HIP_ERR_INVALID_ARGUMENT;
    SAILOR_TRY_HIP(ctx, hipSetDevice(ctx->device));
    if (!bytes) return SAILOR_HIP_OK;
    const size_t count4 = bytes / 16;
    size_t blocks = (count4 + 255) / 256;
    if (blocks > (size_t)ctx->numCUs * 32) blocks = (size_t)ctx->numCUs * 32; // 32 blocks per CU, grid-stride beyond
    sailor_launch(ctx, k_copy_probe, dim3((unsigned)blocks), dim3(256), (const float4*)dSrc, (float4*)dDst, count4);
    SAILOR_CHECK_LAUNCH(ctx, "k_copy_probe");
    return SAILOR_HIP_OK;
}

// sailor_hip_marker: an empty one-wave kernel.  A launch's dispatch-packet timestamps (sailor_hip_context_time_launches, rocprofv3's kernel trace) start when
// the packet is taken up, i.e. they include the wait for the PREDECESSOR's last blocks to drain -- 5-8 us behind k1_tile_cull, whose launch ends in a tail of
// single blocks.  With a marker between the two the drain lands on the marker's reading and the kernel behind it reads its own execution.
__global__ void k_marker() {}

int sailor_hip_marker(SailorHipContext* ctx)
{
    if (!ctx) return SAILOR_HIP_ERR_INVALID_ARGUMENT;
    SAILOR_TRY_HIP(ctx, hipSetDevice(ctx->device));
    sailor_launch(ctx, k_marker, dim3(1), dim3(64));
    SAILOR_CHECK_LAUNCH(ctx, "k_marker");
    return SAILOR_HIP_OK;
}

int sailor_hip_buffer_copy(SailorHipContext* ctx, void* dstDevice, size_t dstOffset, const void* srcDevice, size_t srcOffset, size_t bytes)
{
    if (!ctx || (bytes && (!dstDevice || !srcDevice))) return SAILOR_HIP_ERR_INVALID_ARGUMENT;
    SAILOR_TRY_HIP(ctx, hipSetDevice(ctx->device)); // a host thread may drive several contexts
    if (bytes == 0) return SAILOR_HIP_OK;
    SAILOR_TRY_HIP(ctx, hipMemcpyAsync((char*)dstDevice + dstOffset, (const char*)srcDevice + srcOffset, bytes, hipMemcpyDeviceToDevice, ctx->stream));
    return SAILOR_HIP_OK;
}

int sailor_hip_buffer_fill_u32(SailorHipContext* ctx, void* dstDevice, size_t dstOffset, uint32_t value, size_t count)
{
    if (!ctx || !dstDevice || (dstOffset & 3)) return SAILOR_HIP_ERR_INVALID_ARGUMENT;
    SAILOR_TRY_HIP(ctx, hipSetDevice(ctx->device)); // a host thread may drive several contexts
    if (!count) return SAILOR_HIP_OK;
    size_t blocks = (count + 255) / 256;
    if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL(k_fill_u32, dim3((unsigned)blocks), dim3(256), 0, ctx->stream, (uint32_t*)((char*)dstDevice + dstOffset), value, count);
    SAILOR_CHECK_LAUNCH(ctx, "k_fill_u32");
    return SAILOR_HIP_OK;
}

int sailor_hip_num_tiles(int32_t width, int32_t height, int32_t* outTilesX, int32_t* outTilesY)
{
    if (width <= 0 || height <= 0 || !outTilesX || !outTilesY) return SAILOR_HIP_ERR_INVALID_ARGUMENT;
    *outTilesX = (width - 1) / TILE + 1;  // FrameGraph/LightCullingNode.cpp:56
    *outTilesY = (height - 1) / TILE + 1; // :57
    return SAILOR_HIP_OK;
}

static void band_from_rows(int32_t height, int32_t r0, int32_t r1, SailorBand* b)
{
    b->tileRowBegin = r0;
    b->tileRowEnd = r1;
    // tile row t covers framebuffer rows H-1-16t-15 .. H-1-16t, clamped to [0, H)
    int32_t lo = height - TILE * r1;
    int32_t hi = height - TILE * r0;
    if (lo < 0) lo = 0;
    if (hi > height) hi = height;
    if (hi < lo) hi = lo;
    b->fbRowBegin = lo;
    b->fbRowCount = hi - lo;
}

int sailor_hip_band_whole_frame(int32_t width, int32_t height, SailorBand* outBand)
{
    int32_t tx, ty;
    if (!outBand || sailor_hip_num_tiles(width, height, &tx, &ty) != SAILOR_HIP_OK) return SAILOR_HIP_ERR_INVALID_ARGUMENT;
    band_from_rows(height, 0, ty, outBand);
    return SAILOR_HIP_OK;
}

int sailor_hip_band_from_tile_rows(int32_t width, int32_t height, int32_t tileRowBegin, int32_t tileRowEnd, SailorBand* outBand)
{
    int32_t tx, ty;
    if (!outBand || sailor_hip_num_tiles(width, height, &tx, &ty) != SAILOR_HIP_OK) return SAILOR_HIP_ERR_INVALID_ARGUMENT;
    if (tileRowBegin < 0 || tileRowEnd > ty || tileRowBegin > tileRowEnd) return SAILOR_HIP_ERR_INVALID_ARGUMENT;
    band_from_rows(height, tileRowBegin, tileRowEnd, outBand);
    return SAILOR_HIP_OK;
}

int sailor_hip_band_for_rank(int32_t width, int32_t height, int32_t rank, int32_t worldSize, SailorBand* outBand)
{
    int32_t tx, ty;
    if (!outBand || worldSize <= 0 || rank < 0 || rank >= worldSize) return SAILOR_HIP_ERR_INVALID_ARGUMENT;
    if (sailor_hip_num_tiles(width, height, &tx, &ty) != SAILOR_HIP_OK) return SAILOR_HIP_ERR_INVALID_ARGUMENT;
    const int32_t r0 = (int32_t)(((int64_t)rank * ty) / worldSize);
    const int32_t r1 = (int32_t)(((int64_t)(rank + 1) * ty) / worldSize);
    band_from_rows(height, r0, r1, outBand);
    return SAILOR_HIP_OK;
}

} // extern "C"
