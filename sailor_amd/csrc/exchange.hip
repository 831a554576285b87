// RCCL exchange step for split frames.  librccl is bound lazily so that the library loads (and the
// single-GPU path runs) on hosts without RCCL; the collective itself is recorded on the context stream.
#include "common.h"
#include <dlfcn.h>
#include <link.h>
#include <string.h>

typedef int (*nccl_allgather_fn)(const void*, void*, size_t, int /*ncclDataType_t*/, void* /*ncclComm_t*/, hipStream_t);

// The communicator the caller hands over was made by SOME copy of librccl; the collective must come from the same copy (a PyTorch process carries its own
// under torch/lib beside the one in /opt/rocm/lib: two libraries with separate state).  So: a librccl that is already mapped into the process wins,
// whatever its path; only a process without one gets the loader's default.
static int find_loaded_rccl(struct dl_phdr_info* info, size_t, void* out)
{
    if (info->dlpi_name && strstr(info->dlpi_name, "librccl.so")) { *(const char**)out = info->dlpi_name; return 1; }
    return 0;
}

static nccl_allgather_fn resolve_allgather()
{
    static nccl_allgather_fn fn = nullptr;
    static bool tried = false;
    if (!tried) {
        tried = true;
        void* h = nullptr;
        const char* loaded = nullptr;
        dl_iterate_phdr(find_loaded_rccl, &loaded);
        if (loaded) h = dlopen(loaded, RTLD_NOW | RTLD_NOLOAD | RTLD_GLOBAL);
        if (!h) h = dlopen("librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
        if (!h) h = dlopen("librccl.so", RTLD_NOW | RTLD_GLOBAL);
        if (h) fn = (nccl_allgather_fn)dlsym(h, "ncclAllGather");
    }
    return fn;
}

// the bands of an equal split as tile-row bounds (world + 1 entries)
static void equal_row_bounds(int32_t width, int32_t height, int32_t worldSize, int32_t* bounds)
{
    for (int r = 0; r < worldSize; r++) {
        SailorBand b;
        sailor_hip_band_for_rank(width, height, r, worldSize, &b);
        bounds[r] = b.tileRowBegin;
        bounds[r + 1] = b.tileRowEnd;
    }
}

static bool row_bounds_valid(const int32_t* bounds, int32_t worldSize, int32_t Ty)
{
    if (!bounds || bounds[0] != 0 || bounds[worldSize] != Ty) return false;
    for (int r = 0; r < worldSize; r++)
        if (bounds[r] > bounds[r + 1]) return false;
    return true;
}

extern "C" int sailor_hip_allgather_u32(SailorHipContext* ctx, void* comm, const uint32_t* dSend, uint32_t* dRecv, size_t countPerRank)
{
    if (!ctx || !comm || !dSend || !dRecv) return SAILOR_HIP_ERR_INVALID_ARGUMENT;
    if (countPerRank == 0) return SAILOR_HIP_OK;
    nccl_allgather_fn fn = resolve_allgather();
    if (!fn) { ctx->lastError = "librccl.so not found"; return SAILOR_HIP_ERR_RCCL; }
    const int ncclUint32 = 3; // rccl.h: ncclInt8 0, ncclUint8 1, ncclInt32 2, ncclUint32 3
    const int rc = fn(dSend, dRecv, countPerRank, ncclUint32, comm, ctx->stream);
    if (rc != 0) { ctx->lastError = "ncclAllGather failed"; return SAILOR_HIP_ERR_RCCL; }
    return SAILOR_HIP_OK;
}

// ---- K4 split across the ranks ("Entities: contiguous index ranges ... one all-gather of the visibility bitmask"): every rank has swept
// its slice (sailor_hip_ecs_range_for_rank / sailor_hip_ecs_sweep_range: whole 64-entity words, the same number per rank) into ITS part of dVisibility;
// one in-place ncclAllGather completes the bitmask on every rank.  dVisibility holds worldSize * wordsPerRank uint64.
extern "C" int sailor_hip_exchange_visibility(SailorHipContext* ctx, void* comm, int32_t rank, int32_t worldSize, uint32_t numEntities, uint64_t* dVisibility,
                                              size_t visibilityWords)
{
    if (!ctx || !comm || !dVisibility || worldSize <= 0 || rank < 0 || rank >= worldSize) return SAILOR_HIP_ERR_INVALID_ARGUMENT;
    uint32_t b = 0, e = 0, per = 0;
    if (sailor_hip_ecs_range_for_rank(numEntities, rank, worldSize, &b, &e, &per) != SAILOR_HIP_OK) return SAILOR_HIP_ERR_INVALID_ARGUMENT;
    if (per == 0) return SAILOR_HIP_OK;
    // (the gather writes worldSize * per words: more than ceil(n / 64) when the words do not divide evenly -- ADVICE r05)
    if (visibilityWords < (size_t)worldSize * per) { ctx->lastError = "sailor_hip_exchange_visibility: dVisibility holds fewer than worldSize * wordsPerRank words"; return SAILOR_HIP_ERR_INVALID_ARGUMENT; }
    // (in place: a rank's send buffer is its own slot of the receive buffer, which is what ncclAllGather's in-place form asks for)
    return sailor_hip_allgather_u32(ctx, comm, (const uint32_t*)(dVisibility + (size_t)rank * per), (uint32_t*)dVisibility, (size_t)per * 2);
}

// ---- the whole exchange of a split frame: band lists -> the reference's global lightsGrid / culledLights on every rank --------
// Three all-gathers (band total; index segments in slots of the largest band's total, known from the first; grid padded to the largest band)
// into the workspace, then ONE kernel per rank that turns the gathered slots into the canonical buffers: global offset of band r = sum of the
// totals of the bands before it (read from the gathered totals on the device).  One host read, of the gathered totals, sizes the second gather.
struct StitchArgs {
    const uint32_t* totals;   // [world]
    const uint32_t* segments; // [world][segCap]  (a band's culledLights[1 ..])
    const uint32_t* grids;    // [world][gridCap] ({offset, num} pairs, band-local offsets)
    uint32_t* outGrid; uint32_t* outCulled;
    uint32_t world, segCap, gridCap, outCapacity;
    uint32_t tiles[SAILOR_MAX_SPLIT + 1]; // prefix sums of the bands' tile counts
    uint32_t* status; uint32_t seq;       // the exchange's three status words in pinned host memory (SailorHipContext::exchangeStatus) or null, and its number
};

__global__ __launch_bounds__(256) void k_stitch_lists(const StitchArgs a)
{
    const uint32_t r = blockIdx.y; // band
    uint32_t base = 0, all = 0;
    for (uint32_t q = 0; q < a.world; q++) { const uint32_t t = a.totals[q]; if (q < r) base += t; all += t; }
    const uint32_t total = a.totals[r] < a.segCap ? a.totals[r] : a.segCap;
    const uint32_t stride = gridDim.x * 256u, first = blockIdx.x * 256u + threadIdx.x;
    for (uint32_t i = first; i < total; i += stride)
        if (1u + base + i < a.outCapacity) a.outCulled[1u + base + i] = a.segments[(size_t)r * a.segCap + i];
    const uint32_t t0 = a.tiles[r], nt = a.tiles[r + 1] - t0;
    for (uint32_t i = first; i < nt; i += stride) {
        const uint32_t off = a.grids[(size_t)r * a.gridCap + 2u * i], num = a.grids[(size_t)r * a.gridCap + 2u * i + 1u];
        a.outGrid[2u * (t0 + i)] = off + base; // Appendix A step 6: canonical global offsets
        a.outGrid[2u * (t0 + i) + 1u] = num;
    }
    if (r == 0 && first == 0) {
        a.outCulled[0] = all < a.outCapacity - 1u ? all : a.outCapacity - 1u; // what was written: segments are clipped to the capacity
        if (a.status) { // what sailor_hip_exchange_adapt sizes the NEXT exchange's slots from; a band whose total did not fit its slot arrived clipped
            uint32_t largest = 0;
            for (uint32_t q = 0; q < a.world; q++) largest = a.totals[q] > largest ? a.totals[q] : largest;
            a.status[1] = largest;
            a.status[2] = largest > a.segCap ? 1u : 0u;
            __threadfence_system();
            __hip_atomic_store(&a.status[0], a.seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
        }
    }
}

static size_t exchange_workspace_bytes(int32_t worldSize, size_t maxTiles)
{
    return align_up((size_t)worldSize * 4, 256) + align_up((size_t)worldSize * maxTiles * KEEP * 4, 256) + align_up((size_t)worldSize * maxTiles * 8, 256) +
           align_up(maxTiles * KEEP * 4, 256) + align_up(maxTiles * 8, 256);
}

extern "C" size_t sailor_hip_exchange_workspace_size(int32_t width, int32_t height, int32_t worldSize)
{
    if (width <= 0 || height <= 0 || worldSize < 1 || worldSize > SAILOR_MAX_SPLIT) return 0;
    int32_t Tx = 0, Ty = 0;
    sailor_hip_num_tiles(width, height, &Tx, &Ty);
    const size_t maxRows = ((size_t)Ty + worldSize - 1) / worldSize; // band_for_rank: floor((g+1) Ty / G) - floor(g Ty / G) <= ceil(Ty / G)
    return exchange_workspace_bytes(worldSize, maxRows * Tx);
}

extern "C" size_t sailor_hip_exchange_workspace_size_rows(int32_t width, int32_t height, int32_t worldSize, const int32_t* tileRowBounds)
{
    if (width <= 0 || height <= 0 || worldSize < 1 || worldSize > SAILOR_MAX_SPLIT) return 0;
    int32_t Tx = 0, Ty = 0;
    sailor_hip_num_tiles(width, height, &Tx, &Ty);
    if (!row_bounds_valid(tileRowBounds, worldSize, Ty)) return 0;
    size_t maxRows = 1;
    for (int r = 0; r < worldSize; r++) maxRows = (size_t)(tileRowBounds[r + 1] - tileRowBounds[r]) > maxRows ? (size_t)(tileRowBounds[r + 1] - tileRowBounds[r]) : maxRows;
    return exchange_workspace_bytes(worldSize, maxRows * Tx);
}

extern "C" int sailor_hip_stitch_light_lists(SailorHipContext* ctx, int32_t width, int32_t height, int32_t worldSize, const uint32_t* dTotals,
                                             const uint32_t* dSegments, size_t segmentCapacity, const uint32_t* dGrids, size_t gridCapacity,
                                             SailorLightsGrid* dGlobalGrid, uint32_t* dGlobalCulled, size_t globalCapacity)
{
    if (width <= 0 || height <= 0 || worldSize < 1 || worldSize > SAILOR_MAX_SPLIT) return SAILOR_HIP_ERR_INVALID_ARGUMENT;
    int32_t bounds[SAILOR_MAX_SPLIT + 1], Tx = 0, Ty = 0;
    equal_row_bounds(width, height, worldSize, bounds);
    sailor_hip_num_tiles(width, height, &Tx, &Ty);
    return sailor_hip_stitch_light_lists_rows(ctx, width, height, worldSize, bounds, dTotals, dSegments, segmentCapacity, dGrids, gridCapacity, dGlobalGrid,
                                              (size_t)Tx * Ty, dGlobalCulled, globalCapacity);
}

static int stitch_rows(SailorHipContext* ctx, int32_t width, int32_t height, int32_t worldSize, const int32_t* tileRowBounds, const uint32_t* dTotals,
                       const uint32_t* dSegments, size_t segmentCapacity, const uint32_t* dGrids, size_t gridCapacity, SailorLightsGrid* dGlobalGrid,
                       size_t globalGridTiles, uint32_t* dGlobalCulled, size_t globalCapacity, uint32_t* status, uint32_t seq);

extern "C" int sailor_hip_stitch_light_lists_rows(SailorHipContext* ctx, int32_t width, int32_t height, int32_t worldSize, const int32_t* tileRowBounds,
                                                  const uint32_t* dTotals, const uint32_t* dSegments, size_t segmentCapacity, const uint32_t* dGrids,
                                                  size_t gridCapacity, SailorLightsGrid* dGlobalGrid, size_t globalGridTiles, uint32_t* dGlobalCulled,
                                                  size_t globalCapacity)
{
    return stitch_rows(ctx, width, height, worldSize, tileRowBounds, dTotals, dSegments, segmentCapacity, dGrids, gridCapacity, dGlobalGrid, globalGridTiles, dGlobalCulled,
                       globalCapacity, nullptr, 0u);
}

static int stitch_rows(SailorHipContext* ctx, int32_t width, int32_t height, int32_t worldSize, const int32_t* tileRowBounds, const uint32_t* dTotals,
                       const uint32_t* dSegments, size_t segmentCapacity, const uint32_t* dGrids, size_t gridCapacity, SailorLightsGrid* dGlobalGrid,
                       size_t globalGridTiles, uint32_t* dGlobalCulled, size_t globalCapacity, uint32_t* status, uint32_t seq)
{
    if (!ctx || !dTotals || !dSegments || !dGrids || !dGlobalGrid || !dGlobalCulled || width <= 0 || height <= 0 || worldSize < 1 || worldSize > SAILOR_MAX_SPLIT)
        return SAILOR_HIP_ERR_INVALID_ARGUMENT;
    if (globalCapacity < 1 || segmentCapacity > 0xFFFFFFFFull || gridCapacity > 0xFFFFFFFFull) return SAILOR_HIP_ERR_INVALID_ARGUMENT;
    {
        int32_t Tx = 0, Ty = 0;
        sailor_hip_num_tiles(width, height, &Tx, &Ty);
        if (!row_bounds_valid(tileRowBounds, worldSize, Ty)) return SAILOR_HIP_ERR_INVALID_ARGUMENT;
        if (globalGridTiles < (size_t)Tx * Ty) return SAILOR_HIP_ERR_INVALID_ARGUMENT; // the stitch writes one entry per tile of the frame
    }
    StitchArgs a;
    a.totals = dTotals; a.segments = dSegments; a.grids = dGrids; a.outGrid = (uint32_t*)dGlobalGrid; a.outCulled = dGlobalCulled;
    a.world = (uint32_t)worldSize; a.segCap = (uint32_t)segmentCapacity; a.gridCap = (uint32_t)gridCapacity;
    a.outCapacity = (uint32_t)(globalCapacity > 0xFFFFFFFFull ? 0xFFFFFFFFull : globalCapacity);
    a.status = status; a.seq = seq;
    a.tiles[0] = 0;
    uint32_t maxTiles = 1;
    for (int r = 0; r < worldSize; r++) {
        int32_t Tx = 0, Ty = 0;
        sailor_hip_num_tiles(width, height, &Tx, &Ty);
        const uint32_t nt = (uint32_t)((tileRowBounds[r + 1] - tileRowBounds[r]) * Tx);
        if ((size_t)nt * 2 > gridCapacity) return SAILOR_HIP_ERR_INVALID_ARGUMENT;
        a.tiles[r + 1] = a.tiles[r] + nt;
        if (nt > maxTiles) maxTiles = nt;
    }
    SAILOR_TRY_HIP(ctx, hipSetDevice(ctx->device));
    unsigned bx = (maxTiles * (unsigned)KEEP / 8 + 255) / 256;
    if (bx > 256) bx = 256;
    if (bx < 1) bx = 1;
    hipLaunchKernelGGL(k_stitch_lists, dim3(bx, (unsigned)worldSize), dim3(256), 0, ctx->stream, a);
    SAILOR_CHECK_LAUNCH(ctx, "k_stitch_lists");
    return SAILOR_HIP_OK;
}

__global__ void k_pad_copy(const uint32_t* __restrict__ src, uint32_t* __restrict__ dst, const uint32_t* __restrict__ count, uint32_t fixedCount, uint32_t capacity)
{
    const uint32_t n = count ? (*count < capacity ? *count : capacity) : fixedCount;
    for (uint32_t i = blockIdx.x * 256u + threadIdx.x; i < n; i += gridDim.x * 256u) dst[i] = src[i];
}

extern "C" int sailor_hip_exchange_light_lists(SailorHipContext* ctx, void* comm, int32_t rank, int32_t worldSize, int32_t width, int32_t height,
                                               const SailorLightsGrid* dBandGrid, const uint32_t* dBandCulled, SailorLightsGrid* dGlobalGrid,
                                               uint32_t* dGlobalCulled, size_t globalCapacity, void* dWorkspace, size_t workspaceBytes)
{
    if (width <= 0 || height <= 0 || worldSize < 1 || worldSize > SAILOR_MAX_SPLIT) return SAILOR_HIP_ERR_INVALID_ARGUMENT;
    int32_t bounds[SAILOR_MAX_SPLIT + 1], Tx = 0, Ty = 0;
    equal_row_bounds(width, height, worldSize, bounds);
    sailor_hip_num_tiles(width, height, &Tx, &Ty);
    return sailor_hip_exchange_light_lists_rows(ctx, comm, rank, worldSize, width, height, bounds, dBandGrid, dBandCulled, dGlobalGrid, (size_t)Tx * Ty, dGlobalCulled,
                                                globalCapacity, dWorkspace, workspaceBytes);
}

extern "C" int sailor_hip_exchange_light_lists_rows(SailorHipContext* ctx, void* comm, int32_t rank, int32_t worldSize, int32_t width, int32_t height,
                                                    const int32_t* tileRowBounds, const SailorLightsGrid* dBandGrid, const uint32_t* dBandCulled,
                                                    SailorLightsGrid* dGlobalGrid, size_t globalGridTiles, uint32_t* dGlobalCulled, size_t globalCapacity,
                                                    void* dWorkspace, size_t workspaceBytes)
{
    if (!ctx || !comm || !dBandGrid || !dBandCulled || !dGlobalGrid || !dGlobalCulled || !dWorkspace || rank < 0 || rank >= worldSize) return SAILOR_HIP_ERR_INVALID_ARGUMENT;
    const size_t need = sailor_hip_exchange_workspace_size_rows(width, height, worldSize, tileRowBounds);
    if (need == 0 || ((uintptr_t)dWorkspace & 255)) return SAILOR_HIP_ERR_INVALID_ARGUMENT;
    if (workspaceBytes < need) return SAILOR_HIP_ERR_WORKSPACE_TOO_SMALL;
    int32_t Tx = 0, Ty = 0;
    sailor_hip_num_tiles(width, height, &Tx, &Ty);
    if (globalGridTiles < (size_t)Tx * Ty) return SAILOR_HIP_ERR_INVALID_ARGUMENT;
    size_t maxRows = 1;
    for (int r = 0; r < worldSize; r++) maxRows = (size_t)(tileRowBounds[r + 1] - tileRowBounds[r]) > maxRows ? (size_t)(tileRowBounds[r + 1] - tileRowBounds[r]) : maxRows;
    const size_t maxTiles = maxRows * Tx;
    const size_t segCap = maxTiles * KEEP, gridCap = maxTiles * 2;
    char* ws = (char*)dWorkspace;
    uint32_t* totals = (uint32_t*)ws; ws += align_up((size_t)worldSize * 4, 256);
    uint32_t* segments = (uint32_t*)ws; ws += align_up((size_t)worldSize * segCap * 4, 256);
    uint32_t* grids = (uint32_t*)ws; ws += align_up((size_t)worldSize * gridCap * 4, 256);
    uint32_t* sendSeg = (uint32_t*)ws; ws += align_up(segCap * 4, 256);
    uint32_t* sendGrid = (uint32_t*)ws;
    const uint32_t myTiles = (uint32_t)((tileRowBounds[rank + 1] - tileRowBounds[rank]) * Tx);
    SAILOR_TRY_HIP(ctx, hipSetDevice(ctx->device));
    // send slots (what lies behind the valid part is never read by the stitch)
    hipLaunchKernelGGL(k_pad_copy, dim3(256), dim3(256), 0, ctx->stream, dBandCulled + 1, sendSeg, dBandCulled, 0u, (uint32_t)segCap);
    hipLaunchKernelGGL(k_pad_copy, dim3(64), dim3(256), 0, ctx->stream, (const uint32_t*)dBandGrid, sendGrid, (const uint32_t*)nullptr, myTiles * 2u, (uint32_t)gridCap);
    SAILOR_CHECK_LAUNCH(ctx, "k_pad_copy");
    int rc = sailor_hip_allgather_u32(ctx, comm, dBandCulled, totals, 1);             // collective 1: band totals
    if (rc != SAILOR_HIP_OK) return rc;
    // Collective 2: the index segments travel in slots of ctx->exchangeSegHint words -- what sailor_hip_exchange_adapt made of an EARLIER exchange's gathered
    // totals (every rank read the same totals there, so every rank arrives at the same count) -- or, until that has been called, of the worst case
    // maxTiles * 128: 16.7 MB per rank at C3 for ~3 MB of lists.  Nothing is read back here and nothing waits (round 5 read the totals of collective 1 and
    // synchronised the stream to size collective 2): the call only records, and can be captured into a hipGraph.  A band whose total outgrew the slot
    // arrives clipped; the stitch kernel notes that in the context's status words and the next sailor_hip_exchange_adapt reports it and widens the slots.
    size_t segCount = ctx->exchangeSegHint ? ctx->exchangeSegHint : segCap;
    if (segCount > segCap) segCount = segCap;      // (a band's total cannot exceed its tiles' worst case)
    rc = sailor_hip_allgather_u32(ctx, comm, sendSeg, segments, segCount);             // collective 2: index segments, stride segCount
    if (rc == SAILOR_HIP_OK) rc = sailor_hip_allgather_u32(ctx, comm, sendGrid, grids, gridCap);    // (the grids, 8 bytes per tile)
    if (rc != SAILOR_HIP_OK) return rc;
    hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
    const bool capturing = hipStreamIsCapturing(ctx->stream, &cap) == hipSuccess && cap != hipStreamCaptureStatusNone;
    if (!ctx->exchangeStatus && !capturing) { // (three words of pinned host memory, once per context; a first exchange recorded under capture goes without)
        SAILOR_TRY_HIP(ctx, hipHostMalloc((void**)&ctx->exchangeStatus, 64, hipHostMallocDefault));
        ctx->exchangeStatus[0] = ctx->exchangeStatus[1] = ctx->exchangeStatus[2] = 0u;
    }
    ctx->exchangeSeq++;
    ctx->exchangeLastSegCount = segCount;
    rc = stitch_rows(ctx, width, height, worldSize, tileRowBounds, totals, segments, segCount, grids, gridCap, dGlobalGrid, globalGridTiles, dGlobalCulled, globalCapacity,
                     ctx->exchangeStatus, ctx->exchangeSeq);
    if (rc != SAILOR_HIP_OK) return rc;
    ctx->exchangeEventValid = false;
    if (!capturing) {
        if (!ctx->exchangeEvent) SAILOR_TRY_HIP(ctx, hipEventCreateWithFlags(&ctx->exchangeEvent, hipEventDisableTiming));
        SAILOR_TRY_HIP(ctx, hipEventRecord(ctx->exchangeEvent, ctx->stream));
        ctx->exchangeEventValid = true;
    }
    return SAILOR_HIP_OK;
}

// The one synchronising call of the exchange: waits for the LAST exchange recorded through this context (its event; the whole stream if that exchange was
// captured), reads the three status words its stitch kernel left, and sizes the next exchange's slots from the largest band total it gathered: + 25 %,
// whole 256-byte lines.  If that exchange was clipped (a band outgrew a slot sized from an earlier frame) the next one goes back to the worst case, the
// call says so through *outClipped and the context's error text, and the call after it adapts again.  Every rank of the communicator must call this at the
// same point of its call sequence: the totals -- hence the slot sizes -- are the same on every rank, and only then do the ranks' gathers agree.
extern "C" int sailor_hip_exchange_adapt(SailorHipContext* ctx, uint32_t* outLargestBandTotal, int32_t* outClipped, size_t* outSlotWords)
{
    if (!ctx) return SAILOR_HIP_ERR_INVALID_ARGUMENT;
    if (outLargestBandTotal) *outLargestBandTotal = 0;
    if (outClipped) *outClipped = 0;
    if (outSlotWords) *outSlotWords = ctx->exchangeSegHint;
    if (ctx->exchangeSeq == 0 || !ctx->exchangeStatus) return SAILOR_HIP_OK; // nothing exchanged yet: the worst case stays
    SAILOR_TRY_HIP(ctx, hipSetDevice(ctx->device));
    if (ctx->exchangeEventValid) SAILOR_TRY_HIP(ctx, hipEventSynchronize(ctx->exchangeEvent));
    else SAILOR_TRY_HIP(ctx, hipStreamSynchronize(ctx->stream));
    const uint32_t seq = __atomic_load_n(&ctx->exchangeStatus[0], __ATOMIC_ACQUIRE);
    if (seq != ctx->exchangeSeq) return SAILOR_HIP_OK; // (a captured exchange that has not been replayed yet: nothing new to learn)
    const uint32_t largest = ctx->exchangeStatus[1], clipped = ctx->exchangeStatus[2];
    if (outLargestBandTotal) *outLargestBandTotal = largest;
    if (outClipped) *outClipped = (int32_t)clipped;
    if (clipped) {
        char buf[160];
        snprintf(buf, sizeof buf, "exchange #%u: a band total of %u did not fit its slot of %zu indices; the next exchange uses the worst-case slots", seq, largest,
                 ctx->exchangeLastSegCount);
        ctx->lastError = buf;
        ctx->exchangeSegHint = 0;
    } else {
        size_t want = (size_t)largest + (size_t)largest / 4 + 1;
        ctx->exchangeSegHint = (want + 63) / 64 * 64;
    }
    if (outSlotWords) *outSlotWords = ctx->exchangeSegHint;
    return SAILOR_HIP_OK;
}

// an explicit slot size (0: the worst case) -- for a host that knows its lists, and for the tests
extern "C" int sailor_hip_exchange_set_slot_words(SailorHipContext* ctx, size_t slotWords)
{
    if (!ctx) return SAILOR_HIP_ERR_INVALID_ARGUMENT;
    ctx->exchangeSegHint = slotWords ? (slotWords + 63) / 64 * 64 : 0;
    return SAILOR_HIP_OK;
}
