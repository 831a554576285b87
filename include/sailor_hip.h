/*
 * sailor_hip.h -- C-ABI of the MI355X-native Forward+ lighting path for the Sailor engine.
 *
 * This is the drop-in boundary: the only thing `Runtime/GraphicsDriver/HIP` (the new RHI backend that sits
 * beside the reference's `Runtime/GraphicsDriver/Vulkan`) calls.  Style follows the reference's own exported
 * C surface, Lib/DllMain.cpp:9-160: `extern "C"`, cdecl, POD arguments, integer status returns.
 *
 * Conventions
 *   - every entry point returns `int`: 0 = SAILOR_HIP_OK, negative = error class (table below); nothing throws;
 *   - entry points RECORD work on the context's stream and return without synchronising (mirrors the reference's
 *     record-then-submit model: RHI/GraphicsDriver.h:310-314 Dispatch, :149 SubmitCommandList); only
 *     sailor_hip_context_synchronize / sailor_hip_buffer_download wait;
 *   - no allocation on the caller's behalf except through sailor_hip_buffer_create; kernels take raw device
 *     pointers, so memory owned by another allocator (a torch tensor, an engine heap) is accepted as is;
 *   - thread-compatible per context (one stream per context; the reference calls from the one Render thread,
 *     RHI/Renderer.cpp:264-303);
 *   - matrices are column-major float[16] exactly as glm::mat4 lies in memory.
 *
 * There is NO CPU fallback behind this header: if the HIP runtime or a gfx950 device is missing every
 * device entry point fails with SAILOR_HIP_ERR_NO_DEVICE.
 */
#ifndef SAILOR_HIP_H
#define SAILOR_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#if defined(SAILOR_HIP_BUILD)
#define SAILOR_HIP_API __attribute__((visibility("default")))
#else
#define SAILOR_HIP_API
#endif

/* ---- status codes ------------------------------------------------------------------------------------ */
#define SAILOR_HIP_OK 0
#define SAILOR_HIP_ERR_INVALID_ARGUMENT (-1)
#define SAILOR_HIP_ERR_NO_DEVICE (-2)       /* HIP runtime present but no usable device / wrong ordinal     */
#define SAILOR_HIP_ERR_OUT_OF_MEMORY (-3)
#define SAILOR_HIP_ERR_LAUNCH (-4)          /* kernel launch or stream operation failed                      */
#define SAILOR_HIP_ERR_WORKSPACE_TOO_SMALL (-5)
#define SAILOR_HIP_ERR_RCCL (-6)
#define SAILOR_HIP_ERR_UNSUPPORTED (-7)

/* ---- constants generated into Constants.glsl by AssetRegistry/Shader/ShaderCompiler.cpp:141-167 -------- */
#define SAILOR_LIGHTS_CULLING_TILE_SIZE 16     /* Constants.glsl:13 ; FrameGraph/LightCullingNode.h:16 */
#define SAILOR_LIGHTS_CANDIDATES_PER_TILE 196  /* Constants.glsl:14 */
#define SAILOR_LIGHTS_PER_TILE 128             /* Constants.glsl:15 ; FrameGraph/LightCullingNode.h:15 */
#define SAILOR_GPU_CULLING_GROUP_SIZE 256      /* Constants.glsl:18 ; RHI/Renderer.h:32 */
#define SAILOR_NUM_CSM_CASCADES 4              /* Constants.glsl:23 ; ECS/LightingECS.h:65 */
#define SAILOR_LIGHTS_MAX_NUM 65535            /* ECS/LightingECS.h:54 (the reference's SSBO capacity) */

/* ---- POD mirrors of the reference's GPU-visible structs ------------------------------------------------ */

/* RHI/Types.h:751-761 UboFrameData, filled at FrameGraph/RHIFrameGraph.cpp:60-67.  232 bytes. */
typedef struct SailorUboFrameData {
    float view[16];
    float projection[16];
    float invProjection[16];
    float cameraPosition[4];
    int32_t viewportSize[2];
    float cameraZNearZFar[2];
    float currentTime;
    float deltaTime;
} SailorUboFrameData;

/* FrameGraph/LightCullingNode.h:25-31 PushConstants == ComputeLightCulling.shader:12-18.  88 bytes. */
typedef struct SailorLightCullPushConstants {
    float invViewProjection[16]; /* never read by the shader; kept for layout */
    int32_t viewportSize[2];
    int32_t numTiles[2];
    int32_t lightsNum;
    int32_t _pad;
} SailorLightCullPushConstants;

/* ECS/LightingECS.h:71-81 LightShaderData == Lighting.glsl:4-15 LightData.  std430, 112 bytes. */
typedef struct SailorLightShaderData {
    uint32_t type;       /* Engine/Types.h:31-37: 0 Directional, 1 Point, 2 Spot, 3 Area */
    uint32_t shadowType; /* RHI/SceneView.h:13-18: 0 None, 1 PCF, 2 EVSM */
    uint32_t _pad0[2];
    float worldPosition[3]; float _pad1;
    float direction[3];     float _pad2;
    float intensity[3];     float _pad3;
    float attenuation[3];   float _pad4;
    float cutOff[2];        float _pad5[2]; /* cosines, ECS/LightingECS.cpp:171 */
    float bounds[3];        float _pad6;
} SailorLightShaderData;

/* Lighting.glsl:17-22 LightsGrid */
typedef struct SailorLightsGrid {
    uint32_t offset;
    uint32_t num;
} SailorLightsGrid;

/* FrameGraph/RenderSceneNode.h:16-33 PerInstanceData == ComputeMeshCulling.shader:20-26.  96 bytes. */
typedef struct SailorPerInstanceData {
    float model[16];
    float sphereBounds[4];
    uint32_t materialInstance;
    uint32_t isCulled;
    uint32_t _pad[2];
} SailorPerInstanceData;

/* RHI/Types.h DrawIndexedIndirectData == VkDrawIndexedIndirectCommand == ComputeMeshCulling.shader:28-35.  20 bytes. */
typedef struct SailorDrawIndexedIndirectData {
    uint32_t indexCount;
    uint32_t instanceCount;
    uint32_t firstIndex;
    int32_t vertexOffset;
    uint32_t firstInstance;
} SailorDrawIndexedIndirectData;

/* Math/Transform.h: {vec4 m_position; quat m_rotation (memory x,y,z,w); vec4 m_scale}.  48 bytes. */
typedef struct SailorTransform {
    float position[4];
    float rotation[4];
    float scale[4];
} SailorTransform;

/* Math/Bounds.h:110-113 AABB {vec3 m_min; vec3 m_max}.  24 bytes. */
typedef struct SailorAABB {
    float min[3];
    float max[3];
} SailorAABB;

/* Shadow-map texel formats (ECS/LightingECS.h:57-58 + a plain fp32 form for tests) */
#define SAILOR_SHADOWMAP_R16_SFLOAT 0
#define SAILOR_SHADOWMAP_R32G32B32A32_SFLOAT 1
#define SAILOR_SHADOWMAP_R32_SFLOAT 2

/* The CSM inputs of Standard.shader: binding 6 `lightsMatrices` and binding 8 `shadowMaps[cascade]`
 * (Standard.shader:223-233).  Maps are linear row-major images in device memory, row 0 first.
 * (An R16F cascade of up to 8192 x 8192 texels has the sixteen PCF taps of a pixel read as one 6 x 6-texel window -- Lighting.glsl:168-197's
 * disk is +-2 texels --, a larger one tap by tap: the same bits either way; tests/test_shade_gpu.py runs both against the oracle.) */
typedef struct SailorCsmDesc {
    float lightsMatrices[SAILOR_NUM_CSM_CASCADES][16];
    const void* maps[SAILOR_NUM_CSM_CASCADES]; /* device pointers; NULL = no map bound => shadow factor 1 */
    int32_t width[SAILOR_NUM_CSM_CASCADES];
    int32_t height[SAILOR_NUM_CSM_CASCADES];
    int32_t format[SAILOR_NUM_CSM_CASCADES];
} SailorCsmDesc;

/* The image-based-lighting inputs of Standard.shader's AmbientLighting (:343-372): binding 3 `g_irradianceCubemap`, binding 4
 * `g_brdfSampler`, binding 5 `g_envCubemap`, binding 9 `g_aoSampler` (:219-221,234).  All device pointers, fp32 texels (the
 * reference keeps RGBA16F / RG16F images).  Canonical sampler (identical in the oracle): cube face and (s, t) by the Vulkan
 * major-axis table (ties z over y over x), bilinear inside the face with clamp-to-edge, linear between the two nearest mip
 * levels with lod clamped to [0, levels - 1]; 2-D: bilinear, clamp-to-edge. */
typedef struct SailorIblDesc {
    const float* irradiance; /* 6 faces (+X,-X,+Y,-Y,+Z,-Z) x irrSize^2 float4, face-major, row t = 0 first */
    int32_t irrSize;
    const float* env;        /* mip chain, level-major: level l = 6 faces x max(1, envSize >> l)^2 float4 */
    int32_t envSize;
    int32_t envLevels;       /* textureQueryLevels(g_envCubemap) (:361) */
    const float* brdfLut;    /* lutH x lutW float2 (DFG1, DFG2), u = cosLo, v = roughness (ComputeBrdfLut.shader) */
    int32_t lutW;
    int32_t lutH;
    const float* ao;         /* band rows x width floats: the g_aoSampler target at this pixel (:386); NULL = 1.0 */
} SailorIblDesc;

/*
 * A horizontal band of the frame: tile rows [tileRowBegin, tileRowEnd) of the light grid.  Tile row t covers
 * framebuffer rows H-1-16t-15 .. H-1-16t, so the band's pixels are framebuffer rows
 * [fbRowBegin, fbRowBegin + fbRowCount).  Per-pixel device buffers handed to the band entry points (linear
 * depth, surface planes, radiance) hold exactly those rows, first row = fbRowBegin.  The whole frame is the
 * band {0, Ty, 0, H}; use sailor_hip_band_whole_frame / sailor_hip_band_for_rank to fill one.
 */
typedef struct SailorBand {
    int32_t tileRowBegin;
    int32_t tileRowEnd;
    int32_t fbRowBegin;
    int32_t fbRowCount;
} SailorBand;

typedef struct SailorHipContext SailorHipContext; /* opaque */

/* ---- library / context --------------------------------------------------------------------------------- */

SAILOR_HIP_API int sailor_hip_version(void);                 /* 1000*major + minor */
SAILOR_HIP_API const char* sailor_hip_status_string(int status);
SAILOR_HIP_API int sailor_hip_device_count(int* outCount);

/* Replaces: RHI/Renderer.cpp:60-63 (backend instantiation) + IGraphicsDriver::Initialize (RHI/GraphicsDriver.h:63).
 * flags = 0: record on `stream`, a hipStream_t the caller owns (NULL = the device's default stream, which is what
 * torch.cuda.current_stream() is unless the caller switched streams).  flags = SAILOR_CTX_OWN_STREAM: `stream` is
 * ignored and the context creates (and owns) a non-blocking stream -- the graphics-queue analogue of the reference. */
#define SAILOR_CTX_OWN_STREAM 1u
SAILOR_HIP_API int sailor_hip_context_create(int deviceOrdinal, void* stream, uint32_t flags, SailorHipContext** outContext);
SAILOR_HIP_API int sailor_hip_context_destroy(SailorHipContext* ctx);
/* Replaces: IGraphicsDriver::WaitIdle (RHI/GraphicsDriver.h:83) */
SAILOR_HIP_API int sailor_hip_context_synchronize(SailorHipContext* ctx);
SAILOR_HIP_API int sailor_hip_context_stream(SailorHipContext* ctx, void** outStream);
/* Text of the last HIP/RCCL error seen by this context (never NULL). */
SAILOR_HIP_API const char* sailor_hip_context_last_error(SailorHipContext* ctx);

/* Orders two contexts of one device: everything recorded on `waiter` after this call starts after everything recorded on `signaller` before it
 * (an event on the signaller's stream, a wait on the waiter's: the Vulkan backend's semaphore between two queues, VulkanGraphicsDriver.cpp
 * SubmitCommandList's wait / signal lists).  Used by the HIP backend to run the cull's compaction step beside the shade (SAILOR_CULL_DEFER_PACK). */
SAILOR_HIP_API int sailor_hip_context_wait_for(SailorHipContext* waiter, SailorHipContext* signaller);

/* Measurement aid (no reference counterpart): the next `count` kernel launches recorded through this context carry a HIP event pair on their own
 * dispatch packets (hipExtLaunchKernel's start / stop events) in slots [firstSlot, firstSlot + count), in launch order -- e.g. the four kernels of
 * one sailor_hip_light_cull, or the one of a sailor_hip_shade.  sailor_hip_context_timed_launch_ms waits for a slot's kernel and returns its
 * duration: the command processor's timestamps of THAT kernel -- the figure rocprofv3 --kernel-trace reports -- without draining the stream around
 * it as events recorded in front of and behind a launch do.  Eager launches only (not inside a hipGraph capture); at most 4 096 slots. */
SAILOR_HIP_API int sailor_hip_context_time_launches(SailorHipContext* ctx, int32_t firstSlot, int32_t count);
SAILOR_HIP_API int sailor_hip_context_timed_launch_ms(SailorHipContext* ctx, int32_t slot, float* outMs);
/* Measurement aid (no reference counterpart): *outCount = the number of kernels the path's entry points have launched through this context since it was
 * created (exactly the launches that take a timing slot above); outNames[0 .. n) = the names of the last n = min(maxNames, 16, *outCount) of them,
 * oldest first (static strings; further entries NULL).  A caller that wants the kernels of ONE call reads the count in front of and behind it -- which
 * kernels a cull chain consists of (the band selection's two kernels or not, the wide list builder or not, brute force) is the library's decision, not the caller's guess.
 * sailor_hip_context_time_launches is refused (SAILOR_HIP_ERR_UNSUPPORTED) while the context's stream is being captured into a hipGraph. */
SAILOR_HIP_API int sailor_hip_context_launch_log(SailorHipContext* ctx, uint64_t* outCount, const char** outNames, int32_t maxNames);
/* Measurement aid (no reference counterpart): one float4-per-lane streaming copy of `bytes` (a multiple of 16; both pointers 16-byte aligned) from dSrc to
 * dDst on the context's stream -- the yardstick of THIS box and process for the roofline figures (2 x bytes of HBM traffic per call; time it with
 * sailor_hip_context_time_launches like any kernel of the path).  Boxes of the pool differ by +-5 %; the guide's 6.29 TB/s is one of them. */
SAILOR_HIP_API int sailor_hip_copy_probe(SailorHipContext* ctx, const void* dSrc, void* dDst, size_t bytes);
/* Measurement aid (no reference counterpart): an empty one-wave kernel on the context's stream.  A kernel's dispatch-packet timestamps begin when its packet is
 * taken up and so include the wait for its predecessor's last blocks; a marker in front of a kernel takes that wait onto its own reading. */
SAILOR_HIP_API int sailor_hip_marker(SailorHipContext* ctx);

/* ---- buffers: IGraphicsDriver::CreateBuffer (RHI/GraphicsDriver.h:89-90), AddSsboToShaderBindings (:154),
 *      IGraphicsDriverCommands::UpdateShaderBinding / UpdateBuffer (:303-304) -------------------------------- */
SAILOR_HIP_API int sailor_hip_buffer_create(SailorHipContext* ctx, size_t bytes, void** outDevicePtr);
SAILOR_HIP_API int sailor_hip_buffer_free(SailorHipContext* ctx, void* devicePtr);
/* async host->device copy on the context stream; `src` bytes are staged at record time (the reference copies
 * push-constant / update payloads at record time too: VulkanCommandBuffer.cpp:679-685) */
SAILOR_HIP_API int sailor_hip_buffer_upload(SailorHipContext* ctx, void* dstDevice, size_t dstOffset, const void* src, size_t bytes);
/* synchronous device->host copy (waits for the stream) */
SAILOR_HIP_API int sailor_hip_buffer_download(SailorHipContext* ctx, void* dstHost, const void* srcDevice, size_t srcOffset, size_t bytes);
/* device-to-device copy on the context's stream (the BlitImage of equally sized images, e.g. EnvironmentNode.cpp:200-203) */
SAILOR_HIP_API int sailor_hip_buffer_copy(SailorHipContext* ctx, void* dstDevice, size_t dstOffset, const void* srcDevice, size_t srcOffset, size_t bytes);
SAILOR_HIP_API int sailor_hip_buffer_fill_u32(SailorHipContext* ctx, void* dstDevice, size_t dstOffset, uint32_t value, size_t count);

/* ---- bands ---------------------------------------------------------------------------------------------- */
SAILOR_HIP_API int sailor_hip_num_tiles(int32_t width, int32_t height, int32_t* outTilesX, int32_t* outTilesY); /* LightCullingNode.cpp:56-57 */
SAILOR_HIP_API int sailor_hip_band_whole_frame(int32_t width, int32_t height, SailorBand* outBand);
/* an arbitrary contiguous run of tile rows [tileRowBegin, tileRowEnd) -- for cost-balanced partitions */
SAILOR_HIP_API int sailor_hip_band_from_tile_rows(int32_t width, int32_t height, int32_t tileRowBegin, int32_t tileRowEnd, SailorBand* outBand);
/* contiguous tile-row bands: rank g of G gets rows [floor(g*Ty/G), floor((g+1)*Ty/G)) */
SAILOR_HIP_API int sailor_hip_band_for_rank(int32_t width, int32_t height, int32_t rank, int32_t worldSize, SailorBand* outBand);
/* 1 if `band` is a run of whole tile rows of a width x height frame with the matching framebuffer rows (what the three helpers above
 * produce), else 0.  Every band entry point refuses anything else. */
SAILOR_HIP_API int sailor_hip_band_is_valid(int32_t width, int32_t height, const SailorBand* band);

/* ---- K0 + K1: tile light cull ---------------------------------------------------------------------------
 * Replaces: the Dispatch recorded by LightCullingNode::Process (FrameGraph/LightCullingNode.cpp:74-77) and the
 * whole of Content/Shaders/ComputeLightCulling.shader, under the canonical sequential semantics of SURVEY.md
 * Appendix A (ascending-light-index candidates, first 196, nearest-128 selection, prefix-sum offsets).
 *
 *   frame, pc      : host structs, copied at record time
 *   dLights        : device, pc->lightsNum x SailorLightShaderData (binding set 0 / binding 0 `light`)
 *   dLinearDepth   : device, R32F linear depth rows of `band` (binding set 1 / binding 2 `sceneDepth`)
 *   dLightsGrid    : device out, one SailorLightsGrid per tile OF THE BAND (band-local tile index
 *                    (ty - tileRowBegin)*Tx + tx); offset is band-local: 1 + sum of num over earlier band tiles
 *   dCulledLights  : device out, uint32: [0] = sum of num over the band, [offset+i] = i-th light of the tile;
 *                    capacity culledCapacity uints: 1 + bandTiles*128 holds every possible result.  The reference allocates
 *                    bandTiles*128 (one short, LightCullingNode.cpp:64): that size is accepted, and a list that does not
 *                    fit is cut -- lightsGrid[tile].num and [0] then say what was written, so a consumer never reads past
 *                    the buffer; anything smaller is SAILOR_HIP_ERR_INVALID_ARGUMENT
 *   dWorkspace     : device scratch of at least sailor_hip_light_cull_workspace_size(...) bytes
 *   flags          : SAILOR_CULL_* bits
 * For the whole-frame band the outputs ARE the reference's `lightsGrid` / `culledLights` buffers.
 */
#define SAILOR_CULL_DEFAULT 0u
#define SAILOR_CULL_BRUTE_FORCE 1u /* skip the conservative macro-tile pre-filter (same results, for validation) */
#define SAILOR_CULL_RAW_DEPTH 2u   /* dLinearDepth holds the RAW reversed-Z depth attachment: the depth pass linearises it on
                                      the fly (sailor_hip_linearize_depth's arithmetic, same bits) -- the LinearizeDepth node's
                                      full-screen pass and its 8 bytes per pixel disappear.  x -> zNear / x is monotone, so
                                      the tile's min / max are taken on the raw bits and only two values per tile are divided */
#define SAILOR_CULL_INTERVAL_MASKS 4u /* build the pre-filter masks from per-light band intervals (the default above 262 144 lights) also for small
                                       * light sets with <= 256 bands: same lists, for validation */
#define SAILOR_CULL_DEFER_PACK 8u     /* stop after the per-tile lists: dLightsGrid / dCulledLights are NOT written by this call.  The lists are complete in the
                                       * workspace (sailor_hip_light_cull_tile_lists; sailor_hip_shade_tile_lists shades from them), and
                                       * sailor_hip_light_cull_pack -- on any context / stream ordered after this call, e.g. a second stream beside the shade --
                                       * produces the two canonical buffers from them, bit for bit what the undeferred call writes */

#define SAILOR_CULL_PREPARE_LIGHTS 16u /* sailor_hip_light_cull_prepared only: EVERY light is dirty this frame (LightingECS::Tick re-uploaded the whole set,
                                        * ECS/LightingECS.cpp:152-191).  The cull's per-light pass reads the 112-byte records in dLights and WRITES the prepared
                                        * views of all pc->lightsNum lights into dPreparedLights on the way -- sailor_hip_prepare_lights(0, lightsNum) folded
                                        * into the cull: one pass over the records instead of two, one launch less; the same bits in the views, the same lists */
#define SAILOR_CULL_BAND_SELECT 32u    /* a band of a split frame: select the lights that can reach the band's rows first (k0_band_count + k0_band_scatter: an ordered compaction on the band's
                                        * top / bottom planes) and run the chain on those -- the default from 131 072 lights on; this flag forces it for smaller sets
                                        * (same lists bit for bit; validation) */
#define SAILOR_CULL_NO_BAND_SELECT 64u /* ... never (same lists; A / B) */
#define SAILOR_CULL_PREPARE_SELECTED 128u /* with SAILOR_CULL_PREPARE_LIGHTS on a band whose lights are selected (above): the staged SHADE records are derived only for
                                        * the selected lights -- all this band's shade of THIS frame can read -- instead of for all of them (the 20-byte cull views
                                        * still for all).  For hosts that re-prepare every frame (every light dirty every frame): a record of a light outside the
                                        * selection keeps whatever an earlier call left there, so a later frame that culls WITHOUT re-preparing must not follow.
                                        * Ignored where no selection runs */
SAILOR_HIP_API size_t sailor_hip_light_cull_workspace_size(int32_t width, int32_t height, int32_t lightsCapacity, const SailorBand* band);
SAILOR_HIP_API int sailor_hip_light_cull(SailorHipContext* ctx,
                                         const SailorUboFrameData* frame, const SailorLightCullPushConstants* pc,
                                         const SailorLightShaderData* dLights, const float* dLinearDepth,
                                         SailorLightsGrid* dLightsGrid, uint32_t* dCulledLights, size_t culledCapacity,
                                         void* dWorkspace, size_t workspaceBytes,
                                         const SailorBand* band, uint32_t flags);

/* The cull's own per-tile form of the lists (round 4).  Every tile of the band leaves its list in a fixed 128-entry slot of the workspace --
 * tileLists[bandTile * 128 + i] = i-th light of the tile, the same entries in the same order as culledLights[offset + i] -- and its length (<= 128) in
 * tileNum[bandTile].  A consumer that only needs "the lights of tile t" (the shade: Standard.shader:422-436) reads them there and does not depend on the
 * compaction into the reference's layout, which then leaves the frame's critical path: SAILOR_CULL_DEFER_PACK + sailor_hip_light_cull_pack on a
 * second stream.  The pointers depend on (width, height, band) alone (lightsCapacity: any light count the workspace can hold) and stay valid until the
 * next cull on the same workspace.  The band shade's hint below (the lengths as bytes) is written by k1_tile_cull as well: it is there with a deferred
 * pack too, so a band's shade does not wait for the compaction either. */
SAILOR_HIP_API int sailor_hip_light_cull_tile_lists(int32_t width, int32_t height, int32_t lightsCapacity, const SailorBand* band, const void* dWorkspace,
                                                    const uint32_t** outTileNum, const uint32_t** outTileLists);
/* Replaces: nothing of its own -- the second half of the Dispatch at LightCullingNode.cpp:74-77 (Appendix A step 6: offsets = prefix sum of the list
 * lengths in tile order, ComputeLightCulling.shader:227-238 without its atomic allocation order) when sailor_hip_light_cull ran with
 * SAILOR_CULL_DEFER_PACK.  Arguments as there. */
SAILOR_HIP_API int sailor_hip_light_cull_pack(SailorHipContext* ctx, int32_t width, int32_t height, int32_t lightsCapacity, const SailorBand* band,
                                              const void* dWorkspace, SailorLightsGrid* dLightsGrid, uint32_t* dCulledLights, size_t culledCapacity);

/* Shading hint of a band.  A band of a split frame is a round or two of blocks, so its longest tile is its duration, and a tile in the middle of a light
 * cluster keeps one block busy ~100x longer than an average one.  This returns the band's per-tile list lengths as BYTES (T bytes, tile order, <= 128 each;
 * every sailor_hip_light_cull writes them beside tileNum -- nothing extra, no atomics: round 4's form, a list of the long tiles appended through two
 * device-scope counters, cost the cull up to 12.7 us on half a 4K frame).  Handing the pointer to sailor_hip_shade_ex (band smaller than the frame, no
 * ambient term; with or without shadow maps) switches the BAND FORM of the launch on: the tiles with at least `splitMin` lights -- 64 on a band of up to
 * 12 000 tiles, 96 on a larger one; SAILOR_SPLIT_MIN=<1..128> overrides for A / B runs -- go to "split" blocks -- one per (tile, 8x8
 * quadrant), four waves sharing the quadrant's list -- at the front of the grid, which find them in these bytes.  Lists and every tile below the threshold keep
 * their bits; a split tile's radiance differs from the one-block form by the order of four partial sums per pixel (within the shade tolerance).  With the
 * ambient term the pointer is ignored.  NULL for the whole-frame band (on the whole frame the split measured no gain), for a band of more than 12 000
 * tiles when lightsCapacity >= 131 072 (short lists, no long tiles to split: the whole frame's launch form pipelines better there) and on bad arguments.  Valid until
 * the next sailor_hip_light_cull on the same workspace.  `lightsCapacity` only has to be a light count the workspace can hold: the bytes' place in the
 * workspace depends on (width, height, band) alone.  (The return type is kept from round 2's word array; the data are bytes.) */
SAILOR_HIP_API const uint32_t* sailor_hip_light_cull_tile_order(int32_t width, int32_t height, int32_t lightsCapacity, const SailorBand* band,
                                                                const void* dWorkspace);

/* ---- LinearizeDepth: the pass immediately before K1 ---------------------------------------
 * Replaces: the full-screen draw of LinearizeDepthNode::Process (FrameGraph/LinearizeDepthNode.cpp:22-109) with the
 * fragment shader Content/Shaders/LinearizeDepth.shader:61-73 under its REVERSE_Z_INF_FAR_PLANE define (:6):
 *   linearDepth = -frame.cameraZNearZFar.x / depth;  outColor = vec4(-linearDepth)
 * i.e. out = zNear / raw with one IEEE division (both negations are exact).  The quad's flipped texcoord (:49) and the
 * driver's flipped viewport (GraphicsDriver/Vulkan/VulkanDevice.cpp:681) cancel: output row r reads depth row r.
 *   dRawDepth / dLinearDepth : device, float32, `rows` x `width`, row-major (any contiguous run of framebuffer rows)
 * raw = 0 (nothing drawn, far plane at infinity) gives +inf, as in the reference.
 */
SAILOR_HIP_API int sailor_hip_linearize_depth(SailorHipContext* ctx, const SailorUboFrameData* frame,
                                              const float* dRawDepth, float* dLinearDepth, int32_t width, int32_t rows);

/* Tuning diagnostics (synchronises): out8 = {numBands, mask bits set, numGroups, sum of group list lengths, overflowed
 * groups, longest group list, 64-bit words per band, bits set in the column masks} for the intermediate pre-filter state
 * left in dWorkspace by the last sailor_hip_light_cull of the same geometry. */
/* ---- prepared lights: the per-light half of the path, run where the `light` SSBO is WRITTEN instead of where it is read ------------------
 * The reference re-derives everything per light inside its per-tile / per-fragment loops.  Lights change when LightingECS::Tick uploads a dirty run
 * (ECS/LightingECS.cpp:182-191 -> IGraphicsDriverCommands::UpdateShaderBinding, RHI/GraphicsDriver.h:303), not every frame, so the HIP backend derives
 * two views of the records right behind that copy ("pre-split SoA"), for the lights [firstLight, firstLight + count) it carried:
 *   - the cull's view: (worldPosition, bounds.x) as one float4 and the type, dense arrays -- 20 bytes per light for k01_prepare to stream each frame
 *     instead of 112;
 *   - the shade's staged record, 80 bytes: normalised spot axis, reach threshold, 1 / bounds.x, the "every parameter is finite" bit ... -- what every
 *     shade block otherwise derives again for every slot of its tile's list.
 * dPrepared is caller-owned device memory of sailor_hip_prepared_lights_size(lightsCapacity) bytes, 16-byte aligned; lightsCapacity fixes its layout
 * and must be the same in every call that names the buffer.  sailor_hip_light_cull_prepared / sailor_hip_shade_prepared take it beside dLights (NULL =
 * the entry points above: same lists, same radiance, bit for bit -- the kernels run the same instructions on the same inputs, only elsewhere).
 * The caller keeps it in step with dLights: a record changed without a prepare of its slot is a stale light, as a missed upload would be. */
SAILOR_HIP_API size_t sailor_hip_prepared_lights_size(int32_t lightsCapacity);
SAILOR_HIP_API int sailor_hip_prepare_lights(SailorHipContext* ctx, const SailorLightShaderData* dLights, int32_t firstLight, int32_t count,
                                             int32_t lightsCapacity, void* dPrepared, size_t preparedBytes);
/* the three arrays inside dPrepared (float4 posRadius[capacity]; uint32 type[capacity]; float4 staged[capacity][5]); any out pointer may be NULL */
SAILOR_HIP_API int sailor_hip_prepared_lights_views(int32_t lightsCapacity, const void* dPrepared, const void** outPosRadius, const void** outType,
                                                    const void** outStaged);
SAILOR_HIP_API int sailor_hip_light_cull_prepared(SailorHipContext* ctx,
                                                  const SailorUboFrameData* frame, const SailorLightCullPushConstants* pc,
                                                  const SailorLightShaderData* dLights, const float* dLinearDepth,
                                                  SailorLightsGrid* dLightsGrid, uint32_t* dCulledLights, size_t culledCapacity,
                                                  void* dWorkspace, size_t workspaceBytes,
                                                  const SailorBand* band, uint32_t flags,
                                                  const void* dPreparedLights /* or NULL */, int32_t preparedCapacity);
SAILOR_HIP_API int sailor_hip_light_cull_diagnostics(SailorHipContext* ctx, int32_t width, int32_t height, int32_t lightsNum, const SailorBand* band,
                                                     const void* dWorkspace, uint64_t* out8);
/* The band-local light selection (SAILOR_CULL_BAND_SELECT and its default above) as the last cull with this geometry and pc->lightsNum == lightsNum left
 * it in dWorkspace: *outSelectedCount -> one device uint32, the number M of lights that can reach the band; *outLightMap -> M device uint32, the
 * selected lights' indices in ascending order.  Meaningful only if that cull ran the selection (sailor_hip_context_launch_log names a call's kernels:
 * "k0_band_count", "k0_band_scatter" are the first two of the chain then).  Tests and diagnostics; either out pointer may be NULL. */
SAILOR_HIP_API int sailor_hip_light_cull_band_selection(int32_t width, int32_t height, int32_t lightsNum, const SailorBand* band, const void* dWorkspace,
                                                        const uint32_t** outSelectedCount, const uint32_t** outLightMap);

/* Multi-GPU stitch helpers.  After an all-gather of the per-band totals, rebase a band's grid
 * to the canonical global offsets: offset += globalBase (globalBase = sum of num over all earlier bands). */
SAILOR_HIP_API int sailor_hip_light_grid_rebase(SailorHipContext* ctx, SailorLightsGrid* dLightsGrid, int32_t numTiles, uint32_t globalBase);

/* ---- K2 + K3: PBR shade over per-tile light lists, with CSM sampling --------------------------------------
 * Replaces: the fragment work of Content/Shaders/Standard.shader main() (:377-439) + CalculateLighting (:259-341)
 * + Lighting.glsl:39-76,168-284 for the draws recorded by RenderSceneNode::Process (FrameGraph/RenderSceneNode.cpp:109),
 * executed as compute over a surface buffer instead of rasterised fragments (ambient/IBL term == 0).
 *
 *   dSurface       : device, 3 planes of float4 per pixel, plane-major, each plane = band rows x width:
 *                    P0 = (worldPos.xyz, albedo.a)  P1 = (normal.xyz, roughness)  P2 = (albedo.rgb, metallic)
 *   surfacePlaneStride : distance between planes in float4 elements (>= band rows * width)
 *   dLightsGrid / dCulledLights : the lists of the SAME band as produced by sailor_hip_light_cull (band-local)
 *   csm            : host struct or NULL (no directional shadowing: factor 1)
 *   dRadiance      : device out, float4 per pixel of the band (rgb = sum over lights, a = albedo.a)
 */
SAILOR_HIP_API int sailor_hip_shade(SailorHipContext* ctx, const SailorUboFrameData* frame,
                                    const float* dSurface, size_t surfacePlaneStride,
                                    const SailorLightShaderData* dLights, int32_t lightsNum,
                                    const SailorLightsGrid* dLightsGrid, const uint32_t* dCulledLights,
                                    const SailorCsmDesc* csm, float* dRadiance,
                                    const SailorBand* band);

/* As sailor_hip_shade, plus the ambient term: outColor.rgb = AmbientLighting(...) + sum over lights
 * (Standard.shader:425).  ibl == NULL and dTileOrder == NULL is sailor_hip_shade. */
SAILOR_HIP_API int sailor_hip_shade_ex(SailorHipContext* ctx, const SailorUboFrameData* frame,
                                       const float* dSurface, size_t surfacePlaneStride,
                                       const SailorLightShaderData* dLights, int32_t lightsNum,
                                       const SailorLightsGrid* dLightsGrid, const uint32_t* dCulledLights,
                                       const SailorCsmDesc* csm, const SailorIblDesc* ibl, float* dRadiance,
                                       const SailorBand* band, const uint32_t* dTileOrder /* sailor_hip_light_cull_tile_order or NULL */);

/* As sailor_hip_shade_ex with the lights' staged records (sailor_hip_prepare_lights, above); dPreparedLights == NULL is sailor_hip_shade_ex.
 * With it dLights is not read and may be NULL. */
SAILOR_HIP_API int sailor_hip_shade_prepared(SailorHipContext* ctx, const SailorUboFrameData* frame,
                                             const float* dSurface, size_t surfacePlaneStride,
                                             const SailorLightShaderData* dLights, int32_t lightsNum,
                                             const SailorLightsGrid* dLightsGrid, const uint32_t* dCulledLights,
                                             const SailorCsmDesc* csm, const SailorIblDesc* ibl, float* dRadiance,
                                             const SailorBand* band, const uint32_t* dTileOrder,
                                             const void* dPreparedLights /* or NULL */, int32_t preparedCapacity);

/* As sailor_hip_shade_prepared, reading the lists where the cull left them (sailor_hip_light_cull_tile_lists: dTileNum[t] entries at
 * dTileLists[128 t ..] for band tile t) instead of through lightsGrid / culledLights: the same entries in the same order, hence the same radiance bit
 * for bit -- and no dependence on the compaction step (SAILOR_CULL_DEFER_PACK).  This is the form the HIP backend records for RenderSceneNode's
 * draws when the lists come from its own LightCulling node (Standard.shader:422-436 is the loop it replaces either way). */
SAILOR_HIP_API int sailor_hip_shade_tile_lists(SailorHipContext* ctx, const SailorUboFrameData* frame,
                                               const float* dSurface, size_t surfacePlaneStride,
                                               const SailorLightShaderData* dLights, int32_t lightsNum,
                                               const uint32_t* dTileNum, const uint32_t* dTileLists,
                                               const SailorCsmDesc* csm, const SailorIblDesc* ibl, float* dRadiance,
                                               const SailorBand* band, const uint32_t* dTileOrder,
                                               const void* dPreparedLights /* or NULL */, int32_t preparedCapacity);

/* Self-check of the shade kernels' short forms of exact arithmetic (no reference counterpart: the shader leaves sqrt and 1 / x to the driver).
 * K2 evaluates normalize() as v * (1 / sqrt(dot(v, v))) with a correctly rounded square root and reciprocal, but not through the compiler's
 * 21- and 12-instruction expansions: sqrt as one Newton step on v_rsq_f32 for x in [2^-96, inf), 1 / s as one Newton step on v_rcp_f32 for
 * |s| in [2^-126, 2^126] (sailor_amd/csrc/shade_body.h: sqrt_exact, rcp_of_sqrt).  This runs both against sqrtf and 1.0f / s over EVERY float of
 * those ranges on the device the context is bound to (about a second) and writes the number of inputs whose bits differ:
 *   mismatches[0] = square root, mismatches[1] = reciprocal.  The quotient from a staged reciprocal (div_stored: Markstein's correction step,
 * exact by theorem when nothing over- or underflows) has 2^64 operand pairs; 2^30 pseudo-random ones in the ranges its callers guarantee are
 * compared with the IEEE division: mismatches[2].  All three must be 0; dScratch: device, 24 bytes. */
SAILOR_HIP_API int sailor_hip_self_check_exact_math(SailorHipContext* ctx, void* dScratch, uint64_t mismatches[3]);

/* The split-sum BRDF look-up table sampled by AmbientLighting: Content/Shaders/ComputeBrdfLut.shader:26-71 (1 024 Hammersley /
 * GGX samples per texel), dispatched once at start-up.  dLut: device, height x width float2 (the reference image is RG16F). */
SAILOR_HIP_API int sailor_hip_compute_brdf_lut(SailorHipContext* ctx, float* dLut, int32_t width, int32_t height);

/* Replaces: IGraphicsDriverCommands::ConvertEquirect2Cubemap (GraphicsDriver/Vulkan/VulkanGraphicsDriver.cpp:1662-1686) =
 * Content/Shaders/ComputeEquirect2Cube.shader:20-58, the first step of the raw environment cube (FrameGraph/EnvironmentNode.cpp:131-135).
 *   dEquirect : device in, eqWidth x eqHeight RGBA32F ("src"), rows top to bottom; `repeat` != 0 = the texture's sampler wraps
 *               (TextureAssetInfo.h:30 default), 0 = clamp-to-edge
 *   dCube     : device out, 6 x size x size RGBA32F = level 0 of the cube chain ("dst"; the reference image is RGBA16F)
 *   coverWidth / coverHeight : the reference dispatches equirectExtent / 32 groups (not cubeSize / 32), so only texels with
 *               x < coverWidth and y < coverHeight are written; pass `size` twice for the whole cube */
SAILOR_HIP_API int sailor_hip_equirect_to_cube(SailorHipContext* ctx, const float* dEquirect, int32_t eqWidth, int32_t eqHeight, int32_t repeat,
                                               float* dCube, int32_t size, int32_t coverWidth, int32_t coverHeight);
/* Replaces: IGraphicsDriverCommands::GenerateMipMaps on a cubemap (GraphicsDriver/Vulkan/VulkanCommandBuffer.cpp:814-907; called at
 * EnvironmentNode.cpp:137): level i = a 2:1 VK_FILTER_LINEAR blit of level i - 1, face by face = the mean of 2 x 2 texels.
 *   dCube : device in/out, the level-major RGBA32F chain; level 0 is read, levels 1 .. levels-1 are written */
SAILOR_HIP_API int sailor_hip_generate_mipmaps_cube(SailorHipContext* ctx, float* dCube, int32_t size, int32_t levels);

/* Replaces: Content/Shaders/ComputeIrradianceMap.shader:78-101 for the Dispatch at FrameGraph/EnvironmentNode.cpp:264-269
 * (IrradianceMapSize / 32 groups squared x 6): 65 536 uniform hemisphere samples per texel of the environment cube at lod 0.
 *   dEnv        : device in, RGBA32F cube mip chain ("envMap"): level-major, then face (+X -X +Y -Y +Z -Z), then rows of texels
 *   dIrradiance : device out, 6 x size x size RGBA32F ("irradianceMap"; the reference image is RGBA16F), alpha = 1 */
SAILOR_HIP_API int sailor_hip_compute_irradiance_map(SailorHipContext* ctx, const float* dEnv, int32_t envSize, int32_t envLevels,
                                                     float* dIrradiance, int32_t size);

/* Replaces: the "pre-filter the mip chain" section of FrameGraph/EnvironmentNode.cpp:196-233 -- level 0 copied from the raw cube (:200-203),
 * level l = 1 .. levels-1 by Content/Shaders/ComputeEnvMap_IBL.shader:76-136 with push constants { level - 1, roughness = l / (levels - 1) }:
 * 1 024 GGX importance samples per texel, mip-filtered lookups into all levels of the raw cube.
 *   dRawEnv : device in, RGBA32F cube mip chain ("rawEnvMap", size x size x 6, `levels` levels, layout as above)
 *   dEnv    : device out, the same layout ("envMap[]"); must not alias dRawEnv */
SAILOR_HIP_API int sailor_hip_prefilter_env_map(SailorHipContext* ctx, const float* dRawEnv, float* dEnv, int32_t size, int32_t levels);
/* One Dispatch of that loop (EnvironmentNode.cpp:223-231): output mip `level` (the shader's push constant is level - 1) at `roughness`. */
SAILOR_HIP_API int sailor_hip_prefilter_env_level(SailorHipContext* ctx, const float* dRawEnv, float* dEnv, int32_t size, int32_t levels,
                                                  int32_t level, float roughness);

/* ---- EVSM shadow-map blur ---------------------------------------------------------------
 * Replaces: the two full-screen draws "Blur Horizontal" / "Blur Vertical" of ShadowPrepassNode::Process
 * (FrameGraph/ShadowPrepassNode.cpp:283-356) with Content/Shaders/Blur.shader (defines EVSM + HORIZONTAL | VERTICAL, :66-98) ->
 * GaussianBlur_Evsm (Lighting.glsl:83-127) over the cascade-0 moments map (RGBA32F): .zw blurred with the umbra radius, .xy with
 * the penumbra radius (RHI/SceneView.h:60; ECS/LightingECS.h:68 ShadowCascadeBlur = (2, 5) for cascade 0), radii capped at 12.
 *   dMap  : device, height x width float4, blurred IN PLACE (the node renders into a temporary target and back)
 *   dTemp : device scratch of the same size (the node's blurAttachment)
 * Taps sit on texel centres: the texel itself, clamp-to-edge.  Same sums in the same order as the shader: bit-exact vs the oracle. */
SAILOR_HIP_API int sailor_hip_evsm_blur(SailorHipContext* ctx, float* dMap, float* dTemp, int32_t width, int32_t height,
                                        int32_t radiusUmbra, int32_t radiusPenumbra);
/* one of the two draws: vertical == 0 is Blur.shader with HORIZONTAL (texelSize.y = 0, :72-74), else VERTICAL (:68-70) */
SAILOR_HIP_API int sailor_hip_evsm_blur_pass(SailorHipContext* ctx, const float* dSrc, float* dDst, int32_t width, int32_t height,
                                             int32_t radiusUmbra, int32_t radiusPenumbra, int32_t vertical);

/* ---- K4: ECS transform + bounds + frustum-cull sweep -------------------------------------------------------
 * Replaces: TransformECS::Tick full-sweep branch + CalculateMatrices (ECS/TransformECS.cpp:144-212),
 * Transform::Matrix (Math/Transform.cpp:39-42), the AABB::Apply of StaticMeshRendererECS::Tick
 * (ECS/StaticMeshRendererECS.cpp:40-58, Math/Bounds.cpp:479-492) and the Frustum::OverlapsAABB sweep that
 * RHISceneView::TraceScene performs through the octree (RHI/SceneView.cpp:56, Math/Bounds.cpp:245-260), over flat
 * level-sorted arrays.
 *
 *   entities are level-sorted: levelOffsets[l]..levelOffsets[l+1] is hierarchy level l (host array of
 *   numLevels+1 entries), dParent[i] = index of the parent (in an earlier level) or 0xFFFFFFFF for roots
 *   planes         : host, 6 x vec4 (L,R,T,B,N,F) from sailor_host_extract_frustum_planes
 *   dWorld         : device out, mat4 per entity (m_cachedWorldMatrix)
 *   dWorldAabb     : device out, SailorAABB per entity
 *   dVisibility    : device out, 1 bit per entity, LSB-first in uint64 words, ceil(n/64) words
 */
SAILOR_HIP_API int sailor_hip_ecs_sweep(SailorHipContext* ctx, uint32_t numEntities,
                                        const SailorTransform* dTransforms, const uint32_t* dParent,
                                        const uint32_t* levelOffsets, uint32_t numLevels,
                                        const SailorAABB* dLocalAabb, const float* planes,
                                        float* dWorld, SailorAABB* dWorldAabb, uint64_t* dVisibility);
/* K4 split across the ranks of a node (the reference fans the same sweep out in 1 024-entity chunks over its worker threads,
 * ECS/StaticMeshRendererECS.cpp:17-150): the sweep of the slice [entityBegin, entityEnd) of the entity array only -- its entries of dWorld / dWorldAabb,
 * its bits of dVisibility.  A hierarchy of at most four levels (the one-launch form: an entity rebuilds its ancestors' relative matrices from their TRS
 * records, the same bits as the level-by-level product) takes ANY slice and needs nothing another rank computes; a deeper one reads its parents' world
 * matrices and is refused unless the slice is the whole set (SAILOR_HIP_ERR_UNSUPPORTED: sweep it replicated).
 * sailor_hip_ecs_range_for_rank: rank r's slice of an equal split in whole 64-entity visibility words, ceil(words / worldSize) = *outWordsPerRank per
 * rank (the last ranks may get fewer entities, or none).  sailor_hip_exchange_visibility: ONE in-place ncclAllGather of those words on `comm` and the
 * context's stream -- dVisibility must hold worldSize * wordsPerRank uint64, which is MORE than sailor_hip_ecs_sweep's ceil(n / 64) when the words do
 * not divide evenly (n = 100 on 8 ranks: 8 words, not 2); `visibilityWords` is the buffer's capacity in uint64 and a smaller one is refused
 * (SAILOR_HIP_ERR_INVALID_ARGUMENT) instead of overrun -- after which every rank holds the whole bitmask (C5: 128 KB).  World
 * matrices and boxes stay where they were computed: a consumer that needs ALL of them on every rank (instance data for draws) is better served by the
 * replicated sweep -- 64 B + 24 B per entity over xGMI cost more than the 38 us the whole sweep takes on one GPU. */
SAILOR_HIP_API int sailor_hip_ecs_sweep_range(SailorHipContext* ctx, uint32_t numEntities,
                                              const SailorTransform* dTransforms, const uint32_t* dParent,
                                              const uint32_t* levelOffsets, uint32_t numLevels,
                                              const SailorAABB* dLocalAabb, const float* planes,
                                              float* dWorld, SailorAABB* dWorldAabb, uint64_t* dVisibility,
                                              uint32_t entityBegin, uint32_t entityEnd);
SAILOR_HIP_API int sailor_hip_ecs_range_for_rank(uint32_t numEntities, int32_t rank, int32_t worldSize, uint32_t* outBegin, uint32_t* outEnd,
                                                 uint32_t* outWordsPerRank /* or NULL */);
SAILOR_HIP_API int sailor_hip_exchange_visibility(SailorHipContext* ctx, void* comm, int32_t rank, int32_t worldSize, uint32_t numEntities,
                                                  uint64_t* dVisibility, size_t visibilityWords);

#define SAILOR_RASTER_CLEAR 1u
#define SAILOR_RASTER_CULL_BACK 2u
/* Replaces: the caster draws of one shadow pass, FrameGraph/ShadowPrepassNode.cpp:219-262 with Content/Shaders/ShadowCaster.shader:46-59 (vertex stage:
 * gl_Position = lightMatrix * instance.model * vec4(inPosition, 1)) -- depth only; the rasterisation rules are those of the oracle (1/256-pixel
 * snapping, top-left rule, unfused fp32 depth interpolation, reversed Z: GREATER against a buffer cleared to 0, viewport (0, H, W, -H)).
 *   lightMatrix  : host, mat4 (the pass' push constant, RHIUpdateShadowMapCommand::m_lightMatrix)
 *   dPositions   : device, vec3 per vertex (VertexP3N3T3B3UV2C4::m_position); dIndices: device, 3 x numTriangles
 *   dModels      : device, mat4 per instance (PerInstanceData.model); dInstanceIds: device, numDrawn instance indices, or NULL for 0 .. numDrawn-1
 *   dDepth       : device in/out, width x height floats
 *   flags        : SAILOR_RASTER_CLEAR clears dDepth (and the coarse depth) to 0 first -- a dependent pass (:250-261) draws on top without it;
 *                  SAILOR_RASTER_CULL_BACK discards back faces as the reference's materials do (ECullMode::Back, frontFace counter-clockwise)
 *   dCoarseDepth : device scratch or NULL, sailor_hip_raster_coarse_words(width, height) words belonging to dDepth (cleared with it): lower bounds of the
 *                  depths stored in each 8 x 8 block and each 64 x 64 superblock, used to skip what cannot win any more -- whole triangles, blocks, and
 *                  (round 6, draws of >= 4 096 instances) whole INSTANCES: the box of the mesh's referenced vertices through the instance's matrix -- and,
 *                  behind those, the box itself (8 words) and a queue of "giant" triangles (4 + 24 words per entry, two entries per superblock): a
 *                  triangle that is visible and large on the map is drawn by sixty-four waves of a second kernel instead of block by block by the wave
 *                  that set it up, and a draw of many instances goes out in up to six launches of growing size so that a launch's giants are on the map
 *                  before the next launch starts (a caller that sorts its instances front to back gets most of the later ones skipped).  All of it is
 *                  bounds and scheduling only: the depth buffer is bit for bit the same with and without the workspace, in any drawing order. */
SAILOR_HIP_API size_t sailor_hip_raster_coarse_words(int32_t width, int32_t height);
SAILOR_HIP_API int sailor_hip_raster_depth(SailorHipContext* ctx, const float* lightMatrix, const float* dPositions, const uint32_t* dIndices,
                                           uint32_t numTriangles, const float* dModels, const uint32_t* dInstanceIds, uint32_t numDrawn,
                                           int32_t width, int32_t height, float* dDepth, uint32_t flags, uint32_t* dCoarseDepth);
/* Replaces: the draws of the depth prepass, FrameGraph/DepthPrepassNode.cpp:283-297 with Content/Shaders/DepthOnly.shader:51
 * (gl_Position = frame.projection * (frame.view * (model * position))): the same rasteriser, the camera's matrices from the frame UBO; the reversed-Z
 * projection makes it GREATER against the cleared 0 again.  dDepth is the raw depth attachment LinearizeDepth / SAILOR_CULL_RAW_DEPTH consume.
 * Triangles that cross the near plane (z_clip > w_clip, which includes vertices behind the eye) are cut against it in clip space, as in the oracle. */
SAILOR_HIP_API int sailor_hip_raster_depth_camera(SailorHipContext* ctx, const SailorUboFrameData* frame, const float* dPositions, const uint32_t* dIndices,
                                                  uint32_t numTriangles, const float* dModels, const uint32_t* dInstanceIds, uint32_t numDrawn,
                                                  int32_t width, int32_t height, float* dDepth, uint32_t flags, uint32_t* dCoarseDepth);
/* The fragment stage of ShadowCaster.shader:66-78 on the winning depth of every texel: EVSM moments (format RGBA32F: exp(40 z), its square,
 * -exp(-40 z), its square), or the depth itself (R16F / R32F); texels nothing was drawn to keep the cleared colour 0. */
SAILOR_HIP_API int sailor_hip_shadow_resolve(SailorHipContext* ctx, const float* dDepth, int32_t width, int32_t height, int32_t format, void* dShadowMap);

/* The cascade mesh lists of LightingECS::PrepareCSMPasses (ECS/LightingECS.cpp:287-296: `cascade.m_meshList = sceneView->TraceScene(frustums[k], true)`)
 * as bitmasks over the entities of sailor_hip_ecs_sweep: bit i of mask k = dWorldAabb[i] overlaps the frustum of cascade k (Math/Bounds.cpp:245-260).
 *   cascadePlanes : host, numCascades x 6 x vec4 from sailor_host_extract_frustum_planes_matrix(lightCascadesMatrices[k] * lightMatrix) (:287-292)
 *   dMasks        : device out, numCascades x ceil(numEntities / 64) words, LSB first
 * The removal of meshes already drawn by an earlier cascade of the same kind (:310-327) and the change tracking (:334-366, CSMLightState::Equals :14-38)
 * stay on the host, on these sets. */
SAILOR_HIP_API int sailor_hip_csm_caster_masks(SailorHipContext* ctx, uint32_t numEntities, const SailorAABB* dWorldAabb, const float* cascadePlanes,
                                               uint32_t numCascades, uint64_t* dMasks);

/* Replaces: FrustumCulling of Content/Shaders/ComputeMeshCulling.shader:96-110 (+ CreateViewFrustum, Math.glsl:185-222)
 * for the Dispatch at RHI/Batch.hpp:188; writes PerInstanceData::isCulled in place for the instances
 * [firstInstanceIndex, firstInstanceIndex + numInstances).  Hi-Z occlusion (the shader's OCCLUSION_CULLING define) is out of
 * scope. */
SAILOR_HIP_API int sailor_hip_mesh_frustum_cull(SailorHipContext* ctx, const SailorUboFrameData* frame,
                                                SailorPerInstanceData* dInstances, uint32_t numInstances, uint32_t firstInstanceIndex);

/* Replaces: the whole main() of Content/Shaders/ComputeMeshCulling.shader:119-177 built without OCCLUSION_CULLING, for the same
 * Dispatch (RHI/Batch.hpp:177-188; push constants numBatches / numInstances / firstInstanceIndex):
 *   step 2 (:126-144)  isCulled over the instance window, as sailor_hip_mesh_frustum_cull;
 *   step 3 (:146-177)  per indirect draw a stable in-place compaction of its instance records
 *                      [firstInstance, firstInstance + instanceCount) by isCulled == 0, and instanceCount = number kept.
 * Every flag is written before any batch is compacted (the canonical reading of the shader's cross-workgroup race).  The batches
 * own disjoint instance ranges inside the buffer and their instanceCounts add up to at most numInstances (exactly numInstances at
 * RHI/Batch.hpp:158-159,183).  Records behind a batch's kept prefix keep their old contents, as in the shader.
 *   dBatches   : device in/out, numBatches x SailorDrawIndexedIndirectData ("drawIndexedIndirect", set 2 binding 0)
 *   dWorkspace : device scratch, 256-byte aligned, sailor_hip_mesh_cull_workspace_bytes(numInstances, numBatches) bytes */
/* The Hi-Z pyramid ("depthHighZ", set 0 binding 0 of ComputeMeshCulling.shader; render target DepthHighZ of Content/DefaultRenderer.renderer:51-57:
 * R32_SFLOAT, mips, sampler reduction Min): level-major, level l = max(height >> l, 1) rows of max(width >> l, 1) floats of raw reversed-Z depth. */
typedef struct SailorHiZDesc {
    const float* pyramid;
    int32_t width, height, levels;
} SailorHiZDesc;

/* Replaces: Content/Shaders/ComputeDepthHighZ.shader:22-30 for one Dispatch of FrameGraph/DepthHighZNode.cpp:90-95 -- out(pos) = the minimum
 * over the bilinear footprint of the source at (pos + 0.5) / outputSize (texels with non-zero weight, clamp-to-edge). */
SAILOR_HIP_API int sailor_hip_hiz_downscale(SailorHipContext* ctx, const float* dSrc, int32_t srcWidth, int32_t srcHeight, float* dDst,
                                            int32_t dstWidth, int32_t dstHeight);
/* The node's whole loop (DepthHighZNode.cpp:78-96): mip 0 from the depth attachment "src", mip i + 1 from mip i. */
SAILOR_HIP_API int sailor_hip_hiz_build(SailorHipContext* ctx, const float* dDepth, int32_t depthWidth, int32_t depthHeight, float* dPyramid,
                                        int32_t width, int32_t height, int32_t levels);
/* sailor_hip_mesh_frustum_cull with the shader's OCCLUSION_CULLING define when `hiz` is given: isCulled = FrustumCulling || OcclusionCulling
 * (ComputeMeshCulling.shader:62-94,136-140; ProjectSphere Math.glsl:296-315; floor(log2()) of the mip selection taken from the exponent field). */
SAILOR_HIP_API int sailor_hip_mesh_cull_flags(SailorHipContext* ctx, const SailorUboFrameData* frame, SailorPerInstanceData* dInstances,
                                              uint32_t numInstances, uint32_t firstInstanceIndex, const SailorHiZDesc* hiz);
SAILOR_HIP_API size_t sailor_hip_mesh_cull_workspace_bytes(uint32_t numInstances, uint32_t numBatches);
/* sailor_hip_mesh_cull_compact (below) with the Hi-Z pyramid: the shader as shipped (`defines: - OCCLUSION_CULLING`); hiz == NULL = frustum only */
SAILOR_HIP_API int sailor_hip_mesh_cull_compact_ex(SailorHipContext* ctx, const SailorUboFrameData* frame, SailorPerInstanceData* dInstances,
                                                   uint32_t numInstances, uint32_t firstInstanceIndex, SailorDrawIndexedIndirectData* dBatches,
                                                   uint32_t numBatches, void* dWorkspace, size_t workspaceBytes, const SailorHiZDesc* hiz);
SAILOR_HIP_API int sailor_hip_mesh_cull_compact(SailorHipContext* ctx, const SailorUboFrameData* frame, SailorPerInstanceData* dInstances,
                                                uint32_t numInstances, uint32_t firstInstanceIndex, SailorDrawIndexedIndirectData* dBatches,
                                                uint32_t numBatches, void* dWorkspace, size_t workspaceBytes);

/* ---- RCCL exchange for split frames (only when the frame is split AND a consumer needs the global list) ----
 * `comm` is an ncclComm_t created by the host.  Collective 1: all-gather of one uint32 (band total) per rank.
 * Collective 2: all-gather of the padded band index segments (each rank contributes `segmentCapacity` uints). */
SAILOR_HIP_API int sailor_hip_allgather_u32(SailorHipContext* ctx, void* comm, const uint32_t* dSend, uint32_t* dRecv, size_t countPerRank);

/* The whole exchange of a split frame (the reference's single Dispatch at FrameGraph/LightCullingNode.cpp:74-77 fills ONE pair of
 * buffers, a split frame has one pair per band): every rank hands in the lists of its band (sailor_hip_band_for_rank(width, height, rank,
 * worldSize), as sailor_hip_light_cull left them) and gets the reference's global `lightsGrid` (tiles x {offset, num}) and `culledLights`
 * ([0] = sum of num, then the lists at the canonical offsets) -- three ncclAllGather on the context's stream (band total; index segments in slots
 * sized by sailor_hip_exchange_adapt from an earlier exchange's totals, or of the worst case; grid) and one stitch kernel that takes each band's global
 * base from the gathered totals on the device.  The call only RECORDS (round 6; round 5 read the gathered totals back and synchronised the stream to
 * size the second gather): nothing is read back, nothing waits, and it may be captured into a hipGraph.
 *   comm          : an ncclComm_t of `worldSize` ranks created by the host (RCCL over xGMI)
 *   dGlobalCulled : globalCapacity uints, 1 + tiles * 128 holds every result
 *   dWorkspace    : sailor_hip_exchange_workspace_size(width, height, worldSize) bytes, 256-byte aligned */
#define SAILOR_MAX_SPLIT 64
SAILOR_HIP_API size_t sailor_hip_exchange_workspace_size(int32_t width, int32_t height, int32_t worldSize);
SAILOR_HIP_API int sailor_hip_exchange_light_lists(SailorHipContext* ctx, void* comm, int32_t rank, int32_t worldSize, int32_t width, int32_t height,
                                                   const SailorLightsGrid* dBandGrid, const uint32_t* dBandCulled, SailorLightsGrid* dGlobalGrid,
                                                   uint32_t* dGlobalCulled, size_t globalCapacity, void* dWorkspace, size_t workspaceBytes);
/* The stitch alone, on gathered slots: dTotals[worldSize]; dSegments[worldSize][segmentCapacity] (band r's culledLights[1 ..]);
 * dGrids[worldSize][gridCapacity] (band r's lightsGrid as uint32 pairs, band-local offsets). */
SAILOR_HIP_API int sailor_hip_stitch_light_lists(SailorHipContext* ctx, int32_t width, int32_t height, int32_t worldSize, const uint32_t* dTotals,
                                                 const uint32_t* dSegments, size_t segmentCapacity, const uint32_t* dGrids, size_t gridCapacity,
                                                 SailorLightsGrid* dGlobalGrid, uint32_t* dGlobalCulled, size_t globalCapacity);

/* The same two with the split given explicitly: tileRowBounds[worldSize + 1], rank r's band = tile rows [bounds[r], bounds[r + 1]) (bounds[0] = 0,
 * bounds[worldSize] = tile rows of the frame, non-decreasing) -- equal bands, cost-balanced bands (sailor_hip_band_from_tile_rows), anything in
 * between -- and with the capacity of dGlobalGrid in tiles (the stitch writes one entry per tile of the frame: fewer is an error, not an overrun).
 * The entry points above are these with the bounds of sailor_hip_band_for_rank and a grid of exactly the frame's tiles.  culledLights[0] is clamped
 * to what was written (globalCapacity - 1) when the segments did not fit. */
SAILOR_HIP_API size_t sailor_hip_exchange_workspace_size_rows(int32_t width, int32_t height, int32_t worldSize, const int32_t* tileRowBounds);
SAILOR_HIP_API int sailor_hip_exchange_light_lists_rows(SailorHipContext* ctx, void* comm, int32_t rank, int32_t worldSize, int32_t width, int32_t height,
                                                        const int32_t* tileRowBounds, const SailorLightsGrid* dBandGrid, const uint32_t* dBandCulled,
                                                        SailorLightsGrid* dGlobalGrid, size_t globalGridTiles, uint32_t* dGlobalCulled, size_t globalCapacity,
                                                        void* dWorkspace, size_t workspaceBytes);
SAILOR_HIP_API int sailor_hip_stitch_light_lists_rows(SailorHipContext* ctx, int32_t width, int32_t height, int32_t worldSize, const int32_t* tileRowBounds,
                                                      const uint32_t* dTotals, const uint32_t* dSegments, size_t segmentCapacity, const uint32_t* dGrids,
                                                      size_t gridCapacity, SailorLightsGrid* dGlobalGrid, size_t globalGridTiles, uint32_t* dGlobalCulled,
                                                      size_t globalCapacity);

/* The slot size of the exchange's second gather.  A context starts with the worst case (tiles of the largest band x 128 indices per rank: 16.7 MB gathered
 * per rank at C3 / 8 ranks for ~3 MB of lists).  sailor_hip_exchange_adapt is the exchange's one SYNCHRONISING call (like sailor_hip_context_synchronize):
 * it waits for the last exchange recorded through `ctx` -- its own event, not the stream's later work -- reads the three status words that exchange's
 * stitch kernel left in pinned host memory, and sizes the next exchange's slots from the largest band total it gathered (+ 25 %, whole 256-byte lines).
 * If that exchange was CLIPPED -- a band's total had outgrown a slot sized from an earlier frame: its global lists are incomplete -- *outClipped is 1, the
 * context's error text says which, and the next exchange goes back to the worst case.  The totals are the same on every rank, so the slot sizes are too,
 * PROVIDED every rank of the communicator calls this at the same point of its call sequence (the HIP backend: in front of every exchange but the first).
 * sailor_hip_exchange_set_slot_words sets the size explicitly (0 = the worst case).  Outputs may be NULL. */
SAILOR_HIP_API int sailor_hip_exchange_adapt(SailorHipContext* ctx, uint32_t* outLargestBandTotal, int32_t* outClipped, size_t* outSlotWords);
SAILOR_HIP_API int sailor_hip_exchange_set_slot_words(SailorHipContext* ctx, size_t slotWords);

/* ---- host-side math of the path (pure CPU, no device needed) ----------------------------------------------- */
/* Math/Math.cpp:18-21 PerspectiveRH (reversed Z) */
SAILOR_HIP_API int sailor_host_perspective_rh(float fovRadians, float aspect, float zNear, float zFar, float* outMat4);
/* glm::inverse(mat4) as used at ECS/CameraECS.cpp:20,33 */
SAILOR_HIP_API int sailor_host_mat4_inverse(const float* m, float* outMat4);
SAILOR_HIP_API int sailor_host_mat4_mul(const float* a, const float* b, float* outMat4);
/* Math/Transform.cpp:39-42 */
SAILOR_HIP_API int sailor_host_transform_matrix(const SailorTransform* t, float* outMat4);
/* FrameGraph/RHIFrameGraph.cpp:60-67 FillFrameData from a camera world matrix + lens */
SAILOR_HIP_API int sailor_host_fill_frame_data(const float* cameraWorld, float fovDegrees, float aspect, float zNear, float zFar,
                                               int32_t viewportWidth, int32_t viewportHeight, float currentTime, float deltaTime,
                                               SailorUboFrameData* outFrame);
/* Math/Bounds.cpp:142-193 Frustum::ExtractFrustumPlanes(world, aspect, fovY[deg], zNear, zFar): 6 planes, 8 corners (may be NULL) */
SAILOR_HIP_API int sailor_host_extract_frustum_planes(const float* worldMatrix, float aspect, float fovYDegrees, float zNear, float zFar,
                                                      float* outPlanes24, float* outCorners24);
/* FrameGraph/ShadowPrepassNode.cpp:387-404 + Math/Bounds.cpp:78-109 + ECS/LightingECS.cpp:292: the 4 lightsMatrices */
/* Math/Bounds.cpp:20-67 Frustum::ExtractFrustumPlanes(projectionViewMatrix) (+ CalculateCorners :110-140, reversed Z): planes L R T B N F, 8 corners */
SAILOR_HIP_API int sailor_host_extract_frustum_planes_matrix(const float* projectionViewMatrix, float* outPlanes24, float* outCorners24);
SAILOR_HIP_API int sailor_host_csm_matrices(const float* lightView, const float* cameraWorld, float aspect, float fovYDegrees,
                                            float cameraNear, float cameraFar, float* outMatrices64);
/* ECS/LightingECS.cpp:163-172: pack one light (cutOff in degrees -> cosines) */
SAILOR_HIP_API int sailor_host_pack_light(uint32_t type, uint32_t shadowType, const float* worldPosition, const float* direction,
                                          const float* intensity, const float* attenuation, const float* cutOffDegrees, const float* bounds,
                                          SailorLightShaderData* outLight);


/* Math/Bounds.cpp:211-243 Frustum::OverlapsSphere / ContainsSphere (scalar): sphere4 = (centre.xyz, radius); 1 / 0, negative = bad argument.
 * OverlapsSphere: no plane has the whole sphere behind it.  ContainsSphere: the sphere lies in front of all six planes by at least its radius. */
SAILOR_HIP_API int sailor_host_overlaps_sphere(const float* planes24, const float* sphere4);
SAILOR_HIP_API int sailor_host_contains_sphere(const float* planes24, const float* sphere4);
/* ECS/LightingECS.cpp:209-260 LightingECS::GetLightsInFrustum over flat component arrays (type, shadowType, active flag or NULL = all active,
 * owner position, bounds): shadow-casting directional lights in component order; point and spot lights whose sphere (radius = largest bound)
 * is contained in the frustum, sorted by distance to the camera with the reference's lower_bound insertion (equal distances: the later
 * component goes in front).  Every output array needs room for numLights entries. */
SAILOR_HIP_API int sailor_host_lights_in_frustum(const float* planes24, const float* cameraPosition3, uint32_t numLights, const uint32_t* types,
                                                 const uint32_t* shadowTypes, const uint8_t* active, const float* positions3, const float* bounds3,
                                                 uint32_t* outDirectional, uint32_t* outNumDirectional,
                                                 uint32_t* outPoint, float* outPointDistance, uint32_t* outNumPoint,
                                                 uint32_t* outSpot, float* outSpotDistance, uint32_t* outNumSpot);

/* The change tracking of LightingECS::PrepareCSMPasses (ECS/LightingECS.cpp:299-366) on the cascade overlap sets that
 * sailor_hip_csm_caster_masks produces.  SailorCsmSnapshots is LightingECS::m_csmSnapshots; SailorCsmView the transform half of a
 * CSMLightState (:14-38): light component index, camera position / rotation (xyzw), light position / rotation. */
typedef struct SailorCsmView {
    uint32_t componentIndex;
    float cameraPosition[4], cameraRotation[4], lightPosition[4], lightRotation[4];
} SailorCsmView;
typedef struct SailorCsmSnapshots SailorCsmSnapshots; /* opaque */
SAILOR_HIP_API SailorCsmSnapshots* sailor_host_csm_snapshots_create(void);
SAILOR_HIP_API SailorCsmSnapshots* sailor_host_csm_snapshots_clone(const SailorCsmSnapshots* snapshots);
SAILOR_HIP_API void sailor_host_csm_snapshots_destroy(SailorCsmSnapshots* snapshots);
SAILOR_HIP_API uint32_t sailor_host_csm_snapshots_count(const SailorCsmSnapshots* snapshots);
SAILOR_HIP_API int sailor_host_csm_snapshot_get(const SailorCsmSnapshots* snapshots, uint32_t index, uint32_t capacity, uint32_t* outCount, uint32_t* outMeshes,
                                                uint64_t* outFrames, int32_t* outHasView, SailorCsmView* outView);
/* One directional light: for cascade k = 0 .. numCascades-1 (snapshot slots firstSnapshot + k)
 *   * drop from its overlap set every mesh that an EARLIER cascade of the same shadow type, re-rendered this frame, overlaps (:310-327);
 *   * re-render it iff its (mesh, frame the mesh last changed) list differs from the stored snapshot, or -- with `view` -- the camera moved more
 *     than 15 units / turned past dot(forward, forward') = 0.9995 / the light moved at all since that snapshot was TAKEN (a kept snapshot keeps
 *     its old camera, :353-357).
 * overlapMasks / outMasks: numCascades x ceil(numEntities / 64) words; outRender: numCascades flags; lastChangedFrame: one per entity. */
SAILOR_HIP_API int sailor_host_csm_plan_passes(SailorCsmSnapshots* snapshots, uint32_t firstSnapshot, uint32_t numCascades, uint32_t numEntities,
                                               const uint64_t* overlapMasks, const uint32_t* shadowTypes, const uint64_t* lastChangedFrame,
                                               const SailorCsmView* view, uint32_t* outRender, uint64_t* outMasks);

#ifdef __cplusplus
}
#endif
#endif /* SAILOR_HIP_H */
