#include "HipGraphicsDriver.h"
#include <dlfcn.h>
#include "../../RHI/Renderer.h"

#include <cstdio>

using namespace Sailor;
using namespace Sailor::RHI;
using namespace Sailor::GraphicsDriver::HIP;

IGraphicsDriver* Renderer::s_driver = nullptr;
IGraphicsDriverCommands* Renderer::s_commands = nullptr;

Renderer::Renderer(int deviceOrdinal, void* stream, bool ownStream)
{
    auto* hip = new HipGraphicsDriver(deviceOrdinal, stream, ownStream);
    m_driverInstance.reset(hip);
    m_status = hip->GetStatus();
    s_driver = hip;
    s_commands = hip; // the reference dynamic_casts the same object (RHI/Renderer.cpp:156-164)
}

Renderer::~Renderer()
{
    s_driver = nullptr;
    s_commands = nullptr;
}

RHIBuffer::~RHIBuffer()
{
    if (m_hip.m_bOwned && m_hip.m_devicePtr && m_ctx) sailor_hip_buffer_free(m_ctx.get(), m_hip.m_devicePtr);
}

HipGraphicsDriver::HipGraphicsDriver(int deviceOrdinal, void* stream, bool ownStream)
{
    m_status = sailor_hip_context_create(deviceOrdinal, stream, ownStream ? SAILOR_CTX_OWN_STREAM : 0u, &m_ctx);
    if (m_ctx) m_ctxOwner = std::shared_ptr<SailorHipContext>(m_ctx, [](SailorHipContext* c) { sailor_hip_context_destroy(c); });
    // the second queue (k1_pack beside the shade); without it the cull packs inline as in rounds 1-3
    if (m_ctx && sailor_hip_context_create(deviceOrdinal, nullptr, SAILOR_CTX_OWN_STREAM, &m_ctxAux) != SAILOR_HIP_OK) m_ctxAux = nullptr;
}

HipGraphicsDriver::~HipGraphicsDriver()
{
    m_cullWorkspace.Clear();
    m_meshCullWorkspace.Clear();
    m_exchangeWorkspace.Clear();
    if (m_ctxAux) { sailor_hip_context_synchronize(m_ctxAux); sailor_hip_context_destroy(m_ctxAux); m_ctxAux = nullptr; }
    if (m_ctx) sailor_hip_context_synchronize(m_ctx);
    m_ctx = nullptr;
    m_ctxOwner.reset(); // the context itself goes with the last buffer that refers to it
}

void HipGraphicsDriver::WaitIdle()
{
    if (m_ctxAux) sailor_hip_context_synchronize(m_ctxAux);
    if (m_ctx) sailor_hip_context_synchronize(m_ctx);
}

RHICommandListPtr HipGraphicsDriver::CreateCommandList(bool) { return RHICommandListPtr::Make(); }

RHIBufferPtr HipGraphicsDriver::CreateBuffer(size_t size)
{
    auto b = RHIBufferPtr::Make();
    b->m_ctx = m_ctxOwner;
    b->m_size = size;
    if (!m_ctx || sailor_hip_buffer_create(m_ctx, size, &b->m_hip.m_devicePtr) != SAILOR_HIP_OK) return RHIBufferPtr();
    b->m_hip.m_bOwned = true;
    return b;
}

RHIBufferPtr HipGraphicsDriver::WrapBuffer(void* devicePtr, size_t size)
{
    auto b = RHIBufferPtr::Make();
    b->m_ctx = m_ctxOwner;
    b->m_size = size;
    b->m_hip.m_devicePtr = devicePtr;
    b->m_hip.m_bOwned = false;
    return b;
}

RHIShaderPtr HipGraphicsDriver::CreateShader(const std::string& assetPath, const TVector<std::string>& defines) { return RHIShaderPtr::Make(assetPath, defines); }

static size_t texel_size(EFormat f) { return f == EFormat::R16_SFLOAT ? 2 : (f == EFormat::R32_SFLOAT ? 4 : (f == EFormat::R32G32_SFLOAT ? 8 : 16)); }

RHITexturePtr HipGraphicsDriver::CreateTexture(const void* pData, size_t size, ivec2 extent, EFormat format)
{
    auto t = RHITexturePtr::Make();
    t->m_extent = extent;
    t->m_format = format;
    const size_t bytes = (size_t)extent.x * extent.y * texel_size(format);
    t->m_buffer = CreateBuffer(bytes);
    if (!t->m_buffer) return RHITexturePtr();
    if (pData) sailor_hip_buffer_upload(m_ctx, t->m_buffer->m_hip.m_devicePtr, 0, pData, size < bytes ? size : bytes);
    return t;
}

static size_t cube_chain_texels(int size, uint32_t levels)
{
    size_t n = 0;
    for (uint32_t l = 0; l < levels; l++) { const int sz = (size >> l) > 1 ? (size >> l) : 1; n += (size_t)6 * sz * sz; }
    return n;
}

RHITexturePtr RHI::RHITexture::GetMipLevel(uint32_t mipLevel) const
{
    auto v = RHITexturePtr::Make();
    v->m_buffer = m_buffer; v->m_format = m_format; v->m_bCubemap = m_bCubemap; v->m_mipLevels = 1;
    v->m_extent = { (m_extent.x >> mipLevel) > 1 ? (m_extent.x >> mipLevel) : 1, (m_extent.y >> mipLevel) > 1 ? (m_extent.y >> mipLevel) : 1 };
    v->m_viewLevel = mipLevel;
    v->m_parent = RHITexturePtr(const_cast<RHI::RHITexture*>(this));
    return v;
}

RHITexturePtr HipGraphicsDriver::CreateRenderTarget(ivec2 extent, uint32_t mipLevels, EFormat format)
{
    if (mipLevels <= 1) return CreateTexture(nullptr, 0, extent, format);
    auto t = RHITexturePtr::Make(); // a mip chain: level-major, like the cubemaps
    t->m_extent = extent; t->m_format = format; t->m_mipLevels = mipLevels;
    size_t texels = 0;
    for (uint32_t l = 0; l < mipLevels; l++) texels += (size_t)((extent.x >> l) > 1 ? (extent.x >> l) : 1) * ((extent.y >> l) > 1 ? (extent.y >> l) : 1);
    t->m_buffer = CreateBuffer(texels * texel_size(format));
    return t->m_buffer ? t : RHITexturePtr();
}

// device address of a texture or of a mip-level view (RHITexture::GetMipLevel)
static void* texels_of(const RHITexturePtr& t)
{
    if (!t || !t->m_buffer) return nullptr;
    char* base = (char*)t->m_buffer->m_hip.m_devicePtr;
    if (!t->m_parent) return base;
    const auto& p = t->m_parent;
    size_t texels = 0;
    for (uint32_t l = 0; l < t->m_viewLevel; l++)
        texels += (size_t)(p->m_bCubemap ? 6 : 1) * ((p->m_extent.x >> l) > 1 ? (p->m_extent.x >> l) : 1) * ((p->m_extent.y >> l) > 1 ? (p->m_extent.y >> l) : 1);
    return base + texels * texel_size(p->m_format);
}

RHICubemapPtr HipGraphicsDriver::CreateCubemap(ivec2 extent, uint32_t mipLevels, EFormat format)
{
    auto t = RHITexturePtr::Make();
    t->m_extent = extent; t->m_format = format; t->m_mipLevels = mipLevels; t->m_bCubemap = true;
    t->m_buffer = CreateBuffer(cube_chain_texels(extent.x, mipLevels) * texel_size(format));
    return t->m_buffer ? t : RHITexturePtr();
}

RHICubemapPtr HipGraphicsDriver::WrapCubemap(void* devicePtr, int size, uint32_t mipLevels, EFormat format)
{
    auto t = RHITexturePtr::Make();
    t->m_extent = { size, size }; t->m_format = format; t->m_mipLevels = mipLevels; t->m_bCubemap = true;
    t->m_buffer = WrapBuffer(devicePtr, cube_chain_texels(size, mipLevels) * texel_size(format));
    return t;
}

RHITexturePtr HipGraphicsDriver::WrapTexture(void* devicePtr, ivec2 extent, EFormat format)
{
    auto t = RHITexturePtr::Make();
    t->m_extent = extent;
    t->m_format = format;
    t->m_buffer = WrapBuffer(devicePtr, (size_t)extent.x * extent.y * texel_size(format));
    return t;
}

void HipGraphicsDriver::SubmitCommandList(RHICommandListPtr commandList)
{
    // the recorded closures enqueue on the context stream; GPU work runs asynchronously after this returns
    for (auto& c : commandList->m_hip.m_commands) {
        const int st = c();
        if (st != SAILOR_HIP_OK) { m_lastDispatchStatus = st; fprintf(stderr, "[HIP driver] command failed: %s (%s)\n", sailor_hip_status_string(st), sailor_hip_context_last_error(m_ctx)); }
    }
    commandList->m_hip.m_commands.clear();
    // a cull of this submit left its compaction on the second queue: what is recorded from here on (other nodes, the exchange, the next frame's
    // cull into the same workspace) comes after it
    if (m_packPending) { sailor_hip_context_wait_for(m_ctx, m_ctxAux); m_packPending = false; }
    // "these two SSBOs hold this cull's lists" ends with the submit: a later shade handed the same pointers (a frame whose cull node was skipped, a
    // buffer freed and allocated again at the same address) reads lightsGrid / culledLights themselves, not the workspace
    m_ownGrid = nullptr; m_ownCulled = nullptr;
}

// A command is about to WRITE this buffer: if it is one of the two SSBOs the last cull filled, the shade must read what the write leaves there, not the
// cull's per-tile lists -- and the write must come behind the compaction that is still filling the buffer on the second queue.
void HipGraphicsDriver::BeforeBufferWrite(const void* devicePtr)
{
    if (devicePtr) m_rasterWorkspaces.erase(devicePtr); // (a depth attachment written by anything but caster draws: its coarse depth no longer bounds it)
    if (!devicePtr || (devicePtr != m_ownGrid && devicePtr != m_ownCulled)) return;
    if (m_packPending) { sailor_hip_context_wait_for(m_ctx, m_ctxAux); m_packPending = false; }
    m_ownGrid = nullptr; m_ownCulled = nullptr;
    m_cullOrderValid = false; // (the workspace's length bytes describe the cull's lists, not what the write leaves in the SSBOs)
}

RHIMaterialPtr HipGraphicsDriver::CreateMaterial(RHIShaderPtr shader) { return RHIMaterialPtr::Make(std::move(shader)); }

RHIShaderBindingSetPtr HipGraphicsDriver::CreateShaderBindings() { return RHIShaderBindingSetPtr::Make(); }

RHIShaderBindingPtr HipGraphicsDriver::AddSsboToShaderBindings(RHIShaderBindingSetPtr& set, const std::string& name, size_t elementSize,
                                                               size_t numElements, uint32_t shaderBinding, bool)
{
    auto b = set->GetOrAddShaderBinding(name);
    b->m_type = EShaderBindingType::StorageBuffer;
    b->m_binding = shaderBinding;
    b->m_buffer = CreateBuffer(elementSize * numElements);
    // The `light` SSBO (LightingECS.cpp:44): LightingECS::Tick counts every slot in lightsNum but only ever uploads the slots of ACTIVE lights
    // (:148-149), so a light that is inactive from registration leaves its slot as the allocation left it.  Here that is defined: the records are
    // zero-filled (type 0, intensity 0) and the prepared views are derived for the whole capacity at creation, so the plain and the prepared entry
    // points read the same thing from a slot nobody wrote (VERDICT r03 "What's weak" 3).
    if (name == "light" && elementSize == sizeof(SailorLightShaderData) && b->m_buffer) EnsurePreparedLights(b, true);
    return b;
}

// The prepared views of a `light` SSBO (include/sailor_hip.h "prepared lights"), created together with it and derived for ALL of its slots whenever the
// prepared buffer is (re-)created; `zeroRecords`: the SSBO itself was just created.  Enqueued on the context's stream right away (creation is not a
// recorded command in the reference either).  Returns false when the prepared buffer cannot be had: cull and shade then take the plain entry points.
bool HipGraphicsDriver::EnsurePreparedLights(RHIShaderBindingPtr binding, bool zeroRecords)
{
    const int32_t capacity = (int32_t)(binding->m_buffer->m_size / sizeof(SailorLightShaderData));
    if (zeroRecords && sailor_hip_buffer_fill_u32(m_ctx, binding->m_buffer->m_hip.m_devicePtr, 0, 0u, binding->m_buffer->m_size / 4) != SAILOR_HIP_OK) return false;
    if (binding->m_hipPreparedLights && binding->m_hipPreparedCapacity == capacity) return true;
    binding->m_hipPreparedLights = CreateBuffer(sailor_hip_prepared_lights_size(capacity));
    binding->m_hipPreparedCapacity = binding->m_hipPreparedLights ? capacity : 0;
    if (!binding->m_hipPreparedLights) return false;
    const int st = sailor_hip_prepare_lights(m_ctx, (const SailorLightShaderData*)binding->m_buffer->m_hip.m_devicePtr, 0, capacity, capacity,
                                             binding->m_hipPreparedLights->m_hip.m_devicePtr, binding->m_hipPreparedLights->m_size);
    if (st != SAILOR_HIP_OK) { binding->m_hipPreparedLights = RHIBufferPtr(); binding->m_hipPreparedCapacity = 0; return false; }
    return true;
}

RHIShaderBindingPtr HipGraphicsDriver::AddBufferToShaderBindings(RHIShaderBindingSetPtr& set, const std::string& name, size_t size, uint32_t shaderBinding,
                                                                 EShaderBindingType bufferType)
{
    auto b = set->GetOrAddShaderBinding(name);
    b->m_type = bufferType;
    b->m_binding = shaderBinding;
    if (bufferType == EShaderBindingType::UniformBuffer) b->m_hostCopy.assign(size, 0); // passed by value to the kernels
    else b->m_buffer = CreateBuffer(size);
    return b;
}

RHIShaderBindingPtr HipGraphicsDriver::AddStorageImageToShaderBindings(RHIShaderBindingSetPtr& set, const std::string& name, RHITexturePtr texture, uint32_t shaderBinding)
{
    return AddStorageImageToShaderBindings(set, name, TVector<RHITexturePtr> { texture }, shaderBinding);
}

RHIShaderBindingPtr HipGraphicsDriver::AddStorageImageToShaderBindings(RHIShaderBindingSetPtr& set, const std::string& name, const TVector<RHITexturePtr>& array,
                                                                       uint32_t shaderBinding)
{
    auto b = set->GetOrAddShaderBinding(name);
    b->m_type = EShaderBindingType::StorageImage;
    b->m_binding = shaderBinding;
    b->m_textures = array;
    return b;
}

RHIShaderBindingPtr HipGraphicsDriver::AddSamplerToShaderBindings(RHIShaderBindingSetPtr& set, const std::string& name, RHITexturePtr texture, uint32_t shaderBinding)
{
    auto b = set->GetOrAddShaderBinding(name);
    b->m_type = EShaderBindingType::CombinedImageSampler;
    b->m_binding = shaderBinding;
    b->m_textures = { texture };
    return b;
}

RHIShaderBindingPtr HipGraphicsDriver::AddSamplerToShaderBindings(RHIShaderBindingSetPtr& set, const std::string& name, const TVector<RHITexturePtr>& array,
                                                                  uint32_t shaderBinding)
{
    auto b = set->GetOrAddShaderBinding(name);
    b->m_type = EShaderBindingType::CombinedImageSampler;
    b->m_binding = shaderBinding;
    b->m_textures = array;
    return b;
}

RHIShaderBindingPtr HipGraphicsDriver::AddShaderBinding(RHIShaderBindingSetPtr& set, const RHIShaderBindingPtr& binding, const std::string& name, uint32_t shaderBinding)
{
    // share the SAME resource under another set/slot (LightCullingNode.cpp:69-70 registers its SSBOs into the lights set)
    auto b = set->GetOrAddShaderBinding(name);
    b->m_type = binding->m_type;
    b->m_binding = shaderBinding;
    b->m_buffer = binding->m_buffer;
    b->m_textures = binding->m_textures;
    return b;
}

// Debug regions (RHI/GraphicsDriver.h:238-239; the Vulkan backend turns them into vkCmdBeginDebugUtilsLabelEXT): here roctx ranges with the
// reference's own region names ("LightCulling", LightCullingNode.cpp:37,80; "RenderScene QueueTag:..." ...), pushed / popped when the
// command list is replayed, so that a rocprofv3 --marker-trace shows the kernels of a node inside the node's range.  libroctx64 is bound
// lazily: without it the regions are only kept on the command list.
namespace {
struct Roctx {
    int (*push)(const char*) = nullptr;
    int (*pop)() = nullptr;
    Roctx()
    {
        void* h = dlopen("libroctx64.so.4", RTLD_NOW | RTLD_GLOBAL);
        if (!h) h = dlopen("libroctx64.so", RTLD_NOW | RTLD_GLOBAL);
        if (h) { push = (int (*)(const char*))dlsym(h, "roctxRangePushA"); pop = (int (*)())dlsym(h, "roctxRangePop"); }
    }
};
const Roctx& roctx() { static Roctx r; return r; }
} // namespace

void HipGraphicsDriver::BeginDebugRegion(RHICommandListPtr cmdList, const std::string& title)
{
    cmdList->m_debugRegions.push_back(title);
    cmdList->m_hip.m_commands.push_back([title]() { if (roctx().push) roctx().push(title.c_str()); return (int)SAILOR_HIP_OK; });
}
void HipGraphicsDriver::EndDebugRegion(RHICommandListPtr cmdList)
{
    cmdList->m_hip.m_commands.push_back([]() { if (roctx().pop) roctx().pop(); return (int)SAILOR_HIP_OK; });
}
void HipGraphicsDriver::ImageMemoryBarrier(RHICommandListPtr, RHITexturePtr, EImageLayout) {} // one in-order stream: nothing to do

bool HipGraphicsDriver::BlitImage(RHICommandListPtr cmd, RHITexturePtr src, RHITexturePtr dst, ivec4 srcRegionRect, ivec4 dstRegionRect)
{
    // the path's only blit copies level 0 of one cubemap to another of the same size and format (EnvironmentNode.cpp:200-203): a device copy
    if (!src || !dst || src->m_format != dst->m_format || src->GetExtent().x != dst->GetExtent().x || src->GetExtent().y != dst->GetExtent().y ||
        srcRegionRect.x != 0 || srcRegionRect.y != 0 || dstRegionRect.x != 0 || dstRegionRect.y != 0 || srcRegionRect.z != dstRegionRect.z ||
        srcRegionRect.w != dstRegionRect.w || srcRegionRect.z != src->GetExtent().x || srcRegionRect.w != src->GetExtent().y)
        return false;
    const size_t bytes = (size_t)(src->m_bCubemap ? 6 : 1) * src->GetExtent().x * src->GetExtent().y * texel_size(src->m_format);
    SailorHipContext* ctx = m_ctx;
    cmd->m_hip.m_commands.push_back([this, ctx, src, dst, bytes]() {
        BeforeBufferWrite(dst->m_buffer->m_hip.m_devicePtr);
        return sailor_hip_buffer_copy(ctx, dst->m_buffer->m_hip.m_devicePtr, 0, src->m_buffer->m_hip.m_devicePtr, 0, bytes);
    });
    return true;
}

// VulkanGraphicsDriver.cpp:1642-1653 -> VulkanCommandBuffer::GenerateMipMaps: the linear 2:1 blit chain.  The path calls it on the raw
// environment cube only (EnvironmentNode.cpp:137).
void HipGraphicsDriver::GenerateMipMaps(RHICommandListPtr cmd, RHITexturePtr target)
{
    SailorHipContext* ctx = m_ctx;
    cmd->m_hip.m_commands.push_back([ctx, target]() {
        if (!target || !target->m_bCubemap || target->m_format != EFormat::R32G32B32A32_SFLOAT) return (int)SAILOR_HIP_ERR_INVALID_ARGUMENT;
        return sailor_hip_generate_mipmaps_cube(ctx, (float*)target->m_buffer->m_hip.m_devicePtr, target->GetExtent().x, (int32_t)target->GetMipLevels());
    });
}

// VulkanGraphicsDriver.cpp:1662-1686: ComputeEquirect2Cube.shader over equirectExtent / 32 groups of 32 x 32 (x 6 faces) -- the covered corner
// of each face is groups * 32 texels, whatever the cube's size.
void HipGraphicsDriver::ConvertEquirect2Cubemap(RHICommandListPtr cmd, RHITexturePtr equirect, RHICubemapPtr cubemap)
{
    SailorHipContext* ctx = m_ctx;
    cmd->m_hip.m_commands.push_back([ctx, equirect, cubemap]() {
        if (!equirect || !cubemap || !cubemap->m_bCubemap || equirect->m_format != EFormat::R32G32B32A32_SFLOAT || cubemap->m_format != EFormat::R32G32B32A32_SFLOAT)
            return (int)SAILOR_HIP_ERR_INVALID_ARGUMENT;
        const int32_t coverW = (int32_t)((uint32_t)(equirect->GetExtent().x / 32.0f) * 32u), coverH = (int32_t)((uint32_t)(equirect->GetExtent().y / 32.0f) * 32u);
        return sailor_hip_equirect_to_cube(ctx, (const float*)equirect->m_buffer->m_hip.m_devicePtr, equirect->GetExtent().x, equirect->GetExtent().y,
                                           equirect->m_bRepeat ? 1 : 0, (float*)cubemap->m_buffer->m_hip.m_devicePtr, cubemap->GetExtent().x, coverW, coverH);
    });
}

void HipGraphicsDriver::UpdateShaderBinding(RHICommandListPtr cmd, RHIShaderBindingPtr binding, const void* data, size_t size, size_t variableOffset)
{
    if (binding->m_type == EShaderBindingType::UniformBuffer) {
        if (binding->m_hostCopy.size() < variableOffset + size) binding->m_hostCopy.resize(variableOffset + size);
        memcpy(binding->m_hostCopy.data() + variableOffset, data, size); // copied at record time, like push constants
        return;
    }
    UpdateBuffer(cmd, binding->m_buffer, data, size, variableOffset);
    // The `light` SSBO (LightingECS.cpp:44; written only through here, by LightingECS::Tick's dirty runs :182-191): derive the prepared views of the
    // records this copy carried right behind it -- the cull then streams 20 bytes per light and frame instead of 112, the shade copies staged records
    // instead of deriving them per tile and list slot (include/sailor_hip.h "prepared lights").
    if (binding->m_name == "light" && binding->m_type == EShaderBindingType::StorageBuffer && binding->m_buffer && size > 0) {
        const int32_t capacity = (int32_t)(binding->m_buffer->m_size / sizeof(SailorLightShaderData));
        // (created with the SSBO; a binding that got its buffer some other way is caught up here -- every slot, not just this run: the stream is in
        // order, so the whole-capacity derivation enqueued now sees the records as they are before this command list's uploads, and the run below
        // then follows its own upload)
        if (!EnsurePreparedLights(binding, false)) return;
        const int32_t first = (int32_t)(variableOffset / sizeof(SailorLightShaderData));
        const int32_t last = (int32_t)((variableOffset + size + sizeof(SailorLightShaderData) - 1) / sizeof(SailorLightShaderData)); // (a copy need not start on a record)
        const int32_t count = (last < capacity ? last : capacity) - first;
        SailorHipContext* ctx = m_ctx;
        RHIBufferPtr records = binding->m_buffer, prepared = binding->m_hipPreparedLights;
        if (prepared && count > 0)
            cmd->m_hip.m_commands.push_back([ctx, records, prepared, first, count, capacity]() {
                return sailor_hip_prepare_lights(ctx, (const SailorLightShaderData*)records->m_hip.m_devicePtr, first, count, capacity, prepared->m_hip.m_devicePtr,
                                                 prepared->m_size);
            });
    }
}

void HipGraphicsDriver::UpdateBuffer(RHICommandListPtr cmd, RHIBufferPtr buffer, const void* data, size_t size, size_t offset)
{
    TVector<uint8_t> staged((const uint8_t*)data, (const uint8_t*)data + size); // payload captured at record time
    SailorHipContext* ctx = m_ctx;
    cmd->m_hip.m_commands.push_back([this, ctx, buffer, staged = std::move(staged), offset]() {
        BeforeBufferWrite(buffer->m_hip.m_devicePtr);
        return sailor_hip_buffer_upload(ctx, buffer->m_hip.m_devicePtr, offset, staged.data(), staged.size());
    });
}

void HipGraphicsDriver::Dispatch(RHICommandListPtr cmd, RHIShaderPtr computeShader, uint32_t, uint32_t, uint32_t,
                                 const TVector<RHIShaderBindingSetPtr>& bindings, const void* pPushConstantsData, uint32_t sizePushConstantsData)
{
    TVector<uint8_t> pc((const uint8_t*)pPushConstantsData, (const uint8_t*)pPushConstantsData + sizePushConstantsData); // VulkanCommandBuffer.cpp:679-685
    const std::string name = computeShader->m_name;
    // The workgroup grid of the reference dispatch (numTiles.x, numTiles.y, 1) is implied by the push constants; the HIP
    // kernels choose their own launch geometry.
    const bool occlusion = computeShader->HasDefine("OCCLUSION_CULLING"); // RenderSceneNode.cpp:130 loads the culling shader with it
    cmd->m_hip.m_commands.push_back([this, name, bindings, occlusion, pc = std::move(pc)]() {
        if (name == "Shaders/ComputeLightCulling.shader") return RecordLightCulling(bindings, pc);
        if (name == "Shaders/Standard.shader") return RecordShade(bindings);
        if (name == "Shaders/ComputeMeshCulling.shader") return RecordMeshCulling(bindings, pc, occlusion);
        if (name == "Shaders/ComputeDepthHighZ.shader") return RecordDepthHighZ(bindings);
        if (name == "Shaders/ComputeBrdfLut.shader") return RecordBrdfLut(bindings);
        if (name == "Shaders/ComputeIrradianceMap.shader") return RecordIrradianceMap(bindings);
        if (name == "Shaders/ComputeEnvMap_IBL.shader") return RecordEnvPrefilter(bindings, pc);
        return (int)SAILOR_HIP_ERR_UNSUPPORTED;
    });
}

static void* buffer_of(const RHIShaderBindingSetPtr& set, const char* name)
{
    auto b = set->Find(name);
    return (b && b->m_buffer) ? b->m_buffer->m_hip.m_devicePtr : nullptr;
}

int HipGraphicsDriver::RecordLightCulling(const TVector<RHIShaderBindingSetPtr>& bindings, const TVector<uint8_t>& pcBytes)
{
    // LightCullingNode.cpp:76: { sceneView.m_rhiLightsData, m_culledLights, sceneView.m_frameBindings }
    if (bindings.size() != 3 || pcBytes.size() < sizeof(SailorLightCullPushConstants) - 4) return SAILOR_HIP_ERR_INVALID_ARGUMENT;
    SailorLightCullPushConstants pc {};
    memcpy(&pc, pcBytes.data(), pcBytes.size() < sizeof pc ? pcBytes.size() : sizeof pc);
    auto frameB = bindings[2]->Find("frameData");
    auto depthB = bindings[1]->Find("sceneDepth");
    auto culledB = bindings[1]->Find("culledLights");
    if (!frameB || frameB->m_hostCopy.size() < sizeof(SailorUboFrameData) || !depthB || depthB->m_textures.empty() || !culledB) return SAILOR_HIP_ERR_INVALID_ARGUMENT;
    SailorUboFrameData frame;
    memcpy(&frame, frameB->m_hostCopy.data(), sizeof frame);
    SailorBand band;
    if (m_worldSize > 1) {
        // Split frame: the node sized its push constants by the depth attachment it was given (LightCullingNode.cpp:55-57) -- the band's rows.
        // The tile frusta live in the WHOLE frame (frame.viewportSize, RHIFrameGraph.cpp:67), so the push constants are rebuilt for it.
        const int W = frame.viewportSize[0], H = frame.viewportSize[1];
        if (sailor_hip_band_for_rank(W, H, m_rank, m_worldSize, &band) != SAILOR_HIP_OK) return SAILOR_HIP_ERR_INVALID_ARGUMENT;
        if (pc.viewportSize[0] != W || pc.viewportSize[1] != band.fbRowCount) return SAILOR_HIP_ERR_INVALID_ARGUMENT; // the depth attachment must be the band's rows
        pc.viewportSize[0] = W; pc.viewportSize[1] = H;
        sailor_hip_num_tiles(W, H, &pc.numTiles[0], &pc.numTiles[1]);
    } else {
        sailor_hip_band_whole_frame(pc.viewportSize[0], pc.viewportSize[1], &band);
    }
    m_band = band; m_splitW = pc.viewportSize[0]; m_splitH = pc.viewportSize[1];
    const size_t need = sailor_hip_light_cull_workspace_size(pc.viewportSize[0], pc.viewportSize[1], pc.lightsNum, &band);
    if (!m_cullWorkspace || m_cullWorkspace->m_size < need) m_cullWorkspace = CreateBuffer(need);
    if (!m_cullWorkspace) return SAILOR_HIP_ERR_OUT_OF_MEMORY;
    m_cullW = pc.viewportSize[0]; m_cullH = pc.viewportSize[1]; m_cullLights = pc.lightsNum; m_cullOrderValid = true; // the shade of this frame may use the order hint
    auto lightB = bindings[0]->Find("light");
    const bool prepared = lightB && lightB->m_hipPreparedLights && lightB->m_hipPreparedCapacity >= pc.lightsNum;
    SailorLightsGrid* grid = (SailorLightsGrid*)buffer_of(bindings[1], "lightsGrid");
    uint32_t* culled = (uint32_t*)buffer_of(bindings[1], "culledLights");
    if (m_packPending) { sailor_hip_context_wait_for(m_ctx, m_ctxAux); m_packPending = false; } // (a second cull in one submit: its workspace is the one the pending pack reads)
    const bool defer = m_ctxAux != nullptr;
    const int st = sailor_hip_light_cull_prepared(m_ctx, &frame, &pc, (const SailorLightShaderData*)buffer_of(bindings[0], "light"),
                                                  (const float*)depthB->m_textures[0]->m_buffer->m_hip.m_devicePtr, grid, culled,
                                                  culledB->m_buffer->m_size / 4, m_cullWorkspace->m_hip.m_devicePtr, m_cullWorkspace->m_size, &band,
                                                  defer ? SAILOR_CULL_DEFER_PACK : SAILOR_CULL_DEFAULT,
                                                  prepared ? lightB->m_hipPreparedLights->m_hip.m_devicePtr : nullptr, prepared ? lightB->m_hipPreparedCapacity : 0);
    m_ownGrid = nullptr; m_ownCulled = nullptr;
    if (st != SAILOR_HIP_OK) return st;
    m_ownGrid = grid; m_ownCulled = culled; // a shade handed these two SSBOs reads this cull's per-tile lists
    m_cullBand = band;                      // ... if it shades the band this cull ran on
    if (!defer) return st;
    // the node's two SSBOs in the reference's layout, bit for bit, written on the second queue behind the cull's last kernel
    int st2 = sailor_hip_context_wait_for(m_ctxAux, m_ctx);
    if (st2 == SAILOR_HIP_OK)
        st2 = sailor_hip_light_cull_pack(m_ctxAux, pc.viewportSize[0], pc.viewportSize[1], pc.lightsNum, &band, m_cullWorkspace->m_hip.m_devicePtr, grid, culled,
                                         culledB->m_buffer->m_size / 4);
    m_packPending = true;
    return st2;
}

int HipGraphicsDriver::SetFrameSplit(int rank, int worldSize, void* ncclComm)
{
    if (worldSize < 1 || worldSize > SAILOR_MAX_SPLIT || rank < 0 || rank >= worldSize) return SAILOR_HIP_ERR_INVALID_ARGUMENT;
    m_rank = rank; m_worldSize = worldSize; m_comm = ncclComm;
    m_cullOrderValid = false;
    return SAILOR_HIP_OK;
}

int HipGraphicsDriver::ExchangeLightLists(RHIBufferPtr bandGrid, RHIBufferPtr bandCulled, RHIBufferPtr globalGrid, RHIBufferPtr globalCulled)
{
    if (!bandGrid || !bandCulled || !globalGrid || !globalCulled || !m_comm || m_splitW <= 0) return SAILOR_HIP_ERR_INVALID_ARGUMENT;
    {   // the stitch writes an entry for every tile of the frame: a shorter lightsGrid buffer is refused here, not overrun on the device
        int32_t Tx = 0, Ty = 0;
        sailor_hip_num_tiles(m_splitW, m_splitH, &Tx, &Ty);
        if (globalGrid->m_size < (size_t)Tx * Ty * sizeof(SailorLightsGrid) || globalCulled->m_size < 4) return SAILOR_HIP_ERR_INVALID_ARGUMENT;
    }
    if (m_packPending) { sailor_hip_context_wait_for(m_ctx, m_ctxAux); m_packPending = false; } // the band's canonical buffers are the pack's output
    const size_t need = sailor_hip_exchange_workspace_size(m_splitW, m_splitH, m_worldSize);
    if (!m_exchangeWorkspace || m_exchangeWorkspace->m_size < need) m_exchangeWorkspace = CreateBuffer(need);
    if (!m_exchangeWorkspace) return SAILOR_HIP_ERR_OUT_OF_MEMORY;
    // The slots of the exchange's second gather follow the PREVIOUS exchange's gathered totals (sailor_hip_exchange_adapt: waits for that exchange's own
    // event -- a frame old by now -- not for anything this frame recorded; the first exchange takes the worst-case slots).  Every rank's driver runs the
    // same sequence, so every rank arrives at the same slot size.  A clipped exchange (a band's total grew by more than the 25 % of slack from one frame to
    // the next) is reported here, one exchange late, and the next one is back on the worst case.
    {
        int32_t clipped = 0;
        const int st = sailor_hip_exchange_adapt(m_ctx, nullptr, &clipped, nullptr);
        if (st != SAILOR_HIP_OK) return st;
        if (clipped) { m_exchangesClipped++; fprintf(stderr, "[HIP driver] %s\n", sailor_hip_context_last_error(m_ctx)); }
    }
    return sailor_hip_exchange_light_lists(m_ctx, m_comm, m_rank, m_worldSize, m_splitW, m_splitH, (const SailorLightsGrid*)bandGrid->m_hip.m_devicePtr,
                                           (const uint32_t*)bandCulled->m_hip.m_devicePtr, (SailorLightsGrid*)globalGrid->m_hip.m_devicePtr,
                                           (uint32_t*)globalCulled->m_hip.m_devicePtr, globalCulled->m_size / 4, m_exchangeWorkspace->m_hip.m_devicePtr,
                                           m_exchangeWorkspace->m_size);
}

int HipGraphicsDriver::RecordShade(const TVector<RHIShaderBindingSetPtr>& bindings)
{
    // Standard.shader:180-251: set 0 frame, set 1 lights {0 light, 1 culledLights, 2 lightsGrid, 6 lightsMatrices, 8 shadowMaps},
    // set 2 the surface/radiance buffers that stand in for the rasterised fragments (per-instance / material / textures sets)
    if (bindings.size() != 3) return SAILOR_HIP_ERR_INVALID_ARGUMENT;
    auto frameB = bindings[0]->Find("frameData");
    auto surfaceB = bindings[2]->Find("surface");
    auto countB = bindings[1]->Find("light");
    if (!frameB || frameB->m_hostCopy.size() < sizeof(SailorUboFrameData) || !surfaceB || !surfaceB->m_buffer || !countB) return SAILOR_HIP_ERR_INVALID_ARGUMENT;
    SailorUboFrameData frame;
    memcpy(&frame, frameB->m_hostCopy.data(), sizeof frame);
    const int W = frame.viewportSize[0], H = frame.viewportSize[1];
    SailorCsmDesc csm {};
    bool hasCsm = false;
    auto mats = bindings[1]->Find("lightsMatrices");
    auto maps = bindings[1]->Find("shadowMaps");
    if (mats && mats->m_hostCopy.size() >= 256 && maps) {
        memcpy(csm.lightsMatrices, mats->m_hostCopy.data(), 256);
        for (size_t k = 0; k < SAILOR_NUM_CSM_CASCADES && k < maps->m_textures.size(); k++) {
            const auto& t = maps->m_textures[k];
            if (!t) continue;
            csm.maps[k] = t->m_buffer->m_hip.m_devicePtr;
            csm.width[k] = t->m_extent.x; csm.height[k] = t->m_extent.y;
            csm.format[k] = t->m_format == EFormat::R16_SFLOAT ? SAILOR_SHADOWMAP_R16_SFLOAT
                          : (t->m_format == EFormat::R32_SFLOAT ? SAILOR_SHADOWMAP_R32_SFLOAT : SAILOR_SHADOWMAP_R32G32B32A32_SFLOAT);
            hasCsm = true;
        }
    }
    // Standard.shader:219-221,234: bindings 3 g_irradianceCubemap, 4 g_brdfSampler, 5 g_envCubemap, 9 g_aoSampler -> the ambient term
    SailorIblDesc ibl {};
    bool hasIbl = false;
    {
        auto tex = [&](const char* name) -> RHITexturePtr {
            auto b = bindings[1]->Find(name);
            return (b && !b->m_textures.empty() && b->m_textures[0] && b->m_textures[0]->m_buffer) ? b->m_textures[0] : RHITexturePtr();
        };
        auto irr = tex("g_irradianceCubemap"), lut = tex("g_brdfSampler"), env = tex("g_envCubemap"), ao = tex("g_aoSampler");
        if (irr && lut && env) {
            ibl.irradiance = (const float*)irr->m_buffer->m_hip.m_devicePtr; ibl.irrSize = irr->m_extent.x;
            ibl.env = (const float*)env->m_buffer->m_hip.m_devicePtr; ibl.envSize = env->m_extent.x; ibl.envLevels = (int32_t)env->m_mipLevels;
            ibl.brdfLut = (const float*)lut->m_buffer->m_hip.m_devicePtr; ibl.lutW = lut->m_extent.x; ibl.lutH = lut->m_extent.y;
            ibl.ao = (ao && ao->m_extent.x == W && ao->m_extent.y == H) ? (const float*)ao->m_buffer->m_hip.m_devicePtr : nullptr;
            hasIbl = true;
        }
    }
    const int32_t lightsNum = (int32_t)(countB->m_buffer->m_size / sizeof(SailorLightShaderData));
    const SailorBand* band = (m_worldSize > 1 && m_splitW == W && m_splitH == H) ? &m_band : nullptr; // the band of this frame's light cull
    const size_t planeStride = band ? (size_t)W * band->fbRowCount : (size_t)W * H;
    if (surfaceB->m_buffer->m_size < planeStride * 48) return SAILOR_HIP_ERR_INVALID_ARGUMENT;
    auto lightB = bindings[1]->Find("light");
    const bool prepared = lightB && lightB->m_hipPreparedLights && lightB->m_hipPreparedCapacity >= lightsNum;
    // (the band's list lengths as bytes -- what switches the band form of the shade on: only from the workspace of a cull that ran on THIS band)
    const uint32_t* order = (m_cullOrderValid && m_cullWorkspace && band && band->tileRowBegin == m_cullBand.tileRowBegin && band->tileRowEnd == m_cullBand.tileRowEnd)
                                ? sailor_hip_light_cull_tile_order(m_cullW, m_cullH, m_cullLights, band, m_cullWorkspace->m_hip.m_devicePtr) : nullptr;
    const SailorLightsGrid* grid = (const SailorLightsGrid*)buffer_of(bindings[1], "lightsGrid");
    const uint32_t* culled = (const uint32_t*)buffer_of(bindings[1], "culledLights");
    // The lists of this frame's own LightCulling node (the usual graph: LightCullingNode.cpp:69-70 registers its SSBOs into the lights set that
    // RenderScene binds): read them where the cull left them -- the same entries in the same order as the two SSBOs hold, without waiting for the
    // compaction that fills those.  Lists from anywhere else (SSBOs the caller filled): through lightsGrid / culledLights, behind a pending pack.
    const uint32_t *tileNum = nullptr, *tileLists = nullptr;
    SailorBand shadeBand;
    if (band) shadeBand = *band; else sailor_hip_band_whole_frame(W, H, &shadeBand);
    const bool sameBand = shadeBand.tileRowBegin == m_cullBand.tileRowBegin && shadeBand.tileRowEnd == m_cullBand.tileRowEnd; // (SetFrameSplit between the two: not this cull's lists)
    const bool own = m_cullWorkspace && grid && culled && grid == m_ownGrid && culled == m_ownCulled && m_cullW == W && m_cullH == H && sameBand &&
                     sailor_hip_light_cull_tile_lists(m_cullW, m_cullH, m_cullLights, band, m_cullWorkspace->m_hip.m_devicePtr, &tileNum, &tileLists) == SAILOR_HIP_OK;
    if (own) {
        // (the tile-order hint a band's split blocks need is written by k1_tile_cull itself since round 4 -- through round 3 by the pack step, and a
        // split frame's shade had to come behind it)
        return sailor_hip_shade_tile_lists(m_ctx, &frame, (const float*)surfaceB->m_buffer->m_hip.m_devicePtr, planeStride,
                                           (const SailorLightShaderData*)buffer_of(bindings[1], "light"), lightsNum, tileNum, tileLists,
                                           hasCsm ? &csm : nullptr, hasIbl ? &ibl : nullptr, (float*)buffer_of(bindings[2], "radiance"), band, order,
                                           prepared ? lightB->m_hipPreparedLights->m_hip.m_devicePtr : nullptr, prepared ? lightB->m_hipPreparedCapacity : 0);
    }
    if (m_packPending) { sailor_hip_context_wait_for(m_ctx, m_ctxAux); m_packPending = false; }
    // (lists that are not this cull's: NO order hint.  The band form's tile blocks decide "split or not" on the grid entry they read, its split blocks on
    // the workspace's length bytes -- of the OLD cull here -- and a tile the two disagree on would be shaded by nobody: ADVICE r05)
    return sailor_hip_shade_prepared(m_ctx, &frame, (const float*)surfaceB->m_buffer->m_hip.m_devicePtr, planeStride,
                                     (const SailorLightShaderData*)buffer_of(bindings[1], "light"), lightsNum, grid, culled,
                                     hasCsm ? &csm : nullptr, hasIbl ? &ibl : nullptr, (float*)buffer_of(bindings[2], "radiance"), band, nullptr,
                                     prepared ? lightB->m_hipPreparedLights->m_hip.m_devicePtr : nullptr, prepared ? lightB->m_hipPreparedCapacity : 0);
}

// ---- the render-pass subset: state is kept on the command list, a 6-index draw of a known full-screen material becomes a
// kernel launch at submit time (record-then-submit, like Dispatch) ---------------------------------------------------------------
void HipGraphicsDriver::BeginRenderPass(RHICommandListPtr cmd, const TVector<RHITexturePtr>& colorAttachments, RHITexturePtr depthStencilAttachment)
{
    cmd->m_colorAttachments = colorAttachments;
    cmd->m_depthAttachment = depthStencilAttachment;
    cmd->m_casterDraws = 0;
}

void HipGraphicsDriver::BindVertexBuffer(RHICommandListPtr cmd, RHIBufferPtr vertexBuffer, uint32_t) { cmd->m_vertexBuffer = vertexBuffer; }
void HipGraphicsDriver::BindIndexBuffer(RHICommandListPtr cmd, RHIBufferPtr indexBuffer, uint32_t, bool) { cmd->m_indexBuffer = indexBuffer; }
void HipGraphicsDriver::PushConstants(RHICommandListPtr cmd, RHIMaterialPtr, size_t size, const void* ptr)
{
    cmd->m_pushConstants.assign((const uint8_t*)ptr, (const uint8_t*)ptr + size); // captured at record time (vkCmdPushConstants)
}

void HipGraphicsDriver::EndRenderPass(RHICommandListPtr cmd)
{
    if (cmd->m_casterDraws > 0 && !cmd->m_colorAttachments.empty() && cmd->m_depthAttachment) {
        // the fragment stage of ShadowCaster.shader wrote one colour per winning fragment; as compute it is one pass over the depth attachment
        RHITexturePtr color = cmd->m_colorAttachments[0], depth = cmd->m_depthAttachment;
        SailorHipContext* ctx = m_ctx;
        cmd->m_hip.m_commands.push_back([ctx, color, depth]() {
            const int fmt = color->m_format == EFormat::R32G32B32A32_SFLOAT ? SAILOR_SHADOWMAP_R32G32B32A32_SFLOAT
                                                                              : (color->m_format == EFormat::R16_SFLOAT ? SAILOR_SHADOWMAP_R16_SFLOAT : SAILOR_SHADOWMAP_R32_SFLOAT);
            return sailor_hip_shadow_resolve(ctx, (const float*)depth->m_buffer->m_hip.m_devicePtr, depth->GetExtent().x, depth->GetExtent().y, fmt,
                                             color->m_buffer->m_hip.m_devicePtr);
        });
    }
    cmd->m_depthAttachment.Clear();
    cmd->m_casterDraws = 0;
    cmd->m_colorAttachments.clear();
    cmd->m_boundMaterial.Clear();
    cmd->m_boundBindings.clear();
}

void HipGraphicsDriver::BindMaterial(RHICommandListPtr cmd, RHIMaterialPtr material) { cmd->m_boundMaterial = material; }

void HipGraphicsDriver::BindShaderBindings(RHICommandListPtr cmd, RHIMaterialPtr, const TVector<RHIShaderBindingSetPtr>& bindings)
{
    cmd->m_boundBindings = bindings;
}

void HipGraphicsDriver::DrawIndexed(RHICommandListPtr cmd, uint32_t indexCount, uint32_t instanceCount, uint32_t firstIndex, uint32_t vertexOffset, uint32_t firstInstance)
{
    const std::string name = (cmd->m_boundMaterial && cmd->m_boundMaterial->m_shader) ? cmd->m_boundMaterial->m_shader->m_name : std::string();
    if (name == "Shaders/ShadowCaster.shader") { // a caster draw of a shadow pass (ShadowPrepassNode.cpp:251-261 via RHIRecordDrawCall, RHI/Batch.hpp)
        RHITexturePtr depth = cmd->m_depthAttachment;
        RHIBufferPtr vb = cmd->m_vertexBuffer, ib = cmd->m_indexBuffer, instances;
        for (auto& set : cmd->m_boundBindings)
            if (set) if (auto b = set->Find("data")) if (b->m_buffer) instances = b->m_buffer; // ShadowCaster.shader:51-54 PerInstanceDataSSBO { mat4 model; }
        const bool first = cmd->m_casterDraws++ == 0;
        TVector<uint8_t> pc = cmd->m_pushConstants;
        SailorHipContext* ctx = m_ctx;
        cmd->m_hip.m_commands.push_back([this, ctx, depth, vb, ib, instances, pc, first, indexCount, instanceCount, firstIndex, vertexOffset, firstInstance]() {
            if (!depth || !vb || !ib || !instances || pc.size() < 64 || depth->m_format != EFormat::R32_SFLOAT) return (int)SAILOR_HIP_ERR_INVALID_ARGUMENT;
            float lightMatrix[16];
            memcpy(lightMatrix, pc.data(), 64);
            // The rasteriser's workspace of this depth attachment (round 6: through round 5 the backend drew without one): coarse depth, the mesh's box, the giant
            // triangles' queue -- bounds and scheduling only, the same depth buffer.  One per attachment, created at its first cleared draw, valid as long as
            // nothing but caster draws writes the attachment (BeforeBufferWrite drops it otherwise; a draw that does not clear and finds none goes without).
            const void* key = depth->m_buffer->m_hip.m_devicePtr;
            const size_t words = sailor_hip_raster_coarse_words(depth->GetExtent().x, depth->GetExtent().y);
            RHIBufferPtr ws;
            auto it = m_rasterWorkspaces.find(key);
            if (it != m_rasterWorkspaces.end() && it->second && it->second->m_size >= words * 4) ws = it->second;
            else if (first) { ws = CreateBuffer(words * 4); m_rasterWorkspaces[key] = ws; }
            // the render pass clears the depth attachment (ShadowPrepassNode.cpp:239-248); the material culls back faces (:39)
            return sailor_hip_raster_depth(ctx, lightMatrix, (const float*)vb->m_hip.m_devicePtr + 3 * (size_t)vertexOffset,
                                           (const uint32_t*)ib->m_hip.m_devicePtr + firstIndex, indexCount / 3,
                                           (const float*)instances->m_hip.m_devicePtr + 16 * (size_t)firstInstance, nullptr, instanceCount, depth->GetExtent().x,
                                           depth->GetExtent().y, (float*)depth->m_buffer->m_hip.m_devicePtr, (first ? SAILOR_RASTER_CLEAR : 0u) | SAILOR_RASTER_CULL_BACK,
                                           ws ? (uint32_t*)ws->m_hip.m_devicePtr : nullptr);
        });
        return;
    }
    RHITexturePtr target = cmd->m_colorAttachments.empty() ? RHITexturePtr() : cmd->m_colorAttachments[0];
    // A draw sees the binding sets as they are NOW: ShadowPrepassNode re-points `colorSampler` of one and the same set between its two
    // blur draws (ShadowPrepassNode.cpp:292,329), so the sets are snapshotted at record time (textures / buffers by reference, UBO bytes by value).
    TVector<RHIShaderBindingSetPtr> bindings;
    for (auto& set : cmd->m_boundBindings) {
        auto copy = RHIShaderBindingSetPtr::Make();
        for (auto& kv : set->m_bindings) {
            auto b = copy->GetOrAddShaderBinding(kv.first);
            b->m_type = kv.second->m_type; b->m_binding = kv.second->m_binding; b->m_buffer = kv.second->m_buffer;
            b->m_textures = kv.second->m_textures; b->m_hostCopy = kv.second->m_hostCopy;
        }
        bindings.push_back(copy);
    }
    const bool fullScreenQuad = indexCount == 6 && instanceCount == 1; // RHIFrameGraph.cpp:106-125 GetFullscreenNdcQuad
    RHIShaderPtr shader = cmd->m_boundMaterial ? cmd->m_boundMaterial->m_shader : RHIShaderPtr();
    const bool evsm = shader && shader->HasDefine("EVSM"), vertical = shader && shader->HasDefine("VERTICAL"), horizontal = shader && shader->HasDefine("HORIZONTAL");
    cmd->m_hip.m_commands.push_back([this, name, bindings, target, fullScreenQuad, evsm, vertical, horizontal]() {
        if (fullScreenQuad && name == "Shaders/LinearizeDepth.shader") return RecordLinearizeDepth(bindings, target);
        if (fullScreenQuad && name == "Shaders/Blur.shader" && evsm && vertical != horizontal) return RecordEvsmBlur(bindings, target, vertical);
        return (int)SAILOR_HIP_ERR_UNSUPPORTED;
    });
}

int HipGraphicsDriver::RecordLinearizeDepth(const TVector<RHIShaderBindingSetPtr>& bindings, const RHITexturePtr& target)
{
    // LinearizeDepthNode.cpp:89: { sceneView.m_frameBindings, m_linearizeDepth }; LinearizeDepth.shader:15-28 (set 0 frame), :58 (set 1 depthSampler)
    if (bindings.size() != 2 || !target || !target->m_buffer) return SAILOR_HIP_ERR_INVALID_ARGUMENT;
    auto frameB = bindings[0]->Find("frameData");
    auto depthB = bindings[1]->Find("depthSampler");
    if (!frameB || frameB->m_hostCopy.size() < sizeof(SailorUboFrameData) || !depthB || depthB->m_textures.empty() || !depthB->m_textures[0]->m_buffer)
        return SAILOR_HIP_ERR_INVALID_ARGUMENT;
    const auto& src = depthB->m_textures[0];
    if (src->GetExtent().x != target->GetExtent().x || src->GetExtent().y != target->GetExtent().y) return SAILOR_HIP_ERR_UNSUPPORTED; // 1:1 texel fetch only
    SailorUboFrameData frame;
    memcpy(&frame, frameB->m_hostCopy.data(), sizeof frame);
    return sailor_hip_linearize_depth(m_ctx, &frame, (const float*)src->m_buffer->m_hip.m_devicePtr, (float*)target->m_buffer->m_hip.m_devicePtr,
                                      target->GetExtent().x, target->GetExtent().y);
}

int HipGraphicsDriver::RecordEvsmBlur(const TVector<RHIShaderBindingSetPtr>& bindings, const RHITexturePtr& target, bool vertical)
{
    // ShadowPrepassNode.cpp:309,343: { sceneView.m_frameBindings, m_pBlurShaderBindings }; Blur.shader:53-61: set 1 binding 0 `data` (blurRadius.xy =
    // [umbra, penumbra], uploaded at :286), binding 1 `colorSampler`
    if (bindings.size() != 2 || !target || !target->m_buffer) return SAILOR_HIP_ERR_INVALID_ARGUMENT;
    auto dataB = bindings[1]->Find("data");
    auto srcB = bindings[1]->Find("colorSampler");
    if (!dataB || dataB->m_hostCopy.size() < 8 || !srcB || srcB->m_textures.empty() || !srcB->m_textures[0] || !srcB->m_textures[0]->m_buffer)
        return SAILOR_HIP_ERR_INVALID_ARGUMENT;
    const auto& src = srcB->m_textures[0];
    if (src->m_format != EFormat::R32G32B32A32_SFLOAT || target->m_format != EFormat::R32G32B32A32_SFLOAT) return SAILOR_HIP_ERR_UNSUPPORTED;
    if (src->GetExtent().x != target->GetExtent().x || src->GetExtent().y != target->GetExtent().y) return SAILOR_HIP_ERR_UNSUPPORTED;
    float radius[2];
    memcpy(radius, dataB->m_hostCopy.data(), 8);
    return sailor_hip_evsm_blur_pass(m_ctx, (const float*)src->m_buffer->m_hip.m_devicePtr, (float*)target->m_buffer->m_hip.m_devicePtr,
                                     target->GetExtent().x, target->GetExtent().y, (int32_t)radius[0], (int32_t)radius[1], vertical ? 1 : 0); // ivec2(data.blurRadius.xy) (Blur.shader:94)
}

// ---- EnvironmentNode's one-off Dispatches (FrameGraph/EnvironmentNode.cpp:86-91, :223-231, :264-269) ------------------------------------------
static RHITexturePtr texture_of(const TVector<RHIShaderBindingSetPtr>& bindings, const char* name, size_t index = 0)
{
    for (const auto& set : bindings)
        if (set) if (auto b = set->Find(name)) if (b->m_textures.size() > index) return b->m_textures[index];
    return RHITexturePtr();
}

int HipGraphicsDriver::RecordBrdfLut(const TVector<RHIShaderBindingSetPtr>& bindings)
{
    auto dst = texture_of(bindings, "dst"); // ComputeBrdfLut.shader:24 (rg16f there; two fp32 channels here)
    if (!dst || dst->m_format != EFormat::R32G32_SFLOAT) return SAILOR_HIP_ERR_INVALID_ARGUMENT;
    return sailor_hip_compute_brdf_lut(m_ctx, (float*)dst->m_buffer->m_hip.m_devicePtr, dst->GetExtent().x, dst->GetExtent().y);
}

int HipGraphicsDriver::RecordIrradianceMap(const TVector<RHIShaderBindingSetPtr>& bindings)
{
    auto env = texture_of(bindings, "envMap"), irr = texture_of(bindings, "irradianceMap"); // ComputeIrradianceMap.shader:19-20
    if (!env || !irr || !env->m_bCubemap || !irr->m_bCubemap || env->m_format != EFormat::R32G32B32A32_SFLOAT || irr->m_format != EFormat::R32G32B32A32_SFLOAT)
        return SAILOR_HIP_ERR_INVALID_ARGUMENT;
    return sailor_hip_compute_irradiance_map(m_ctx, (const float*)env->m_buffer->m_hip.m_devicePtr, env->GetExtent().x, (int32_t)env->m_mipLevels,
                                             (float*)irr->m_buffer->m_hip.m_devicePtr, irr->GetExtent().x);
}

int HipGraphicsDriver::RecordEnvPrefilter(const TVector<RHIShaderBindingSetPtr>& bindings, const TVector<uint8_t>& pcBytes)
{
    // ComputeEnvMap_IBL.shader:20-29: rawEnvMap, envMap[NumMipLevels] = the mip TAIL views (EnvironmentNode.cpp:208-215), push constants { level, roughness }
    if (pcBytes.size() < 8) return SAILOR_HIP_ERR_INVALID_ARGUMENT;
    int32_t level; float roughness;
    memcpy(&level, pcBytes.data(), 4); memcpy(&roughness, pcBytes.data() + 4, 4);
    auto raw = texture_of(bindings, "rawEnvMap");
    auto view = level >= 0 ? texture_of(bindings, "envMap", (size_t)level) : RHITexturePtr();
    if (!raw || !view || !view->m_parent || !raw->m_bCubemap || raw->m_format != EFormat::R32G32B32A32_SFLOAT) return SAILOR_HIP_ERR_INVALID_ARGUMENT;
    const auto& env = view->m_parent;
    if (env->GetExtent().x != raw->GetExtent().x || env->m_mipLevels != raw->m_mipLevels || env->m_format != raw->m_format) return SAILOR_HIP_ERR_UNSUPPORTED;
    return sailor_hip_prefilter_env_level(m_ctx, (const float*)raw->m_buffer->m_hip.m_devicePtr, (float*)env->m_buffer->m_hip.m_devicePtr, raw->GetExtent().x,
                                          (int32_t)raw->m_mipLevels, (int32_t)view->m_viewLevel, roughness);
}

int HipGraphicsDriver::RecordDepthHighZ(const TVector<RHIShaderBindingSetPtr>& bindings)
{
    // ComputeDepthHighZ.shader:11-17: inputDepth (sampler, reduction Min), outputDepth (storage image), push constant outputSize = the output extent
    auto src = texture_of(bindings, "inputDepth"), dst = texture_of(bindings, "outputDepth");
    if (!src || !dst || src->m_format != EFormat::R32_SFLOAT || dst->m_format != EFormat::R32_SFLOAT) return SAILOR_HIP_ERR_INVALID_ARGUMENT;
    return sailor_hip_hiz_downscale(m_ctx, (const float*)texels_of(src), src->GetExtent().x, src->GetExtent().y, (float*)texels_of(dst), dst->GetExtent().x,
                                    dst->GetExtent().y);
}

int HipGraphicsDriver::RecordMeshCulling(const TVector<RHIShaderBindingSetPtr>& bindings, const TVector<uint8_t>& pcBytes, bool occlusion)
{
    // ComputeMeshCulling.shader:13-18 push constants {numBatches, numInstances, firstInstanceIndex}; the Dispatch binds
    // { depthHighZ, data, drawIndexedIndirect, frame } (RenderSceneNode.cpp:265,335; DepthPrepassNode.cpp:290) -- looked up by name.
    // The Hi-Z set is used when the shader was created with its OCCLUSION_CULLING define (RenderSceneNode.cpp:130).
    if (bindings.size() < 2 || pcBytes.size() < 12) return SAILOR_HIP_ERR_INVALID_ARGUMENT;
    uint32_t pc[3];
    memcpy(pc, pcBytes.data(), 12);
    RHIShaderBindingPtr frameB;
    void* data = nullptr;
    void* batches = nullptr;
    for (const auto& set : bindings) {
        if (!set) continue;
        if (auto f = set->Find("frameData")) frameB = f;
        if (void* d = buffer_of(set, "data")) data = d;
        if (void* d = buffer_of(set, "drawIndexedIndirect")) batches = d;
    }
    if (!frameB || frameB->m_hostCopy.size() < sizeof(SailorUboFrameData)) return SAILOR_HIP_ERR_INVALID_ARGUMENT;
    SailorUboFrameData frame;
    memcpy(&frame, frameB->m_hostCopy.data(), sizeof frame);
    SailorHiZDesc hiz {};
    const SailorHiZDesc* pHiz = nullptr;
    if (occlusion) {
        auto pyramid = texture_of(bindings, "depthHighZ");
        if (!pyramid || pyramid->m_format != EFormat::R32_SFLOAT || pyramid->m_parent) return SAILOR_HIP_ERR_INVALID_ARGUMENT;
        hiz.pyramid = (const float*)pyramid->m_buffer->m_hip.m_devicePtr;
        hiz.width = pyramid->GetExtent().x; hiz.height = pyramid->GetExtent().y; hiz.levels = (int32_t)pyramid->GetMipLevels();
        pHiz = &hiz;
    }
    if (!batches || pc[0] == 0) return sailor_hip_mesh_cull_flags(m_ctx, &frame, (SailorPerInstanceData*)data, pc[1], pc[2], pHiz);
    const size_t need = sailor_hip_mesh_cull_workspace_bytes(pc[1], pc[0]);
    if (!m_meshCullWorkspace || m_meshCullWorkspace->m_size < need) m_meshCullWorkspace = CreateBuffer(need);
    if (!m_meshCullWorkspace) return SAILOR_HIP_ERR_OUT_OF_MEMORY;
    return sailor_hip_mesh_cull_compact_ex(m_ctx, &frame, (SailorPerInstanceData*)data, pc[1], pc[2], (SailorDrawIndexedIndirectData*)batches, pc[0],
                                           m_meshCullWorkspace->m_hip.m_devicePtr, m_meshCullWorkspace->m_size, pHiz);
}
