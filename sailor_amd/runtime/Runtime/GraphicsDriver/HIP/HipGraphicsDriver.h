// Runtime/GraphicsDriver/HIP -- the new RHI backend, sibling of Runtime/GraphicsDriver/Vulkan.
// Like `class VulkanGraphicsDriver : public IGraphicsDriver, public IGraphicsDriverCommands`
// (GraphicsDriver/Vulkan/VulkanGraphicsDriver.h:22) one object implements both interfaces.  It owns a SailorHipContext and
// translates recorded Dispatches of the path's shaders into calls of the C-ABI (include/sailor_hip.h):
//   "Shaders/ComputeLightCulling.shader" -> sailor_hip_light_cull        (binding contract: ComputeLightCulling.shader:20-47)
//   "Shaders/Standard.shader"            -> sailor_hip_shade             (binding contract: Standard.shader:180-251)
//   "Shaders/ComputeMeshCulling.shader"  -> sailor_hip_mesh_cull_compact (binding contract: ComputeMeshCulling.shader:37-58;
//                                           sailor_hip_mesh_frustum_cull when no indirect buffer is bound)
//   "Shaders/ComputeDepthHighZ.shader"   -> sailor_hip_hiz_downscale     (binding contract: ComputeDepthHighZ.shader:11-17)
//   "Shaders/ComputeBrdfLut.shader" / "ComputeIrradianceMap.shader" / "ComputeEnvMap_IBL.shader" (EnvironmentNode's one-off Dispatches)
//                                        -> sailor_hip_compute_brdf_lut / _compute_irradiance_map / _prefilter_env_level
// and the one full-screen DRAW in front of the path (6 indices with the material of)
//   "Shaders/LinearizeDepth.shader"      -> sailor_hip_linearize_depth   (binding contract: LinearizeDepth.shader:15-59)
//   "Shaders/Blur.shader" {EVSM, HORIZONTAL | VERTICAL} -> sailor_hip_evsm_blur_pass (binding contract: Blur.shader:53-61)
// and the depth-only instanced draws of the shadow passes (material of)
//   "Shaders/ShadowCaster.shader" [EVSM]  -> sailor_hip_raster_depth into the pass' depth attachment, and at EndRenderPass sailor_hip_shadow_resolve
//                                            into its colour attachment (push constant lightMatrix, set 1 `data`, vertex positions, 32-bit indices)
#pragma once
#include <map>
#include <memory>
#include "../../RHI/GraphicsDriver.h"

namespace Sailor::GraphicsDriver::HIP {

class HipGraphicsDriver : public RHI::IGraphicsDriver, public RHI::IGraphicsDriverCommands {
public:
    HipGraphicsDriver(int deviceOrdinal, void* stream, bool ownStream);
    ~HipGraphicsDriver() override;
    int GetStatus() const { return m_status; }
    SailorHipContext* GetContext() const { return m_ctx; }
    int GetLastDispatchStatus() const { return m_lastDispatchStatus; }
    // the prepared views of a `light` SSBO: created with it, derived for every slot on (re-)creation (HipGraphicsDriver.cpp)
    bool EnsurePreparedLights(RHI::RHIShaderBindingPtr binding, bool zeroRecords);

    // IGraphicsDriver
    void WaitIdle() override;
    RHI::RHICommandListPtr CreateCommandList(bool bIsSecondary = false) override;
    RHI::RHIBufferPtr CreateBuffer(size_t size) override;
    RHI::RHIShaderPtr CreateShader(const std::string& assetPath, const TVector<std::string>& defines = {}) override;
    RHI::RHITexturePtr CreateTexture(const void* pData, size_t size, RHI::ivec2 extent, RHI::EFormat format) override;
    RHI::RHITexturePtr CreateRenderTarget(RHI::ivec2 extent, uint32_t mipLevels, RHI::EFormat format) override;
    RHI::RHICubemapPtr CreateCubemap(RHI::ivec2 extent, uint32_t mipLevels, RHI::EFormat format) override;
    void SubmitCommandList(RHI::RHICommandListPtr commandList) override;
    RHI::RHIMaterialPtr CreateMaterial(RHI::RHIShaderPtr shader) override;
    RHI::RHIShaderBindingSetPtr CreateShaderBindings() override;
    RHI::RHIShaderBindingPtr AddSsboToShaderBindings(RHI::RHIShaderBindingSetPtr& set, const std::string& name, size_t elementSize, size_t numElements,
                                                     uint32_t shaderBinding, bool bBindSsboWithOffset = false) override;
    RHI::RHIShaderBindingPtr AddBufferToShaderBindings(RHI::RHIShaderBindingSetPtr& set, const std::string& name, size_t size, uint32_t shaderBinding,
                                                       RHI::EShaderBindingType bufferType) override;
    RHI::RHIShaderBindingPtr AddSamplerToShaderBindings(RHI::RHIShaderBindingSetPtr& set, const std::string& name, RHI::RHITexturePtr texture,
                                                        uint32_t shaderBinding) override;
    RHI::RHIShaderBindingPtr AddSamplerToShaderBindings(RHI::RHIShaderBindingSetPtr& set, const std::string& name,
                                                        const TVector<RHI::RHITexturePtr>& array, uint32_t shaderBinding) override;
    RHI::RHIShaderBindingPtr AddStorageImageToShaderBindings(RHI::RHIShaderBindingSetPtr& set, const std::string& name, RHI::RHITexturePtr texture,
                                                             uint32_t shaderBinding) override;
    RHI::RHIShaderBindingPtr AddStorageImageToShaderBindings(RHI::RHIShaderBindingSetPtr& set, const std::string& name,
                                                             const TVector<RHI::RHITexturePtr>& array, uint32_t shaderBinding) override;
    RHI::RHIShaderBindingPtr AddShaderBinding(RHI::RHIShaderBindingSetPtr& set, const RHI::RHIShaderBindingPtr& binding, const std::string& name,
                                              uint32_t shaderBinding) override;
    // Split frame: this driver renders tile-row band `rank` of `worldSize` of every frame.  The per-pixel targets bound afterwards
    // (sceneDepth, surface, radiance) hold the band's framebuffer rows only; `frame.viewportSize` stays the whole frame's.  `ncclComm` (an
    // ncclComm_t of worldSize ranks, may be null until an exchange is wanted) is only used by ExchangeLightLists.
    int SetFrameSplit(int rank, int worldSize, void* ncclComm);
    const SailorBand* GetBand() const { return m_worldSize > 1 ? &m_band : nullptr; }
    // The RCCL step of a split frame: the band's lightsGrid / culledLights of the last light cull -> the reference's global buffers (on every
    // rank), recorded on the driver's stream (sailor_hip_exchange_light_lists).  Buffers: tiles x 8 bytes and (1 + tiles x 128) x 4 bytes.
    int ExchangeLightLists(RHI::RHIBufferPtr bandGrid, RHI::RHIBufferPtr bandCulled, RHI::RHIBufferPtr globalGrid, RHI::RHIBufferPtr globalCulled);
    // wrap memory owned by someone else (a torch tensor in the tests, an engine heap in the real thing)
    RHI::RHIBufferPtr WrapBuffer(void* devicePtr, size_t size);
    RHI::RHITexturePtr WrapTexture(void* devicePtr, RHI::ivec2 extent, RHI::EFormat format);
    RHI::RHICubemapPtr WrapCubemap(void* devicePtr, int size, uint32_t mipLevels, RHI::EFormat format);

    // IGraphicsDriverCommands
    void BeginDebugRegion(RHI::RHICommandListPtr cmdList, const std::string& title) override;
    void EndDebugRegion(RHI::RHICommandListPtr cmdList) override;
    void ImageMemoryBarrier(RHI::RHICommandListPtr cmd, RHI::RHITexturePtr image, RHI::EImageLayout newLayout) override;
    bool BlitImage(RHI::RHICommandListPtr cmd, RHI::RHITexturePtr src, RHI::RHITexturePtr dst, RHI::ivec4 srcRegionRect, RHI::ivec4 dstRegionRect) override;
    void GenerateMipMaps(RHI::RHICommandListPtr cmd, RHI::RHITexturePtr target) override;
    void ConvertEquirect2Cubemap(RHI::RHICommandListPtr cmd, RHI::RHITexturePtr equirect, RHI::RHICubemapPtr cubemap) override;
    void UpdateShaderBinding(RHI::RHICommandListPtr cmd, RHI::RHIShaderBindingPtr binding, const void* data, size_t size, size_t variableOffset = 0) override;
    void UpdateBuffer(RHI::RHICommandListPtr cmd, RHI::RHIBufferPtr buffer, const void* data, size_t size, size_t offset = 0) override;
    void BeginRenderPass(RHI::RHICommandListPtr cmd, const TVector<RHI::RHITexturePtr>& colorAttachments, RHI::RHITexturePtr depthStencilAttachment) override;
    void BindVertexBuffer(RHI::RHICommandListPtr cmd, RHI::RHIBufferPtr vertexBuffer, uint32_t offset) override;
    void BindIndexBuffer(RHI::RHICommandListPtr cmd, RHI::RHIBufferPtr indexBuffer, uint32_t offset, bool bUint16InsteadOfUint32 = false) override;
    void PushConstants(RHI::RHICommandListPtr cmd, RHI::RHIMaterialPtr material, size_t size, const void* ptr) override;
    void EndRenderPass(RHI::RHICommandListPtr cmd) override;
    void BindMaterial(RHI::RHICommandListPtr cmd, RHI::RHIMaterialPtr material) override;
    void BindShaderBindings(RHI::RHICommandListPtr cmd, RHI::RHIMaterialPtr material, const TVector<RHI::RHIShaderBindingSetPtr>& bindings) override;
    void DrawIndexed(RHI::RHICommandListPtr cmd, uint32_t indexCount, uint32_t instanceCount, uint32_t firstIndex, uint32_t vertexOffset,
                     uint32_t firstInstance) override;
    void Dispatch(RHI::RHICommandListPtr cmd, RHI::RHIShaderPtr computeShader, uint32_t groupSizeX, uint32_t groupSizeY, uint32_t groupSizeZ,
                  const TVector<RHI::RHIShaderBindingSetPtr>& bindings, const void* pPushConstantsData = nullptr,
                  uint32_t sizePushConstantsData = 0) override;

private:
    int RecordLightCulling(const TVector<RHI::RHIShaderBindingSetPtr>& bindings, const TVector<uint8_t>& pc);
    int RecordShade(const TVector<RHI::RHIShaderBindingSetPtr>& bindings);
    int RecordBrdfLut(const TVector<RHI::RHIShaderBindingSetPtr>& bindings);
    int RecordIrradianceMap(const TVector<RHI::RHIShaderBindingSetPtr>& bindings);
    int RecordEnvPrefilter(const TVector<RHI::RHIShaderBindingSetPtr>& bindings, const TVector<uint8_t>& pc);
    int RecordMeshCulling(const TVector<RHI::RHIShaderBindingSetPtr>& bindings, const TVector<uint8_t>& pc, bool occlusion);
    int RecordDepthHighZ(const TVector<RHI::RHIShaderBindingSetPtr>& bindings);
    int RecordLinearizeDepth(const TVector<RHI::RHIShaderBindingSetPtr>& bindings, const RHI::RHITexturePtr& target);
    int RecordEvsmBlur(const TVector<RHI::RHIShaderBindingSetPtr>& bindings, const RHI::RHITexturePtr& target, bool vertical);

    SailorHipContext* m_ctx = nullptr;              // == m_ctxOwner.get(): what the C-ABI calls take
    std::shared_ptr<SailorHipContext> m_ctxOwner;   // destroyed with the last buffer that still refers to it
    int m_status = 0;
    int m_lastDispatchStatus = 0;
    RHI::RHIBufferPtr m_cullWorkspace;
    RHI::RHIBufferPtr m_meshCullWorkspace;
    int32_t m_cullW = 0, m_cullH = 0, m_cullLights = 0; // geometry of the last light cull: locates its shading-order hint in the workspace
    bool m_cullOrderValid = false;
    std::map<const void*, RHI::RHIBufferPtr> m_rasterWorkspaces; // depth attachment -> the rasteriser's workspace (coarse depth, mesh box, giants' queue)
    uint32_t m_exchangesClipped = 0; // exchanges whose global lists arrived clipped (reported by the exchange after them)
    // Round 4: the light cull stops after its per-tile lists (SAILOR_CULL_DEFER_PACK); the compaction into the node's `lightsGrid` / `culledLights`
    // SSBOs is recorded on a second context (own stream) and runs BESIDE the RenderScene shade, which reads the per-tile lists when the two SSBOs
    // it is handed are the ones this frame's cull owns.  The main stream joins the second one at the end of the submit that recorded the cull.
    SailorHipContext* m_ctxAux = nullptr;
    bool m_packPending = false;
    void BeforeBufferWrite(const void* devicePtr);
    SailorBand m_cullBand {};          // the band the last recorded cull ran on
    const void* m_ownGrid = nullptr;   // the SSBOs the last recorded cull fills
    const void* m_ownCulled = nullptr;
    int m_rank = 0, m_worldSize = 1; // the frame split
    void* m_comm = nullptr;
    SailorBand m_band {};
    int32_t m_splitW = 0, m_splitH = 0; // the frame size m_band was computed for
    RHI::RHIBufferPtr m_exchangeWorkspace;
};

} // namespace Sailor::GraphicsDriver::HIP
