// The two interfaces every FrameGraph node talks to -- the subset the Forward+ path calls.
// Mirrors Runtime/RHI/GraphicsDriver.h:59-224 (IGraphicsDriver) and :226-346 (IGraphicsDriverCommands): same method names,
// argument order and meaning; methods the path never calls (swapchain, vertex buffers, ...) are omitted.  The render-pass
// subset exists for the one full-screen draw in front of the path, LinearizeDepthNode (FrameGraph/LinearizeDepthNode.cpp:22-109).
#pragma once
#include "Types.h"

namespace Sailor::RHI {

class IGraphicsDriver {
public:
    virtual ~IGraphicsDriver() = default;
    virtual void WaitIdle() = 0;                                                                           // :83
    virtual RHICommandListPtr CreateCommandList(bool bIsSecondary = false) = 0;                            // :88
    virtual RHIBufferPtr CreateBuffer(size_t size) = 0;                                                    // :89
    virtual RHIShaderPtr CreateShader(const std::string& assetPath, const TVector<std::string>& defines = {}) = 0; // :97 (SPIR-V there; here the asset path + permutation)
    virtual RHITexturePtr CreateTexture(const void* pData, size_t size, ivec2 extent, EFormat format) = 0; // :98-108
    virtual RHITexturePtr CreateRenderTarget(ivec2 extent, uint32_t mipLevels, EFormat format) = 0;        // :110-117 (filtration, clamping, usage dropped)
    virtual RHICubemapPtr CreateCubemap(ivec2 extent, uint32_t mipLevels, EFormat format) = 0;             // :137-143
    virtual void SubmitCommandList(RHICommandListPtr commandList) = 0;                                     // :149
    virtual RHIMaterialPtr CreateMaterial(RHIShaderPtr shader) = 0;                                        // :138-141 (vertex layout / topology / render state dropped)
    virtual RHIShaderBindingSetPtr CreateShaderBindings() = 0;                                             // :152
    virtual RHIShaderBindingPtr AddSsboToShaderBindings(RHIShaderBindingSetPtr& pShaderBindings, const std::string& name, size_t elementSize,
                                                        size_t numElements, uint32_t shaderBinding, bool bBindSsboWithOffset = false) = 0; // :154
    virtual RHIShaderBindingPtr AddBufferToShaderBindings(RHIShaderBindingSetPtr& pShaderBindings, const std::string& name, size_t size,
                                                          uint32_t shaderBinding, EShaderBindingType bufferType) = 0;                       // :155
    virtual RHIShaderBindingPtr AddSamplerToShaderBindings(RHIShaderBindingSetPtr& pShaderBindings, const std::string& name, RHITexturePtr texture,
                                                           uint32_t shaderBinding) = 0;                                                     // :156
    virtual RHIShaderBindingPtr AddSamplerToShaderBindings(RHIShaderBindingSetPtr& pShaderBindings, const std::string& name,
                                                           const TVector<RHITexturePtr>& array, uint32_t shaderBinding) = 0;               // :157
    virtual RHIShaderBindingPtr AddStorageImageToShaderBindings(RHIShaderBindingSetPtr& pShaderBindings, const std::string& name, RHITexturePtr texture,
                                                                uint32_t shaderBinding) = 0;                                                // :158
    virtual RHIShaderBindingPtr AddStorageImageToShaderBindings(RHIShaderBindingSetPtr& pShaderBindings, const std::string& name,
                                                                const TVector<RHITexturePtr>& array, uint32_t shaderBinding) = 0;          // :159
    virtual RHIShaderBindingPtr AddShaderBinding(RHIShaderBindingSetPtr& pShaderBindings, const RHIShaderBindingPtr& binding, const std::string& name,
                                                 uint32_t shaderBinding) = 0;                                                               // :160
};

class IGraphicsDriverCommands {
public:
    virtual ~IGraphicsDriverCommands() = default;
    virtual void BeginDebugRegion(RHICommandListPtr cmdList, const std::string& title) = 0; // :238
    virtual void EndDebugRegion(RHICommandListPtr cmdList) = 0;                              // :239
    virtual void ImageMemoryBarrier(RHICommandListPtr cmd, RHITexturePtr image, EImageLayout newLayout) = 0; // :290
    virtual bool BlitImage(RHICommandListPtr cmd, RHITexturePtr src, RHITexturePtr dst, ivec4 srcRegionRect, ivec4 dstRegionRect) = 0;                 // :293 (equal regions only: a copy)
    virtual void GenerateMipMaps(RHICommandListPtr cmd, RHITexturePtr target) = 0;                                                                  // :296 (cubemaps)
    virtual void ConvertEquirect2Cubemap(RHICommandListPtr cmd, RHITexturePtr equirect, RHICubemapPtr cubemap) = 0;                                 // :297
    virtual void UpdateShaderBinding(RHICommandListPtr cmd, RHIShaderBindingPtr binding, const void* data, size_t size, size_t variableOffset = 0) = 0; // :303
    virtual void UpdateBuffer(RHICommandListPtr cmd, RHIBufferPtr buffer, const void* data, size_t size, size_t offset = 0) = 0;                       // :304
    virtual void BeginRenderPass(RHICommandListPtr cmd, const TVector<RHITexturePtr>& colorAttachments, RHITexturePtr depthStencilAttachment) = 0; // :246-255 (area, clear values dropped)
    virtual void BindVertexBuffer(RHICommandListPtr cmd, RHIBufferPtr vertexBuffer, uint32_t offset) = 0;                                           // :316
    virtual void BindIndexBuffer(RHICommandListPtr cmd, RHIBufferPtr indexBuffer, uint32_t offset, bool bUint16InsteadOfUint32 = false) = 0;          // :317
    virtual void PushConstants(RHICommandListPtr cmd, RHIMaterialPtr material, size_t size, const void* ptr) = 0;                                      // :324
    virtual void EndRenderPass(RHICommandListPtr cmd) = 0;                                                                                          // :273
    virtual void BindMaterial(RHICommandListPtr cmd, RHIMaterialPtr material) = 0;                                                                  // :276
    virtual void BindShaderBindings(RHICommandListPtr cmd, RHIMaterialPtr material, const TVector<RHIShaderBindingSetPtr>& bindings) = 0;           // :282
    virtual void DrawIndexed(RHICommandListPtr cmd, uint32_t indexCount, uint32_t instanceCount, uint32_t firstIndex, uint32_t vertexOffset,
                             uint32_t firstInstance) = 0;                                                                                           // :285
    virtual void Dispatch(RHICommandListPtr cmd, RHIShaderPtr computeShader, uint32_t groupSizeX, uint32_t groupSizeY, uint32_t groupSizeZ,
                          const TVector<RHIShaderBindingSetPtr>& bindings, const void* pPushConstantsData = nullptr,
                          uint32_t sizePushConstantsData = 0) = 0;                                                                                       // :310-314
};

} // namespace Sailor::RHI
