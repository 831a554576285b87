// Linux-buildable mirror of the slice of Sailor's RHI that the Forward+ lighting path touches.
// Mirrors (signatures, names, argument meaning): Runtime/RHI/Types.h:14-30,751-761,781-799 (RHIResource + TRefPtr aliases,
// UboFrameData), RHI/Buffer.h, RHI/Shader.h, RHI/CommandList.h -- with an `m_hip` member where the reference's classes
// carry `m_vulkan` (RHI/Shader.h:17-27,71-76, RHI/Buffer.h:15-24, RHI/CommandList.h:15-20).
#pragma once
#include <memory>
#include <atomic>
#include <cstdint>
#include <cstring>
#include <functional>
#include <map>
#include <string>
#include <utility>
#include <vector>

#include "../../../../include/sailor_hip.h"

namespace Sailor {

// minimal intrusive pointer (the reference: Memory/RefPtr.hpp, TRefPtr<T> over TRefBase)
class TRefBase {
public:
    virtual ~TRefBase() = default;
    void AddRef() const { m_refs.fetch_add(1, std::memory_order_relaxed); }
    void Release() const { if (m_refs.fetch_sub(1, std::memory_order_acq_rel) == 1) delete this; }
private:
    mutable std::atomic<int> m_refs { 0 };
};

template <typename T>
class TRefPtr {
public:
    TRefPtr() = default;
    TRefPtr(T* p) : m_p(p) { if (m_p) m_p->AddRef(); }
    TRefPtr(const TRefPtr& o) : m_p(o.m_p) { if (m_p) m_p->AddRef(); }
    template <typename U> TRefPtr(const TRefPtr<U>& o) : m_p(o.GetRawPtr()) { if (m_p) m_p->AddRef(); }
    TRefPtr(TRefPtr&& o) noexcept : m_p(o.m_p) { o.m_p = nullptr; }
    ~TRefPtr() { if (m_p) m_p->Release(); }
    TRefPtr& operator=(TRefPtr o) { std::swap(m_p, o.m_p); return *this; }
    template <typename... A> static TRefPtr Make(A&&... a) { return TRefPtr(new T(std::forward<A>(a)...)); }
    T* operator->() const { return m_p; }
    T& operator*() const { return *m_p; }
    T* GetRawPtr() const { return m_p; }
    explicit operator bool() const { return m_p != nullptr; }
    bool IsValid() const { return m_p != nullptr; }
    void Clear() { if (m_p) m_p->Release(); m_p = nullptr; }
    template <typename U> TRefPtr<U> DynamicCast() const { return TRefPtr<U>(dynamic_cast<U*>(m_p)); }
private:
    T* m_p = nullptr;
};

template <typename T> using TVector = std::vector<T>;

namespace RHI {

class RHIResource : public TRefBase {};
using RHIResourcePtr = TRefPtr<RHIResource>;

struct ivec2 { int32_t x = 0, y = 0; };
struct ivec4 { int32_t x = 0, y = 0, z = 0, w = 0; };

using UboFrameData = SailorUboFrameData; // RHI/Types.h:751-761

// RHI/Buffer.h: a device allocation.  m_hip.m_devicePtr is either owned (created through IGraphicsDriver) or wrapped.
class RHIBuffer : public RHIResource {
public:
    struct { void* m_devicePtr = nullptr; bool m_bOwned = false; } m_hip;
    size_t m_size = 0;
    ~RHIBuffer() override;
    // The context the allocation belongs to, shared with the driver that made it: a buffer (or a texture / binding holding one) may outlive
    // the driver object, and the context is destroyed by whoever lets go of it last.
    std::shared_ptr<SailorHipContext> m_ctx;
};
using RHIBufferPtr = TRefPtr<RHIBuffer>;

enum class EFormat { R32_SFLOAT, R16_SFLOAT, R32G32B32A32_SFLOAT, R32G32_SFLOAT };
enum class EImageLayout { Undefined, ShaderReadOnlyOptimal, General, ColorAttachmentOptimal, ComputeWrite, TransferSrcOptimal, TransferDstOptimal };

// RHI/Texture.h: here a linear row-major image in device memory (row 0 = top)
class RHITexture : public RHIResource {
public:
    RHIBufferPtr m_buffer;
    ivec2 m_extent;
    EFormat m_format = EFormat::R32_SFLOAT;
    uint32_t m_mipLevels = 1; // a cubemap is 6 faces x m_extent.x^2 texels per level, level-major (include/sailor_hip.h SailorIblDesc)
    bool m_bCubemap = false;
    bool m_bRepeat = true;    // ETextureClamping::Repeat (the TextureAssetInfo default) vs Clamp: read by ConvertEquirect2Cubemap's sampler
    // a mip-level view of a cubemap (RHI/Cubemap.h:24 GetMipLevel): shares m_buffer with its parent
    TRefPtr<RHITexture> m_parent;
    uint32_t m_viewLevel = 0;
    ivec2 GetExtent() const { return m_extent; }
    uint32_t GetMipLevels() const { return m_mipLevels; }
    TRefPtr<RHITexture> GetMipLevel(uint32_t mipLevel) const;
};
using RHITexturePtr = TRefPtr<RHITexture>;
using RHICubemapPtr = RHITexturePtr; // RHI/Cubemap.h:15: class RHICubemap : public RHITexture

enum class EShaderBindingType { UniformBuffer, StorageBuffer, CombinedImageSampler, StorageImage };

// RHI/ShaderBinding.h: one named resource at one binding slot
class RHIShaderBinding : public RHIResource {
public:
    std::string m_name;
    EShaderBindingType m_type = EShaderBindingType::StorageBuffer;
    uint32_t m_binding = 0;
    RHIBufferPtr m_buffer;              // SSBO / UBO storage
    TVector<RHITexturePtr> m_textures;  // sampler (array)
    TVector<uint8_t> m_hostCopy;        // UBOs keep the last uploaded bytes: the HIP entry points take them by value
    // HIP backend, the `light` SSBO only: the prepared views of its records (sailor_hip_prepare_lights), kept in step by UpdateShaderBinding
    TRefPtr<class RHIBuffer> m_hipPreparedLights;
    int32_t m_hipPreparedCapacity = 0;
    size_t GetBufferOffset() const { return 0; }
};
using RHIShaderBindingPtr = TRefPtr<RHIShaderBinding>;

class RHIShaderBindingSet : public RHIResource {
public:
    RHIShaderBindingPtr GetOrAddShaderBinding(const std::string& name)
    {
        auto it = m_bindings.find(name);
        if (it != m_bindings.end()) return it->second;
        auto b = RHIShaderBindingPtr::Make();
        b->m_name = name;
        m_bindings[name] = b;
        return b;
    }
    RHIShaderBindingPtr Find(const std::string& name) const
    {
        auto it = m_bindings.find(name);
        return it == m_bindings.end() ? RHIShaderBindingPtr() : it->second;
    }
    std::map<std::string, RHIShaderBindingPtr> m_bindings;
};
using RHIShaderBindingSetPtr = TRefPtr<RHIShaderBindingSet>;

// RHI/Shader.h: identified by the asset path the reference loads it from (e.g. "Shaders/ComputeLightCulling.shader")
class RHIShader : public RHIResource {
public:
    explicit RHIShader(std::string name, std::vector<std::string> defines = {}) : m_name(std::move(name)), m_defines(std::move(defines)) {}
    std::string m_name;
    std::vector<std::string> m_defines; // the permutation (ShaderCompiler::LoadShader_Immediate(fileId, shader, { "VERTICAL", "EVSM" }), ShadowPrepassNode.cpp:60)
    bool HasDefine(const std::string& d) const { for (auto& x : m_defines) if (x == d) return true; return false; }
};
using RHIShaderPtr = TRefPtr<RHIShader>;

// RHI/Material.h: for this path a material is its shader (render state, vertex layout and topology have no meaning for a
// full-screen pass executed as a compute kernel)
class RHIMaterial : public RHIResource {
public:
    explicit RHIMaterial(RHIShaderPtr shader) : m_shader(std::move(shader)) {}
    RHIShaderPtr m_shader;
};
using RHIMaterialPtr = TRefPtr<RHIMaterial>;

// RHI/CommandList.h: recorded work, executed at SubmitCommandList (record-then-submit, RHI/GraphicsDriver.h:149)
class RHICommandList : public RHIResource {
public:
    struct { TVector<std::function<int()>> m_commands; } m_hip;
    TVector<std::string> m_debugRegions; // names seen by BeginDebugRegion, for tests
    // state of the render pass being recorded (BeginRenderPass .. EndRenderPass)
    TVector<TRefPtr<class RHITexture>> m_colorAttachments;
    TRefPtr<class RHITexture> m_depthAttachment;
    TRefPtr<class RHIBuffer> m_vertexBuffer, m_indexBuffer;   // BindVertexBuffer / BindIndexBuffer
    TVector<uint8_t> m_pushConstants;                          // PushConstants(material, size, ptr)
    uint32_t m_casterDraws = 0;                                // depth-only draws recorded in the current pass (the first one clears)
    TRefPtr<class RHIMaterial> m_boundMaterial;
    TVector<TRefPtr<class RHIShaderBindingSet>> m_boundBindings;
};
using RHICommandListPtr = TRefPtr<RHICommandList>;

} // namespace RHI
} // namespace Sailor
