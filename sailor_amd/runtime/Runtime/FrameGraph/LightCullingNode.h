// Mirrors Runtime/FrameGraph/LightCullingNode.h:10-37.
#pragma once
#include "FrameGraphNode.h"

namespace Sailor::Framegraph {

class LightCullingNode : public TFrameGraphNode<LightCullingNode> {
public:
    static const uint32_t LightsPerTile = 128; // LightCullingNode.h:15
    static const uint32_t TileSize = 16;       // LightCullingNode.h:16

    static const char* GetName() { return m_name; }

    void Process(RHIFrameGraphPtr frameGraph, RHI::RHICommandListPtr transferCommandList, RHI::RHICommandListPtr commandList,
                 const RHI::RHISceneViewSnapshot& sceneView) override;
    void Clear() override;

    RHI::RHIShaderBindingSetPtr GetCulledLights() const { return m_culledLights; }

protected:
    using PushConstants = SailorLightCullPushConstants; // LightCullingNode.h:25-31 (88 bytes)

    static const char* m_name;
    RHI::RHIShaderPtr m_pComputeShader;
    RHI::RHIShaderBindingSetPtr m_culledLights;
};

// The pass in front of the cull: Runtime/FrameGraph/LinearizeDepthNode.h.  Resources: "depthStencil" = the raw reversed-Z
// depth attachment, "target" = the LinearDepth render target (DefaultRenderer.renderer: LinearizeDepth node).
class LinearizeDepthNode : public TFrameGraphNode<LinearizeDepthNode> {
public:
    static const char* GetName() { return m_name; }
    void Process(RHIFrameGraphPtr frameGraph, RHI::RHICommandListPtr transferCommandList, RHI::RHICommandListPtr commandList,
                 const RHI::RHISceneViewSnapshot& sceneView) override;
    void Clear() override;

protected:
    static const char* m_name;
    RHI::RHIShaderPtr m_pLinearizeDepthShader;
    RHI::RHIMaterialPtr m_postEffectMaterial;
    RHI::RHIShaderBindingSetPtr m_linearizeDepth;
};

// The shading consumer of the lists: RenderSceneNode (FrameGraph/RenderSceneNode.cpp:109) records raster draws whose
// fragment shader is Standard.shader; here the fragment work is one compute dispatch over a surface buffer
// (resources "surface" = 3 float4 planes, "radiance" = float4 per pixel).
class RenderSceneNode : public TFrameGraphNode<RenderSceneNode> {
public:
    static const char* GetName() { return m_name; }
    void Process(RHIFrameGraphPtr frameGraph, RHI::RHICommandListPtr transferCommandList, RHI::RHICommandListPtr commandList,
                 const RHI::RHISceneViewSnapshot& sceneView) override;
    void Clear() override;

protected:
    static const char* m_name;
    RHI::RHIShaderPtr m_pShader;
    RHI::RHIShaderBindingSetPtr m_surfaceBindings;
};

// The one-off image-based-lighting bake: Runtime/FrameGraph/EnvironmentNode.h.  Raw environment = the frame graph's sampler
// "g_skyCubemap" (EnvironmentNode.cpp:139-142; the equirect-file branch needs the asset pipeline and is not mirrored); results are
// published as samplers "g_brdfSampler", "g_envCubemap", "g_irradianceCubemap" (:79, :187, :247), which RHIFrameGraph::Process binds
// into the lights set for Standard.shader's ambient term.
class EnvironmentNode : public TFrameGraphNode<EnvironmentNode> {
public:
    static constexpr uint32_t EnvMapSize = 512;       // EnvironmentNode.h:16 } the raw cube made from an equirect panorama; a sky cubemap
    static constexpr uint32_t EnvMapLevels = 10;      // EnvironmentNode.h:17 } published as g_skyCubemap brings its own size
    static constexpr uint32_t IrradianceMapSize = 32; // EnvironmentNode.h:19
    static constexpr uint32_t BrdfLutSize = 256;      // EnvironmentNode.h:20 (EnvMapSize / EnvMapLevels (:16-17) are taken from the raw cubemap here)
    static const char* GetName() { return m_name; }
    void Process(RHIFrameGraphPtr frameGraph, RHI::RHICommandListPtr transferCommandList, RHI::RHICommandListPtr commandList,
                 const RHI::RHISceneViewSnapshot& sceneView) override;
    void Clear() override;
    void MarkDirty() { m_bIsDirty = true; } // EnvironmentNode.h:28
    // EnvironmentNode.cpp:100-111 loads the "EnvironmentMap" asset through the TextureImporter (out of scope): the loaded texture is handed in
    void SetEnvironmentMap(RHI::RHITexturePtr equirect) { m_envMapTexture = equirect; m_envCubemap.Clear(); m_irradianceCubemap.Clear(); m_bIsDirty = true; }

protected:
    static const char* m_name;
    RHI::RHIShaderPtr m_pComputeIrradianceShader, m_pComputeSpecularShader, m_pComputeBrdfShader;
    RHI::RHIShaderBindingSetPtr m_computeIrradianceBindings, m_computeSpecularBindings, m_computeBrdfBindings;
    RHI::RHICubemapPtr m_envCubemap, m_irradianceCubemap; // (keyed by the sky parameters in the reference: one sky here)
    RHI::RHITexturePtr m_brdfSampler;
    RHI::RHITexturePtr m_envMapTexture; // EnvironmentNode.h:43
    bool m_bIsDirty = false;
};

// The Hi-Z pyramid builder: Runtime/FrameGraph/DepthHighZNode.h.  Resources: "src" = the (half-resolution) depth target, "dst" = the
// DepthHighZ render target with its mip chain (DefaultRenderer.renderer:51-57, :213-217).
class DepthHighZNode : public TFrameGraphNode<DepthHighZNode> {
public:
    static const char* GetName() { return m_name; }
    void Process(RHIFrameGraphPtr frameGraph, RHI::RHICommandListPtr transferCommandList, RHI::RHICommandListPtr commandList,
                 const RHI::RHISceneViewSnapshot& sceneView) override;
    void Clear() override;

protected:
    struct PushConstantsDownscale { float m_outputSize[2]; }; // DepthHighZNode.h
    static const char* m_name;
    RHI::RHIShaderPtr m_pComputeDepthHighZShader;
    TVector<RHI::RHIShaderBindingSetPtr> m_computeDepthHighZBindings;
    RHI::RHIShaderBindingSetPtr m_computePrepassDepthHighZBindings;
};

} // namespace Sailor::Framegraph
