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

} // namespace Sailor::Framegraph
