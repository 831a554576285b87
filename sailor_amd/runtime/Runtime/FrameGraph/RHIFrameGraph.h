// Mirrors the parts of Runtime/FrameGraph/RHIFrameGraph.{h,cpp} the path needs: the ordered node list, named render
// targets, FillFrameData (RHIFrameGraph.cpp:50-73) and the per-frame node walk (RHIFrameGraph.cpp:95,250-252).  The Vulkan-
// specific barrier balancing (:201-246) and command-list chaining (:254-322) are out of scope.
#pragma once
#include <map>
#include <string>
#include <vector>
#include "BaseFrameGraphNode.h"
#include "../RHI/GraphicsDriver.h"

namespace Sailor::Framegraph {

class RHIFrameGraph {
public:
    void AddNode(FrameGraphNodePtr node) { m_graph.push_back(node); }
    const std::vector<FrameGraphNodePtr>& GetGraph() const { return m_graph; }
    void SetRenderTarget(const std::string& name, RHI::RHITexturePtr rt) { m_renderTargets[name] = rt; }
    RHI::RHITexturePtr GetRenderTarget(const std::string& name) const
    {
        auto it = m_renderTargets.find(name);
        return it == m_renderTargets.end() ? RHI::RHITexturePtr() : it->second;
    }
    // named samplers published by nodes (EnvironmentNode.cpp:79,169-170 SetSampler("g_brdfSampler" / "g_envCubemap" / "g_irradianceCubemap"))
    void SetSampler(const std::string& name, RHI::RHITexturePtr tex) { m_samplers[name] = tex; }
    RHI::RHITexturePtr GetSampler(const std::string& name) const
    {
        auto it = m_samplers.find(name);
        return it == m_samplers.end() ? RHI::RHITexturePtr() : it->second;
    }
    void SetValue(const std::string& name, float v) { m_values[name] = v; } // RHIFrameGraph.h SetValue (FrameGraphParser.cpp:130)
    float GetValue(const std::string& name, float fallback = 0.0f) const { auto it = m_values.find(name); return it == m_values.end() ? fallback : it->second; }
    void SetViewport(int32_t width, int32_t height) { m_viewport = { width, height }; } // App::GetMainWindow()->GetRenderArea()
    RHI::ivec2 GetViewport() const { return m_viewport; }

    RHI::UboFrameData FillFrameData(RHI::RHICommandListPtr transferCmdList, RHI::RHISceneViewSnapshot& snapshot, float deltaTime, float worldTime) const;
    // One frame: FillFrameData, then node->Process in graph order, then submit (record-then-submit)
    void Process(RHI::RHISceneViewSnapshot& snapshot);
    void Clear();

private:
    std::vector<FrameGraphNodePtr> m_graph;
    std::map<std::string, RHI::RHITexturePtr> m_renderTargets;
    std::map<std::string, RHI::RHITexturePtr> m_samplers;
    std::map<std::string, float> m_values;
    RHI::ivec2 m_viewport;
};

} // namespace Sailor::Framegraph
