// Mirrors Runtime/FrameGraph/BaseFrameGraphNode.h:12-52: the node base class with its string / vec4 / float / resource
// parameter bags and the Prepare / Process / Clear virtuals.  (Prepare returns a task in the reference; the path's nodes
// do not override it, so it is a plain no-op here.)
#pragma once
#include <map>
#include <string>
#include "../RHI/Types.h"
#include "../RHI/SceneView.h"

namespace Sailor::Framegraph {

class RHIFrameGraph;
using RHIFrameGraphPtr = RHIFrameGraph*;

struct vec4 { float x = 0, y = 0, z = 0, w = 0; };

class BaseFrameGraphNode : public RHI::RHIResource {
public:
    ~BaseFrameGraphNode() override = default;

    void SetString(const std::string& name, const std::string& value) { m_stringParams[name] = value; }
    void SetVec4(const std::string& name, const vec4& value) { m_vectorParams[name] = value; }
    void SetFloat(const std::string& name, float value) { m_floatParams[name] = value; }
    void SetRHIResource(const std::string& name, RHI::RHIResourcePtr value) { m_resourceParams[name] = value; }
    // BaseFrameGraphNode.h SetRHIResource_Unresolved: per-frame targets (DepthBuffer, BackBuffer, ...) are looked up by name when the node runs
    void SetRHIResource_Unresolved(const std::string& name, const std::string& renderTargetName) { m_unresolvedResourceParams[name] = renderTargetName; }
    const std::map<std::string, std::string>& GetUnresolvedResources() const { return m_unresolvedResourceParams; }

    RHI::RHIResourcePtr GetRHIResource(const std::string& name) const
    {
        auto it = m_resourceParams.find(name);
        return it == m_resourceParams.end() ? RHI::RHIResourcePtr() : it->second;
    }
    const vec4& GetVec4(const std::string& name) const { return m_vectorParams.at(name); }
    float GetFloat(const std::string& name) const { return m_floatParams.at(name); }
    const std::string& GetString(const std::string& name) const { return m_stringParams.at(name); }
    bool TryGetString(const std::string& name, std::string& out) const
    {
        auto it = m_stringParams.find(name);
        if (it == m_stringParams.end()) return false;
        out = it->second;
        return true;
    }

    bool TryGetFloat(const std::string& name, float& out) const
    {
        auto it = m_floatParams.find(name);
        if (it == m_floatParams.end()) return false;
        out = it->second;
        return true;
    }

    virtual void Prepare(RHIFrameGraphPtr, const RHI::RHISceneViewSnapshot&) {}
    virtual void Process(RHIFrameGraphPtr frameGraph, RHI::RHICommandListPtr transferCommandList, RHI::RHICommandListPtr commandList,
                         const RHI::RHISceneViewSnapshot& sceneView) = 0;
    virtual void Clear() = 0;
    virtual std::string GetDebugName() const = 0;

    const std::string& GetTag() const { return m_tag; }
    void SetTag(const std::string& tag) { m_tag = tag; }

protected:
    std::map<std::string, std::string> m_stringParams;
    std::map<std::string, vec4> m_vectorParams;
    std::map<std::string, float> m_floatParams;
    std::map<std::string, RHI::RHIResourcePtr> m_resourceParams;
    std::map<std::string, std::string> m_unresolvedResourceParams;
    std::string m_tag;
};

using FrameGraphNodePtr = TRefPtr<BaseFrameGraphNode>;

} // namespace Sailor::Framegraph
