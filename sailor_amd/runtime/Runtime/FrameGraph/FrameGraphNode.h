// Mirrors Runtime/FrameGraph/FrameGraphNode.h:13-68: the name -> factory registry (FrameGraphBuilder) and the CRTP base
// TFrameGraphNode<T> whose static registration object makes a node creatable by the name used in the .renderer YAML
// (AssetRegistry/FrameGraph/FrameGraphParser.cpp:153 CreateNode(node.m_name)).
#pragma once
#include <functional>
#include <map>
#include <string>
#include "BaseFrameGraphNode.h"

namespace Sailor::Framegraph {

class FrameGraphBuilder {
public:
    static void RegisterFrameGraphNode(const std::string& nodeName, std::function<FrameGraphNodePtr(void)> factoryMethod);
    static FrameGraphNodePtr CreateNode(const std::string& nodeName);
    static bool IsRegistered(const std::string& nodeName);
private:
    static std::map<std::string, std::function<FrameGraphNodePtr(void)>>& Registry();
};

template <typename TRenderNode>
class TFrameGraphNode : public BaseFrameGraphNode {
public:
    TFrameGraphNode() { s_registrationFactoryMethod.DoWork(); }
    static const char* GetName() { return TRenderNode::GetName(); }
    std::string GetDebugName() const override { return TRenderNode::GetName(); }

protected:
    class RegistrationFactoryMethod {
    public:
        RegistrationFactoryMethod()
        {
            FrameGraphBuilder::RegisterFrameGraphNode(std::string(TRenderNode::GetName()), []() { return FrameGraphNodePtr(new TRenderNode()); });
        }
        void DoWork() {} // keeps the static object (and thereby the registration) alive in every build
    };
    static RegistrationFactoryMethod s_registrationFactoryMethod;
};

template <typename T>
typename TFrameGraphNode<T>::RegistrationFactoryMethod TFrameGraphNode<T>::s_registrationFactoryMethod;

} // namespace Sailor::Framegraph
