// The `.renderer` frame-graph description (Content/DefaultRenderer.renderer) and its importer: mirrors
// Runtime/AssetRegistry/FrameGraph/FrameGraphParser.{h,cpp} -- FrameGraphAsset::Deserialize (FrameGraphParser.cpp:23-78), its RenderTarget /
// Node records (FrameGraphParser.h:64-213) and FrameGraphImporter::BuildFrameGraph (FrameGraphParser.cpp:80-203).
// The reference reads the file with yaml-cpp; the files use a small, regular subset of YAML (block sequences of flat maps, one level of
// nested `key:` dictionaries written as sequences of single-pair maps, flow sequences for vec4, `#` comments, `~`), which the reader below
// handles directly.  Texture samplers (`samplers:`) need the asset pipeline and are recorded but not loaded.
#pragma once
#include <map>
#include <string>
#include <vector>
#include "../../FrameGraph/RHIFrameGraph.h"

namespace Sailor {

class FrameGraphAsset {
public:
    struct RenderTarget { // FrameGraphParser.h:64-160
        std::string m_name;
        uint32_t m_width = 1, m_height = 1;
        std::string m_format = "R8G8B8A8_SRGB", m_filtration = "Linear", m_clamping = "Clamp", m_reduction = "Average";
        bool m_bIsSurface = false, m_bIsCompatibleWithComputeShaders = false, m_bGenerateMips = false;
        uint32_t m_maxMipLevel = 10000;
    };
    struct Node { // FrameGraphParser.h:162-213
        std::string m_name, m_tag;
        std::map<std::string, std::string> m_strings;
        std::map<std::string, float> m_floats;
        std::map<std::string, Framegraph::vec4> m_vectors;
        std::vector<std::pair<std::string, std::string>> m_renderTargets; // parameter name -> render-target name
    };

    // FrameGraphAsset::Deserialize; `ViewportWidth[/k]`, `ViewportHeight[/k]` resolve against the render area (FrameGraphParser.h:82-108).
    // Returns false (and a message) on text outside the subset described above.
    bool Deserialize(const std::string& yamlText, int32_t viewportWidth, int32_t viewportHeight, std::string* outError = nullptr);

    std::vector<std::string> m_samplers;                 // names only
    std::map<std::string, float> m_values;               // top-level `float:`
    std::vector<RenderTarget> m_renderTargets;           // file order
    std::vector<Node> m_nodes;                           // frame order
};

struct FrameGraphBuildReport {
    int m_nodesCreated = 0, m_nodesNotImplemented = 0, m_renderTargets = 0, m_unresolvedTargets = 0;
    std::vector<std::string> m_notImplemented;
};

class FrameGraphImporter {
public:
    // BuildFrameGraph (FrameGraphParser.cpp:80-203): render targets through IGraphicsDriver::CreateRenderTarget (formats the HIP backend has no
    // image type for are created as their fp32 canonical form: R16G16B16A16_SFLOAT -> R32G32B32A32_SFLOAT, depth / R8 / R16 -> R32_SFLOAT), values, then one
    // node per `frame` entry through FrameGraphBuilder::CreateNode -- names without a node class are logged and skipped (:155-159) -- with its tag,
    // parameters and render targets; targets that do not exist yet (DepthBuffer, BackBuffer, ...) stay unresolved by name (:190-195).
    static FrameGraphBuildReport BuildFrameGraph(const FrameGraphAsset& asset, Framegraph::RHIFrameGraph& outGraph);
};

} // namespace Sailor
