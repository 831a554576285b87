#include "FrameGraphParser.h"
#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <sstream>
#include "../../FrameGraph/FrameGraphNode.h"
#include "../../RHI/Renderer.h"

using namespace Sailor;
using namespace Sailor::RHI;
using namespace Sailor::Framegraph;

namespace {

std::string trim(const std::string& s)
{
    size_t a = 0, b = s.size();
    while (a < b && (s[a] == ' ' || s[a] == '\t' || s[a] == '\r')) a++;
    while (b > a && (s[b - 1] == ' ' || s[b - 1] == '\t' || s[b - 1] == '\r')) b--;
    return s.substr(a, b - a);
}

std::string strip_comment(const std::string& line)
{
    bool sq = false, dq = false;
    for (size_t i = 0; i < line.size(); i++) {
        const char c = line[i];
        if (c == '\'' && !dq) sq = !sq;
        else if (c == '"' && !sq) dq = !dq;
        else if (c == '#' && !sq && !dq && (i == 0 || line[i - 1] == ' ' || line[i - 1] == '\t')) return line.substr(0, i);
    }
    return line;
}

std::string unquote(const std::string& v)
{
    if (v.size() >= 2 && ((v.front() == '\'' && v.back() == '\'') || (v.front() == '"' && v.back() == '"'))) return v.substr(1, v.size() - 2);
    return v;
}

bool split_pair(const std::string& text, std::string& key, std::string& value)
{
    const size_t c = text.find(':');
    if (c == std::string::npos) return false;
    key = trim(text.substr(0, c));
    value = unquote(trim(text.substr(c + 1)));
    if (value == "~") value.clear();
    return !key.empty();
}

bool parse_bool(const std::string& v) { return v == "true" || v == "True" || v == "TRUE" || v == "yes" || v == "1"; }

bool parse_vec4(const std::string& v, vec4& out)
{
    std::string t = v;
    for (auto& ch : t) if (ch == '[' || ch == ']' || ch == ',') ch = ' ';
    std::stringstream ss(t);
    float f[4] = { 0.0f, 0.0f, 0.0f, 1.0f };
    int n = 0;
    while (n < 4 && (ss >> f[n])) n++;
    if (n == 0) return false;
    out.x = f[0]; out.y = f[1]; out.z = f[2]; out.w = f[3];
    return true;
}

// FrameGraphParser.h:82-108 ParseUintValue
uint32_t parse_extent(const std::string& str, int32_t vw, int32_t vh)
{
    std::stringstream ss(str);
    uint32_t res = 1;
    if (ss >> res) return res;
    const size_t slash = str.find('/');
    const float multiplier = slash != std::string::npos ? 1.0f / (float)std::atof(str.c_str() + slash + 1) : 1.0f;
    if (str.rfind("ViewportWidth", 0) == 0) return std::max(1u, (uint32_t)((float)vw * multiplier));
    if (str.rfind("ViewportHeight", 0) == 0) return std::max(1u, (uint32_t)((float)vh * multiplier));
    return 1;
}

struct Item { // one `- ...` entry of a top-level sequence
    std::vector<std::pair<std::string, std::string>> scalars;
    std::vector<std::pair<std::string, std::vector<std::pair<std::string, std::string>>>> dicts;
};

} // namespace

bool FrameGraphAsset::Deserialize(const std::string& yamlText, int32_t vw, int32_t vh, std::string* outError)
{
    std::map<std::string, std::vector<Item>> sections;
    std::vector<Item>* section = nullptr;
    Item* item = nullptr;
    std::vector<std::pair<std::string, std::string>>* dict = nullptr;
    int itemIndent = -1;
    std::stringstream in(yamlText);
    std::string raw;
    int lineNo = 0;
    auto fail = [&](const std::string& what) {
        if (outError) *outError = "line " + std::to_string(lineNo) + ": " + what;
        return false;
    };
    while (std::getline(in, raw)) {
        lineNo++;
        std::string line = strip_comment(raw);
        while (!line.empty() && (line.back() == ' ' || line.back() == '\t' || line.back() == '\r')) line.pop_back();
        if (line.empty() || line == "---" || line == "...") continue;
        int indent = 0;
        while (indent < (int)line.size() && line[indent] == ' ') indent++;
        if (indent < (int)line.size() && line[indent] == '\t') return fail("tab indentation");
        std::string text = line.substr(indent);
        const bool dash = text.rfind("- ", 0) == 0 || text == "-";
        std::string key, value;
        if (!dash && indent == 0) { // `section:`
            if (!split_pair(text, key, value) || !value.empty()) return fail("expected `section:`");
            section = &sections[key];
            item = nullptr; dict = nullptr; itemIndent = -1;
            continue;
        }
        if (!section) return fail("entry outside a section");
        if (dash && (itemIndent < 0 || indent <= itemIndent) && !(dict && item && indent > itemIndent)) {
            // a new item of the section's sequence (the first one fixes the sequence's indentation)
            if (itemIndent < 0) itemIndent = indent;
            if (indent == itemIndent) {
                section->emplace_back();
                item = &section->back();
                dict = nullptr;
                const std::string rest = trim(text.substr(1));
                if (!rest.empty()) {
                    if (!split_pair(rest, key, value)) return fail("expected `- key: value`");
                    item->scalars.emplace_back(key, value);
                }
                continue;
            }
        }
        if (!item) return fail("entry outside a sequence item");
        if (dash) { // an entry of the item's current nested dictionary: `- key: value`
            if (!dict) return fail("sequence entry without a dictionary key in front of it");
            if (!split_pair(trim(text.substr(1)), key, value)) return fail("expected `- key: value`");
            dict->emplace_back(key, value);
            continue;
        }
        if (!split_pair(text, key, value)) return fail("expected `key: value`");
        const size_t colon = text.find(':');
        const bool opensDict = trim(text.substr(colon + 1)).empty();
        if (opensDict) { // `string:` / `float:` / `vec4:` / `renderTargets:` of a node (a `~` value is an explicit null: plain scalar)
            item->dicts.emplace_back(key, std::vector<std::pair<std::string, std::string>>());
            dict = &item->dicts.back().second;
        } else {
            item->scalars.emplace_back(key, value);
            dict = nullptr;
        }
    }

    for (const auto& it : sections["samplers"])
        for (const auto& kv : it.scalars) if (kv.first == "name") m_samplers.push_back(kv.second);
    for (const auto& it : sections["float"])
        for (const auto& kv : it.scalars) m_values[kv.first] = (float)std::atof(kv.second.c_str());
    for (const auto& it : sections["renderTargets"]) {
        RenderTarget rt;
        for (const auto& kv : it.scalars) {
            if (kv.first == "name") rt.m_name = kv.second;
            else if (kv.first == "width") rt.m_width = parse_extent(kv.second, vw, vh);
            else if (kv.first == "height") rt.m_height = parse_extent(kv.second, vw, vh);
            else if (kv.first == "format") rt.m_format = kv.second;
            else if (kv.first == "filtration") rt.m_filtration = kv.second;
            else if (kv.first == "reduction") rt.m_reduction = kv.second;
            else if (kv.first == "clamping") rt.m_clamping = kv.second;
            else if (kv.first == "bIsSurface") rt.m_bIsSurface = parse_bool(kv.second);
            else if (kv.first == "bGenerateMips") rt.m_bGenerateMips = parse_bool(kv.second);
            else if (kv.first == "maxMipLevel") rt.m_maxMipLevel = (uint32_t)std::atoi(kv.second.c_str());
            else if (kv.first == "bIsCompatibleWithComputeShaders") rt.m_bIsCompatibleWithComputeShaders = parse_bool(kv.second);
        }
        if (rt.m_name.empty()) return fail("render target without a name");
        m_renderTargets.push_back(rt);
    }
    for (const auto& it : sections["frame"]) {
        Node node;
        for (const auto& kv : it.scalars) {
            if (kv.first == "name") node.m_name = kv.second;
            else if (kv.first == "tag") node.m_tag = kv.second;
        }
        if (node.m_name.empty()) return fail("frame entry without a name");
        for (const auto& d : it.dicts) {
            for (const auto& kv : d.second) {
                if (d.first == "string") node.m_strings[kv.first] = kv.second;
                else if (d.first == "float") node.m_floats[kv.first] = (float)std::atof(kv.second.c_str());
                else if (d.first == "vec4") { vec4 v; if (parse_vec4(kv.second, v)) node.m_vectors[kv.first] = v; }
                else if (d.first == "renderTargets") node.m_renderTargets.emplace_back(kv.first, kv.second);
            }
        }
        m_nodes.push_back(node);
    }
    return true;
}

static EFormat canonical_format(const std::string& f)
{
    if (f == "R32G32B32A32_SFLOAT" || f == "R16G16B16A16_SFLOAT" || f == "R8G8B8A8_UNORM" || f == "R8G8B8A8_SRGB" || f == "B8G8R8A8_SRGB") return EFormat::R32G32B32A32_SFLOAT;
    if (f == "R32G32_SFLOAT" || f == "R16G16_SFLOAT") return EFormat::R32G32_SFLOAT;
    return EFormat::R32_SFLOAT; // R32_SFLOAT, R16_SFLOAT, R8_UNORM and the depth formats
}

FrameGraphBuildReport FrameGraphImporter::BuildFrameGraph(const FrameGraphAsset& asset, RHIFrameGraph& graph)
{
    FrameGraphBuildReport report;
    auto driver = Renderer::GetDriver();
    for (const auto& rt : asset.m_renderTargets) { // (:87-126)
        const uint32_t maxExtent = std::max(rt.m_width, rt.m_height);
        const uint32_t numMips = std::min(rt.m_maxMipLevel, rt.m_bGenerateMips ? (uint32_t)std::floor(std::log2((float)maxExtent)) + 1 : 1u);
        auto target = driver->CreateRenderTarget({ (int32_t)rt.m_width, (int32_t)rt.m_height }, numMips, canonical_format(rt.m_format));
        if (!target) continue;
        graph.SetRenderTarget(rt.m_name, target);
        report.m_renderTargets++;
    }
    for (const auto& v : asset.m_values) graph.SetValue(v.first, v.second); // (:128-131)
    for (const auto& n : asset.m_nodes) { // (:151-199)
        auto pNewNode = FrameGraphBuilder::CreateNode(n.m_name);
        if (!pNewNode) { // "FrameGraph Node %s is not implemented!" (:155-159)
            report.m_nodesNotImplemented++;
            report.m_notImplemented.push_back(n.m_name);
            continue;
        }
        pNewNode->SetTag(n.m_tag.empty() ? n.m_name : n.m_tag);
        for (const auto& p : n.m_vectors) pNewNode->SetVec4(p.first, p.second);
        for (const auto& p : n.m_floats) pNewNode->SetFloat(p.first, p.second);
        for (const auto& p : n.m_strings) pNewNode->SetString(p.first, p.second);
        for (const auto& p : n.m_renderTargets) {
            if (auto target = graph.GetRenderTarget(p.second)) pNewNode->SetRHIResource(p.first, target);
            else if (auto tex = graph.GetSampler(p.second)) pNewNode->SetRHIResource(p.first, tex);
            else { pNewNode->SetRHIResource_Unresolved(p.first, p.second); report.m_unresolvedTargets++; } // per-frame targets, resolved by name later (:190-195)
        }
        graph.AddNode(pNewNode);
        report.m_nodesCreated++;
    }
    return report;
}
