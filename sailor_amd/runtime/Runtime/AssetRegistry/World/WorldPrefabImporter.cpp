#include "WorldPrefabImporter.h"
#include <cstdlib>
#include <cstring>

using namespace Sailor;

// ---- the YAML subset -------------------------------------------------------------------------------------------------------------
namespace {

struct Line { int indent; std::string text; int number; };

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

// position of the `:` that ends a map key (followed by a space or the end of the line, outside quotes), or npos
size_t key_colon(const std::string& text)
{
    bool sq = false, dq = false;
    for (size_t i = 0; i < text.size(); i++) {
        const char c = text[i];
        if (c == '\'' && !dq) sq = !sq;
        else if (c == '"' && !sq) dq = !dq;
        else if (c == ':' && !sq && !dq && (i + 1 == text.size() || text[i + 1] == ' ')) return i;
    }
    return std::string::npos;
}

YamlNode scalar_node(const std::string& raw)
{
    YamlNode n;
    std::string v = trim(raw);
    if (v.empty() || v == "~" || v == "null") return n;
    if (v.size() >= 2 && v.front() == '[' && v.back() == ']') { // flow sequence of scalars
        n.m_type = YamlNode::EType::Sequence;
        std::string item;
        for (size_t i = 1; i + 1 < v.size(); i++) {
            if (v[i] == ',') { n.m_sequence.push_back(scalar_node(item)); item.clear(); }
            else item += v[i];
        }
        if (!trim(item).empty()) n.m_sequence.push_back(scalar_node(item));
        return n;
    }
    if (v.size() >= 2 && ((v.front() == '"' && v.back() == '"') || (v.front() == '\'' && v.back() == '\''))) v = v.substr(1, v.size() - 2);
    n.m_type = YamlNode::EType::Scalar;
    n.m_scalar = v;
    return n;
}

struct Parser {
    std::vector<Line> lines;
    size_t cur = 0;
    std::string error;

    bool fail(const std::string& what)
    {
        if (error.empty()) error = "line " + std::to_string(cur < lines.size() ? lines[cur].number : 0) + ": " + what;
        return false;
    }

    // the block that starts at lines[cur], whose entries sit at exactly `indent`
    bool parse_block(int indent, YamlNode& out)
    {
        if (cur >= lines.size() || lines[cur].indent != indent) return fail("unexpected indentation");
        const bool sequence = lines[cur].text == "-" || lines[cur].text.rfind("- ", 0) == 0;
        if (sequence) {
            out.m_type = YamlNode::EType::Sequence;
            while (cur < lines.size() && lines[cur].indent == indent && (lines[cur].text == "-" || lines[cur].text.rfind("- ", 0) == 0)) {
                const std::string rest = trim(lines[cur].text.substr(1));
                out.m_sequence.emplace_back();
                YamlNode& item = out.m_sequence.back();
                if (rest.empty()) { // the item is the block below
                    cur++;
                    if (cur < lines.size() && lines[cur].indent > indent) { if (!parse_block(lines[cur].indent, item)) return false; }
                } else if (key_colon(rest) != std::string::npos) { // `- key: value`: a map whose first entry shares the dash's line
                    const int inner = indent + (int)(lines[cur].text.size() - rest.size());
                    lines[cur].indent = inner;
                    lines[cur].text = rest;
                    if (!parse_block(inner, item)) return false;
                } else {
                    item = scalar_node(rest);
                    cur++;
                }
            }
            if (cur < lines.size() && lines[cur].indent > indent) return fail("unexpected indentation");
            return true;
        }
        out.m_type = YamlNode::EType::Map;
        while (cur < lines.size() && lines[cur].indent == indent) {
            const std::string& text = lines[cur].text;
            if (text == "-" || text.rfind("- ", 0) == 0) break; // a sequence of the enclosing key at the same indentation ends this map
            const size_t colon = key_colon(text);
            if (colon == std::string::npos) return fail("expected `key: value`");
            std::string key = trim(text.substr(0, colon));
            if (key.size() >= 2 && (key.front() == '"' || key.front() == '\'')) key = key.substr(1, key.size() - 2);
            const std::string value = trim(text.substr(colon + 1));
            out.m_map.emplace_back(key, YamlNode());
            cur++;
            if (!value.empty()) { out.m_map.back().second = scalar_node(value); continue; }
            if (cur < lines.size() && lines[cur].indent > indent) {
                YamlNode child;
                if (!parse_block(lines[cur].indent, child)) return false;
                out.m_map.back().second = std::move(child);
            } else if (cur < lines.size() && lines[cur].indent == indent && (lines[cur].text == "-" || lines[cur].text.rfind("- ", 0) == 0)) {
                YamlNode child; // `key:` followed by `- item` at the key's own indentation
                if (!parse_block(indent, child)) return false;
                out.m_map.back().second = std::move(child);
            }
        }
        if (cur < lines.size() && lines[cur].indent > indent) return fail("unexpected indentation");
        return true;
    }
};

} // namespace

const YamlNode* YamlNode::Find(const std::string& key) const
{
    for (const auto& e : m_map) if (e.first == key) return &e.second;
    return nullptr;
}

float YamlNode::AsFloat(float fallback) const
{
    if (m_type != EType::Scalar) return fallback;
    char* end = nullptr;
    const float v = strtof(m_scalar.c_str(), &end);
    return end == m_scalar.c_str() ? fallback : v;
}

bool YamlNode::AsFloats(float* out, int count) const
{
    if (m_type != EType::Sequence) return false;
    for (int i = 0; i < count && i < (int)m_sequence.size(); i++) out[i] = m_sequence[i].AsFloat(out[i]);
    return true;
}

bool Sailor::ParseYamlSubset(const std::string& text, YamlNode& outRoot, std::string* outError)
{
    Parser p;
    size_t pos = 0;
    int number = 0;
    while (pos <= text.size()) {
        const size_t nl = text.find('\n', pos);
        std::string raw = text.substr(pos, nl == std::string::npos ? std::string::npos : nl - pos);
        pos = nl == std::string::npos ? text.size() + 1 : nl + 1;
        number++;
        std::string line = strip_comment(raw);
        while (!line.empty() && (line.back() == ' ' || line.back() == '\t' || line.back() == '\r')) line.pop_back();
        if (line.empty() || line == "---" || line == "...") continue;
        int indent = 0;
        while (indent < (int)line.size() && line[indent] == ' ') indent++;
        if (line[indent] == '\t') { if (outError) *outError = "line " + std::to_string(number) + ": tab indentation"; return false; }
        p.lines.push_back({ indent, line.substr(indent), number });
    }
    outRoot = YamlNode();
    if (p.lines.empty()) return true;
    if (!p.parse_block(p.lines[0].indent, outRoot) || p.cur != p.lines.size()) {
        if (p.error.empty()) p.fail("unexpected indentation");
        if (outError) *outError = p.error;
        return false;
    }
    return true;
}

// ---- Prefab / WorldPrefab ----------------------------------------------------------------------------------------------------------
bool Prefab::Deserialize(const YamlNode& inData, std::string* outError)
{
    auto fail = [&](const std::string& what) { if (outError) *outError = what; return false; };
    const YamlNode* gos = inData.Find("gameObjects");
    const YamlNode* comps = inData.Find("components");
    if (!gos || !gos->IsSequence()) return fail("prefab without a gameObjects sequence");
    for (const auto& g : gos->m_sequence) { // ReflectedGameObject::Deserialize (PrefabImporter.cpp:33-42)
        if (!g.IsMap()) return fail("game object is not a map");
        ReflectedGameObject go;
        if (auto n = g.Find("name")) go.m_name = n->m_scalar;
        if (auto n = g.Find("position")) n->AsFloats(go.m_position, 4);
        if (auto n = g.Find("rotation")) n->AsFloats(go.m_rotation, 4);
        if (auto n = g.Find("scale")) n->AsFloats(go.m_scale, 4);
        if (auto n = g.Find("parentIndex")) go.m_parentIndex = (uint32_t)strtoull(n->m_scalar.c_str(), nullptr, 10);
        if (auto n = g.Find("instanceId")) go.m_instanceId = n->m_scalar;
        if (auto n = g.Find("components"))
            for (const auto& c : n->m_sequence) go.m_components.push_back((uint32_t)strtoul(c.m_scalar.c_str(), nullptr, 10));
        m_gameObjects.push_back(std::move(go));
    }
    if (comps && comps->IsSequence())
        for (const auto& c : comps->m_sequence) {
            ReflectedData d;
            if (auto n = c.Find("typename")) d.m_typename = n->m_scalar;
            if (auto n = c.Find("overrideProperties")) d.m_overrideProperties = *n;
            m_components.push_back(std::move(d));
        }
    for (const auto& go : m_gameObjects) {
        if (go.m_parentIndex != 0xFFFFFFFFu && go.m_parentIndex >= m_gameObjects.size()) return fail("parentIndex out of range in " + go.m_name);
        for (uint32_t c : go.m_components) if (c >= m_components.size()) return fail("component index out of range in " + go.m_name);
    }
    return true;
}

bool WorldPrefab::Deserialize(const std::string& yamlText, std::string* outError)
{
    YamlNode root;
    if (!ParseYamlSubset(yamlText, root, outError)) return false;
    if (auto n = root.Find("name")) m_name = n->m_scalar; // WorldPrefabImporter.cpp:36
    const YamlNode* prefabs = root.Find("prefabs");
    if (!prefabs || !prefabs->IsSequence()) { if (outError) *outError = "`prefabs` is not a sequence"; return false; } // :38
    for (const auto& p : prefabs->m_sequence) {
        Prefab prefab;
        if (!prefab.Deserialize(p, outError)) return false;
        m_gameObjects.push_back(std::move(prefab));
    }
    return true;
}

// ---- World::Instantiate, flattened -------------------------------------------------------------------------------------------------
static ELightType parse_light_type(const std::string& s, ELightType fallback)
{
    if (s == "Directional") return ELightType::Directional;
    if (s == "Point") return ELightType::Point;
    if (s == "Spot") return ELightType::Spot;
    if (s == "Area") return ELightType::Area;
    return fallback;
}

bool WorldScene::Instantiate(const WorldPrefab& world, std::string* outError)
{
    m_name = world.m_name;
    for (const Prefab& prefab : world.m_gameObjects) {
        const uint32_t base = (uint32_t)m_gameObjects.size();
        for (const auto& r : prefab.m_gameObjects) { // Engine/World.cpp:160-190 -- objects, then :209-224 parents by index inside the prefab
            GameObject go;
            go.m_name = r.m_name;
            memcpy(go.m_transform.position, r.m_position, 16);
            memcpy(go.m_transform.rotation, r.m_rotation, 16);
            memcpy(go.m_transform.scale, r.m_scale, 16);
            go.m_parent = r.m_parentIndex == 0xFFFFFFFFu ? 0xFFFFFFFFu : base + r.m_parentIndex;
            float local[16];
            sailor_host_transform_matrix(&go.m_transform, local);
            if (go.m_parent != 0xFFFFFFFFu) {
                if (go.m_parent >= m_gameObjects.size()) { if (outError) *outError = "parent of " + r.m_name + " comes after it"; return false; }
                sailor_host_mat4_mul(m_gameObjects[go.m_parent].m_world, local, go.m_world); // TransformECS: parent world * local
            } else memcpy(go.m_world, local, sizeof local);
            const uint32_t owner = (uint32_t)m_gameObjects.size();
            for (uint32_t ci : r.m_components) {
                const ReflectedData& c = prefab.m_components[ci];
                const YamlNode& props = c.m_overrideProperties;
                go.m_componentTypes.push_back(c.m_typename);
                if (c.m_typename == "Sailor::CameraComponent") { // Components/CameraComponent.h:55-62
                    Camera cam;
                    cam.m_owner = owner;
                    if (auto n = props.Find("fov")) cam.m_fov = n->AsFloat(cam.m_fov);
                    if (auto n = props.Find("zNear")) cam.m_zNear = n->AsFloat(cam.m_zNear);
                    if (auto n = props.Find("zFar")) cam.m_zFar = n->AsFloat(cam.m_zFar);
                    m_cameras.push_back(cam);
                } else if (c.m_typename == "Sailor::LightComponent") { // Components/LightComponent.h:51-64
                    Light l;
                    l.m_owner = owner;
                    if (auto n = props.Find("intensity")) n->AsFloats(l.m_data.m_intensity, 3);
                    if (auto n = props.Find("attenuation")) n->AsFloats(l.m_data.m_attenuation, 3);
                    if (auto n = props.Find("bounds")) n->AsFloats(l.m_data.m_bounds, 3);
                    if (auto n = props.Find("cutOff")) n->AsFloats(l.m_data.m_cutOff, 2);
                    if (auto n = props.Find("lightType")) l.m_data.m_type = parse_light_type(n->m_scalar, l.m_data.m_type);
                    // ECS/LightingECS.cpp:169-170: direction = world * vec4_Forward (0, 0, -1, 0), position = world[3]
                    for (int k = 0; k < 3; k++) { l.m_data.m_direction[k] = -go.m_world[2 * 4 + k]; l.m_data.m_worldPosition[k] = go.m_world[3 * 4 + k]; }
                    m_lights.push_back(l);
                } else if (c.m_typename == "Sailor::MeshRendererComponent") {
                    MeshRenderer m;
                    m.m_owner = owner;
                    if (auto model = props.Find("model")) if (auto id = model->Find("fileId")) m.m_modelFileId = id->m_scalar;
                    m_meshRenderers.push_back(m);
                } else m_otherComponents++;
            }
            m_gameObjects.push_back(std::move(go));
        }
    }
    return true;
}
