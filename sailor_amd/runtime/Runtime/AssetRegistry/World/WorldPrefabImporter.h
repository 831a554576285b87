// The `.world` scene description (Content/Editor.world) and what the path needs from it: mirrors WorldPrefab::Deserialize
// (Runtime/AssetRegistry/World/WorldPrefabImporter.cpp:34-49), Prefab / ReflectedGameObject::Deserialize (AssetRegistry/Prefab/PrefabImporter.cpp:33-59) and the
// game-object half of World::Instantiate (Engine/World.cpp:160-232: objects in file order, parents by index inside their prefab).
// The reference reads the file with yaml-cpp and builds components through its reflection registry; here a reader for the YAML subset the
// serializer writes (block maps and sequences, plain / quoted scalars, flow sequences) fills the flat arrays the path consumes: transforms +
// parent indices for the ECS sweep (K4), the camera (CameraComponent: fov, zNear, zFar) and the lights (LightComponent -> LightingECS's LightData).
// Mesh renderers keep their model's file id (the model importer and its bounds are outside the path).
#pragma once
#include <map>
#include <memory>
#include <string>
#include <utility>
#include <vector>
#include "../../ECS/LightingECS.h"

namespace Sailor {

// one node of the parsed text: scalar, ordered map or sequence
struct YamlNode {
    enum class EType { Null, Scalar, Map, Sequence } m_type = EType::Null;
    std::string m_scalar;
    std::vector<std::pair<std::string, YamlNode>> m_map;
    std::vector<YamlNode> m_sequence;
    const YamlNode* Find(const std::string& key) const;
    bool IsSequence() const { return m_type == EType::Sequence; }
    bool IsMap() const { return m_type == EType::Map; }
    float AsFloat(float fallback = 0.0f) const;
    // a sequence of up to `count` numbers (vec2 / vec3 / vec4 / quat as Core/YamlSerializable.h:250-335 writes them)
    bool AsFloats(float* out, int count) const;
};
bool ParseYamlSubset(const std::string& text, YamlNode& outRoot, std::string* outError = nullptr);

struct ReflectedData { // Core/Reflection.h: a component's type name and its overridden properties
    std::string m_typename;
    YamlNode m_overrideProperties;
};

class Prefab {
public:
    struct ReflectedGameObject { // AssetRegistry/Prefab/PrefabImporter.h:36-48
        std::string m_name;
        float m_position[4] = { 0, 0, 0, 1 };
        float m_rotation[4] = { 0, 0, 0, 1 }; // x, y, z, w (YamlSerializable.h:314-335)
        float m_scale[4] = { 1, 1, 1, 1 };
        uint32_t m_parentIndex = 0xFFFFFFFFu;
        std::string m_instanceId;
        std::vector<uint32_t> m_components;
    };
    bool Deserialize(const YamlNode& inData, std::string* outError);
    std::vector<ReflectedGameObject> m_gameObjects;
    std::vector<ReflectedData> m_components;
};

class WorldPrefab {
public:
    bool Deserialize(const std::string& yamlText, std::string* outError = nullptr);
    std::string m_name;
    std::vector<Prefab> m_gameObjects; // one prefab per root object (WorldPrefabImporter.h: TVector<PrefabPtr> m_gameObjects)
};

// What World::Instantiate leaves behind, flattened for the path
struct WorldScene {
    struct GameObject {
        std::string m_name;
        SailorTransform m_transform;
        uint32_t m_parent = 0xFFFFFFFFu; // index into m_gameObjects
        float m_world[16];               // parent chain * Transform::Matrix()
        std::vector<std::string> m_componentTypes;
    };
    struct Camera { uint32_t m_owner = 0; float m_fov = 90.0f, m_zNear = 0.1f, m_zFar = 50000.0f; }; // Components/CameraComponent.cpp:51-53 defaults
    struct MeshRenderer { uint32_t m_owner = 0; std::string m_modelFileId; };
    struct Light { uint32_t m_owner = 0; LightData m_data; };
    std::string m_name;
    std::vector<GameObject> m_gameObjects;
    std::vector<Camera> m_cameras;
    std::vector<Light> m_lights;
    std::vector<MeshRenderer> m_meshRenderers;
    uint32_t m_otherComponents = 0;
    bool Instantiate(const WorldPrefab& world, std::string* outError = nullptr);
};

} // namespace Sailor
