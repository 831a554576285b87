// Mirrors the GPU-facing half of Runtime/ECS/LightingECS.{h,cpp}: the `light` SSBO (capacity LightsMaxNum records of
// LightShaderData, LightingECS.cpp:44), Tick's streaming of dirty runs (LightingECS.cpp:79-195: the skip list of static lights, inactive
// slots, dirtiness from the component or from its owner's transform, one UpdateShaderBinding per contiguous run) and FillLightingData
// (:373-406).  The owning game object is reduced to the two things Tick asks it: its mobility and the frame its transform last changed.
#pragma once
#include <vector>
#include "../RHI/GraphicsDriver.h"
#include "../RHI/SceneView.h"

namespace Sailor {

enum class ELightType : uint32_t { Directional = 0, Point = 1, Spot = 2, Area = 3 };          // Engine/Types.h:31-37
enum class EShadowType : uint32_t { None = 0, PCF = 1, EVSM = 2 };                            // RHI/SceneView.h:13-18
enum class EMobilityType : uint8_t { Static = 0, Stationary = 1, Dynamic = 2 };               // Engine/Types.h:24-29

struct LightData { // ECS/LightingECS.h:18-33 (+ the owner transform's position / forward vector)
    float m_intensity[3] = { 100.0f, 100.0f, 100.0f };
    float m_attenuation[3] = { 1.0f, 0.022f, 0.0019f };
    float m_bounds[3] = { 100.0f, 100.0f, 100.0f };
    float m_cutOff[2] = { 30.0f, 45.0f }; // degrees
    ELightType m_type = ELightType::Point;
    EShadowType m_shadowType = EShadowType::PCF;
    float m_worldPosition[3] = { 0, 0, 0 };
    float m_direction[3] = { 0, 0, -1 };
    bool m_bIsDirty = true;                                   // ECS/ECS.h:38 (LightComponent marks a new light dirty)
    bool m_bIsActive = true;                                  // ECS/ECS.h:37
    size_t m_frameLastChange = 0;                             // ECS/ECS.h:36: the owner's frame this record was packed for
    size_t m_ownerFrameLastChange = 0;                        // GameObject::GetFrameLastChange(): the frame the owner's transform last changed
    EMobilityType m_ownerMobility = EMobilityType::Stationary; // Engine/GameObject.h:124
};

// One UpdateShaderBinding of LightingECS::Tick: `m_count` consecutive records written from record slot `m_startIndex` on.
struct LightUploadRun {
    size_t m_startIndex = 0;
    size_t m_count = 0;
};

class LightingECS {
public:
    static constexpr uint32_t LightsMaxNum = SAILOR_LIGHTS_MAX_NUM; // ECS/LightingECS.h:54
    using LightShaderData = SailorLightShaderData;                  // ECS/LightingECS.h:71-81

    explicit LightingECS(uint32_t capacity = LightsMaxNum);
    size_t RegisterComponent(const LightData& data);
    LightData& GetComponentData(size_t index) { return m_components[index]; }
    size_t Num() const { return m_components.size(); }
    // packs dirty lights and records one UpdateShaderBinding per contiguous dirty run into cmdList
    void Tick(RHI::RHICommandListPtr cmdList);
    // The loop of Tick without a driver (LightingECS.cpp:93-192): packs what is dirty, clears the flags, keeps the skip list of static lights up to
    // date and returns the runs in the order Tick issues them; `records` receives the runs' records back to back.
    static std::vector<LightUploadRun> CollectDirtyRuns(std::vector<LightData>& components, std::vector<std::pair<uint32_t, uint32_t>>& skipList,
                                                        std::vector<LightShaderData>& records);
    const std::vector<LightUploadRun>& GetLastUploads() const { return m_lastUploads; } // what the last Tick recorded
    // hands an already packed array over (synthetic frames): one upload
    void SetPacked(RHI::RHICommandListPtr cmdList, const LightShaderData* records, size_t count);
    void SetShadowMaps(const TVector<RHI::RHITexturePtr>& maps, const float* lightsMatrices64);
    void FillLightingData(RHI::RHISceneViewSnapshot& snapshot) const; // LightingECS.cpp:403-405
    RHI::RHIShaderBindingSetPtr GetLightsData() const { return m_lightsData; }

private:
    std::vector<LightData> m_components;
    std::vector<std::pair<uint32_t, uint32_t>> m_skipList; // (first slot, count) runs of static lights Tick steps over (LightingECS.h:106)
    std::vector<LightUploadRun> m_lastUploads;
    size_t m_packedCount = 0;
    RHI::RHIShaderBindingSetPtr m_lightsData;
};

// Flat form of TransformECS + StaticMeshRendererECS for the sweep (ECS/TransformECS.cpp:144-212, StaticMeshRendererECS.cpp:40-58)
class EcsSweepSystem {
public:
    EcsSweepSystem(const SailorTransform* transforms, const uint32_t* parent, const SailorAABB* localAabb, uint32_t count,
                   const uint32_t* levelOffsets, uint32_t numLevels);
    int Tick(const float* cameraWorld, float aspect, float fovDegrees, float zNear, float zFar);
    RHI::RHIBufferPtr m_transforms, m_parent, m_localAabb, m_world, m_worldAabb, m_visibility;
    uint32_t m_count = 0;
private:
    std::vector<uint32_t> m_levelOffsets;
};

} // namespace Sailor
