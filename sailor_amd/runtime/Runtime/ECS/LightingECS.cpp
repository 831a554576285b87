#include "LightingECS.h"
#include "../RHI/Renderer.h"
#include "../GraphicsDriver/HIP/HipGraphicsDriver.h"

using namespace Sailor;
using namespace Sailor::RHI;

LightingECS::LightingECS(uint32_t capacity)
{
    auto driver = Renderer::GetDriver();
    m_lightsData = driver->CreateShaderBindings();
    driver->AddSsboToShaderBindings(m_lightsData, "light", sizeof(LightShaderData), capacity, 0, true); // LightingECS.cpp:44
}

size_t LightingECS::RegisterComponent(const LightData& data)
{
    m_components.push_back(data);
    return m_components.size() - 1;
}

// LightingECS.cpp:93-192, literally -- including what follows from where its `continue` and its flush sit:
//   * a slot of a static owner joins the skip list the first time the loop meets it (it is still packed on that pass) and is stepped over from
//     the next Tick on, dirty or not; only ONE skip-list run is applied per loop step;
//   * an inactive slot leaves the loop body before the "close the run" check (:148-149), so it neither ends a run nor flushes one: a dirty run
//     that continues behind it is written as ONE copy whose later records land one slot early, and a run still open when the LAST slot is
//     inactive (or when the skip list steps past the end) is never written, although its lights were marked clean;
//   * a light is packed when it is dirty OR its owner's transform changed after the frame it was last packed for (:152).
std::vector<LightUploadRun> LightingECS::CollectDirtyRuns(std::vector<LightData>& components, std::vector<std::pair<uint32_t, uint32_t>>& skipList,
                                                          std::vector<LightShaderData>& records)
{
    std::vector<LightUploadRun> runs;
    size_t pending = 0; // records packed since the last copy (the reference's shaderDataBatch)
    bool bShouldWrite = true;
    size_t startIndex = 0;
    uint32_t skipIndex = 0;
    const size_t num = components.size();
    for (size_t index = 0; index < num; index++) {
        if (skipIndex < skipList.size() && index == skipList[skipIndex].first) { // :95-103
            index += skipList[skipIndex].second;
            if (index >= num) break;
            skipIndex++;
        }
        LightData& data = components[index];
        if (data.m_ownerMobility == EMobilityType::Static) { // :108-146
            bool bPlaced = false;
            if (skipIndex > 0 && index == (size_t)skipList[skipIndex - 1].first + skipList[skipIndex - 1].second) { // grows the run it follows
                skipList[skipIndex - 1].second++;
                bPlaced = true;
            }
            if (!bPlaced && skipIndex > 0) { // between two runs (the bound uses run skipIndex-1's length for every i, as the reference does)
                for (int32_t i = (int32_t)skipIndex - 1; i < (int32_t)skipList.size() - 1; i++) {
                    const uint32_t start = skipList[i].first + skipList[skipIndex - 1].second, end = skipList[i + 1].first;
                    if (index > start && index < end) {
                        skipList.insert(skipList.begin() + (i + 1), std::make_pair((uint32_t)index, 1u));
                        bPlaced = true;
                        skipIndex++;
                        break;
                    }
                }
            }
            if (!bPlaced) {
                skipList.emplace_back((uint32_t)index, 1u);
                skipIndex++;
            }
        }
        if (!data.m_bIsActive) continue; // :148-149
        if (data.m_bIsDirty || data.m_frameLastChange < data.m_ownerFrameLastChange) { // :152
            if (bShouldWrite) { bShouldWrite = false; startIndex = index; }
            LightShaderData shaderData;
            sailor_host_pack_light((uint32_t)data.m_type, (uint32_t)data.m_shadowType, data.m_worldPosition, data.m_direction, data.m_intensity,
                                   data.m_attenuation, data.m_cutOff, data.m_bounds, &shaderData); // :162-172
            records.push_back(shaderData);
            pending++;
            data.m_frameLastChange = data.m_ownerFrameLastChange; // :174
            data.m_bIsDirty = false;
        } else bShouldWrite = true;
        if ((bShouldWrite || index == num - 1) && pending > 0) { // :182-191: one copy per run
            runs.push_back({ startIndex, pending });
            pending = 0;
        }
    }
    records.resize(records.size() - pending); // a run that was never closed is never written
    return runs;
}

void LightingECS::Tick(RHICommandListPtr cmdList)
{
    auto binding = m_lightsData->GetOrAddShaderBinding("light");
    std::vector<LightShaderData> records;
    m_lastUploads = CollectDirtyRuns(m_components, m_skipList, records);
    size_t at = 0;
    for (const LightUploadRun& run : m_lastUploads) {
        Renderer::GetDriverCommands()->UpdateShaderBinding(cmdList, binding, records.data() + at, sizeof(LightShaderData) * run.m_count,
                                                           binding->GetBufferOffset() + sizeof(LightShaderData) * run.m_startIndex);
        at += run.m_count;
    }
    m_packedCount = m_components.size(); // FillLightingData: m_totalNumLights counts every slot, active or not (:404)
}

void LightingECS::SetPacked(RHICommandListPtr cmdList, const LightShaderData* records, size_t count)
{
    auto binding = m_lightsData->GetOrAddShaderBinding("light");
    if (count) Renderer::GetDriverCommands()->UpdateShaderBinding(cmdList, binding, records, sizeof(LightShaderData) * count, 0);
    m_packedCount = count;
}

void LightingECS::SetShadowMaps(const TVector<RHITexturePtr>& maps, const float* lightsMatrices64)
{
    auto driver = Renderer::GetDriver();
    driver->AddSamplerToShaderBindings(m_lightsData, "shadowMaps", maps, 8);                                               // LightingECS.cpp:71
    auto b = driver->AddBufferToShaderBindings(m_lightsData, "lightsMatrices", 256, 6, EShaderBindingType::UniformBuffer); // :74 (SSBO there)
    memcpy(b->m_hostCopy.data(), lightsMatrices64, 256);
}

void LightingECS::FillLightingData(RHISceneViewSnapshot& snapshot) const
{
    snapshot.m_totalNumLights = (uint32_t)m_packedCount; // includes inactive slots in the reference (LightingECS.cpp:404)
    snapshot.m_rhiLightsData = m_lightsData;
}

EcsSweepSystem::EcsSweepSystem(const SailorTransform* transforms, const uint32_t* parent, const SailorAABB* localAabb, uint32_t count,
                               const uint32_t* levelOffsets, uint32_t numLevels)
    : m_count(count), m_levelOffsets(levelOffsets, levelOffsets + numLevels + 1)
{
    auto driver = Renderer::GetDriver();
    auto cmds = Renderer::GetDriverCommands();
    auto cmd = driver->CreateCommandList();
    m_transforms = driver->CreateBuffer(sizeof(SailorTransform) * (size_t)count);
    m_parent = driver->CreateBuffer(4 * (size_t)count);
    m_localAabb = driver->CreateBuffer(sizeof(SailorAABB) * (size_t)count);
    m_world = driver->CreateBuffer(64 * (size_t)count);
    m_worldAabb = driver->CreateBuffer(sizeof(SailorAABB) * (size_t)count);
    m_visibility = driver->CreateBuffer(8 * (((size_t)count + 63) / 64));
    cmds->UpdateBuffer(cmd, m_transforms, transforms, sizeof(SailorTransform) * (size_t)count);
    cmds->UpdateBuffer(cmd, m_parent, parent, 4 * (size_t)count);
    cmds->UpdateBuffer(cmd, m_localAabb, localAabb, sizeof(SailorAABB) * (size_t)count);
    driver->SubmitCommandList(cmd);
}

int EcsSweepSystem::Tick(const float* cameraWorld, float aspect, float fovDegrees, float zNear, float zFar)
{
    float planes[24];
    sailor_host_extract_frustum_planes(cameraWorld, aspect, fovDegrees, zNear, zFar, planes, nullptr); // RHI/SceneView.cpp:158
    auto* hip = dynamic_cast<GraphicsDriver::HIP::HipGraphicsDriver*>(Renderer::GetDriver());
    if (!hip) return SAILOR_HIP_ERR_UNSUPPORTED;
    return sailor_hip_ecs_sweep(hip->GetContext(), m_count, (const SailorTransform*)m_transforms->m_hip.m_devicePtr, (const uint32_t*)m_parent->m_hip.m_devicePtr,
                                m_levelOffsets.data(), (uint32_t)m_levelOffsets.size() - 1, (const SailorAABB*)m_localAabb->m_hip.m_devicePtr, planes,
                                (float*)m_world->m_hip.m_devicePtr, (SailorAABB*)m_worldAabb->m_hip.m_devicePtr, (uint64_t*)m_visibility->m_hip.m_devicePtr);
}
