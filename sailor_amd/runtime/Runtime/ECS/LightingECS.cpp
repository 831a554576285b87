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

void LightingECS::Tick(RHICommandListPtr cmdList)
{
    auto binding = m_lightsData->GetOrAddShaderBinding("light");
    std::vector<LightShaderData> batch;
    bool bShouldWrite = true;
    size_t startIndex = 0;
    for (size_t index = 0; index < m_components.size(); index++) {
        auto& data = m_components[index];
        if (data.m_bIsDirty) {
            if (bShouldWrite) { bShouldWrite = false; startIndex = index; }
            LightShaderData shaderData;
            sailor_host_pack_light((uint32_t)data.m_type, (uint32_t)data.m_shadowType, data.m_worldPosition, data.m_direction, data.m_intensity,
                                   data.m_attenuation, data.m_cutOff, data.m_bounds, &shaderData); // LightingECS.cpp:163-172
            batch.push_back(shaderData);
            data.m_bIsDirty = false;
        } else bShouldWrite = true;
        if ((bShouldWrite || index == m_components.size() - 1) && !batch.empty()) { // LightingECS.cpp:182-191: one copy per dirty run
            Renderer::GetDriverCommands()->UpdateShaderBinding(cmdList, binding, batch.data(), sizeof(LightShaderData) * batch.size(),
                                                               binding->GetBufferOffset() + sizeof(LightShaderData) * startIndex);
            batch.clear();
        }
    }
    m_packedCount = m_components.size();
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
