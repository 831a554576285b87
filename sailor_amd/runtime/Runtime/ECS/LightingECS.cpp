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

// ---- which light slots one Tick uploads, as contiguous runs (the boundary contract of row L7: every run becomes ONE UpdateShaderBinding, and the HIP backend
// re-derives the prepared views of exactly those slots -- sailor_hip_prepare_lights(first, count)).  The rules are the reference's (ECS/LightingECS.cpp:93-192),
// restated around two small pieces of state instead of transliterated: the list of static-owner runs the loop steps over, and the run of packed
// records that is currently open.  oracle/oracle.py:lighting_tick_runs is an independent restatement; tests/test_host_cpu.py and
// tests/test_runtime_gpu.py::test_lighting_ecs_streams_dirty_runs_into_the_light_ssbo hold the two against each other slot for slot.
namespace {

// Slots whose owner is static are packed once -- on the pass that first meets them -- and stepped over from the next Tick on, dirty or not.  The list holds them
// as (first slot, length) runs in slot order; `passed` counts the runs the current pass has stepped over or opened.
struct StaticRuns {
    std::vector<std::pair<uint32_t, uint32_t>>& runs;
    uint32_t passed = 0;

    // (:95-103) if the next run starts at `index`, jump behind it; false = the jump left the array.  (ONE run per loop step, as the reference: two runs that touch
    // are stepped over in two steps.)
    bool StepOver(size_t& index, size_t num)
    {
        if (passed < runs.size() && index == runs[passed].first) {
            index += runs[passed].second;
            if (index >= num) return false;
            passed++;
        }
        return true;
    }

    // (:108-146) slot `index` has a static owner and was not stepped over: it extends the run just passed when it follows it directly, opens a run of its own in
    // the gap between two later runs, or is appended.  (The gap's lower bound uses the length of the run just passed for EVERY gap -- the reference's arithmetic,
    // kept because it decides which later Ticks step over what.)
    void Note(size_t index)
    {
        if (passed > 0) {
            std::pair<uint32_t, uint32_t>& last = runs[passed - 1];
            if (index == (size_t)last.first + last.second) { last.second++; return; }
            for (size_t i = passed - 1; i + 1 < runs.size(); i++) {
                const uint32_t gapBegin = runs[i].first + last.second, gapEnd = runs[i + 1].first;
                if (index > gapBegin && index < gapEnd) {
                    runs.insert(runs.begin() + (long)(i + 1), std::make_pair((uint32_t)index, 1u));
                    passed++;
                    return;
                }
            }
        }
        runs.emplace_back((uint32_t)index, 1u);
        passed++;
    }
};

// The run of freshly packed records that is open: closed -- one copy -- when a slot that needs no upload follows it, or at the last slot.
struct OpenRun {
    std::vector<LightUploadRun>& out;
    size_t start = 0, count = 0;
    bool open = false;
    void Add(size_t index) { if (!open) { open = true; start = index; } count++; }
    void Close() { open = false; }
    void FlushIf(bool atEnd) { if ((!open || atEnd) && count > 0) { out.push_back({ start, count }); count = 0; } }
};

} // namespace

// Two consequences of where the reference's `continue` sits (:148-149), reproduced because they decide what reaches the SSBO: an INACTIVE slot neither closes nor
// flushes the open run -- a dirty run that continues behind it is written as one copy whose later records land one slot early -- and a run still open when the
// last slot is inactive (or when the static runs step past the end) is never written although its lights were marked clean.
std::vector<LightUploadRun> LightingECS::CollectDirtyRuns(std::vector<LightData>& components, std::vector<std::pair<uint32_t, uint32_t>>& skipList,
                                                          std::vector<LightShaderData>& records)
{
    std::vector<LightUploadRun> runs;
    StaticRuns statics { skipList };
    OpenRun run { runs };
    const size_t num = components.size();
    for (size_t index = 0; index < num; index++) {
        if (!statics.StepOver(index, num)) break;
        LightData& data = components[index];
        if (data.m_ownerMobility == EMobilityType::Static) statics.Note(index);
        if (!data.m_bIsActive) continue;
        const bool upload = data.m_bIsDirty || data.m_frameLastChange < data.m_ownerFrameLastChange; // (:152) dirty, or the owner moved after the last packing
        if (upload) {
            LightShaderData shaderData;
            sailor_host_pack_light((uint32_t)data.m_type, (uint32_t)data.m_shadowType, data.m_worldPosition, data.m_direction, data.m_intensity,
                                   data.m_attenuation, data.m_cutOff, data.m_bounds, &shaderData); // (:162-172)
            records.push_back(shaderData);
            run.Add(index);
            data.m_frameLastChange = data.m_ownerFrameLastChange; // (:174)
            data.m_bIsDirty = false;
        } else run.Close();
        run.FlushIf(index == num - 1); // (:182-191) one copy per run
    }
    records.resize(records.size() - run.count); // a run that was never flushed is never written
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
