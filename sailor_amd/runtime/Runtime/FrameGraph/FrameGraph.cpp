#include <cmath>
#include "FrameGraphNode.h"
#include "LightCullingNode.h"
#include "RHIFrameGraph.h"
#include "../RHI/Renderer.h"

using namespace Sailor;
using namespace Sailor::RHI;
using namespace Sailor::Framegraph;

// ---- FrameGraphBuilder (FrameGraph/FrameGraphNode.cpp:16-41) -----------------------------------------------------------------
std::map<std::string, std::function<FrameGraphNodePtr(void)>>& FrameGraphBuilder::Registry()
{
    static std::map<std::string, std::function<FrameGraphNodePtr(void)>> s_nodes; // function-local: safe during static init
    return s_nodes;
}

void FrameGraphBuilder::RegisterFrameGraphNode(const std::string& nodeName, std::function<FrameGraphNodePtr(void)> factoryMethod)
{
    Registry()[nodeName] = std::move(factoryMethod);
}

FrameGraphNodePtr FrameGraphBuilder::CreateNode(const std::string& nodeName)
{
    auto it = Registry().find(nodeName);
    return it == Registry().end() ? FrameGraphNodePtr() : it->second();
}

bool FrameGraphBuilder::IsRegistered(const std::string& nodeName) { return Registry().count(nodeName) != 0; }

// force the registration objects of the path's nodes into the library
template class Sailor::Framegraph::TFrameGraphNode<LightCullingNode>;
template class Sailor::Framegraph::TFrameGraphNode<RenderSceneNode>;
template class Sailor::Framegraph::TFrameGraphNode<LinearizeDepthNode>;
template class Sailor::Framegraph::TFrameGraphNode<EnvironmentNode>;
template class Sailor::Framegraph::TFrameGraphNode<DepthHighZNode>;

// ---- RHIFrameGraph ----------------------------------------------------------------------------------------------------------
UboFrameData RHIFrameGraph::FillFrameData(RHICommandListPtr transferCmdList, RHISceneViewSnapshot& snapshot, float deltaTime, float worldTime) const
{
    // RHIFrameGraph.cpp:56-70
    UboFrameData frameData {};
    snapshot.m_frameBindings = Renderer::GetDriver()->CreateShaderBindings();
    Renderer::GetDriver()->AddBufferToShaderBindings(snapshot.m_frameBindings, "frameData", sizeof(UboFrameData), 0, EShaderBindingType::UniformBuffer);
    sailor_host_fill_frame_data(snapshot.m_camera.m_world, snapshot.m_camera.m_fov, snapshot.m_camera.m_aspect, snapshot.m_camera.m_zNear,
                                snapshot.m_camera.m_zFar, m_viewport.x, m_viewport.y, worldTime, deltaTime, &frameData);
    Renderer::GetDriverCommands()->UpdateShaderBinding(transferCmdList, snapshot.m_frameBindings->GetOrAddShaderBinding("frameData"), &frameData, sizeof(frameData));
    return frameData;
}

void RHIFrameGraph::Process(RHISceneViewSnapshot& snapshot)
{
    auto driver = Renderer::GetDriver();
    auto transferCmdList = driver->CreateCommandList();
    auto cmdList = driver->CreateCommandList();
    FillFrameData(transferCmdList, snapshot, snapshot.m_deltaTime, snapshot.m_currentTime);
    if (snapshot.m_rhiLightsData) { // RHIFrameGraph.cpp:128-163: the IBL samplers and the AO target join the lights set (bindings 3, 4, 5, 9)
        struct { const char* name; RHITexturePtr tex; uint32_t binding; } ibl[] = {
            { "g_irradianceCubemap", GetSampler("g_irradianceCubemap"), 3 }, { "g_brdfSampler", GetSampler("g_brdfSampler"), 4 },
            { "g_envCubemap", GetSampler("g_envCubemap"), 5 }, { "g_aoSampler", GetRenderTarget("g_AO"), 9 } };
        for (auto& e : ibl) {
            if (!e.tex) continue;
            auto b = snapshot.m_rhiLightsData->Find(e.name);
            if (!b || b->m_textures.empty() || b->m_textures[0].GetRawPtr() != e.tex.GetRawPtr())
                driver->AddSamplerToShaderBindings(snapshot.m_rhiLightsData, e.name, e.tex, e.binding);
        }
    }
    for (auto& node : m_graph) node->Prepare(this, snapshot);
    for (auto& node : m_graph) node->Process(this, transferCmdList, cmdList, snapshot); // RHIFrameGraph.cpp:250-252
    driver->SubmitCommandList(transferCmdList);
    driver->SubmitCommandList(cmdList);
}

void RHIFrameGraph::Clear()
{
    for (auto& node : m_graph) node->Clear();
    m_graph.clear();
    m_renderTargets.clear();
}

// ---- EnvironmentNode (FrameGraph/EnvironmentNode.cpp:19-281) --------------------------------------------------------------------
const char* EnvironmentNode::m_name = "Environment";

void EnvironmentNode::Process(RHIFrameGraphPtr frameGraph, RHICommandListPtr, RHICommandListPtr commandList, const RHISceneViewSnapshot&)
{
    auto driver = Renderer::GetDriver();
    auto commands = Renderer::GetDriverCommands();
    commands->BeginDebugRegion(commandList, GetName());
    if (!m_pComputeBrdfShader) { m_pComputeBrdfShader = driver->CreateShader("Shaders/ComputeBrdfLut.shader"); m_computeBrdfBindings = driver->CreateShaderBindings(); }                 // (:31-39)
    if (!m_pComputeSpecularShader) { m_pComputeSpecularShader = driver->CreateShader("Shaders/ComputeEnvMap_IBL.shader"); m_computeSpecularBindings = driver->CreateShaderBindings(); } // (:41-49)
    if (!m_pComputeIrradianceShader) { m_pComputeIrradianceShader = driver->CreateShader("Shaders/ComputeIrradianceMap.shader"); m_computeIrradianceBindings = driver->CreateShaderBindings(); } // (:51-59)

    if (!m_brdfSampler) { // (:69-98)
        m_brdfSampler = driver->CreateRenderTarget({ (int32_t)BrdfLutSize, (int32_t)BrdfLutSize }, 1, EFormat::R32G32_SFLOAT);
        commands->ImageMemoryBarrier(commandList, m_brdfSampler, EImageLayout::ShaderReadOnlyOptimal);
        frameGraph->SetSampler("g_brdfSampler", m_brdfSampler);
        commands->BeginDebugRegion(commandList, "Generate Cook-Torrance BRDF 2D LUT for split-sum approximation");
        driver->AddStorageImageToShaderBindings(m_computeBrdfBindings, "dst", m_brdfSampler, 0);
        commands->ImageMemoryBarrier(commandList, m_brdfSampler, EImageLayout::ComputeWrite);
        commands->Dispatch(commandList, m_pComputeBrdfShader, (uint32_t)(m_brdfSampler->GetExtent().x / 32.0f), (uint32_t)(m_brdfSampler->GetExtent().y / 32.0f), 6u,
                           { m_computeBrdfBindings }, nullptr, 0);
        commands->ImageMemoryBarrier(commandList, m_brdfSampler, EImageLayout::ShaderReadOnlyOptimal);
        commands->EndDebugRegion(commandList);
    }

    if (m_bIsDirty) { // (:100-276)
        RHICubemapPtr rawEnvCubemap;
        if (m_envMapTexture) { // (:116-138) the panorama -> the raw cube and its mip chain
            rawEnvCubemap = driver->CreateCubemap({ (int32_t)EnvMapSize, (int32_t)EnvMapSize }, EnvMapLevels, EFormat::R32G32B32A32_SFLOAT);
            commands->ImageMemoryBarrier(commandList, rawEnvCubemap, EImageLayout::ShaderReadOnlyOptimal);
            commands->BeginDebugRegion(commandList, "Generate Raw Env Cubemap from Equirect");
            commands->ImageMemoryBarrier(commandList, rawEnvCubemap, EImageLayout::ComputeWrite);
            commands->ConvertEquirect2Cubemap(commandList, m_envMapTexture, rawEnvCubemap);
            commands->ImageMemoryBarrier(commandList, rawEnvCubemap, EImageLayout::TransferDstOptimal);
            commands->GenerateMipMaps(commandList, rawEnvCubemap);
            commands->EndDebugRegion(commandList);
            frameGraph->SetSampler("g_rawEnvCubemap", rawEnvCubemap); // not in the reference: lets the harness read the intermediate back
        } else {
            rawEnvCubemap = frameGraph->GetSampler("g_skyCubemap"); // (:139-142)
        }
        if (!rawEnvCubemap || !rawEnvCubemap->m_bCubemap) { commands->EndDebugRegion(commandList); return; } // (:143-146)
        const int32_t EnvMapSize = rawEnvCubemap->GetExtent().x;
        const uint32_t EnvMapLevels = rawEnvCubemap->GetMipLevels();
        const bool bShouldUpdateEnvCubemap = !m_envCubemap, bShouldUpdateIrradianceCubemap = !m_irradianceCubemap; // (:162-163)
        if (!bShouldUpdateEnvCubemap && !bShouldUpdateIrradianceCubemap) { // (:165-173)
            frameGraph->SetSampler("g_envCubemap", m_envCubemap);
            frameGraph->SetSampler("g_irradianceCubemap", m_irradianceCubemap);
            m_bIsDirty = false;
            commands->EndDebugRegion(commandList);
            return;
        }
        if (bShouldUpdateEnvCubemap) { // (:176-236)
            m_envCubemap = driver->CreateCubemap({ EnvMapSize, EnvMapSize }, EnvMapLevels, EFormat::R32G32B32A32_SFLOAT);
            frameGraph->SetSampler("g_envCubemap", m_envCubemap);
            commands->ImageMemoryBarrier(commandList, m_envCubemap, EImageLayout::General);
            commands->BeginDebugRegion(commandList, "Compute pre-filtered specular environment map");
            struct PushConstants { int32_t level {}; float roughness {}; };
            const uint32_t NumMipTailLevels = EnvMapLevels - 1;
            commands->ImageMemoryBarrier(commandList, rawEnvCubemap, EImageLayout::TransferSrcOptimal);
            commands->ImageMemoryBarrier(commandList, m_envCubemap, EImageLayout::TransferDstOptimal);
            commands->BlitImage(commandList, rawEnvCubemap, m_envCubemap, { 0, 0, EnvMapSize, EnvMapSize }, { 0, 0, EnvMapSize, EnvMapSize }); // (:200-203) mip 0
            commands->ImageMemoryBarrier(commandList, rawEnvCubemap, EImageLayout::ShaderReadOnlyOptimal);
            commands->ImageMemoryBarrier(commandList, m_envCubemap, EImageLayout::ComputeWrite);
            TVector<RHITexturePtr> envMapMips; // the mip tail (:208-212)
            for (uint32_t level = 1; level < EnvMapLevels; ++level) envMapMips.push_back(m_envCubemap->GetMipLevel(level));
            driver->AddSamplerToShaderBindings(m_computeSpecularBindings, "rawEnvMap", rawEnvCubemap, 0);
            driver->AddStorageImageToShaderBindings(m_computeSpecularBindings, "envMap", envMapMips, 1);
            const float deltaRoughness = 1.0f / std::max(float(NumMipTailLevels), 1.0f);
            for (uint32_t level = 1, size = (uint32_t)EnvMapSize / 2; level < EnvMapLevels; ++level, size /= 2) { // (:220-233)
                const uint32_t numGroups = std::max<uint32_t>(1u, size / 32u);
                const PushConstants pushConstants = { (int32_t)(level - 1u), level * deltaRoughness };
                commands->Dispatch(commandList, m_pComputeSpecularShader, numGroups, numGroups, 6u, { m_computeSpecularBindings }, &pushConstants, sizeof(PushConstants));
            }
            commands->EndDebugRegion(commandList);
        }
        if (bShouldUpdateIrradianceCubemap) { // (:238-273)
            int32_t irradianceSize = (int32_t)IrradianceMapSize;
            float overrideSize = 0.0f;
            if (TryGetFloat("IrradianceMapSize", overrideSize) && overrideSize >= 1.0f) irradianceSize = (int32_t)overrideSize; // test knob (65 536 samples per texel)
            m_irradianceCubemap = driver->CreateCubemap({ irradianceSize, irradianceSize }, 1, EFormat::R32G32B32A32_SFLOAT);
            commands->ImageMemoryBarrier(commandList, m_irradianceCubemap, EImageLayout::ShaderReadOnlyOptimal);
            frameGraph->SetSampler("g_irradianceCubemap", m_irradianceCubemap);
            commands->ImageMemoryBarrier(commandList, m_irradianceCubemap, EImageLayout::General);
            commands->BeginDebugRegion(commandList, "Compute diffuse irradiance cubemap");
            commands->ImageMemoryBarrier(commandList, m_envCubemap, EImageLayout::ShaderReadOnlyOptimal);
            commands->ImageMemoryBarrier(commandList, m_irradianceCubemap, EImageLayout::ComputeWrite);
            driver->AddSamplerToShaderBindings(m_computeIrradianceBindings, "envMap", m_envCubemap, 0);
            driver->AddStorageImageToShaderBindings(m_computeIrradianceBindings, "irradianceMap", m_irradianceCubemap, 1);
            commands->Dispatch(commandList, m_pComputeIrradianceShader, std::max(1u, (uint32_t)irradianceSize / 32u), std::max(1u, (uint32_t)irradianceSize / 32u), 6u,
                               { m_computeIrradianceBindings });
            commands->EndDebugRegion(commandList);
        }
        m_bIsDirty = false;
    }
    commands->EndDebugRegion(commandList);
}

void EnvironmentNode::Clear()
{
    m_pComputeIrradianceShader.Clear(); m_pComputeSpecularShader.Clear(); m_pComputeBrdfShader.Clear();
    m_computeIrradianceBindings.Clear(); m_computeSpecularBindings.Clear(); m_computeBrdfBindings.Clear();
    m_envCubemap.Clear(); m_irradianceCubemap.Clear(); m_brdfSampler.Clear(); m_envMapTexture.Clear();
}

// ---- DepthHighZNode (FrameGraph/DepthHighZNode.cpp:17-103) -----------------------------------------------------------------------
const char* DepthHighZNode::m_name = "DepthHighZ";

void DepthHighZNode::Process(RHIFrameGraphPtr frameGraph, RHICommandListPtr, RHICommandListPtr commandList, const RHISceneViewSnapshot&)
{
    auto driver = Renderer::GetDriver();
    auto commands = Renderer::GetDriverCommands();
    auto depthAttachment = GetRHIResource("src").DynamicCast<RHITexture>(); // (:28-32)
    if (!depthAttachment) depthAttachment = frameGraph->GetRenderTarget("DepthBuffer");
    auto highZRenderTarget = GetRHIResource("dst").DynamicCast<RHITexture>(); // (:34)
    if (!depthAttachment || !highZRenderTarget) return;
    if (!m_pComputeDepthHighZShader) m_pComputeDepthHighZShader = driver->CreateShader("Shaders/ComputeDepthHighZ.shader"); // (:38-44)
    if (m_computeDepthHighZBindings.empty()) { // (:51-67)
        m_computeDepthHighZBindings.resize(highZRenderTarget->GetMipLevels() - 1);
        for (uint32_t i = 0; i < highZRenderTarget->GetMipLevels() - 1; ++i) {
            auto readMipLevel = highZRenderTarget->GetMipLevel(i), writeMipLevel = highZRenderTarget->GetMipLevel(i + 1); // GetMipLayer
            m_computeDepthHighZBindings[i] = driver->CreateShaderBindings();
            driver->AddSamplerToShaderBindings(m_computeDepthHighZBindings[i], "inputDepth", readMipLevel, 0);
            driver->AddStorageImageToShaderBindings(m_computeDepthHighZBindings[i], "outputDepth", writeMipLevel, 1);
        }
        m_computePrepassDepthHighZBindings = driver->CreateShaderBindings();
        driver->AddSamplerToShaderBindings(m_computePrepassDepthHighZBindings, "inputDepth", depthAttachment, 0);
        driver->AddStorageImageToShaderBindings(m_computePrepassDepthHighZBindings, "outputDepth", highZRenderTarget->GetMipLevel(0), 1);
    }
    commands->BeginDebugRegion(commandList, GetName());
    commands->ImageMemoryBarrier(commandList, highZRenderTarget, EImageLayout::General);
    for (int32_t i = -1; i < (int32_t)highZRenderTarget->GetMipLevels() - 1; ++i) { // Depth Downscale (:78-96)
        const bool bFirst = i == -1;
        auto readMipLevel = bFirst ? depthAttachment : highZRenderTarget->GetMipLevel((uint32_t)i);
        auto writeMipLevel = highZRenderTarget->GetMipLevel((uint32_t)(i + 1));
        PushConstantsDownscale params {};
        params.m_outputSize[0] = (float)writeMipLevel->GetExtent().x; params.m_outputSize[1] = (float)writeMipLevel->GetExtent().y;
        commands->ImageMemoryBarrier(commandList, readMipLevel, EImageLayout::ShaderReadOnlyOptimal);
        commands->ImageMemoryBarrier(commandList, writeMipLevel, EImageLayout::ComputeWrite);
        commands->Dispatch(commandList, m_pComputeDepthHighZShader, (uint32_t)std::ceil(params.m_outputSize[0] / 8), (uint32_t)std::ceil(params.m_outputSize[1] / 8), 1u,
                           { bFirst ? m_computePrepassDepthHighZBindings : m_computeDepthHighZBindings[(size_t)i] }, &params, sizeof(PushConstantsDownscale));
    }
    commands->EndDebugRegion(commandList);
}

void DepthHighZNode::Clear()
{
    m_pComputeDepthHighZShader.Clear();
    m_computeDepthHighZBindings.clear();
    m_computePrepassDepthHighZBindings.Clear();
}

// ---- LinearizeDepthNode (FrameGraph/LinearizeDepthNode.cpp:18-109) -------------------------------------------------------------
const char* LinearizeDepthNode::m_name = "LinearizeDepth";

void LinearizeDepthNode::Process(RHIFrameGraphPtr frameGraph, RHICommandListPtr, RHICommandListPtr commandList, const RHISceneViewSnapshot& sceneView)
{
    auto driver = Renderer::GetDriver();
    auto commands = Renderer::GetDriverCommands();
    commands->BeginDebugRegion(commandList, GetName());

    auto depthAttachment = GetRHIResource("depthStencil").DynamicCast<RHITexture>(); // (:29-37)
    for (const auto& r : m_unresolvedResourceParams)
        if (r.first == "depthStencil") { depthAttachment = frameGraph->GetRenderTarget(r.second); break; }
    if (!depthAttachment) depthAttachment = frameGraph->GetRenderTarget("DepthBuffer");
    if (!m_pLinearizeDepthShader) m_pLinearizeDepthShader = driver->CreateShader("Shaders/LinearizeDepth.shader"); // (:39-43)
    auto target = GetRHIResource("target").DynamicCast<RHITexture>();                                                // (:45)
    if (!m_pLinearizeDepthShader || !target || !depthAttachment) { // (:47-50) silent early return
        commands->EndDebugRegion(commandList);
        return;
    }
    if (!m_linearizeDepth) { // (:52-56)
        m_linearizeDepth = driver->CreateShaderBindings();
        driver->AddSamplerToShaderBindings(m_linearizeDepth, "depthSampler", depthAttachment, 0);
    }
    if (!m_postEffectMaterial) m_postEffectMaterial = driver->CreateMaterial(m_pLinearizeDepthShader); // (:58-63)

    commands->ImageMemoryBarrier(commandList, depthAttachment, EImageLayout::ShaderReadOnlyOptimal); // (:79)
    commands->ImageMemoryBarrier(commandList, target, EImageLayout::ColorAttachmentOptimal);          // (:80)
    commands->BeginRenderPass(commandList, TVector<RHITexturePtr> { target }, RHITexturePtr());       // (:84-92)
    commands->BindMaterial(commandList, m_postEffectMaterial);                                         // (:94)
    commands->BindShaderBindings(commandList, m_postEffectMaterial, { sceneView.m_frameBindings, m_linearizeDepth }); // (:98)
    commands->DrawIndexed(commandList, 6, 1, 0, 0, 0);                                                 // (:105) the full-screen NDC quad
    commands->EndRenderPass(commandList);                                                              // (:106)
    commands->EndDebugRegion(commandList);
}

void LinearizeDepthNode::Clear()
{
    m_linearizeDepth.Clear();
    m_postEffectMaterial.Clear();
    m_pLinearizeDepthShader.Clear();
}

// ---- LightCullingNode (FrameGraph/LightCullingNode.cpp:17-87) ------------------------------------------------------------------
const char* LightCullingNode::m_name = "LightCulling";

void LightCullingNode::Process(RHIFrameGraphPtr frameGraph, RHICommandListPtr, RHICommandListPtr commandList, const RHISceneViewSnapshot& sceneView)
{
    if (!sceneView.m_rhiLightsData) return; // no point to cull lights if we have no lights in the scene (:24-28)

    auto driver = Renderer::GetDriver();
    if (!m_pComputeShader) m_pComputeShader = driver->CreateShader("Shaders/ComputeLightCulling.shader"); // (:30-34)

    auto commands = Renderer::GetDriverCommands();
    commands->BeginDebugRegion(commandList, GetName());

    auto depthAttachment = GetRHIResource("depthStencil").DynamicCast<RHITexture>(); // DefaultRenderer.renderer:127-130 -> LinearDepth
    if (!depthAttachment) depthAttachment = frameGraph->GetRenderTarget("DepthBuffer");
    if (depthAttachment) {
        PushConstants pushConstants {};
        pushConstants.lightsNum = (int32_t)sceneView.m_totalNumLights;
        pushConstants.viewportSize[0] = depthAttachment->GetExtent().x;
        pushConstants.viewportSize[1] = depthAttachment->GetExtent().y;
        pushConstants.numTiles[0] = (depthAttachment->GetExtent().x - 1) / (int32_t)TileSize + 1; // (:56)
        pushConstants.numTiles[1] = (depthAttachment->GetExtent().y - 1) / (int32_t)TileSize + 1; // (:57)

        if (!m_culledLights) {
            const size_t numTiles = (size_t)pushConstants.numTiles[0] * pushConstants.numTiles[1];
            m_culledLights = driver->CreateShaderBindings();
            // +1: the reference's buffer is one uint short when every tile is full (:64)
            auto culledLightsSSBO = driver->AddSsboToShaderBindings(m_culledLights, "culledLights", sizeof(uint32_t) * (numTiles * LightsPerTile + 1), 1, 0, true);
            auto lightsGridSSBO = driver->AddSsboToShaderBindings(m_culledLights, "lightsGrid", sizeof(uint32_t) * (numTiles * 2 + 1), 1, 1, true);
            driver->AddSamplerToShaderBindings(m_culledLights, "sceneDepth", depthAttachment, 2);
            auto shaderBindingSet = sceneView.m_rhiLightsData;
            driver->AddShaderBinding(shaderBindingSet, culledLightsSSBO, "culledLights", 1); // (:69) so that Standard.shader sees them
            driver->AddShaderBinding(shaderBindingSet, lightsGridSSBO, "lightsGrid", 2);      // (:70)
        }

        commands->ImageMemoryBarrier(commandList, depthAttachment, EImageLayout::ShaderReadOnlyOptimal);
        commands->Dispatch(commandList, m_pComputeShader, (uint32_t)pushConstants.numTiles[0], (uint32_t)pushConstants.numTiles[1], 1,
                           { sceneView.m_rhiLightsData, m_culledLights, sceneView.m_frameBindings }, &pushConstants, sizeof(PushConstants));
    }
    commands->EndDebugRegion(commandList);
}

void LightCullingNode::Clear()
{
    m_pComputeShader.Clear();
    m_culledLights.Clear();
}

// ---- RenderSceneNode (shading consumer) ---------------------------------------------------------------------------------------
const char* RenderSceneNode::m_name = "RenderScene";

void RenderSceneNode::Process(RHIFrameGraphPtr, RHICommandListPtr, RHICommandListPtr commandList, const RHISceneViewSnapshot& sceneView)
{
    if (!sceneView.m_rhiLightsData || !sceneView.m_rhiLightsData->Find("culledLights")) return; // RenderSceneNode.cpp:142-146: silent early return
    auto driver = Renderer::GetDriver();
    auto commands = Renderer::GetDriverCommands();
    if (!m_pShader) m_pShader = driver->CreateShader("Shaders/Standard.shader");
    auto surface = GetRHIResource("surface").DynamicCast<RHIBuffer>();
    auto radiance = GetRHIResource("radiance").DynamicCast<RHIBuffer>();
    if (!surface || !radiance) return;
    if (!m_surfaceBindings) {
        m_surfaceBindings = driver->CreateShaderBindings();
        m_surfaceBindings->GetOrAddShaderBinding("surface")->m_buffer = surface;
        m_surfaceBindings->GetOrAddShaderBinding("radiance")->m_buffer = radiance;
    }
    std::string tag;
    commands->BeginDebugRegion(commandList, std::string(GetName()) + (TryGetString("Tag", tag) ? " QueueTag:" + tag : ""));
    // binding sets as RenderSceneNode.cpp:181-185: { frame, lights(+culled+grid+shadowMaps+lightsMatrices), per-draw data }
    commands->Dispatch(commandList, m_pShader, 0, 0, 0, { sceneView.m_frameBindings, sceneView.m_rhiLightsData, m_surfaceBindings });
    commands->EndDebugRegion(commandList);
}

void RenderSceneNode::Clear()
{
    m_pShader.Clear();
    m_surfaceBindings.Clear();
}
