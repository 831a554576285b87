// C entry points that let the Python test-suite drive the C++ host mirror exactly the way the engine would:
// create the renderer (HIP backend), build a frame graph BY NODE NAME, hand it a scene snapshot, process frames, read back.
#include <memory>
#include <string>
#include "Runtime/ECS/LightingECS.h"
#include "Runtime/FrameGraph/LightCullingNode.h"
#include "Runtime/FrameGraph/RHIFrameGraph.h"
#include "Runtime/GraphicsDriver/HIP/HipGraphicsDriver.h"
#include "Runtime/RHI/Renderer.h"
#include "Runtime/AssetRegistry/FrameGraph/FrameGraphParser.h"
#include "Runtime/AssetRegistry/World/WorldPrefabImporter.h"
#include <cstdio>

using namespace Sailor;
using namespace Sailor::RHI;
using namespace Sailor::Framegraph;

#define RT_API extern "C" __attribute__((visibility("default")))

struct SailorRuntime {
    std::unique_ptr<Renderer> renderer;
    std::unique_ptr<LightingECS> lighting;
    RHIFrameGraph graph;
    RHISceneViewSnapshot snapshot;
    FrameGraphNodePtr lightCulling, renderScene, linearizeDepth, environment;
    RHITexturePtr depthHighZ;
    RHITexturePtr depth, rawDepth;
    RHIBufferPtr surface, radiance;
    std::unique_ptr<EcsSweepSystem> sweep;
    int frames = 0;
};

RT_API SailorRuntime* sailor_rt_create(int device, void* stream, int ownStream, int* outStatus)
{
    auto* rt = new SailorRuntime();
    rt->renderer.reset(new Renderer(device, stream, ownStream != 0));
    if (outStatus) *outStatus = rt->renderer->GetStatus();
    if (rt->renderer->GetStatus() != SAILOR_HIP_OK) { delete rt; return nullptr; }
    rt->lighting.reset(new LightingECS());
    return rt;
}

RT_API void sailor_rt_destroy(SailorRuntime* rt)
{
    if (!rt) return;
    Renderer::GetDriver()->WaitIdle();
    rt->graph.Clear();
    rt->sweep.reset();
    rt->lighting.reset();
    rt->depth.Clear(); rt->surface.Clear(); rt->radiance.Clear(); rt->depthHighZ.Clear();
    rt->lightCulling.Clear(); rt->renderScene.Clear(); rt->linearizeDepth.Clear(); rt->environment.Clear(); rt->rawDepth.Clear();
    rt->snapshot = RHISceneViewSnapshot();
    delete rt;
}

RT_API int sailor_rt_node_registered(const char* name) { return FrameGraphBuilder::IsRegistered(name) ? 1 : 0; }

// builds the graph from node names, as FrameGraphImporter does from the .renderer YAML (FrameGraphParser.cpp:153)
RT_API int sailor_rt_build_graph(SailorRuntime* rt, const char** nodeNames, int count)
{
    for (int i = 0; i < count; i++) {
        auto node = FrameGraphBuilder::CreateNode(nodeNames[i]);
        if (!node) return -1;
        if (std::string(nodeNames[i]) == "LightCulling") rt->lightCulling = node;
        if (std::string(nodeNames[i]) == "LinearizeDepth") rt->linearizeDepth = node;
        if (std::string(nodeNames[i]) == "Environment") rt->environment = node;
        if (std::string(nodeNames[i]) == "RenderScene") { node->SetString("Tag", "Opaque"); rt->renderScene = node; }
        rt->graph.AddNode(node);
    }
    return 0;
}

// FrameGraphAsset::Deserialize alone (no device): a one-line summary of what a `.renderer` text declares, for the CPU tests:
//   "targets=Name:WxH:format:mips,...;nodes=Name[tag]{string k=v;float k=v;vec4 k=x y z w;rt k=v},...;values=k=v,...;samplers=a,b"
// WorldPrefab::Deserialize + World::Instantiate of a `.world` text (no device needed): a one-line summary for the tests, -1 + message on an error
static std::string world_summary(const WorldScene& scene)
{
    char b[768];
    std::string s = "name=" + scene.m_name + ";objects=" + std::to_string(scene.m_gameObjects.size()) + ";";
    for (const auto& go : scene.m_gameObjects) {
        snprintf(b, sizeof b, "%s[parent=%d pos=%g %g %g rot=%.9g %.9g %.9g %.9g scale=%g %g %g comps=", go.m_name.c_str(), (int)go.m_parent, go.m_transform.position[0],
                 go.m_transform.position[1], go.m_transform.position[2], go.m_transform.rotation[0], go.m_transform.rotation[1], go.m_transform.rotation[2],
                 go.m_transform.rotation[3], go.m_transform.scale[0], go.m_transform.scale[1], go.m_transform.scale[2]);
        s += b;
        for (size_t i = 0; i < go.m_componentTypes.size(); i++) s += (i ? "," : "") + go.m_componentTypes[i];
        s += "];";
    }
    for (const auto& c : scene.m_cameras) {
        snprintf(b, sizeof b, "camera{owner=%u fov=%g zNear=%g zFar=%g};", c.m_owner, c.m_fov, c.m_zNear, c.m_zFar);
        s += b;
    }
    for (const auto& l : scene.m_lights) {
        snprintf(b, sizeof b, "light{owner=%u type=%u intensity=%g %g %g attenuation=%.9g %.9g %.9g bounds=%g %g %g cutOff=%g %g dir=%.9g %.9g %.9g pos=%g %g %g};", l.m_owner,
                 (unsigned)l.m_data.m_type, l.m_data.m_intensity[0], l.m_data.m_intensity[1], l.m_data.m_intensity[2], l.m_data.m_attenuation[0], l.m_data.m_attenuation[1],
                 l.m_data.m_attenuation[2], l.m_data.m_bounds[0], l.m_data.m_bounds[1], l.m_data.m_bounds[2], l.m_data.m_cutOff[0], l.m_data.m_cutOff[1], l.m_data.m_direction[0],
                 l.m_data.m_direction[1], l.m_data.m_direction[2], l.m_data.m_worldPosition[0], l.m_data.m_worldPosition[1], l.m_data.m_worldPosition[2]);
        s += b;
    }
    for (const auto& m : scene.m_meshRenderers) s += "mesh{owner=" + std::to_string(m.m_owner) + " model=" + m.m_modelFileId + "};";
    s += "other=" + std::to_string(scene.m_otherComponents);
    return s;
}

RT_API int sailor_rt_parse_world(const char* yamlText, char* out, int outSize)
{
    WorldPrefab prefab;
    WorldScene scene;
    std::string err;
    if (!prefab.Deserialize(yamlText ? yamlText : "", &err) || !scene.Instantiate(prefab, &err)) {
        if (out && outSize > 0) snprintf(out, (size_t)outSize, "error: %s", err.c_str());
        return -1;
    }
    if (out && outSize > 0) snprintf(out, (size_t)outSize, "%s", world_summary(scene).c_str());
    return (int)scene.m_gameObjects.size();
}

// Loads a `.world` into the runtime the way Sailor.cpp:185-190 + World::Instantiate do for the parts the path reads: the first camera becomes the scene view's
// camera (aspect from the viewport), every LightComponent registers with the lighting system and is packed by LightingECS::Tick, and the game objects' transforms /
// parent indices come back flat for the ECS sweep (outTransforms: 12 floats each, outParents; up to maxObjects).  Returns the number of game objects.
RT_API int sailor_rt_load_world(SailorRuntime* rt, const char* yamlText, int width, int height, float* outTransforms, uint32_t* outParents, int maxObjects, int* outLights,
                                int* outMeshes)
{
    WorldPrefab prefab;
    WorldScene scene;
    if (!prefab.Deserialize(yamlText ? yamlText : "") || !scene.Instantiate(prefab)) return -1;
    if (!scene.m_cameras.empty()) {
        const auto& c = scene.m_cameras[0];
        memcpy(rt->snapshot.m_camera.m_world, scene.m_gameObjects[c.m_owner].m_world, 64);
        rt->snapshot.m_camera.m_fov = c.m_fov; rt->snapshot.m_camera.m_aspect = (float)width / (float)height;
        rt->snapshot.m_camera.m_zNear = c.m_zNear; rt->snapshot.m_camera.m_zFar = c.m_zFar;
    }
    rt->graph.SetViewport(width, height);
    for (const auto& l : scene.m_lights) rt->lighting->RegisterComponent(l.m_data);
    auto cmd = Renderer::GetDriver()->CreateCommandList();
    rt->lighting->Tick(cmd);
    Renderer::GetDriver()->SubmitCommandList(cmd);
    rt->lighting->FillLightingData(rt->snapshot);
    for (int i = 0; i < (int)scene.m_gameObjects.size() && i < maxObjects; i++) {
        if (outTransforms) memcpy(outTransforms + 12 * i, &scene.m_gameObjects[i].m_transform, 48);
        if (outParents) outParents[i] = scene.m_gameObjects[i].m_parent;
    }
    if (outLights) *outLights = (int)scene.m_lights.size();
    if (outMeshes) *outMeshes = (int)scene.m_meshRenderers.size();
    return (int)scene.m_gameObjects.size();
}

RT_API int sailor_rt_parse_renderer(const char* yamlText, int viewportWidth, int viewportHeight, char* out, int outSize)
{
    FrameGraphAsset asset;
    std::string err;
    if (!asset.Deserialize(yamlText ? yamlText : "", viewportWidth, viewportHeight, &err)) {
        if (out && outSize > 0) snprintf(out, (size_t)outSize, "error: %s", err.c_str());
        return -1;
    }
    std::string s = "targets=";
    for (size_t i = 0; i < asset.m_renderTargets.size(); i++) {
        const auto& t = asset.m_renderTargets[i];
        const uint32_t maxExtent = t.m_width > t.m_height ? t.m_width : t.m_height;
        uint32_t mips = 1;
        if (t.m_bGenerateMips) { mips = 0; for (uint32_t e = maxExtent; e; e >>= 1) mips++; }
        if (mips > t.m_maxMipLevel) mips = t.m_maxMipLevel;
        s += (i ? "," : "") + t.m_name + ":" + std::to_string(t.m_width) + "x" + std::to_string(t.m_height) + ":" + t.m_format + ":" + std::to_string(mips) +
             (t.m_reduction != "Average" ? ":" + t.m_reduction : "");
    }
    s += ";nodes=";
    for (size_t i = 0; i < asset.m_nodes.size(); i++) {
        const auto& n = asset.m_nodes[i];
        s += (i ? "," : "") + n.m_name + "[" + n.m_tag + "]{";
        for (const auto& p : n.m_strings) s += "string " + p.first + "=" + p.second + ";";
        for (const auto& p : n.m_floats) { char b[64]; snprintf(b, sizeof b, "%g", p.second); s += "float " + p.first + "=" + b + ";"; }
        for (const auto& p : n.m_vectors) { char b[128]; snprintf(b, sizeof b, "%g %g %g %g", p.second.x, p.second.y, p.second.z, p.second.w); s += "vec4 " + p.first + "=" + b + ";"; }
        for (const auto& p : n.m_renderTargets) s += "rt " + p.first + "=" + p.second + ";";
        s += "}";
    }
    s += ";values=";
    { bool first = true; for (const auto& v : asset.m_values) { char b[64]; snprintf(b, sizeof b, "%g", v.second); s += (first ? "" : ",") + v.first + "=" + b; first = false; } }
    s += ";samplers=";
    for (size_t i = 0; i < asset.m_samplers.size(); i++) s += (i ? "," : "") + asset.m_samplers[i];
    if (out && outSize > 0) snprintf(out, (size_t)outSize, "%s", s.c_str());
    return (int)asset.m_nodes.size();
}

// FrameGraphImporter: build the runtime's frame graph from a `.renderer` text -- render targets, values, one node per `frame` entry that has a
// node class here (the others are reported, as the reference logs them).  Returns the number of nodes created, < 0 on a parse error.
RT_API int sailor_rt_load_renderer(SailorRuntime* rt, const char* yamlText, int* outNotImplemented, int* outRenderTargets)
{
    FrameGraphAsset asset;
    if (!asset.Deserialize(yamlText ? yamlText : "", rt->graph.GetViewport().x, rt->graph.GetViewport().y)) return -1;
    rt->graph.Clear();
    rt->lightCulling.Clear(); rt->renderScene.Clear(); rt->linearizeDepth.Clear(); rt->environment.Clear();
    const FrameGraphBuildReport report = FrameGraphImporter::BuildFrameGraph(asset, rt->graph);
    for (const auto& node : rt->graph.GetGraph()) { // the harness' setters address these nodes directly
        const std::string name = node->GetDebugName();
        if (name == "LightCulling") rt->lightCulling = node;
        if (name == "LinearizeDepth") rt->linearizeDepth = node;
        if (name == "Environment") rt->environment = node;
        if (name == "RenderScene" && !rt->renderScene) rt->renderScene = node;
    }
    if (rt->renderScene && rt->surface) { rt->renderScene->SetRHIResource("surface", rt->surface); rt->renderScene->SetRHIResource("radiance", rt->radiance); }
    if (outNotImplemented) *outNotImplemented = report.m_nodesNotImplemented;
    if (outRenderTargets) *outRenderTargets = report.m_renderTargets;
    return report.m_nodesCreated;
}

// a render target of the loaded graph, by the name the `.renderer` text gives it
RT_API void* sailor_rt_render_target(SailorRuntime* rt, const char* name, int* outWidth, int* outHeight, int* outLevels)
{
    auto t = rt->graph.GetRenderTarget(name);
    if (!t || !t->m_buffer) return nullptr;
    if (outWidth) *outWidth = t->GetExtent().x;
    if (outHeight) *outHeight = t->GetExtent().y;
    if (outLevels) *outLevels = (int)t->GetMipLevels();
    return t->m_buffer->m_hip.m_devicePtr;
}

// publish a wrapped device image as a named render target of the graph (the per-frame targets a `.renderer` file leaves unresolved: DepthBuffer, ...)
RT_API void sailor_rt_set_render_target(SailorRuntime* rt, const char* name, void* devicePtr, int width, int height)
{
    auto* hip = static_cast<GraphicsDriver::HIP::HipGraphicsDriver*>(Renderer::GetDriver());
    rt->graph.SetRenderTarget(name, hip->WrapTexture(devicePtr, { width, height }, EFormat::R32_SFLOAT));
}

RT_API void sailor_rt_set_camera(SailorRuntime* rt, const float* world16, float fov, float aspect, float zNear, float zFar, int width, int height)
{
    memcpy(rt->snapshot.m_camera.m_world, world16, 64);
    rt->snapshot.m_camera.m_fov = fov; rt->snapshot.m_camera.m_aspect = aspect;
    rt->snapshot.m_camera.m_zNear = zNear; rt->snapshot.m_camera.m_zFar = zFar;
    rt->graph.SetViewport(width, height);
}

// lights: packed LightShaderData records (host memory); goes through UpdateShaderBinding like LightingECS::Tick
RT_API void sailor_rt_set_lights(SailorRuntime* rt, const void* records, int count)
{
    auto cmd = Renderer::GetDriver()->CreateCommandList();
    rt->lighting->SetPacked(cmd, (const SailorLightShaderData*)records, (size_t)count);
    Renderer::GetDriver()->SubmitCommandList(cmd);
    rt->lighting->FillLightingData(rt->snapshot);
}

// lights as components (degrees, defaults of ECS/LightingECS.h:23-28), packed by LightingECS::Tick
RT_API int sailor_rt_add_light(SailorRuntime* rt, uint32_t type, uint32_t shadowType, const float* pos, const float* dir, const float* intensity,
                               const float* bounds, const float* cutOffDegrees)
{
    LightData d;
    d.m_type = (ELightType)type; d.m_shadowType = (EShadowType)shadowType;
    memcpy(d.m_worldPosition, pos, 12); memcpy(d.m_direction, dir, 12); memcpy(d.m_intensity, intensity, 12); memcpy(d.m_bounds, bounds, 12);
    if (cutOffDegrees) memcpy(d.m_cutOff, cutOffDegrees, 8);
    return (int)rt->lighting->RegisterComponent(d);
}

RT_API void sailor_rt_tick_lights(SailorRuntime* rt)
{
    auto cmd = Renderer::GetDriver()->CreateCommandList();
    rt->lighting->Tick(cmd);
    Renderer::GetDriver()->SubmitCommandList(cmd);
    rt->lighting->FillLightingData(rt->snapshot);
}

// LightComponent setters + MarkDirty (Components/LightComponent.h): new parameters for one light; null pointers leave a field alone
RT_API int sailor_rt_update_light(SailorRuntime* rt, int index, const float* pos, const float* dir, const float* intensity, const float* bounds,
                                  const float* cutOffDegrees)
{
    if (!rt || index < 0 || (size_t)index >= rt->lighting->Num()) return -1;
    LightData& d = rt->lighting->GetComponentData((size_t)index);
    if (pos) memcpy(d.m_worldPosition, pos, 12);
    if (dir) memcpy(d.m_direction, dir, 12);
    if (intensity) memcpy(d.m_intensity, intensity, 12);
    if (bounds) memcpy(d.m_bounds, bounds, 12);
    if (cutOffDegrees) memcpy(d.m_cutOff, cutOffDegrees, 8);
    d.m_bIsDirty = true;
    return 0;
}

// What Tick asks of the component and its owner besides the parameters: TComponent::SetActive, MarkDirty, the owner's mobility and the frame
// its transform last changed (GameObject::GetFrameLastChange); a negative value leaves the field alone.
RT_API int sailor_rt_set_light_state(SailorRuntime* rt, int index, int active, int dirty, int mobility, long long ownerFrameLastChange)
{
    if (!rt || index < 0 || (size_t)index >= rt->lighting->Num()) return -1;
    LightData& d = rt->lighting->GetComponentData((size_t)index);
    if (active >= 0) d.m_bIsActive = active != 0;
    if (dirty >= 0) d.m_bIsDirty = dirty != 0;
    if (mobility >= 0) d.m_ownerMobility = (EMobilityType)mobility;
    if (ownerFrameLastChange >= 0) d.m_ownerFrameLastChange = (size_t)ownerFrameLastChange;
    return 0;
}

// the copies the last sailor_rt_tick_lights recorded: (first record slot, record count) pairs in issue order; returns their number
RT_API int sailor_rt_light_uploads(SailorRuntime* rt, uint32_t* outStartCount, int maxRuns)
{
    const auto& runs = rt->lighting->GetLastUploads();
    for (int i = 0; i < (int)runs.size() && i < maxRuns; i++) {
        outStartCount[2 * i] = (uint32_t)runs[(size_t)i].m_startIndex;
        outStartCount[2 * i + 1] = (uint32_t)runs[(size_t)i].m_count;
    }
    return (int)runs.size();
}

RT_API uint32_t sailor_rt_total_num_lights(SailorRuntime* rt) { return rt->snapshot.m_totalNumLights; }

// LightingECS::Tick's loop without a renderer (no device needed): `count` default lights with the given flags, `ticks` passes in a row with the
// skip list carried from pass to pass; a light's position is (its slot, pass, 0).  Per pass: outRunCounts[pass] runs appended to outStartCount
// (pairs), the records of those runs appended to outRecords (112 B each).  Flags are updated in place.  Returns the total number of runs.
RT_API int sailor_rt_plan_light_uploads(int count, uint8_t* dirty, const uint8_t* active, const uint8_t* mobility, uint64_t* frameLastChange,
                                        const uint64_t* ownerFrameLastChange, int ticks, const uint8_t* redirtyPerTick, int* outRunCounts,
                                        uint32_t* outStartCount, int maxRuns, void* outRecords, int maxRecords)
{
    std::vector<LightData> comps((size_t)count);
    for (int i = 0; i < count; i++) {
        comps[(size_t)i].m_bIsDirty = dirty[i] != 0; comps[(size_t)i].m_bIsActive = active[i] != 0;
        comps[(size_t)i].m_ownerMobility = (EMobilityType)mobility[i];
        comps[(size_t)i].m_frameLastChange = (size_t)frameLastChange[i]; comps[(size_t)i].m_ownerFrameLastChange = (size_t)ownerFrameLastChange[i];
    }
    std::vector<std::pair<uint32_t, uint32_t>> skipList;
    int total = 0, totalRecords = 0;
    for (int t = 0; t < ticks; t++) {
        for (int i = 0; i < count; i++) {
            comps[(size_t)i].m_worldPosition[0] = (float)i; comps[(size_t)i].m_worldPosition[1] = (float)t;
            if (t > 0 && redirtyPerTick && redirtyPerTick[(size_t)(t - 1) * count + i]) comps[(size_t)i].m_bIsDirty = true;
        }
        std::vector<SailorLightShaderData> records;
        const auto runs = LightingECS::CollectDirtyRuns(comps, skipList, records);
        outRunCounts[t] = (int)runs.size();
        for (const auto& r : runs) {
            if (total < maxRuns) { outStartCount[2 * total] = (uint32_t)r.m_startIndex; outStartCount[2 * total + 1] = (uint32_t)r.m_count; }
            total++;
        }
        for (const auto& rec : records) {
            if (totalRecords < maxRecords) memcpy((uint8_t*)outRecords + sizeof(SailorLightShaderData) * (size_t)totalRecords, &rec, sizeof rec);
            totalRecords++;
        }
    }
    for (int i = 0; i < count; i++) { dirty[i] = comps[(size_t)i].m_bIsDirty; frameLastChange[i] = comps[(size_t)i].m_frameLastChange; }
    return total;
}

// the LinearDepth render target (device memory owned by the caller), wired as the node's "depthStencil" parameter
RT_API void sailor_rt_set_depth(SailorRuntime* rt, void* devicePtr, int width, int height)
{
    auto* hip = static_cast<GraphicsDriver::HIP::HipGraphicsDriver*>(Renderer::GetDriver());
    rt->depth = hip->WrapTexture(devicePtr, { width, height }, EFormat::R32_SFLOAT);
    if (rt->lightCulling) rt->lightCulling->SetRHIResource("depthStencil", rt->depth);
}

// the raw reversed-Z depth attachment (caller's device memory) in front of a "LinearizeDepth" node whose target is the
// LinearDepth render target set with sailor_rt_set_depth (DefaultRenderer.renderer: LinearizeDepth -> LightCulling)
RT_API void sailor_rt_set_raw_depth(SailorRuntime* rt, void* devicePtr, int width, int height)
{
    auto* hip = static_cast<GraphicsDriver::HIP::HipGraphicsDriver*>(Renderer::GetDriver());
    rt->rawDepth = hip->WrapTexture(devicePtr, { width, height }, EFormat::R32_SFLOAT);
    if (rt->linearizeDepth) {
        rt->linearizeDepth->SetRHIResource("depthStencil", rt->rawDepth);
        if (rt->depth) rt->linearizeDepth->SetRHIResource("target", rt->depth);
    }
}

RT_API void sailor_rt_set_surface(SailorRuntime* rt, void* surfaceDevicePtr, void* radianceDevicePtr, int width, int height)
{
    auto* hip = static_cast<GraphicsDriver::HIP::HipGraphicsDriver*>(Renderer::GetDriver());
    rt->surface = hip->WrapBuffer(surfaceDevicePtr, (size_t)width * height * 48);
    rt->radiance = hip->WrapBuffer(radianceDevicePtr, (size_t)width * height * 16);
    if (rt->renderScene) { rt->renderScene->SetRHIResource("surface", rt->surface); rt->renderScene->SetRHIResource("radiance", rt->radiance); }
}

RT_API void sailor_rt_set_shadow_maps(SailorRuntime* rt, void* const* mapDevicePtrs, const int* sizes, const int* formats, const float* lightsMatrices64)
{
    auto* hip = static_cast<GraphicsDriver::HIP::HipGraphicsDriver*>(Renderer::GetDriver());
    TVector<RHITexturePtr> maps;
    for (int k = 0; k < SAILOR_NUM_CSM_CASCADES; k++) {
        const EFormat f = formats[k] == SAILOR_SHADOWMAP_R16_SFLOAT ? EFormat::R16_SFLOAT : (formats[k] == SAILOR_SHADOWMAP_R32_SFLOAT ? EFormat::R32_SFLOAT : EFormat::R32G32B32A32_SFLOAT);
        maps.push_back(hip->WrapTexture(mapDevicePtrs[k], { sizes[k], sizes[k] }, f));
    }
    rt->lighting->SetShadowMaps(maps, lightsMatrices64);
}

// the image-based-lighting inputs: published to the frame graph the way EnvironmentNode does (SetSampler, EnvironmentNode.cpp:79,169-170),
// the AO target as render target "g_AO" (RHIFrameGraph.cpp:155)
RT_API void sailor_rt_set_ibl(SailorRuntime* rt, void* irradiance, int irrSize, void* env, int envSize, int envLevels, void* lut, int lutW, int lutH,
                              void* ao, int width, int height)
{
    auto* hip = static_cast<GraphicsDriver::HIP::HipGraphicsDriver*>(Renderer::GetDriver());
    auto irr = hip->WrapTexture(irradiance, { irrSize, irrSize * 6 }, EFormat::R32G32B32A32_SFLOAT);
    size_t envTexels = 0;
    for (int l = 0; l < envLevels; l++) { const int s = (envSize >> l) > 1 ? (envSize >> l) : 1; envTexels += (size_t)6 * s * s; }
    auto e = hip->WrapTexture(env, { 1, (int32_t)envTexels }, EFormat::R32G32B32A32_SFLOAT);
    e->m_extent = { envSize, envSize };
    e->m_mipLevels = (uint32_t)envLevels;
    irr->m_extent = { irrSize, irrSize };
    rt->graph.SetSampler("g_irradianceCubemap", irr);
    rt->graph.SetSampler("g_envCubemap", e);
    rt->graph.SetSampler("g_brdfSampler", hip->WrapTexture(lut, { lutW, lutH }, EFormat::R32G32_SFLOAT));
    if (ao) rt->graph.SetRenderTarget("g_AO", hip->WrapTexture(ao, { width, height }, EFormat::R32_SFLOAT));
}

// The raw environment the Environment node bakes from: published as the sampler "g_skyCubemap" the way SkyNode does, and the node is marked dirty
// (SkyNode::MarkDirty -> EnvironmentNode::MarkDirty).  irradianceSize > 0 overrides the node's 32 x 32 x 6 output (a test knob).
RT_API int sailor_rt_set_sky_cubemap(SailorRuntime* rt, void* cubeChain, int size, int levels, int irradianceSize, void* ao, int width, int height)
{
    auto* hip = static_cast<GraphicsDriver::HIP::HipGraphicsDriver*>(Renderer::GetDriver());
    if (!rt->environment) return -1;
    rt->graph.SetSampler("g_skyCubemap", hip->WrapCubemap(cubeChain, size, (uint32_t)levels, EFormat::R32G32B32A32_SFLOAT));
    if (irradianceSize > 0) rt->environment->SetFloat("IrradianceMapSize", (float)irradianceSize);
    static_cast<EnvironmentNode*>(rt->environment.GetRawPtr())->MarkDirty();
    if (ao) rt->graph.SetRenderTarget("g_AO", hip->WrapTexture(ao, { width, height }, EFormat::R32_SFLOAT));
    return 0;
}

// The other source of the raw environment: an equirect panorama (the node's "EnvironmentMap" texture, EnvironmentNode.cpp:100-138).  The harness
// hands over the loaded RGBA32F texture; the node converts it to a 512 x 512 x 6 cube with 10 mips and bakes from that.
RT_API int sailor_rt_set_environment_map(SailorRuntime* rt, void* equirect, int width, int height, int repeat, int irradianceSize)
{
    auto* hip = static_cast<GraphicsDriver::HIP::HipGraphicsDriver*>(Renderer::GetDriver());
    if (!rt->environment || !equirect) return -1;
    auto tex = hip->WrapTexture(equirect, { width, height }, EFormat::R32G32B32A32_SFLOAT);
    tex->m_bRepeat = repeat != 0;
    if (irradianceSize > 0) rt->environment->SetFloat("IrradianceMapSize", (float)irradianceSize);
    static_cast<EnvironmentNode*>(rt->environment.GetRawPtr())->SetEnvironmentMap(tex);
    return 0;
}

// device pointer + geometry of a sampler the graph's nodes published (g_brdfSampler, g_envCubemap, g_irradianceCubemap, ...)
RT_API void* sailor_rt_sampler(SailorRuntime* rt, const char* name, int* outWidth, int* outHeight, int* outLevels)
{
    auto t = rt->graph.GetSampler(name);
    if (!t || !t->m_buffer) return nullptr;
    if (outWidth) *outWidth = t->GetExtent().x;
    if (outHeight) *outHeight = t->GetExtent().y;
    if (outLevels) *outLevels = (int)t->GetMipLevels();
    return t->m_buffer->m_hip.m_devicePtr;
}

// The blur section of ShadowPrepassNode::Process (FrameGraph/ShadowPrepassNode.cpp:283-356) for one EVSM cascade, command for command: the
// radius upload (:286), "Blur Horizontal" (shadow map -> temporary target) and "Blur Vertical" (temporary target -> shadow map).  (The caster draws
// in front of it need a rasteriser and are not part of this path.)
RT_API int sailor_rt_blur_shadow_map(SailorRuntime* rt, void* mapDevicePtr, void* tempDevicePtr, int size, float radiusUmbra, float radiusPenumbra)
{
    auto* hip = static_cast<GraphicsDriver::HIP::HipGraphicsDriver*>(Renderer::GetDriver());
    auto driver = Renderer::GetDriver();
    auto commands = Renderer::GetDriverCommands();
    auto shadowMap = hip->WrapTexture(mapDevicePtr, { size, size }, EFormat::R32G32B32A32_SFLOAT);
    auto blurAttachment = hip->WrapTexture(tempDevicePtr, { size, size }, EFormat::R32G32B32A32_SFLOAT); // GetOrAddTemporaryRenderTarget (:285)
    auto vertical = driver->CreateMaterial(driver->CreateShader("Shaders/Blur.shader", { "VERTICAL", "EVSM" }));     // :60-62
    auto horizontal = driver->CreateMaterial(driver->CreateShader("Shaders/Blur.shader", { "HORIZONTAL", "EVSM" })); // :65-67
    auto blurBindings = driver->CreateShaderBindings();                                                             // :70
    auto blurData = driver->AddBufferToShaderBindings(blurBindings, "data", 48, 0, EShaderBindingType::UniformBuffer); // :73 (3 x vec4)
    auto frameBindings = driver->CreateShaderBindings();
    auto cmd = driver->CreateCommandList();
    const float blurRadius[2] = { radiusUmbra, radiusPenumbra };
    if (radiusUmbra * radiusUmbra + radiusPenumbra * radiusPenumbra > 0.01f) { // m_blurRadius.length() > 0.1f (:283)
        commands->UpdateShaderBinding(cmd, blurData, blurRadius, sizeof blurRadius); // :286
        commands->BeginDebugRegion(cmd, "Blur Horizontal");
        driver->AddSamplerToShaderBindings(blurBindings, "colorSampler", shadowMap, 1); // :292
        commands->ImageMemoryBarrier(cmd, shadowMap, EImageLayout::ShaderReadOnlyOptimal);
        commands->ImageMemoryBarrier(cmd, blurAttachment, EImageLayout::ColorAttachmentOptimal);
        commands->BeginRenderPass(cmd, TVector<RHITexturePtr> { blurAttachment }, RHITexturePtr());
        commands->BindMaterial(cmd, horizontal);
        commands->BindShaderBindings(cmd, horizontal, { frameBindings, blurBindings });
        commands->DrawIndexed(cmd, 6, 1, 0, 0, 0);
        commands->EndRenderPass(cmd);
        commands->EndDebugRegion(cmd);
        commands->BeginDebugRegion(cmd, "Blur Vertical");
        driver->AddSamplerToShaderBindings(blurBindings, "colorSampler", blurAttachment, 1); // :329
        commands->BeginRenderPass(cmd, TVector<RHITexturePtr> { shadowMap }, RHITexturePtr());
        commands->BindMaterial(cmd, vertical);
        commands->BindShaderBindings(cmd, vertical, { frameBindings, blurBindings });
        commands->DrawIndexed(cmd, 6, 1, 0, 0, 0);
        commands->EndRenderPass(cmd);
        commands->EndDebugRegion(cmd);
    }
    driver->SubmitCommandList(cmd);
    return hip->GetLastDispatchStatus();
}

// The "GPU Culling" section of RHIRecordDrawCallGPUCulling (RHI/Batch.hpp:146-188) over resident buffers: the indirect buffer is bound as
// "drawIndexedIndirect" (:86-89), the push constants are { numBatches, numInstances, firstInstanceIndex } (:176-184) and the Dispatch binds
// { computeMeshCullingBindings (depthHighZ), perInstanceData, indirect buffer, frame } (RenderSceneNode.cpp:335).  The frame UBO is the one
// RHIFrameGraph::FillFrameData uploads for the current camera / viewport.
RT_API int sailor_rt_gpu_culling(SailorRuntime* rt, void* instancesDevicePtr, uint32_t numInstances, uint32_t firstInstanceIndex, void* batchesDevicePtr,
                                 uint32_t numBatches)
{
    auto* hip = static_cast<GraphicsDriver::HIP::HipGraphicsDriver*>(Renderer::GetDriver());
    auto driver = Renderer::GetDriver();
    auto commands = Renderer::GetDriverCommands();
    auto transferCmdList = driver->CreateCommandList();
    rt->graph.FillFrameData(transferCmdList, rt->snapshot, rt->snapshot.m_deltaTime, rt->snapshot.m_currentTime);
    // RenderSceneNode.cpp:126-139: the shader is loaded with its OCCLUSION_CULLING define and the node's "depthHighZ" attachment is bound as a sampler
    // (here: when sailor_rt_build_depth_highz has produced a pyramid; otherwise the frustum-only build)
    auto computeCullingShader = rt->depthHighZ ? driver->CreateShader("Shaders/ComputeMeshCulling.shader", { "OCCLUSION_CULLING" })
                                               : driver->CreateShader("Shaders/ComputeMeshCulling.shader");
    auto computeMeshCullingBindings = driver->CreateShaderBindings();
    if (rt->depthHighZ) driver->AddSamplerToShaderBindings(computeMeshCullingBindings, "depthHighZ", rt->depthHighZ, 0);
    auto perInstanceData = driver->CreateShaderBindings();
    perInstanceData->GetOrAddShaderBinding("data")->m_buffer = hip->WrapBuffer(instancesDevicePtr, (size_t)(firstInstanceIndex + numInstances) * sizeof(SailorPerInstanceData));
    auto indirectCommandBufferBinding = driver->CreateShaderBindings();
    if (batchesDevicePtr)
        indirectCommandBufferBinding->GetOrAddShaderBinding("drawIndexedIndirect")->m_buffer = hip->WrapBuffer(batchesDevicePtr, (size_t)numBatches * sizeof(SailorDrawIndexedIndirectData));
    struct PushConstants { uint32_t m_numBatches = 0, m_numInstances = 0, m_firstInstanceIndex = 0; } constants;
    constants.m_numBatches = numBatches;
    constants.m_numInstances = numInstances;
    constants.m_firstInstanceIndex = firstInstanceIndex;
    commands->BeginDebugRegion(transferCmdList, "GPU Culling");
    commands->Dispatch(transferCmdList, computeCullingShader, 256, 1, 1,
                       { computeMeshCullingBindings, perInstanceData, indirectCommandBufferBinding, rt->snapshot.m_frameBindings }, &constants, sizeof constants);
    commands->EndDebugRegion(transferCmdList);
    driver->SubmitCommandList(transferCmdList);
    return hip->GetLastDispatchStatus();
}

// DefaultRenderer.renderer's DepthHighZ target (R32_SFLOAT, mips, reduction Min) + one run of the DepthHighZ node over a wrapped depth image.
// The pyramid stays bound for the following sailor_rt_gpu_culling calls; levels <= 0 drops it again.
RT_API int sailor_rt_build_depth_highz(SailorRuntime* rt, void* depthDevicePtr, int depthWidth, int depthHeight, int width, int height, int levels, void** outPyramid)
{
    auto* hip = static_cast<GraphicsDriver::HIP::HipGraphicsDriver*>(Renderer::GetDriver());
    auto driver = Renderer::GetDriver();
    if (levels <= 0) { rt->depthHighZ.Clear(); return 0; }
    auto node = FrameGraphBuilder::CreateNode("DepthHighZ");
    if (!node) return -1;
    rt->depthHighZ = driver->CreateRenderTarget({ width, height }, (uint32_t)levels, EFormat::R32_SFLOAT);
    if (!rt->depthHighZ) return -1;
    node->SetRHIResource("src", hip->WrapTexture(depthDevicePtr, { depthWidth, depthHeight }, EFormat::R32_SFLOAT));
    node->SetRHIResource("dst", rt->depthHighZ);
    auto cmd = driver->CreateCommandList();
    node->Process(&rt->graph, cmd, cmd, rt->snapshot);
    driver->SubmitCommandList(cmd);
    node->Clear();
    if (outPyramid) *outPyramid = rt->depthHighZ->m_buffer->m_hip.m_devicePtr;
    return hip->GetLastDispatchStatus();
}

// One shadow pass of ShadowPrepassNode::Process (FrameGraph/ShadowPrepassNode.cpp:219-365), command for command: barriers, BeginRenderPass(shadow map, temporary
// depth attachment, clear), the light matrix as push constant (:249), the caster draw (vertex / index buffer, per-instance SSBO `data`, instanced DrawIndexed --
// what RHIRecordDrawCall records per batch), EndRenderPass, then for an EVSM pass with a blur radius the two blur draws (:283-356).
RT_API int sailor_rt_shadow_pass(SailorRuntime* rt, const float* lightMatrix, void* positions, uint32_t numVertices, void* indices, uint32_t numIndices, void* models,
                                 uint32_t firstInstance, uint32_t instanceCount, void* shadowMapDevicePtr, int size, int evsm, float radiusUmbra, float radiusPenumbra)
{
    auto* hip = static_cast<GraphicsDriver::HIP::HipGraphicsDriver*>(Renderer::GetDriver());
    auto driver = Renderer::GetDriver();
    auto commands = Renderer::GetDriverCommands();
    auto shadowMap = hip->WrapTexture(shadowMapDevicePtr, { size, size }, evsm ? EFormat::R32G32B32A32_SFLOAT : EFormat::R16_SFLOAT);
    auto depthAttachment = driver->CreateRenderTarget({ size, size }, 1, EFormat::R32_SFLOAT); // GetOrAddTemporaryRenderTarget(depth format, extent) (:229)
    auto material = evsm ? driver->CreateMaterial(driver->CreateShader("Shaders/ShadowCaster.shader", { "EVSM" }))
                         : driver->CreateMaterial(driver->CreateShader("Shaders/ShadowCaster.shader")); // GetOrAddShadowMaterial (:20-45)
    auto perInstanceData = driver->CreateShaderBindings();
    perInstanceData->GetOrAddShaderBinding("data")->m_buffer = hip->WrapBuffer(models, (size_t)(firstInstance + instanceCount) * 64);
    auto cmd = driver->CreateCommandList();
    commands->BeginDebugRegion(cmd, "Record Shadow Map Pass 0");
    commands->ImageMemoryBarrier(cmd, shadowMap, EImageLayout::ColorAttachmentOptimal);
    commands->ImageMemoryBarrier(cmd, depthAttachment, EImageLayout::General);
    commands->BeginRenderPass(cmd, TVector<RHITexturePtr> { shadowMap }, depthAttachment);
    commands->PushConstants(cmd, material, 64, lightMatrix);
    commands->BindMaterial(cmd, material);
    commands->BindShaderBindings(cmd, material, { rt->snapshot.m_frameBindings ? rt->snapshot.m_frameBindings : driver->CreateShaderBindings(), perInstanceData });
    commands->BindVertexBuffer(cmd, hip->WrapBuffer(positions, (size_t)numVertices * 12), 0);
    commands->BindIndexBuffer(cmd, hip->WrapBuffer(indices, (size_t)numIndices * 4), 0);
    commands->DrawIndexed(cmd, numIndices, instanceCount, 0, 0, firstInstance);
    commands->EndRenderPass(cmd);
    commands->EndDebugRegion(cmd);
    driver->SubmitCommandList(cmd);
    int st = hip->GetLastDispatchStatus();
    if (st == 0 && evsm && radiusUmbra * radiusUmbra + radiusPenumbra * radiusPenumbra > 0.01f) {
        auto temp = driver->CreateRenderTarget({ size, size }, 1, EFormat::R32G32B32A32_SFLOAT);
        st = sailor_rt_blur_shadow_map(rt, shadowMapDevicePtr, temp->m_buffer->m_hip.m_devicePtr, size, radiusUmbra, radiusPenumbra);
        driver->WaitIdle(); // `temp` is released on return
    }
    return st;
}

// split frame: this runtime renders tile-row band `rank` of `worldSize` (targets set afterwards hold the band's rows); `comm` = ncclComm_t or null
RT_API int sailor_rt_set_frame_split(SailorRuntime* rt, int rank, int worldSize, void* comm)
{
    (void)rt;
    return static_cast<GraphicsDriver::HIP::HipGraphicsDriver*>(Renderer::GetDriver())->SetFrameSplit(rank, worldSize, comm);
}

// the RCCL step: the band lists of the last frame -> the reference's global buffers, in caller-owned device memory
RT_API int sailor_rt_exchange_light_lists(SailorRuntime* rt, void* globalGridDevicePtr, size_t gridBytes, void* globalCulledDevicePtr, size_t culledBytes)
{
    auto* hip = static_cast<GraphicsDriver::HIP::HipGraphicsDriver*>(Renderer::GetDriver());
    auto node = rt->lightCulling.DynamicCast<LightCullingNode>();
    if (!node || !node->GetCulledLights()) return SAILOR_HIP_ERR_INVALID_ARGUMENT;
    auto g = node->GetCulledLights()->Find("lightsGrid"), c = node->GetCulledLights()->Find("culledLights");
    if (!g || !c || !g->m_buffer || !c->m_buffer) return SAILOR_HIP_ERR_INVALID_ARGUMENT;
    return hip->ExchangeLightLists(g->m_buffer, c->m_buffer, hip->WrapBuffer(globalGridDevicePtr, gridBytes), hip->WrapBuffer(globalCulledDevicePtr, culledBytes));
}

RT_API int sailor_rt_process_frame(SailorRuntime* rt)
{
    rt->graph.Process(rt->snapshot);
    rt->frames++;
    return static_cast<GraphicsDriver::HIP::HipGraphicsDriver*>(Renderer::GetDriver())->GetLastDispatchStatus();
}

// One frame in which ANOTHER command overwrites the cull's two SSBOs between the LightCulling node and RenderScene (what a debug pass or a host-side
// list editor would record): the shade must read what the write leaves there -- on a split frame too, where the band form's tile blocks and split
// blocks must decide "long tile" on the same bytes.  gridData / culledData: host bytes for lightsGrid / culledLights (either may be null).
RT_API int sailor_rt_process_frame_overwriting_lists(SailorRuntime* rt, const void* gridData, size_t gridBytes, const void* culledData, size_t culledBytes)
{
    auto driver = Renderer::GetDriver();
    auto commands = Renderer::GetDriverCommands();
    auto node = rt->lightCulling.DynamicCast<LightCullingNode>();
    if (!node) return SAILOR_HIP_ERR_INVALID_ARGUMENT;
    auto transferCmdList = driver->CreateCommandList();
    auto cmdList = driver->CreateCommandList();
    rt->graph.FillFrameData(transferCmdList, rt->snapshot, rt->snapshot.m_deltaTime, rt->snapshot.m_currentTime);
    for (auto& n : rt->graph.GetGraph()) n->Prepare(&rt->graph, rt->snapshot);
    for (auto& n : rt->graph.GetGraph()) {
        n->Process(&rt->graph, transferCmdList, cmdList, rt->snapshot);
        if (n.GetRawPtr() != rt->lightCulling.GetRawPtr() || !node->GetCulledLights()) continue;
        auto g = node->GetCulledLights()->Find("lightsGrid"), c = node->GetCulledLights()->Find("culledLights");
        if (gridData && g && g->m_buffer && gridBytes <= g->m_buffer->m_size) commands->UpdateBuffer(cmdList, g->m_buffer, gridData, gridBytes, 0);
        if (culledData && c && c->m_buffer && culledBytes <= c->m_buffer->m_size) commands->UpdateBuffer(cmdList, c->m_buffer, culledData, culledBytes, 0);
    }
    driver->SubmitCommandList(transferCmdList);
    driver->SubmitCommandList(cmdList);
    rt->frames++;
    return static_cast<GraphicsDriver::HIP::HipGraphicsDriver*>(driver)->GetLastDispatchStatus();
}

RT_API void sailor_rt_wait_idle(SailorRuntime*) { Renderer::GetDriver()->WaitIdle(); }

// device pointers of the node-owned SSBOs, for read-back by the tests
RT_API void* sailor_rt_buffer(SailorRuntime* rt, const char* name, size_t* outBytes)
{
    if (name && !strcmp(name, "light")) { // LightingECS's `light` SSBO (LightingECS.cpp:44)
        auto lb = rt->lighting->GetLightsData()->Find("light");
        if (!lb || !lb->m_buffer) return nullptr;
        if (outBytes) *outBytes = lb->m_buffer->m_size;
        return lb->m_buffer->m_hip.m_devicePtr;
    }
    auto node = rt->lightCulling.DynamicCast<LightCullingNode>();
    if (!node || !node->GetCulledLights()) return nullptr;
    auto b = node->GetCulledLights()->Find(name);
    if (!b || !b->m_buffer) return nullptr;
    if (outBytes) *outBytes = b->m_buffer->m_size;
    return b->m_buffer->m_hip.m_devicePtr;
}

RT_API int sailor_rt_ecs_sweep(SailorRuntime* rt, const void* transforms, const uint32_t* parent, const void* localAabb, uint32_t count,
                               const uint32_t* levelOffsets, uint32_t numLevels, void** outWorld, void** outWorldAabb, void** outVisibility)
{
    rt->sweep.reset(new EcsSweepSystem((const SailorTransform*)transforms, parent, (const SailorAABB*)localAabb, count, levelOffsets, numLevels));
    const auto& c = rt->snapshot.m_camera;
    const int st = rt->sweep->Tick(c.m_world, c.m_aspect, c.m_fov, c.m_zNear, c.m_zFar);
    *outWorld = rt->sweep->m_world->m_hip.m_devicePtr;
    *outWorldAabb = rt->sweep->m_worldAabb->m_hip.m_devicePtr;
    *outVisibility = rt->sweep->m_visibility->m_hip.m_devicePtr;
    return st;
}
