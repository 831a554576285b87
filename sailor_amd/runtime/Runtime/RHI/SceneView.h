// Mirrors RHI/SceneView.h:60-83 RHISceneViewSnapshot -- the per-camera snapshot handed to every node's Process.
#pragma once
#include "Types.h"

namespace Sailor::RHI {

struct CameraData { // ECS/CameraECS.h: what the nodes read from sceneView.m_camera
    float m_world[16];
    float m_fov = 90.0f, m_aspect = 1.0f, m_zNear = 1.0f, m_zFar = 20000.0f;
};

struct RHISceneViewSnapshot {
    CameraData m_camera;
    float m_deltaTime = 0.0f, m_currentTime = 0.0f;
    uint32_t m_totalNumLights = 0;           // RHI/SceneView.h:75, filled at ECS/LightingECS.cpp:404
    RHIShaderBindingSetPtr m_frameBindings;  // :78, filled by RHIFrameGraph::FillFrameData
    RHIShaderBindingSetPtr m_rhiLightsData;  // :79, LightingECS::m_lightsData (binding 0 `light`, 6 `lightsMatrices`, 8 `shadowMaps`)
};

} // namespace Sailor::RHI
