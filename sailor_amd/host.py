"""Host-side set-up of the path: frame constants, frusta, CSM matrices, light packing, bands.

Thin numpy-facing wrappers over the `sailor_host_*` / band entry points of the C-ABI (sailor_amd/csrc/host_math.cpp,
context.hip) -- the math itself is native, mirroring what the reference does on its main/render threads before any
GPU work is recorded (FrameGraph/RHIFrameGraph.cpp:60-67, Math/Bounds.cpp:142-193, FrameGraph/ShadowPrepassNode.cpp:387-404,
ECS/LightingECS.cpp:163-172).  Nothing here needs a GPU.
"""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import _lib
from ._lib import Band, CsmDesc, LightCullPushConstants, UboFrameData

# ECS/LightingECS.h:71-81 / Lighting.glsl:4-15: std430 record, stride 112
LIGHT_DTYPE = np.dtype({
    "names": ["type", "shadowType", "worldPosition", "direction", "intensity", "attenuation", "cutOff", "bounds"],
    "formats": ["<u4", "<u4", ("<f4", 3), ("<f4", 3), ("<f4", 3), ("<f4", 3), ("<f4", 2), ("<f4", 3)],
    "offsets": [0, 4, 16, 32, 48, 64, 80, 96],
    "itemsize": 112,
})
# FrameGraph/RenderSceneNode.h:16-33
INSTANCE_DTYPE = np.dtype({
    "names": ["model", "sphereBounds", "materialInstance", "isCulled"],
    "formats": [("<f4", 16), ("<f4", 4), "<u4", "<u4"],
    "offsets": [0, 64, 80, 84],
    "itemsize": 96,
})

LIGHT_DIRECTIONAL, LIGHT_POINT, LIGHT_SPOT, LIGHT_AREA = 0, 1, 2, 3  # Engine/Types.h:31-37
SHADOW_NONE, SHADOW_PCF, SHADOW_EVSM = 0, 1, 2                         # RHI/SceneView.h:13-18


def _fp(a: np.ndarray):
    return a.ctypes.data_as(C.POINTER(C.c_float))


def _f32(a, n) -> np.ndarray:
    a = np.ascontiguousarray(a, dtype=np.float32).reshape(-1)
    assert a.size == n, (a.size, n)
    return a


def num_tiles(width: int, height: int) -> tuple[int, int]:
    tx, ty = C.c_int32(), C.c_int32()
    _lib.check(_lib.load().sailor_hip_num_tiles(width, height, C.byref(tx), C.byref(ty)), "num_tiles")
    return tx.value, ty.value


def band_whole_frame(width: int, height: int) -> Band:
    b = Band()
    _lib.check(_lib.load().sailor_hip_band_whole_frame(width, height, C.byref(b)), "band_whole_frame")
    return b


def band_from_tile_rows(width: int, height: int, row_begin: int, row_end: int) -> Band:
    b = Band()
    _lib.check(_lib.load().sailor_hip_band_from_tile_rows(width, height, row_begin, row_end, C.byref(b)), "band_from_tile_rows")
    return b


def band_for_rank(width: int, height: int, rank: int, world_size: int) -> Band:
    b = Band()
    _lib.check(_lib.load().sailor_hip_band_for_rank(width, height, rank, world_size, C.byref(b)), "band_for_rank")
    return b


def transform_matrix(position, rotation_xyzw, scale) -> np.ndarray:
    """Math/Transform.cpp:39-42; returns a column-major float32[16]."""
    trs = np.concatenate([_f32(position, 4), _f32(rotation_xyzw, 4), _f32(scale, 4)])
    out = np.empty(16, np.float32)
    _lib.check(_lib.load().sailor_host_transform_matrix(_fp(trs), _fp(out)), "transform_matrix")
    return out


def mat4_inverse(m) -> np.ndarray:
    m = _f32(m, 16)
    out = np.empty(16, np.float32)
    _lib.check(_lib.load().sailor_host_mat4_inverse(_fp(m), _fp(out)), "mat4_inverse")
    return out


def mat4_mul(a, b) -> np.ndarray:
    a, b = _f32(a, 16), _f32(b, 16)
    out = np.empty(16, np.float32)
    _lib.check(_lib.load().sailor_host_mat4_mul(_fp(a), _fp(b), _fp(out)), "mat4_mul")
    return out


def fill_frame_data(camera_world, fov_degrees: float, z_near: float, z_far: float, width: int, height: int,
                    current_time: float = 0.0, delta_time: float = 0.0, aspect: float | None = None) -> UboFrameData:
    """FrameGraph/RHIFrameGraph.cpp:60-67 (+ ECS/CameraECS.cpp:20,33, Math/Math.cpp:18-21)."""
    cw = _f32(camera_world, 16)
    frame = UboFrameData()
    asp = np.float32(width) / np.float32(height) if aspect is None else np.float32(aspect)
    _lib.check(_lib.load().sailor_host_fill_frame_data(_fp(cw), fov_degrees, float(asp), z_near, z_far, width, height,
                                                        current_time, delta_time, C.byref(frame)), "fill_frame_data")
    return frame


def push_constants(frame: UboFrameData, width: int, height: int, lights_num: int) -> LightCullPushConstants:
    """FrameGraph/LightCullingNode.cpp:51-57."""
    pc = LightCullPushConstants()
    pc.viewportSize[0], pc.viewportSize[1] = width, height
    tx, ty = num_tiles(width, height)
    pc.numTiles[0], pc.numTiles[1] = tx, ty
    pc.lightsNum = lights_num
    return pc


def extract_frustum_planes(world_matrix, aspect: float, fov_y_degrees: float, z_near: float, z_far: float):
    """Math/Bounds.cpp:142-193 -> (planes float32[6,4] in L,R,T,B,N,F order, corners float32[8,3])."""
    wm = _f32(world_matrix, 16)
    planes = np.empty(24, np.float32)
    corners = np.empty(24, np.float32)
    _lib.check(_lib.load().sailor_host_extract_frustum_planes(_fp(wm), aspect, fov_y_degrees, z_near, z_far, _fp(planes), _fp(corners)),
               "extract_frustum_planes")
    return planes.reshape(6, 4), corners.reshape(8, 3)


def extract_frustum_planes_matrix(projection_view_matrix):
    """Math/Bounds.cpp:20-67 Frustum::ExtractFrustumPlanes(projectionViewMatrix) -> (planes float32[6,4] L,R,T,B,N,F, corners float32[8,3])."""
    m = _f32(projection_view_matrix, 16)
    planes = np.empty(24, np.float32)
    corners = np.empty(24, np.float32)
    _lib.check(_lib.load().sailor_host_extract_frustum_planes_matrix(_fp(m), _fp(planes), _fp(corners)), "extract_frustum_planes_matrix")
    return planes.reshape(6, 4), corners.reshape(8, 3)


def plan_csm_passes(overlap_masks: np.ndarray, shadow_types, last_changed_frame: np.ndarray, previous):
    """The bookkeeping of LightingECS::PrepareCSMPasses (ECS/LightingECS.cpp:299-366) for one directional light, on the cascade overlap sets
    of sailor_hip_csm_caster_masks (uint64 [cascades, words]):
      * cascade k > 0 drops every mesh that an EARLIER cascade of the same shadow type, re-rendered this frame, already overlaps (:310-327);
      * a cascade is re-rendered iff its mesh list, as (mesh index, frame the mesh last changed) pairs, differs from last frame's snapshot
        (CSMLightState::Equals :14-38; the camera / light transform thresholds of that comparison are the caller's `previous is None`).
    Returns (list of cascades to render, their final uint64 masks [cascades, words], the new snapshots)."""
    masks = np.array(overlap_masks, np.uint64, copy=True)
    n_casc = masks.shape[0]
    added = [None] * n_casc                      # bCascadeAdded[z]: the shadow type of a cascade that is rendered this frame
    snapshots, render = [], []
    frames = np.asarray(last_changed_frame)
    for k in range(n_casc):
        for z in range(k):
            if added[z] is not None and added[z] == shadow_types[k]:
                masks[k] &= ~np.asarray(overlap_masks[z], np.uint64)
        bits = np.unpackbits(masks[k].view(np.uint8), bitorder="little")[: len(frames)].astype(bool)
        idx = np.nonzero(bits)[0]
        snap = (idx.copy(), frames[idx].copy())
        same = previous is not None and k < len(previous) and np.array_equal(previous[k][0], snap[0]) and np.array_equal(previous[k][1], snap[1])
        snapshots.append(snap)
        if not same:
            added[k] = shadow_types[k]
            render.append(k)
    return render, masks, snapshots


def csm_matrices(light_view, camera_world, aspect: float, fov_y_degrees: float, camera_near: float, camera_far: float) -> np.ndarray:
    """The 4 `lightsMatrices` (FrameGraph/ShadowPrepassNode.cpp:387-404, ECS/LightingECS.cpp:292) as float32[4,16]."""
    lv, cw = _f32(light_view, 16), _f32(camera_world, 16)
    out = np.empty(64, np.float32)
    _lib.check(_lib.load().sailor_host_csm_matrices(_fp(lv), _fp(cw), aspect, fov_y_degrees, camera_near, camera_far, _fp(out)), "csm_matrices")
    return out.reshape(4, 16)


def cutoff_cosines(inner_degrees: float, outer_degrees: float) -> tuple[np.float32, np.float32]:
    """ECS/LightingECS.cpp:171 through the native packer."""
    z3 = np.zeros(3, np.float32)
    cut = np.array([inner_degrees, outer_degrees], np.float32)
    rec = _lib.LightShaderData()
    _lib.check(_lib.load().sailor_host_pack_light(1, 0, _fp(z3), _fp(z3), _fp(z3), _fp(z3), _fp(cut), _fp(z3), C.byref(rec)), "pack_light")
    return np.float32(rec.cutOff[0]), np.float32(rec.cutOff[1])


def make_csm_desc(lights_matrices: np.ndarray, maps: list) -> CsmDesc:
    """maps: list of 4 (device_ptr | 0, width, height, format)."""
    d = CsmDesc()
    lm = np.ascontiguousarray(lights_matrices, np.float32).reshape(4, 16)
    for k in range(4):
        for i in range(16):
            d.lightsMatrices[k][i] = float(lm[k, i])
        ptr, w, h, fmt = maps[k]
        d.maps[k] = ptr or None
        d.width[k], d.height[k], d.format[k] = w, h, fmt
    return d
