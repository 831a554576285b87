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


def quat_rotate(q_xyzw, v) -> np.ndarray:
    """glm::rotate(quat, vec3) = v + ((cross(q.xyz, v) * q.w) + cross(q.xyz, cross(q.xyz, v))) * 2, in float32 (Math/Transform.cpp:74 GetForward)"""
    q = np.asarray(q_xyzw, np.float32)
    v = np.asarray(v, np.float32)
    uv = np.cross(q[:3], v).astype(np.float32)
    uuv = np.cross(q[:3], uv).astype(np.float32)
    return (v + (uv * q[3] + uuv) * np.float32(2.0)).astype(np.float32)


CSM_CAMERA_POS_DELTA = np.float32(15.0)        # ECS/LightingECS.cpp:16
CSM_CAMERA_ROTATION_DELTA = np.float32(0.9995)  # ECS/LightingECS.cpp:17


def csm_view_unchanged(prev_view, view) -> bool:
    """The transform half of CSMLightState::Equals (ECS/LightingECS.cpp:14-24).  A view is (component index, camera position[4], camera rotation xyzw,
    light position[4], light rotation xyzw): unchanged iff same light component, the camera moved at most 15 units and its forward vector keeps a
    dot product of at least 0.9995 with the old one, and the light's position and rotation are exactly equal."""
    if prev_view is None or view is None:
        return prev_view is None and view is None
    (ci0, cp0, cr0, lp0, lr0), (ci1, cp1, cr1, lp1, lr1) = prev_view, view
    if ci0 != ci1:
        return False
    d = np.asarray(cp0, np.float32) - np.asarray(cp1, np.float32)
    if np.sqrt(np.float32(np.dot(d, d)), dtype=np.float32) > CSM_CAMERA_POS_DELTA:
        return False
    f0, f1 = quat_rotate(cr0, [0, 0, -1]), quat_rotate(cr1, [0, 0, -1])
    if np.float32(np.dot(f0, f1)) < CSM_CAMERA_ROTATION_DELTA:
        return False
    return bool(np.array_equal(np.asarray(lp0, np.float32), np.asarray(lp1, np.float32)) and np.array_equal(np.asarray(lr0, np.float32), np.asarray(lr1, np.float32)))


class CsmSnapshots:
    """LightingECS::m_csmSnapshots behind the C-ABI (sailor_host_csm_snapshots_*): `snaps[k]` = (mesh indices, their last-changed frames, the view the
    snapshot was taken with or None)."""

    def __init__(self, handle=None):
        lib = _lib.load()
        self._lib = lib
        self.handle = handle if handle is not None else lib.sailor_host_csm_snapshots_create()
        if not self.handle:
            raise MemoryError("sailor_host_csm_snapshots_create")

    def clone(self) -> "CsmSnapshots":
        return CsmSnapshots(self._lib.sailor_host_csm_snapshots_clone(self.handle))

    def __len__(self):
        return int(self._lib.sailor_host_csm_snapshots_count(self.handle))

    def __getitem__(self, k: int):
        n = C.c_uint32()
        _lib.check(self._lib.sailor_host_csm_snapshot_get(self.handle, k, 0, C.byref(n), None, None, None, None), "csm_snapshot_get")
        idx = np.zeros(max(n.value, 1), np.uint32); frames = np.zeros(max(n.value, 1), np.uint64)
        has = C.c_int32(); v = _lib.CsmView()
        _lib.check(self._lib.sailor_host_csm_snapshot_get(self.handle, k, n.value, C.byref(n), idx.ctypes.data_as(C.c_void_p), frames.ctypes.data_as(C.c_void_p),
                                                          C.byref(has), C.byref(v)), "csm_snapshot_get")
        view = (int(v.componentIndex), np.float32(list(v.cameraPosition)), np.float32(list(v.cameraRotation)), np.float32(list(v.lightPosition)),
                np.float32(list(v.lightRotation))) if has.value else None
        return idx[: n.value], frames[: n.value], view

    def __del__(self):
        try:
            if self.handle:
                self._lib.sailor_host_csm_snapshots_destroy(self.handle)
                self.handle = None
        except Exception:
            pass


def _csm_view_struct(view):
    v = _lib.CsmView()
    v.componentIndex = int(view[0])
    for name, vals in zip(("cameraPosition", "cameraRotation", "lightPosition", "lightRotation"), view[1:]):
        a = np.asarray(vals, np.float32)
        for i in range(4):
            getattr(v, name)[i] = float(a[i])
    return v


def plan_csm_passes(overlap_masks: np.ndarray, shadow_types, last_changed_frame: np.ndarray, previous, view=None):
    """The bookkeeping of LightingECS::PrepareCSMPasses (ECS/LightingECS.cpp:299-366) for one directional light -- sailor_host_csm_plan_passes
    (host_math.cpp) -- on the cascade overlap sets of sailor_hip_csm_caster_masks (uint64 [cascades, words]):
      * cascade k > 0 drops every mesh that an EARLIER cascade of the same shadow type, re-rendered this frame, already overlaps (:310-327);
      * a cascade is re-rendered iff its mesh list, as (mesh index, frame the mesh last changed) pairs, differs from last frame's snapshot
        or -- with `view` = (light component index, camera position, camera rotation, light position, light rotation) -- the camera moved more than 15
        units / turned past dot 0.9995 / the light moved at all since that snapshot was taken (CSMLightState::Equals :14-38, csm_view_unchanged).
        A cascade that is NOT re-rendered keeps its old snapshot, camera included (:353-357): slow camera drift accumulates until it crosses the threshold.
    `previous`: the CsmSnapshots a former call returned, or None.  Returns (list of cascades to render, their final uint64 masks [cascades, words],
    the new CsmSnapshots; `previous` is left as it was)."""
    masks_in = np.ascontiguousarray(overlap_masks, np.uint64)
    n_casc, words = masks_in.shape
    frames = np.ascontiguousarray(last_changed_frame, np.uint64)
    n = len(frames)
    assert words == (n + 63) // 64
    types = np.ascontiguousarray(shadow_types, np.uint32)
    snaps = previous.clone() if previous is not None else CsmSnapshots()
    render = np.zeros(n_casc, np.uint32)
    out = np.zeros_like(masks_in)
    vs = _csm_view_struct(view) if view is not None else None
    _lib.check(_lib.load().sailor_host_csm_plan_passes(snaps.handle, 0, n_casc, n, masks_in.ctypes.data_as(C.c_void_p), types.ctypes.data_as(C.c_void_p),
                                                       frames.ctypes.data_as(C.c_void_p), C.byref(vs) if vs is not None else None,
                                                       render.ctypes.data_as(C.c_void_p), out.ctypes.data_as(C.c_void_p)), "csm_plan_passes")
    return [int(k) for k in np.nonzero(render)[0]], out, snaps


def overlaps_sphere(planes, center, radius: float) -> bool:
    """Math/Bounds.cpp:211-226 Frustum::OverlapsSphere"""
    pl = _f32(planes, 24); sp = np.float32([center[0], center[1], center[2], radius])
    return bool(_lib.load().sailor_host_overlaps_sphere(_fp(pl), _fp(sp)))


def contains_sphere(planes, center, radius: float) -> bool:
    """Math/Bounds.cpp:228-243 Frustum::ContainsSphere"""
    pl = _f32(planes, 24); sp = np.float32([center[0], center[1], center[2], radius])
    return bool(_lib.load().sailor_host_contains_sphere(_fp(pl), _fp(sp)))


def lights_in_frustum(planes, camera_position, types, shadow_types, positions, bounds, active=None):
    """LightingECS::GetLightsInFrustum (ECS/LightingECS.cpp:209-260) -> (directional indices, (point indices, distances), (spot indices, distances))"""
    n = len(types)
    pl = _f32(planes, 24); cp = np.float32(camera_position)[:3].copy()
    t = np.ascontiguousarray(types, np.uint32); st = np.ascontiguousarray(shadow_types, np.uint32)
    pos = np.ascontiguousarray(positions, np.float32).reshape(n, 3); bd = np.ascontiguousarray(bounds, np.float32).reshape(n, 3)
    act = None if active is None else np.ascontiguousarray(active, np.uint8)
    od, op, os_ = (np.zeros(max(n, 1), np.uint32) for _ in range(3))
    dp, ds = np.zeros(max(n, 1), np.float32), np.zeros(max(n, 1), np.float32)
    nd, npt, ns = C.c_uint32(), C.c_uint32(), C.c_uint32()
    vp = lambda a: None if a is None else a.ctypes.data_as(C.c_void_p)
    _lib.check(_lib.load().sailor_host_lights_in_frustum(_fp(pl), _fp(cp), n, vp(t), vp(st), vp(act), vp(pos), vp(bd), vp(od), C.byref(nd), vp(op), vp(dp), C.byref(npt),
                                                         vp(os_), vp(ds), C.byref(ns)), "lights_in_frustum")
    return od[: nd.value].copy(), (op[: npt.value].copy(), dp[: npt.value].copy()), (os_[: ns.value].copy(), ds[: ns.value].copy())


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
