"""ctypes binding of the C-ABI declared in include/sailor_hip.h (libsailor_hip.so).

This is the same binding a maintainer of the reference would write for its HIP backend;
the Python layer above it only does plumbing: device memory and streams come from torch, the arithmetic is in the
hand-written HIP kernels behind these entry points.  There is no CPU fallback: if the shared library is missing,
or a device entry point reports an error, a SailorHipError is raised.
"""
from __future__ import annotations

import ctypes as C
import os
from pathlib import Path

_CSRC = Path(__file__).resolve().parent / "csrc"
LIB_PATH = _CSRC / "libsailor_hip.so"

TILE = 16            # Constants.glsl:13
CANDIDATES = 196     # Constants.glsl:14
LIGHTS_PER_TILE = 128  # Constants.glsl:15
NUM_CASCADES = 4     # Constants.glsl:23

CTX_OWN_STREAM = 1
CULL_DEFAULT = 0
CULL_BRUTE_FORCE = 1
CULL_RAW_DEPTH = 2
CULL_INTERVAL_MASKS = 4
CULL_DEFER_PACK = 8
CULL_PREPARE_LIGHTS = 16
CULL_BAND_SELECT = 32
CULL_NO_BAND_SELECT = 64
CULL_PREPARE_SELECTED = 128

RASTER_CLEAR, RASTER_CULL_BACK = 1, 2
SHADOWMAP_R16F = 0
SHADOWMAP_RGBA32F = 1
SHADOWMAP_R32F = 2


class SailorHipError(RuntimeError):
    def __init__(self, status: int, where: str, detail: str = ""):
        self.status = status
        super().__init__(f"{where}: status {status} ({status_string(status)}) {detail}".rstrip())


class UboFrameData(C.Structure):  # RHI/Types.h:751-761
    _fields_ = [("view", C.c_float * 16), ("projection", C.c_float * 16), ("invProjection", C.c_float * 16),
                ("cameraPosition", C.c_float * 4), ("viewportSize", C.c_int32 * 2), ("cameraZNearZFar", C.c_float * 2),
                ("currentTime", C.c_float), ("deltaTime", C.c_float)]


class LightCullPushConstants(C.Structure):  # FrameGraph/LightCullingNode.h:25-31
    _fields_ = [("invViewProjection", C.c_float * 16), ("viewportSize", C.c_int32 * 2), ("numTiles", C.c_int32 * 2),
                ("lightsNum", C.c_int32), ("_pad", C.c_int32)]


class LightShaderData(C.Structure):  # ECS/LightingECS.h:71-81
    _fields_ = [("type", C.c_uint32), ("shadowType", C.c_uint32), ("_pad0", C.c_uint32 * 2),
                ("worldPosition", C.c_float * 3), ("_pad1", C.c_float),
                ("direction", C.c_float * 3), ("_pad2", C.c_float),
                ("intensity", C.c_float * 3), ("_pad3", C.c_float),
                ("attenuation", C.c_float * 3), ("_pad4", C.c_float),
                ("cutOff", C.c_float * 2), ("_pad5", C.c_float * 2),
                ("bounds", C.c_float * 3), ("_pad6", C.c_float)]


class Band(C.Structure):
    _fields_ = [("tileRowBegin", C.c_int32), ("tileRowEnd", C.c_int32), ("fbRowBegin", C.c_int32), ("fbRowCount", C.c_int32)]

    def __repr__(self):
        return f"Band(tileRows=[{self.tileRowBegin},{self.tileRowEnd}), fbRows=[{self.fbRowBegin},{self.fbRowBegin + self.fbRowCount}))"


class CsmDesc(C.Structure):
    _fields_ = [("lightsMatrices", (C.c_float * 16) * NUM_CASCADES), ("maps", C.c_void_p * NUM_CASCADES),
                ("width", C.c_int32 * NUM_CASCADES), ("height", C.c_int32 * NUM_CASCADES), ("format", C.c_int32 * NUM_CASCADES)]


class CsmView(C.Structure):  # include/sailor_hip.h SailorCsmView (the transform half of CSMLightState, ECS/LightingECS.cpp:14-38)
    _fields_ = [("componentIndex", C.c_uint32), ("cameraPosition", C.c_float * 4), ("cameraRotation", C.c_float * 4),
                ("lightPosition", C.c_float * 4), ("lightRotation", C.c_float * 4)]


class HiZDesc(C.Structure):
    _fields_ = [("pyramid", C.c_void_p), ("width", C.c_int32), ("height", C.c_int32), ("levels", C.c_int32)]


class IblDesc(C.Structure):  # include/sailor_hip.h SailorIblDesc (Standard.shader bindings 3, 4, 5, 9)
    _fields_ = [("irradiance", C.c_void_p), ("irrSize", C.c_int32), ("env", C.c_void_p), ("envSize", C.c_int32), ("envLevels", C.c_int32),
                ("brdfLut", C.c_void_p), ("lutW", C.c_int32), ("lutH", C.c_int32), ("ao", C.c_void_p)]


assert C.sizeof(UboFrameData) == 232 and C.sizeof(LightCullPushConstants) == 88 and C.sizeof(LightShaderData) == 112

_P = C.c_void_p
# name -> (restype, argtypes); one entry per symbol declared in include/sailor_hip.h
SIGNATURES = {
    "sailor_hip_version": (C.c_int, []),
    "sailor_hip_status_string": (C.c_char_p, [C.c_int]),
    "sailor_hip_device_count": (C.c_int, [C.POINTER(C.c_int)]),
    "sailor_hip_context_create": (C.c_int, [C.c_int, _P, C.c_uint32, C.POINTER(_P)]),
    "sailor_hip_context_destroy": (C.c_int, [_P]),
    "sailor_hip_context_synchronize": (C.c_int, [_P]),
    "sailor_hip_context_stream": (C.c_int, [_P, C.POINTER(_P)]),
    "sailor_hip_context_last_error": (C.c_char_p, [_P]),
    "sailor_hip_buffer_create": (C.c_int, [_P, C.c_size_t, C.POINTER(_P)]),
    "sailor_hip_buffer_free": (C.c_int, [_P, _P]),
    "sailor_hip_buffer_upload": (C.c_int, [_P, _P, C.c_size_t, _P, C.c_size_t]),
    "sailor_hip_buffer_download": (C.c_int, [_P, _P, _P, C.c_size_t, C.c_size_t]),
    "sailor_hip_buffer_fill_u32": (C.c_int, [_P, _P, C.c_size_t, C.c_uint32, C.c_size_t]),
    "sailor_hip_num_tiles": (C.c_int, [C.c_int32, C.c_int32, C.POINTER(C.c_int32), C.POINTER(C.c_int32)]),
    "sailor_hip_band_whole_frame": (C.c_int, [C.c_int32, C.c_int32, C.POINTER(Band)]),
    "sailor_hip_band_from_tile_rows": (C.c_int, [C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.POINTER(Band)]),
    "sailor_hip_band_for_rank": (C.c_int, [C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.POINTER(Band)]),
    "sailor_hip_band_is_valid": (C.c_int, [C.c_int32, C.c_int32, C.POINTER(Band)]),
    "sailor_hip_light_cull_workspace_size": (C.c_size_t, [C.c_int32, C.c_int32, C.c_int32, C.POINTER(Band)]),
    "sailor_hip_light_cull": (C.c_int, [_P, C.POINTER(UboFrameData), C.POINTER(LightCullPushConstants), _P, _P, _P, _P, C.c_size_t,
                                        _P, C.c_size_t, C.POINTER(Band), C.c_uint32]),
    "sailor_hip_linearize_depth": (C.c_int, [_P, C.POINTER(UboFrameData), _P, _P, C.c_int32, C.c_int32]),
    "sailor_hip_light_cull_diagnostics": (C.c_int, [_P, C.c_int32, C.c_int32, C.c_int32, C.POINTER(Band), _P, C.POINTER(C.c_uint64)]),
    "sailor_hip_light_grid_rebase": (C.c_int, [_P, _P, C.c_int32, C.c_uint32]),
    "sailor_hip_shade": (C.c_int, [_P, C.POINTER(UboFrameData), _P, C.c_size_t, _P, C.c_int32, _P, _P, C.POINTER(CsmDesc), _P, C.POINTER(Band)]),
    "sailor_hip_shade_ex": (C.c_int, [_P, C.POINTER(UboFrameData), _P, C.c_size_t, _P, C.c_int32, _P, _P, C.POINTER(CsmDesc), C.POINTER(IblDesc), _P,
                                      C.POINTER(Band), _P]),
    "sailor_hip_prepared_lights_size": (C.c_size_t, [C.c_int32]),
    "sailor_hip_prepare_lights": (C.c_int, [_P, _P, C.c_int32, C.c_int32, C.c_int32, _P, C.c_size_t]),
    "sailor_hip_prepared_lights_views": (C.c_int, [C.c_int32, _P, C.POINTER(_P), C.POINTER(_P), C.POINTER(_P)]),
    "sailor_hip_light_cull_prepared": (C.c_int, [_P, C.POINTER(UboFrameData), C.POINTER(LightCullPushConstants), _P, _P, _P, _P, C.c_size_t,
                                                 _P, C.c_size_t, C.POINTER(Band), C.c_uint32, _P, C.c_int32]),
    "sailor_hip_shade_prepared": (C.c_int, [_P, C.POINTER(UboFrameData), _P, C.c_size_t, _P, C.c_int32, _P, _P, C.POINTER(CsmDesc), C.POINTER(IblDesc), _P,
                                            C.POINTER(Band), _P, _P, C.c_int32]),
    "sailor_hip_light_cull_tile_order": (_P, [C.c_int32, C.c_int32, C.c_int32, C.POINTER(Band), _P]),
    "sailor_hip_light_cull_tile_lists": (C.c_int, [C.c_int32, C.c_int32, C.c_int32, C.POINTER(Band), _P, C.POINTER(_P), C.POINTER(_P)]),
    "sailor_hip_light_cull_pack": (C.c_int, [_P, C.c_int32, C.c_int32, C.c_int32, C.POINTER(Band), _P, _P, _P, C.c_size_t]),
    "sailor_hip_shade_tile_lists": (C.c_int, [_P, C.POINTER(UboFrameData), _P, C.c_size_t, _P, C.c_int32, _P, _P, C.POINTER(CsmDesc), C.POINTER(IblDesc), _P,
                                              C.POINTER(Band), _P, _P, C.c_int32]),
    "sailor_hip_context_wait_for": (C.c_int, [_P, _P]),
    "sailor_hip_context_time_launches": (C.c_int, [_P, C.c_int32, C.c_int32]),
    "sailor_hip_context_timed_launch_ms": (C.c_int, [_P, C.c_int32, C.POINTER(C.c_float)]),
    "sailor_hip_copy_probe": (C.c_int, [_P, _P, _P, C.c_size_t]),
    "sailor_hip_marker": (C.c_int, [_P]),
    "sailor_hip_context_launch_log": (C.c_int, [_P, C.POINTER(C.c_uint64), C.POINTER(C.c_char_p), C.c_int32]),
    "sailor_hip_light_cull_band_selection": (C.c_int, [C.c_int32, C.c_int32, C.c_int32, C.POINTER(Band), _P, C.POINTER(_P), C.POINTER(_P)]),
    "sailor_hip_evsm_blur": (C.c_int, [_P, _P, _P, C.c_int32, C.c_int32, C.c_int32, C.c_int32]),
    "sailor_hip_evsm_blur_pass": (C.c_int, [_P, _P, _P, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32]),
    "sailor_hip_compute_brdf_lut": (C.c_int, [_P, _P, C.c_int32, C.c_int32]),
    "sailor_hip_self_check_exact_math": (C.c_int, [_P, _P, C.POINTER(C.c_uint64)]),
    "sailor_hip_compute_irradiance_map": (C.c_int, [_P, _P, C.c_int32, C.c_int32, _P, C.c_int32]),
    "sailor_hip_prefilter_env_map": (C.c_int, [_P, _P, _P, C.c_int32, C.c_int32]),
    "sailor_hip_equirect_to_cube": (C.c_int, [_P, _P, C.c_int32, C.c_int32, C.c_int32, _P, C.c_int32, C.c_int32, C.c_int32]),
    "sailor_hip_generate_mipmaps_cube": (C.c_int, [_P, _P, C.c_int32, C.c_int32]),
    "sailor_hip_prefilter_env_level": (C.c_int, [_P, _P, _P, C.c_int32, C.c_int32, C.c_int32, C.c_float]),
    "sailor_hip_buffer_copy": (C.c_int, [_P, _P, C.c_size_t, _P, C.c_size_t, C.c_size_t]),
    "sailor_hip_ecs_sweep": (C.c_int, [_P, C.c_uint32, _P, _P, C.POINTER(C.c_uint32), C.c_uint32, _P, C.POINTER(C.c_float), _P, _P, _P]),
    "sailor_hip_ecs_sweep_range": (C.c_int, [_P, C.c_uint32, _P, _P, C.POINTER(C.c_uint32), C.c_uint32, _P, C.POINTER(C.c_float), _P, _P, _P, C.c_uint32, C.c_uint32]),
    "sailor_hip_ecs_range_for_rank": (C.c_int, [C.c_uint32, C.c_int32, C.c_int32, C.POINTER(C.c_uint32), C.POINTER(C.c_uint32), C.POINTER(C.c_uint32)]),
    "sailor_hip_exchange_adapt": (C.c_int, [_P, _P, _P, _P]),
    "sailor_hip_exchange_set_slot_words": (C.c_int, [_P, C.c_size_t]),
    "sailor_hip_exchange_visibility": (C.c_int, [_P, _P, C.c_int32, C.c_int32, C.c_uint32, _P, C.c_size_t]),
    "sailor_hip_mesh_frustum_cull": (C.c_int, [_P, C.POINTER(UboFrameData), _P, C.c_uint32, C.c_uint32]),
    "sailor_hip_raster_coarse_words": (C.c_size_t, [C.c_int32, C.c_int32]),
    "sailor_hip_raster_depth": (C.c_int, [_P, C.POINTER(C.c_float), _P, _P, C.c_uint32, _P, _P, C.c_uint32, C.c_int32, C.c_int32, _P, C.c_uint32, _P]),
    "sailor_hip_raster_depth_camera": (C.c_int, [_P, C.POINTER(UboFrameData), _P, _P, C.c_uint32, _P, _P, C.c_uint32, C.c_int32, C.c_int32, _P, C.c_uint32, _P]),
    "sailor_hip_shadow_resolve": (C.c_int, [_P, _P, C.c_int32, C.c_int32, C.c_int32, _P]),
    "sailor_hip_csm_caster_masks": (C.c_int, [_P, C.c_uint32, _P, C.POINTER(C.c_float), C.c_uint32, _P]),
    "sailor_hip_hiz_downscale": (C.c_int, [_P, _P, C.c_int32, C.c_int32, _P, C.c_int32, C.c_int32]),
    "sailor_hip_hiz_build": (C.c_int, [_P, _P, C.c_int32, C.c_int32, _P, C.c_int32, C.c_int32, C.c_int32]),
    "sailor_hip_mesh_cull_flags": (C.c_int, [_P, C.POINTER(UboFrameData), _P, C.c_uint32, C.c_uint32, C.POINTER(HiZDesc)]),
    "sailor_hip_mesh_cull_compact_ex": (C.c_int, [_P, C.POINTER(UboFrameData), _P, C.c_uint32, C.c_uint32, _P, C.c_uint32, _P, C.c_size_t, C.POINTER(HiZDesc)]),
    "sailor_hip_mesh_cull_workspace_bytes": (C.c_size_t, [C.c_uint32, C.c_uint32]),
    "sailor_hip_mesh_cull_compact": (C.c_int, [_P, C.POINTER(UboFrameData), _P, C.c_uint32, C.c_uint32, _P, C.c_uint32, _P, C.c_size_t]),
    "sailor_hip_allgather_u32": (C.c_int, [_P, _P, _P, _P, C.c_size_t]),
    "sailor_hip_exchange_workspace_size": (C.c_size_t, [C.c_int32, C.c_int32, C.c_int32]),
    "sailor_hip_exchange_light_lists": (C.c_int, [_P, _P, C.c_int32, C.c_int32, C.c_int32, C.c_int32, _P, _P, _P, _P, C.c_size_t, _P, C.c_size_t]),
    "sailor_hip_stitch_light_lists": (C.c_int, [_P, C.c_int32, C.c_int32, C.c_int32, _P, _P, C.c_size_t, _P, C.c_size_t, _P, _P, C.c_size_t]),
    "sailor_hip_exchange_workspace_size_rows": (C.c_size_t, [C.c_int32, C.c_int32, C.c_int32, _P]),
    "sailor_hip_exchange_light_lists_rows": (C.c_int, [_P, _P, C.c_int32, C.c_int32, C.c_int32, C.c_int32, _P, _P, _P, _P, C.c_size_t, _P, C.c_size_t, _P, C.c_size_t]),
    "sailor_hip_stitch_light_lists_rows": (C.c_int, [_P, C.c_int32, C.c_int32, C.c_int32, _P, _P, _P, C.c_size_t, _P, C.c_size_t, _P, C.c_size_t, _P, C.c_size_t]),
    "sailor_host_perspective_rh": (C.c_int, [C.c_float, C.c_float, C.c_float, C.c_float, C.POINTER(C.c_float)]),
    "sailor_host_mat4_inverse": (C.c_int, [C.POINTER(C.c_float), C.POINTER(C.c_float)]),
    "sailor_host_mat4_mul": (C.c_int, [C.POINTER(C.c_float), C.POINTER(C.c_float), C.POINTER(C.c_float)]),
    "sailor_host_transform_matrix": (C.c_int, [C.POINTER(C.c_float), C.POINTER(C.c_float)]),
    "sailor_host_fill_frame_data": (C.c_int, [C.POINTER(C.c_float), C.c_float, C.c_float, C.c_float, C.c_float, C.c_int32, C.c_int32,
                                              C.c_float, C.c_float, C.POINTER(UboFrameData)]),
    "sailor_host_extract_frustum_planes": (C.c_int, [C.POINTER(C.c_float), C.c_float, C.c_float, C.c_float, C.c_float,
                                                     C.POINTER(C.c_float), C.POINTER(C.c_float)]),
    "sailor_host_extract_frustum_planes_matrix": (C.c_int, [C.POINTER(C.c_float), C.POINTER(C.c_float), C.POINTER(C.c_float)]),
    "sailor_host_csm_matrices": (C.c_int, [C.POINTER(C.c_float), C.POINTER(C.c_float), C.c_float, C.c_float, C.c_float, C.c_float,
                                           C.POINTER(C.c_float)]),
    "sailor_host_overlaps_sphere": (C.c_int, [C.POINTER(C.c_float), C.POINTER(C.c_float)]),
    "sailor_host_contains_sphere": (C.c_int, [C.POINTER(C.c_float), C.POINTER(C.c_float)]),
    "sailor_host_lights_in_frustum": (C.c_int, [C.POINTER(C.c_float), C.POINTER(C.c_float), C.c_uint32, _P, _P, _P, _P, _P, _P, C.POINTER(C.c_uint32),
                                                _P, _P, C.POINTER(C.c_uint32), _P, _P, C.POINTER(C.c_uint32)]),
    "sailor_host_csm_snapshots_create": (_P, []),
    "sailor_host_csm_snapshots_clone": (_P, [_P]),
    "sailor_host_csm_snapshots_destroy": (None, [_P]),
    "sailor_host_csm_snapshots_count": (C.c_uint32, [_P]),
    "sailor_host_csm_snapshot_get": (C.c_int, [_P, C.c_uint32, C.c_uint32, C.POINTER(C.c_uint32), _P, _P, C.POINTER(C.c_int32), C.POINTER(CsmView)]),
    "sailor_host_csm_plan_passes": (C.c_int, [_P, C.c_uint32, C.c_uint32, C.c_uint32, _P, _P, _P, C.POINTER(CsmView), _P, _P]),
    "sailor_host_pack_light": (C.c_int, [C.c_uint32, C.c_uint32] + [C.POINTER(C.c_float)] * 6 + [C.POINTER(LightShaderData)]),
}

_lib = None


def load() -> C.CDLL:
    """Load libsailor_hip.so (built in-tree by __graft_entry__.build() / `make -C sailor_amd/csrc`)."""
    global _lib
    if _lib is None:
        # torch ships its own HIP runtime (torch/lib/libamdhip64.so).  Streams and device pointers are handed across this
        # boundary, so both sides must live in ONE runtime instance: import torch first and let the loader resolve our
        # libamdhip64 dependency to the copy that is already mapped.
        import torch  # noqa: F401
        path = Path(os.environ.get("SAILOR_HIP_LIB", LIB_PATH))
        if not path.exists():
            raise SailorHipError(-2, "load", f"{path} is missing: run `python -c 'import __graft_entry__ as g; g.build()'` "
                                             "or `make -C sailor_amd/csrc` (there is no CPU fallback)")
        lib = C.CDLL(str(path))
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(lib, name)  # AttributeError here == the library does not export a declared symbol
            fn.restype = res
            fn.argtypes = args
        _lib = lib
    return _lib


def status_string(status: int) -> str:
    try:
        return load().sailor_hip_status_string(status).decode()
    except Exception:  # library itself unavailable
        return "?"


def check(status: int, where: str, ctx=None) -> None:
    if status != 0:
        detail = ""
        if ctx is not None:
            detail = load().sailor_hip_context_last_error(ctx).decode()
        raise SailorHipError(status, where, detail)
