"""ctypes binding of sailor_amd/runtime/libsailor_runtime.so -- the C++ host mirror (RHI / FrameGraph / GraphicsDriver/HIP / ECS).

The harness drives the path the way the engine does: a Renderer with the HIP backend, a frame graph built from node NAMES
("LightCulling", "RenderScene"), a scene snapshot, `RHIFrameGraph::Process` per frame.  Used by the tests; device memory is
torch's."""
from __future__ import annotations

import ctypes as C
from pathlib import Path

import numpy as np

from . import _lib

LIB_PATH = Path(__file__).resolve().parent / "runtime" / "libsailor_runtime.so"
_rt = None


def load() -> C.CDLL:
    global _rt
    if _rt is None:
        _lib.load()  # torch's HIP runtime + libsailor_hip.so first
        if not LIB_PATH.exists():
            raise _lib.SailorHipError(-2, "runtime", f"{LIB_PATH} is missing: make -C sailor_amd/runtime")
        rt = C.CDLL(str(LIB_PATH))
        P = C.c_void_p
        rt.sailor_rt_create.restype = P
        rt.sailor_rt_create.argtypes = [C.c_int, P, C.c_int, C.POINTER(C.c_int)]
        rt.sailor_rt_destroy.argtypes = [P]
        rt.sailor_rt_node_registered.argtypes = [C.c_char_p]
        rt.sailor_rt_build_graph.argtypes = [P, C.POINTER(C.c_char_p), C.c_int]
        rt.sailor_rt_set_camera.argtypes = [P, P, C.c_float, C.c_float, C.c_float, C.c_float, C.c_int, C.c_int]
        rt.sailor_rt_set_lights.argtypes = [P, P, C.c_int]
        rt.sailor_rt_add_light.argtypes = [P, C.c_uint32, C.c_uint32, P, P, P, P, P]
        rt.sailor_rt_tick_lights.argtypes = [P]
        rt.sailor_rt_update_light.argtypes = [P, C.c_int, P, P, P, P, P]
        rt.sailor_rt_set_light_state.argtypes = [P, C.c_int, C.c_int, C.c_int, C.c_int, C.c_longlong]
        rt.sailor_rt_light_uploads.argtypes = [P, P, C.c_int]
        rt.sailor_rt_total_num_lights.restype = C.c_uint32
        rt.sailor_rt_total_num_lights.argtypes = [P]
        rt.sailor_rt_set_depth.argtypes = [P, P, C.c_int, C.c_int]
        rt.sailor_rt_set_raw_depth.argtypes = [P, P, C.c_int, C.c_int]
        rt.sailor_rt_set_surface.argtypes = [P, P, P, C.c_int, C.c_int]
        rt.sailor_rt_set_shadow_maps.argtypes = [P, P, P, P, P]
        rt.sailor_rt_set_ibl.argtypes = [P, P, C.c_int, P, C.c_int, C.c_int, P, C.c_int, C.c_int, P, C.c_int, C.c_int]
        rt.sailor_rt_blur_shadow_map.argtypes = [P, P, P, C.c_int, C.c_float, C.c_float]
        rt.sailor_rt_set_sky_cubemap.argtypes = [P, P, C.c_int, C.c_int, C.c_int, P, C.c_int, C.c_int]
        rt.sailor_rt_set_environment_map.argtypes = [P, P, C.c_int, C.c_int, C.c_int, C.c_int]
        rt.sailor_rt_sampler.restype = P
        rt.sailor_rt_sampler.argtypes = [P, C.c_char_p, C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_int)]
        rt.sailor_rt_build_depth_highz.argtypes = [P, P, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(P)]
        rt.sailor_rt_parse_renderer.argtypes = [C.c_char_p, C.c_int, C.c_int, C.c_char_p, C.c_int]
        rt.sailor_rt_parse_world.argtypes = [C.c_char_p, C.c_char_p, C.c_int]
        rt.sailor_rt_load_world.argtypes = [P, C.c_char_p, C.c_int, C.c_int, P, P, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int)]
        rt.sailor_rt_load_renderer.argtypes = [P, C.c_char_p, C.POINTER(C.c_int), C.POINTER(C.c_int)]
        rt.sailor_rt_render_target.restype = P
        rt.sailor_rt_render_target.argtypes = [P, C.c_char_p, C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_int)]
        rt.sailor_rt_set_render_target.argtypes = [P, C.c_char_p, P, C.c_int, C.c_int]
        rt.sailor_rt_shadow_pass.argtypes = [P, C.POINTER(C.c_float), P, C.c_uint32, P, C.c_uint32, P, C.c_uint32, C.c_uint32, P, C.c_int, C.c_int, C.c_float, C.c_float]
        rt.sailor_rt_gpu_culling.argtypes = [P, P, C.c_uint32, C.c_uint32, P, C.c_uint32]
        rt.sailor_rt_process_frame.argtypes = [P]
        rt.sailor_rt_process_frame_overwriting_lists.argtypes = [P, P, C.c_size_t, P, C.c_size_t]
        rt.sailor_rt_set_frame_split.argtypes = [P, C.c_int, C.c_int, P]
        rt.sailor_rt_exchange_light_lists.argtypes = [P, P, C.c_size_t, P, C.c_size_t]
        rt.sailor_rt_wait_idle.argtypes = [P]
        rt.sailor_rt_buffer.restype = P
        rt.sailor_rt_buffer.argtypes = [P, C.c_char_p, C.POINTER(C.c_size_t)]
        rt.sailor_rt_ecs_sweep.argtypes = [P, P, P, P, C.c_uint32, P, C.c_uint32, C.POINTER(P), C.POINTER(P), C.POINTER(P)]
        _rt = rt
    return _rt


def parse_world(text: str):
    """WorldPrefab::Deserialize + World::Instantiate of a `.world` text (no device needed): (number of game objects, summary string)"""
    rt = load()
    buf = C.create_string_buffer(1 << 16)
    n = rt.sailor_rt_parse_world(text.encode(), buf, len(buf))
    if n < 0:
        raise ValueError(buf.value.decode())
    return n, buf.value.decode()


def parse_renderer(text: str, viewport_width: int, viewport_height: int):
    """FrameGraphAsset::Deserialize of a `.renderer` text (no device needed): (number of frame nodes, summary string); raises on a parse error"""
    rt = load()
    buf = C.create_string_buffer(1 << 16)
    n = rt.sailor_rt_parse_renderer(text.encode(), viewport_width, viewport_height, buf, len(buf))
    if n < 0:
        raise ValueError(buf.value.decode())
    return n, buf.value.decode()


class Runtime:
    def __init__(self, device_index: int = 0, stream_handle: int = 0):
        self.rt = load()
        st = C.c_int(0)
        self.h = self.rt.sailor_rt_create(device_index, C.c_void_p(stream_handle), 0, C.byref(st))
        if not self.h:
            raise _lib.SailorHipError(st.value, "sailor_rt_create")

    def close(self):
        if self.h:
            self.rt.sailor_rt_destroy(self.h)
            self.h = None

    def build_graph(self, names):
        arr = (C.c_char_p * len(names))(*[n.encode() for n in names])
        if self.rt.sailor_rt_build_graph(self.h, arr, len(names)) != 0:
            raise ValueError(f"unknown frame graph node in {names}")

    def set_camera(self, cam):
        w = np.ascontiguousarray(cam.world, np.float32)
        self.rt.sailor_rt_set_camera(self.h, w.ctypes.data, cam.fov, cam.aspect, cam.z_near, cam.z_far, cam.width, cam.height)

    def set_lights(self, lights: np.ndarray):
        raw = np.ascontiguousarray(lights).view(np.uint8)
        self.rt.sailor_rt_set_lights(self.h, raw.ctypes.data, len(lights))

    @staticmethod
    def _f3(v):
        return None if v is None else np.ascontiguousarray(v, np.float32).ctypes.data_as(C.c_void_p)

    def add_light(self, light_type: int, shadow_type: int, position, direction, intensity, bounds, cut_off_degrees=None) -> int:
        """register one light as a component (LightData; cut-off in degrees); packed by the next tick_lights()"""
        keep = [np.ascontiguousarray(v, np.float32) for v in (position, direction, intensity, bounds)]
        cut = np.ascontiguousarray(cut_off_degrees, np.float32) if cut_off_degrees is not None else None
        return self.rt.sailor_rt_add_light(self.h, light_type, shadow_type, *[k.ctypes.data for k in keep], cut.ctypes.data if cut is not None else None)

    def update_light(self, index: int, position=None, direction=None, intensity=None, bounds=None, cut_off_degrees=None):
        """new parameters for one light + MarkDirty"""
        keep = [None if v is None else np.ascontiguousarray(v, np.float32) for v in (position, direction, intensity, bounds, cut_off_degrees)]
        assert self.rt.sailor_rt_update_light(self.h, index, *[None if k is None else k.ctypes.data for k in keep]) == 0

    def set_light_state(self, index: int, active=None, dirty=None, mobility=None, owner_frame_last_change=None):
        """TComponent::SetActive / MarkDirty, the owner's mobility (0 static, 1 stationary, 2 dynamic) and GameObject::GetFrameLastChange"""
        enc = lambda v: -1 if v is None else int(v)
        assert self.rt.sailor_rt_set_light_state(self.h, index, enc(active), enc(dirty), enc(mobility), enc(owner_frame_last_change)) == 0

    def tick_lights(self):
        """LightingECS::Tick + FillLightingData; returns the copies it recorded as [(first record slot, record count)]"""
        self.rt.sailor_rt_tick_lights(self.h)
        buf = np.zeros(2 * 4096, np.uint32)
        n = self.rt.sailor_rt_light_uploads(self.h, buf.ctypes.data, 4096)
        return [(int(buf[2 * i]), int(buf[2 * i + 1])) for i in range(min(n, 4096))]

    def total_num_lights(self) -> int:
        return self.rt.sailor_rt_total_num_lights(self.h)

    def set_depth(self, depth_tensor):
        self.rt.sailor_rt_set_depth(self.h, depth_tensor.data_ptr(), depth_tensor.shape[1], depth_tensor.shape[0])

    def set_raw_depth(self, raw_tensor):
        """after set_depth: the LinearizeDepth node reads this and writes the tensor given to set_depth"""
        self.rt.sailor_rt_set_raw_depth(self.h, raw_tensor.data_ptr(), raw_tensor.shape[1], raw_tensor.shape[0])

    def set_ibl(self, irradiance, env_chain, env_size, env_levels, lut, ao):
        """device tensors: irradiance [6,S,S,4], env mip chain (flat), lut [H,W,2], ao [H,W] or None"""
        self.rt.sailor_rt_set_ibl(self.h, irradiance.data_ptr(), irradiance.shape[1], env_chain.data_ptr(), env_size, env_levels,
                                  lut.data_ptr(), lut.shape[1], lut.shape[0], ao.data_ptr() if ao is not None else None,
                                  ao.shape[1] if ao is not None else 0, ao.shape[0] if ao is not None else 0)

    def blur_shadow_map(self, moments, temp, radius_umbra, radius_penumbra):
        """the blur section of ShadowPrepassNode::Process over a float32 [S, S, 4] device tensor, in place"""
        return self.rt.sailor_rt_blur_shadow_map(self.h, moments.data_ptr(), temp.data_ptr(), moments.shape[0], radius_umbra, radius_penumbra)

    def set_sky_cubemap(self, chain, size, levels, irradiance_size=0, ao=None):
        """publish the raw environment cube (flat RGBA32F mip chain, device tensor) as "g_skyCubemap" and mark the Environment node dirty"""
        return self.rt.sailor_rt_set_sky_cubemap(self.h, chain.data_ptr(), size, levels, irradiance_size, ao.data_ptr() if ao is not None else None,
                                                 ao.shape[1] if ao is not None else 0, ao.shape[0] if ao is not None else 0)

    def set_environment_map(self, equirect, repeat=True, irradiance_size=0):
        """hand the Environment node its "EnvironmentMap" panorama (float32 [H, W, 4] device tensor): it converts it to the raw 512 x 512 x 6 cube"""
        return self.rt.sailor_rt_set_environment_map(self.h, equirect.data_ptr(), equirect.shape[1], equirect.shape[0], 1 if repeat else 0, irradiance_size)

    def sampler(self, name: str):
        """(device pointer, width, height, mip levels) of a sampler published by the graph's nodes"""
        w, h, l = C.c_int(0), C.c_int(0), C.c_int(0)
        p = self.rt.sailor_rt_sampler(self.h, name.encode(), C.byref(w), C.byref(h), C.byref(l))
        return p, w.value, h.value, l.value

    def build_depth_highz(self, depth, width, height, levels):
        """run the DepthHighZ node over a float32 [h, w] device tensor; returns (status, device pointer of the level-major pyramid)"""
        p = C.c_void_p()
        st = self.rt.sailor_rt_build_depth_highz(self.h, depth.data_ptr() if depth is not None else None, depth.shape[1] if depth is not None else 0,
                                                 depth.shape[0] if depth is not None else 0, width, height, levels, C.byref(p))
        return st, p.value

    def load_world(self, text: str, width: int, height: int, max_objects: int = 4096):
        """load a `.world`: camera -> scene view, LightComponents -> LightingECS (packed by Tick); returns (transforms float32 [n, 12], parents uint32 [n],
        number of lights, number of mesh renderers) for the ECS sweep"""
        import numpy as np
        tr = np.zeros((max_objects, 12), np.float32)
        par = np.zeros(max_objects, np.uint32)
        nl, nm = C.c_int(0), C.c_int(0)
        n = self.rt.sailor_rt_load_world(self.h, text.encode(), width, height, tr.ctypes.data, par.ctypes.data, max_objects, C.byref(nl), C.byref(nm))
        if n < 0:
            raise ValueError("the .world text does not parse")
        return tr[:n], par[:n], nl.value, nm.value

    def load_renderer(self, text: str):
        """FrameGraphImporter::BuildFrameGraph from a `.renderer` text: (nodes created, nodes without a class here, render targets created)"""
        skipped, targets = C.c_int(0), C.c_int(0)
        n = self.rt.sailor_rt_load_renderer(self.h, text.encode(), C.byref(skipped), C.byref(targets))
        if n < 0:
            raise ValueError("the .renderer text does not parse")
        return n, skipped.value, targets.value

    def render_target(self, name: str):
        w, h, l = C.c_int(0), C.c_int(0), C.c_int(0)
        p = self.rt.sailor_rt_render_target(self.h, name.encode(), C.byref(w), C.byref(h), C.byref(l))
        return p, w.value, h.value, l.value

    def set_render_target(self, name: str, tensor):
        """publish a float32 [h, w] device tensor as a named render target (DepthBuffer, ...)"""
        self.rt.sailor_rt_set_render_target(self.h, name.encode(), tensor.data_ptr(), tensor.shape[1], tensor.shape[0])

    def shadow_pass(self, light_matrix, positions, indices, models, first_instance, instance_count, shadow_map, evsm, radius_umbra=0.0, radius_penumbra=0.0):
        """one shadow pass of ShadowPrepassNode (caster draw + fragment stage + blur) over device tensors; the map is float32 [S, S, 4] (EVSM) or float16 [S, S]"""
        lm = np.ascontiguousarray(light_matrix, np.float32).reshape(16)
        return self.rt.sailor_rt_shadow_pass(self.h, lm.ctypes.data_as(C.POINTER(C.c_float)), positions.data_ptr(), positions.shape[0], indices.data_ptr(),
                                             indices.numel(), models.data_ptr(), first_instance, instance_count, shadow_map.data_ptr(), shadow_map.shape[0],
                                             1 if evsm else 0, radius_umbra, radius_penumbra)

    def gpu_culling(self, instances, num_instances, first_instance, batches, num_batches):
        """the "GPU Culling" Dispatch of RHIRecordDrawCallGPUCulling over uint8 / int32 device tensors, in place"""
        return self.rt.sailor_rt_gpu_culling(self.h, instances.data_ptr(), num_instances, first_instance,
                                             batches.data_ptr() if batches is not None else None, num_batches)

    def set_surface(self, surface_tensor, radiance_tensor):
        self.rt.sailor_rt_set_surface(self.h, surface_tensor.data_ptr(), radiance_tensor.data_ptr(), surface_tensor.shape[2], surface_tensor.shape[1])

    def set_shadow_maps(self, map_tensors, formats, lights_matrices):
        ptrs = (C.c_void_p * 4)(*[t.data_ptr() for t in map_tensors])
        sizes = (C.c_int * 4)(*[t.shape[0] for t in map_tensors])
        fmts = (C.c_int * 4)(*formats)
        lm = np.ascontiguousarray(lights_matrices, np.float32).reshape(64)
        self.rt.sailor_rt_set_shadow_maps(self.h, ptrs, sizes, fmts, lm.ctypes.data)

    def set_frame_split(self, rank: int, world_size: int, comm=None):
        """this runtime renders tile-row band `rank` of `world_size`; per-pixel targets set afterwards hold the band's rows"""
        _lib.check(self.rt.sailor_rt_set_frame_split(self.h, rank, world_size, C.c_void_p(comm) if comm else None), "sailor_rt_set_frame_split")

    def exchange_light_lists(self, global_grid, global_culled):
        _lib.check(self.rt.sailor_rt_exchange_light_lists(self.h, global_grid.data_ptr(), global_grid.numel() * global_grid.element_size(),
                                                          global_culled.data_ptr(), global_culled.numel() * global_culled.element_size()), "sailor_rt_exchange_light_lists")

    def process_frame(self) -> int:
        return self.rt.sailor_rt_process_frame(self.h)

    def process_frame_overwriting_lists(self, grid, culled) -> int:
        """one frame with an UpdateBuffer of lightsGrid / culledLights (numpy uint32 arrays or None) recorded between the LightCulling node and RenderScene"""
        import numpy as np
        g = None if grid is None else np.ascontiguousarray(grid, np.uint32)
        c = None if culled is None else np.ascontiguousarray(culled, np.uint32)
        return self.rt.sailor_rt_process_frame_overwriting_lists(self.h, None if g is None else g.ctypes.data, 0 if g is None else g.nbytes,
                                                                 None if c is None else c.ctypes.data, 0 if c is None else c.nbytes)

    def wait_idle(self):
        self.rt.sailor_rt_wait_idle(self.h)

    def buffer(self, name: str):
        n = C.c_size_t(0)
        p = self.rt.sailor_rt_buffer(self.h, name.encode(), C.byref(n))
        return p, n.value
