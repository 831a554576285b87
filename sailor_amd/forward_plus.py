"""Torch-facing handle on the HIP Forward+ path: device memory and streams come from torch, every kernel comes from
libsailor_hip.so through the C-ABI (include/sailor_hip.h).  One object = one GPU = one band of the frame.

Mirrors the two reference passes that own these buffers:
  * LightCullingNode (FrameGraph/LightCullingNode.cpp:59-77): owns `culledLights` / `lightsGrid`, dispatches the cull;
  * RenderSceneNode + Standard.shader (FrameGraph/RenderSceneNode.cpp:109): consumes them while shading.
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np
import torch

from . import _lib, host
from ._lib import Band, CsmDesc, UboFrameData

LIGHTS_PER_TILE = _lib.LIGHTS_PER_TILE


def _ptr(t: torch.Tensor | None) -> int | None:
    return None if t is None else t.data_ptr()


class HipContext:
    """RAII wrapper of SailorHipContext bound to a torch device and stream."""

    def __init__(self, device: torch.device | str | int = "cuda:0", stream: torch.cuda.Stream | None = None):
        self.device = torch.device(device)
        if self.device.type != "cuda":
            raise _lib.SailorHipError(-2, "HipContext", "a HIP device is required (there is no CPU fallback)")
        index = self.device.index if self.device.index is not None else torch.cuda.current_device()
        self.device = torch.device("cuda", index)
        self.stream = stream if stream is not None else torch.cuda.current_stream(self.device)
        handle = C.c_void_p()
        lib = _lib.load()
        _lib.check(lib.sailor_hip_context_create(index, C.c_void_p(self.stream.cuda_stream), 0, C.byref(handle)), "sailor_hip_context_create")
        self.handle = handle
        self._lib = lib

    def synchronize(self):
        _lib.check(self._lib.sailor_hip_context_synchronize(self.handle), "synchronize", self.handle)

    def time_launches(self, first_slot: int, count: int):
        """the next `count` kernel launches through this context carry event pairs on their dispatch packets (sailor_hip_context_time_launches)"""
        _lib.check(self._lib.sailor_hip_context_time_launches(self.handle, first_slot, count), "sailor_hip_context_time_launches", self.handle)

    def timed_launch_ms(self, slot: int) -> float:
        ms = C.c_float()
        _lib.check(self._lib.sailor_hip_context_timed_launch_ms(self.handle, slot, C.byref(ms)), "sailor_hip_context_timed_launch_ms", self.handle)
        return float(ms.value)

    def launch_log(self, max_names: int = 16):
        """(count, names): how many kernels the path's entry points have launched through this context, and the names of the last few, oldest first
        (sailor_hip_context_launch_log).  The kernels of ONE call = the names behind the count read in front of it."""
        count = C.c_uint64()
        names = (C.c_char_p * max_names)()
        _lib.check(self._lib.sailor_hip_context_launch_log(self.handle, C.byref(count), names, max_names), "sailor_hip_context_launch_log", self.handle)
        return int(count.value), [n.decode() for n in names if n is not None]

    def launches_of(self, fn):
        """the names of the kernels fn() launches through this context (at most 16)"""
        before, _ = self.launch_log(0)
        fn()
        after, names = self.launch_log(16)
        n = after - before
        assert n <= 16, n
        return names[len(names) - n:] if n else []

    def close(self):
        if getattr(self, "handle", None):
            self._lib.sailor_hip_context_destroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def linearize_depth(ctx: "HipContext", frame: UboFrameData, raw: torch.Tensor, out: torch.Tensor | None = None) -> torch.Tensor:
    """LinearizeDepthNode (FrameGraph/LinearizeDepthNode.cpp:22): raw reversed-Z depth rows -> positive view distance."""
    assert raw.dtype == torch.float32 and raw.dim() == 2 and raw.is_contiguous()
    if out is None:
        out = torch.empty_like(raw)
    _lib.check(ctx._lib.sailor_hip_linearize_depth(ctx.handle, C.byref(frame), _ptr(raw), _ptr(out), raw.shape[1], raw.shape[0]),
               "sailor_hip_linearize_depth", ctx.handle)
    return out


_ENV_CULL_FLAGS = int(os.environ.get("SAILOR_CULL_FLAGS", "0"))


class PreparedLights:
    """sailor_hip_prepare_lights' output for a `light` SSBO of `capacity` records: the cull's 20-byte view and the shade's staged records, derived where
    the records are uploaded (the HIP backend does it behind UpdateShaderBinding) instead of in every frame's kernels.  prepare(first, count) after
    every change of the records [first, first + count)."""

    def __init__(self, ctx: "HipContext", lights: torch.Tensor, lights_num: int, capacity: int | None = None):
        self.ctx, self.lights, self.capacity = ctx, lights, max(int(capacity if capacity is not None else lights_num), 1)
        n = ctx._lib.sailor_hip_prepared_lights_size(self.capacity)
        self.buffer = torch.empty(n, dtype=torch.uint8, device=ctx.device)
        self.prepare(0, lights_num)

    def prepare(self, first: int, count: int, ctx: "HipContext | None" = None):
        ctx = ctx or self.ctx
        _lib.check(ctx._lib.sailor_hip_prepare_lights(ctx.handle, _ptr(self.lights), first, count, self.capacity, _ptr(self.buffer), self.buffer.numel()),
                   "sailor_hip_prepare_lights", ctx.handle)

    def views(self):
        """(posRadius float32 [capacity, 4], type int32 [capacity], staged float32 [capacity, 5, 4]) as tensors over the buffer"""
        lib = self.ctx._lib
        a, b, c = C.c_void_p(), C.c_void_p(), C.c_void_p()
        _lib.check(lib.sailor_hip_prepared_lights_views(self.capacity, _ptr(self.buffer), C.byref(a), C.byref(b), C.byref(c)), "sailor_hip_prepared_lights_views")
        base = self.buffer.data_ptr()
        n = self.capacity
        f = self.buffer
        return (f[a.value - base: a.value - base + 16 * n].view(torch.float32).view(n, 4), f[b.value - base: b.value - base + 4 * n].view(torch.int32),
                f[c.value - base: c.value - base + 80 * n].view(torch.float32).view(n, 5, 4))


class ForwardPlus:
    """Cull + shade for one band of a W x H frame on one GPU."""

    def __init__(self, ctx: HipContext, width: int, height: int, max_lights: int, band: Band | None = None, prepared: "PreparedLights | None" = None):
        """prepared: the prepared views of the light buffer this object will be used with -- cull() and shade() then go through the prepared entry
        points unless told otherwise per call"""
        self.ctx, self.W, self.H, self.max_lights = ctx, width, height, max_lights
        self.prepared = prepared
        self.Tx, self.Ty = host.num_tiles(width, height)
        self.band = band if band is not None else host.band_whole_frame(width, height)
        self.band_tiles = (self.band.tileRowEnd - self.band.tileRowBegin) * self.Tx
        lib = ctx._lib
        ws = lib.sailor_hip_light_cull_workspace_size(width, height, max_lights, C.byref(self.band))
        if ws == 0:
            raise _lib.SailorHipError(-1, "light_cull_workspace_size")
        dev = ctx.device
        self.workspace = torch.empty(ws, dtype=torch.uint8, device=dev)
        # LightCullingNode.cpp:64-65 (+1: the reference's buffer is one uint short when every tile is full)
        self.grid = torch.zeros(max(self.band_tiles, 1) * 2, dtype=torch.int32, device=dev)
        self.culled = torch.zeros(1 + max(self.band_tiles, 1) * LIGHTS_PER_TILE, dtype=torch.int32, device=dev)
        self.radiance = None
        # the cull's shading-order hint (long lists first) lives in the workspace; None = raster order
        self.tile_order = lib.sailor_hip_light_cull_tile_order(width, height, max_lights, C.byref(self.band), _ptr(self.workspace))
        self.use_tile_order = os.environ.get("SAILOR_NO_TILE_ORDER") is None
        self._culled_once = False
        # the cull's own per-tile form of the lists (tileNum[t] entries at tileLists[128 t ..]): what the shade reads by default -- the canonical
        # grid / culledLights are then only needed by other consumers, and k1_pack can run beside the shade (cull(..., defer_pack=True) + pack())
        a, b = C.c_void_p(), C.c_void_p()
        _lib.check(lib.sailor_hip_light_cull_tile_lists(width, height, max_lights, C.byref(self.band), _ptr(self.workspace), C.byref(a), C.byref(b)),
                   "sailor_hip_light_cull_tile_lists")
        self.tile_num, self.tile_lists = a.value, b.value
        self.shade_from_tile_lists = os.environ.get("SAILOR_SHADE_CANONICAL_LISTS") is None

    # -- K0 + K1 --------------------------------------------------------------------------------------------------
    def cull(self, frame: UboFrameData, lights: torch.Tensor, lights_num: int, depth: torch.Tensor, flags: int = _lib.CULL_DEFAULT,
             ctx: "HipContext | None" = None, prepared: "PreparedLights | None" = None, defer_pack: bool = False, prepare_lights: bool = False):
        """lights: uint8/any tensor holding lights_num 112-byte records; depth: float32 [band rows, W].
        ctx: record on another context's stream (frames in flight: next frame's cull beside this frame's shade).
        prepared: the lights' prepared views (the kernel then streams 20 bytes per light instead of the 112-byte records)."""
        assert depth.dtype == torch.float32 and depth.is_contiguous() and depth.shape == (self.band.fbRowCount, self.W), depth.shape
        assert lights_num <= self.max_lights
        prepared = prepared if prepared is not None else self.prepared
        flags |= _ENV_CULL_FLAGS   # (diagnostics)
        if defer_pack:             # the canonical buffers are written by pack(), wherever the caller records it
            flags |= _lib.CULL_DEFER_PACK
        if prepare_lights:         # every light dirty: the prepared views of all lights_num lights are (re)derived inside this cull
            assert prepared is not None
            flags |= _lib.CULL_PREPARE_LIGHTS
        pc = host.push_constants(frame, self.W, self.H, lights_num)
        ctx = ctx or self.ctx
        lib = ctx._lib
        if prepared is not None:
            assert lights_num <= prepared.capacity
            _lib.check(lib.sailor_hip_light_cull_prepared(ctx.handle, C.byref(frame), C.byref(pc), _ptr(lights), _ptr(depth), _ptr(self.grid), _ptr(self.culled),
                                                          self.culled.numel(), _ptr(self.workspace), self.workspace.numel(), C.byref(self.band), flags,
                                                          _ptr(prepared.buffer), prepared.capacity),
                       "sailor_hip_light_cull_prepared", ctx.handle)
        else:
            _lib.check(lib.sailor_hip_light_cull(ctx.handle, C.byref(frame), C.byref(pc), _ptr(lights), _ptr(depth), _ptr(self.grid), _ptr(self.culled),
                                                 self.culled.numel(), _ptr(self.workspace), self.workspace.numel(), C.byref(self.band), flags),
                       "sailor_hip_light_cull", ctx.handle)
        self._culled_once = True
        return self.grid, self.culled

    def pack(self, ctx: "HipContext | None" = None):
        """the second half of a cull(..., defer_pack=True): lightsGrid / culledLights from the per-tile lists, on ctx's stream (default: the cull's)"""
        ctx = ctx or self.ctx
        _lib.check(ctx._lib.sailor_hip_light_cull_pack(ctx.handle, self.W, self.H, self.max_lights, C.byref(self.band), _ptr(self.workspace), _ptr(self.grid),
                                                       _ptr(self.culled), self.culled.numel()), "sailor_hip_light_cull_pack", ctx.handle)

    # -- K2 + K3 --------------------------------------------------------------------------------------------------
    def shade(self, frame: UboFrameData, surface: torch.Tensor, lights: torch.Tensor, lights_num: int, csm: CsmDesc | None = None,
              out: torch.Tensor | None = None, ibl: "_lib.IblDesc | None" = None, prepared: "PreparedLights | None" = None,
              ctx: "HipContext | None" = None) -> torch.Tensor:
        """surface: float32 [3, band rows, W, 4]; returns radiance float32 [band rows, W, 4].  ibl: ambient term (its `ao`
        pointer, if any, holds the band's rows).  ctx: record on another context's stream."""
        rows = self.band.fbRowCount
        own_ctx = self.ctx
        if ctx is not None:
            self.ctx = ctx
        try:
            return self._shade(frame, surface, lights, lights_num, csm, out, ibl, prepared, rows)
        finally:
            self.ctx = own_ctx

    def _shade(self, frame, surface, lights, lights_num, csm, out, ibl, prepared, rows):
        assert surface.dtype == torch.float32 and surface.is_contiguous() and surface.shape == (3, rows, self.W, 4), surface.shape
        if out is None:
            if self.radiance is None:
                self.radiance = torch.empty((rows, self.W, 4), dtype=torch.float32, device=self.ctx.device)
            out = self.radiance
        lib = self.ctx._lib
        prepared = prepared if prepared is not None else self.prepared
        order = self.tile_order if (self.use_tile_order and self._culled_once) else None  # only lists made by THIS object's cull have an order
        if self.shade_from_tile_lists and self._culled_once:
            # the lists where the cull left them (the same entries in the same order as grid / culledLights: the same radiance bit for bit)
            if prepared is not None:
                assert lights_num <= prepared.capacity
            _lib.check(lib.sailor_hip_shade_tile_lists(self.ctx.handle, C.byref(frame), _ptr(surface), rows * self.W, _ptr(lights), lights_num, self.tile_num,
                                                       self.tile_lists, C.byref(csm) if csm is not None else None, C.byref(ibl) if ibl is not None else None,
                                                       _ptr(out), C.byref(self.band), order, _ptr(prepared.buffer) if prepared is not None else None,
                                                       prepared.capacity if prepared is not None else 0),
                       "sailor_hip_shade_tile_lists", self.ctx.handle)
            return out
        if prepared is not None:
            assert lights_num <= prepared.capacity
            _lib.check(lib.sailor_hip_shade_prepared(self.ctx.handle, C.byref(frame), _ptr(surface), rows * self.W, _ptr(lights), lights_num, _ptr(self.grid),
                                                     _ptr(self.culled), C.byref(csm) if csm is not None else None, C.byref(ibl) if ibl is not None else None,
                                                     _ptr(out), C.byref(self.band), order, _ptr(prepared.buffer), prepared.capacity),
                       "sailor_hip_shade_prepared", self.ctx.handle)
            return out
        if ibl is not None or order is not None:
            _lib.check(lib.sailor_hip_shade_ex(self.ctx.handle, C.byref(frame), _ptr(surface), rows * self.W, _ptr(lights), lights_num, _ptr(self.grid),
                                               _ptr(self.culled), C.byref(csm) if csm is not None else None, C.byref(ibl) if ibl is not None else None,
                                               _ptr(out), C.byref(self.band), order),
                       "sailor_hip_shade_ex", self.ctx.handle)
            return out
        _lib.check(lib.sailor_hip_shade(self.ctx.handle, C.byref(frame), _ptr(surface), rows * self.W, _ptr(lights), lights_num, _ptr(self.grid),
                                        _ptr(self.culled), C.byref(csm) if csm is not None else None, _ptr(out), C.byref(self.band)),
                   "sailor_hip_shade", self.ctx.handle)
        return out

    # -- helpers ----------------------------------------------------------------------------------------------------
    def cull_diagnostics(self, lights_num: int) -> dict:
        out = (C.c_uint64 * 8)()
        _lib.check(self.ctx._lib.sailor_hip_light_cull_diagnostics(self.ctx.handle, self.W, self.H, lights_num, C.byref(self.band),
                                                                   _ptr(self.workspace), out), "light_cull_diagnostics", self.ctx.handle)
        keys = ["bands", "mask_bits", "groups", "group_list_sum", "groups_overflowed", "group_list_max", "words_per_band", "column_mask_bits"]
        return dict(zip(keys, [int(v) for v in out]))

    def band_selection(self, lights_num: int):
        """(M, lightMap uint32[M]) of the last cull with lights_num lights, if that cull ran the band selection (the caller knows: HipContext.launches_of)"""
        a, b = C.c_void_p(), C.c_void_p()
        _lib.check(self.ctx._lib.sailor_hip_light_cull_band_selection(self.W, self.H, lights_num, C.byref(self.band), _ptr(self.workspace), C.byref(a), C.byref(b)),
                   "sailor_hip_light_cull_band_selection")
        self.ctx.synchronize()
        base = self.workspace.data_ptr()
        m = int(self.workspace[a.value - base: a.value - base + 4].view(torch.int32).cpu().numpy().view(np.uint32)[0])
        lm = self.workspace[b.value - base: b.value - base + 4 * m].view(torch.int32).cpu().numpy().view(np.uint32).copy()
        return m, lm

    def lists_to_host(self):
        """(grid uint32[T,2], indices uint32[1 + total]) of this band, band-local offsets."""
        self.ctx.synchronize()
        g = self.grid.cpu().numpy().view(np.uint32).reshape(-1, 2)[: self.band_tiles]
        c = self.culled.cpu().numpy().view(np.uint32)
        return g.copy(), c[: 1 + int(c[0])].copy()


def upload_lights(lights: np.ndarray, device) -> torch.Tensor:
    assert lights.dtype.itemsize == 112
    raw = np.ascontiguousarray(lights).view(np.uint8).reshape(-1)
    if raw.size == 0:
        raw = np.zeros(112, np.uint8)
    return torch.from_numpy(raw.copy()).to(device)


def upload_shadow_maps(shadows, device) -> tuple[CsmDesc, list]:
    """-> (CsmDesc with device pointers, tensors to keep alive)."""
    keep, maps = [], []
    for m in shadows.maps:
        t = torch.from_numpy(np.ascontiguousarray(m)).to(device)
        keep.append(t)
        fmt = _lib.SHADOWMAP_R16F if m.dtype == np.float16 else (_lib.SHADOWMAP_RGBA32F if m.ndim == 3 else _lib.SHADOWMAP_R32F)
        maps.append((t.data_ptr(), m.shape[1], m.shape[0], fmt))
    return host.make_csm_desc(shadows.lights_matrices, maps), keep


def upload_ibl(ibl_set, device, ao_rows: tuple[int, int] | None = None) -> tuple["_lib.IblDesc", list]:
    """synth.IblSet -> (IblDesc with device pointers, tensors to keep alive); ao_rows = framebuffer rows of the band."""
    irr = torch.from_numpy(np.ascontiguousarray(ibl_set.irradiance)).to(device)
    env = torch.from_numpy(np.ascontiguousarray(ibl_set.env_chain)).to(device)
    lut = torch.from_numpy(np.ascontiguousarray(ibl_set.brdf_lut)).to(device)
    keep = [irr, env, lut]
    d = _lib.IblDesc()
    d.irradiance, d.irrSize = irr.data_ptr(), ibl_set.irradiance.shape[1]
    d.env, d.envSize, d.envLevels = env.data_ptr(), ibl_set.env_size, ibl_set.env_levels
    d.brdfLut, d.lutW, d.lutH = lut.data_ptr(), ibl_set.brdf_lut.shape[1], ibl_set.brdf_lut.shape[0]
    if ibl_set.ao is not None:
        a = ibl_set.ao if ao_rows is None else ibl_set.ao[ao_rows[0]:ao_rows[1]]
        ao = torch.from_numpy(np.ascontiguousarray(a)).to(device)
        keep.append(ao)
        d.ao = ao.data_ptr()
    return d, keep


def evsm_blur(ctx: "HipContext", moments: torch.Tensor, radius_umbra: int, radius_penumbra: int, temp: torch.Tensor | None = None) -> torch.Tensor:
    """ShadowPrepassNode's blur of the cascade-0 EVSM map, in place; moments float32 [H, W, 4]."""
    assert moments.dtype == torch.float32 and moments.dim() == 3 and moments.shape[2] == 4 and moments.is_contiguous()
    if temp is None:
        temp = torch.empty_like(moments)
    _lib.check(ctx._lib.sailor_hip_evsm_blur(ctx.handle, _ptr(moments), _ptr(temp), moments.shape[1], moments.shape[0], radius_umbra, radius_penumbra),
               "sailor_hip_evsm_blur", ctx.handle)
    return moments


def compute_brdf_lut(ctx: "HipContext", width: int, height: int) -> torch.Tensor:
    """ComputeBrdfLut.shader on the GPU -> float32 [height, width, 2]"""
    out = torch.empty((height, width, 2), dtype=torch.float32, device=ctx.device)
    _lib.check(ctx._lib.sailor_hip_compute_brdf_lut(ctx.handle, _ptr(out), width, height), "sailor_hip_compute_brdf_lut", ctx.handle)
    return out


def compute_irradiance_map(ctx: "HipContext", env_chain: torch.Tensor, env_size: int, env_levels: int, size: int) -> torch.Tensor:
    """ComputeIrradianceMap.shader on the GPU: flat RGBA32F cube mip chain -> float32 [6, size, size, 4]"""
    out = torch.empty((6, size, size, 4), dtype=torch.float32, device=ctx.device)
    _lib.check(ctx._lib.sailor_hip_compute_irradiance_map(ctx.handle, _ptr(env_chain), env_size, env_levels, _ptr(out), size),
               "sailor_hip_compute_irradiance_map", ctx.handle)
    return out


def prefilter_env_map(ctx: "HipContext", raw_chain: torch.Tensor, size: int, levels: int) -> torch.Tensor:
    """EnvironmentNode's specular pre-filter (ComputeEnvMap_IBL.shader per mip) on the GPU: flat chain -> flat chain"""
    out = torch.empty_like(raw_chain)
    _lib.check(ctx._lib.sailor_hip_prefilter_env_map(ctx.handle, _ptr(raw_chain), _ptr(out), size, levels), "sailor_hip_prefilter_env_map", ctx.handle)
    return out


def raw_env_cubemap(ctx: "HipContext", equirect: torch.Tensor, size: int, levels: int, repeat: bool = True, cover=None) -> torch.Tensor:
    """EnvironmentNode.cpp:116-140: ConvertEquirect2Cubemap + GenerateMipMaps -> the flat RGBA32F chain of `rawEnvCubemap`.
    `equirect` is float32 [H, W, 4]; `cover` = (w, h) mirrors the reference's equirectExtent / 32 dispatch (default: the whole cube)."""
    assert equirect.dtype == torch.float32 and equirect.dim() == 3 and equirect.shape[2] == 4 and equirect.is_contiguous()
    total = sum(6 * max(size >> l, 1) ** 2 * 4 for l in range(levels))
    chain = torch.zeros(total, dtype=torch.float32, device=ctx.device)
    cw, ch = (size, size) if cover is None else cover
    _lib.check(ctx._lib.sailor_hip_equirect_to_cube(ctx.handle, _ptr(equirect), equirect.shape[1], equirect.shape[0], 1 if repeat else 0,
                                                    _ptr(chain), size, cw, ch), "sailor_hip_equirect_to_cube", ctx.handle)
    _lib.check(ctx._lib.sailor_hip_generate_mipmaps_cube(ctx.handle, _ptr(chain), size, levels), "sailor_hip_generate_mipmaps_cube", ctx.handle)
    return chain


def ecs_range_for_rank(n: int, rank: int, world: int):
    """(begin, end, words per rank) of rank's slice of an equal split of n entities in whole visibility words (sailor_hip_ecs_range_for_rank; pure host
    arithmetic, no device)"""
    b, e, per = C.c_uint32(), C.c_uint32(), C.c_uint32()
    _lib.check(_lib.load().sailor_hip_ecs_range_for_rank(n, rank, world, C.byref(b), C.byref(e), C.byref(per)), "sailor_hip_ecs_range_for_rank")
    return int(b.value), int(e.value), int(per.value)


class EcsSweep:
    """K4 on one GPU over level-sorted entities.  rank / world: this GPU sweeps its slice of an equal split only (sailor_hip_ecs_sweep_range); the
    visibility buffer then has room for every rank's words and exchange_visibility() completes it."""

    def __init__(self, ctx: HipContext, entities, rank: int = 0, world: int = 1):
        self.ctx = ctx
        dev = ctx.device
        self.n = len(entities.parent)
        self.rank, self.world_size = rank, world
        self.begin, self.end, self.words_per_rank = ecs_range_for_rank(self.n, rank, world)
        self.trs = torch.from_numpy(entities.transforms).to(dev)
        self.parent = torch.from_numpy(entities.parent.view(np.int32)).to(dev)
        self.local_aabb = torch.from_numpy(entities.local_aabb).to(dev)
        self.level_offsets = np.ascontiguousarray(entities.level_offsets, np.uint32)
        self.world = torch.empty((self.n, 16), dtype=torch.float32, device=dev)
        self.world_aabb = torch.empty((self.n, 6), dtype=torch.float32, device=dev)
        self.visibility = torch.zeros(max((self.n + 63) // 64, world * self.words_per_rank), dtype=torch.int64, device=dev)

    def run(self, planes: np.ndarray):
        planes = np.ascontiguousarray(planes, np.float32).reshape(24)
        lib = self.ctx._lib
        if self.world_size == 1:
            _lib.check(lib.sailor_hip_ecs_sweep(self.ctx.handle, self.n, _ptr(self.trs), _ptr(self.parent),
                                                self.level_offsets.ctypes.data_as(C.POINTER(C.c_uint32)), len(self.level_offsets) - 1,
                                                _ptr(self.local_aabb), planes.ctypes.data_as(C.POINTER(C.c_float)),
                                                _ptr(self.world), _ptr(self.world_aabb), _ptr(self.visibility)),
                       "sailor_hip_ecs_sweep", self.ctx.handle)
        else:
            _lib.check(lib.sailor_hip_ecs_sweep_range(self.ctx.handle, self.n, _ptr(self.trs), _ptr(self.parent),
                                                      self.level_offsets.ctypes.data_as(C.POINTER(C.c_uint32)), len(self.level_offsets) - 1,
                                                      _ptr(self.local_aabb), planes.ctypes.data_as(C.POINTER(C.c_float)),
                                                      _ptr(self.world), _ptr(self.world_aabb), _ptr(self.visibility), self.begin, self.end),
                       "sailor_hip_ecs_sweep_range", self.ctx.handle)
        return self.world, self.world_aabb, self.visibility

    def exchange_visibility(self, comm=None, group=None):
        """every rank's visibility words -> the whole bitmask on every rank: sailor_hip_exchange_visibility on an ncclComm_t (dist.RcclComm), or the same
        all-gather over torch.distributed when there is none (gloo tests, ranks sharing a GPU)"""
        if self.world_size == 1:
            return self.visibility
        if comm is not None:
            _lib.check(self.ctx._lib.sailor_hip_exchange_visibility(self.ctx.handle, comm.handle, self.rank, self.world_size, self.n, _ptr(self.visibility),
                                                                        self.visibility.numel()),
                       "sailor_hip_exchange_visibility", self.ctx.handle)
        else:
            from . import dist as sdist
            sdist.allgather_visibility(self.visibility, self.rank, self.world_size, self.words_per_rank, group)
        return self.visibility


def raster_depth(ctx: "HipContext", light_matrix, positions: torch.Tensor, indices: torch.Tensor, models: torch.Tensor, width: int, height: int,
                 instance_ids: torch.Tensor | None = None, depth: torch.Tensor | None = None, coarse: torch.Tensor | None = None,
                 cull_back: bool = False) -> torch.Tensor:
    """sailor_hip_raster_depth: the caster draws of one shadow pass -> float32 [height, width] depth (reversed Z, 0 = nothing drawn).
    `depth` given = draw on top of it (a dependent pass); otherwise a cleared buffer is used.  `coarse`: int32 [sailor_hip_raster_coarse_words(w, h)] scratch that
    belongs to the depth buffer (hierarchical depth; same result, much less fill)."""
    lm = np.ascontiguousarray(light_matrix, np.float32).reshape(16)
    out = depth if depth is not None else torch.empty((height, width), dtype=torch.float32, device=ctx.device)
    n = models.shape[0] if instance_ids is None else instance_ids.numel()
    _lib.check(ctx._lib.sailor_hip_raster_depth(ctx.handle, lm.ctypes.data_as(C.POINTER(C.c_float)), _ptr(positions), _ptr(indices), indices.numel() // 3,
                                                _ptr(models), _ptr(instance_ids) if instance_ids is not None else None, n, width, height, _ptr(out),
                                                (0 if depth is not None else _lib.RASTER_CLEAR) | (_lib.RASTER_CULL_BACK if cull_back else 0), _ptr(coarse)),
               "sailor_hip_raster_depth", ctx.handle)
    return out


def raster_depth_camera(ctx: "HipContext", frame, positions: torch.Tensor, indices: torch.Tensor, models: torch.Tensor, width: int, height: int,
                        instance_ids: torch.Tensor | None = None, coarse: torch.Tensor | None = None, cull_back: bool = False) -> torch.Tensor:
    """sailor_hip_raster_depth_camera: the depth prepass -> raw reversed-Z depth float32 [height, width] (0 = nothing drawn)"""
    out = torch.empty((height, width), dtype=torch.float32, device=ctx.device)
    n = models.shape[0] if instance_ids is None else instance_ids.numel()
    _lib.check(ctx._lib.sailor_hip_raster_depth_camera(ctx.handle, C.byref(frame), _ptr(positions), _ptr(indices), indices.numel() // 3, _ptr(models),
                                                       _ptr(instance_ids) if instance_ids is not None else None, n, width, height, _ptr(out),
                                                       _lib.RASTER_CLEAR | (_lib.RASTER_CULL_BACK if cull_back else 0), _ptr(coarse)),
               "sailor_hip_raster_depth_camera", ctx.handle)
    return out


def shadow_resolve(ctx: "HipContext", depth: torch.Tensor, fmt: int) -> torch.Tensor:
    """ShadowCaster.shader's fragment stage on the winning depths: RGBA32F EVSM moments, R16F or R32F depth"""
    h, w = depth.shape
    if fmt == _lib.SHADOWMAP_RGBA32F:
        out = torch.empty((h, w, 4), dtype=torch.float32, device=ctx.device)
    elif fmt == _lib.SHADOWMAP_R16F:
        out = torch.empty((h, w), dtype=torch.float16, device=ctx.device)
    else:
        out = torch.empty((h, w), dtype=torch.float32, device=ctx.device)
    _lib.check(ctx._lib.sailor_hip_shadow_resolve(ctx.handle, _ptr(depth), w, h, fmt, _ptr(out)), "sailor_hip_shadow_resolve", ctx.handle)
    return out


def csm_caster_masks(ctx: "HipContext", world_aabb: torch.Tensor, cascade_planes: np.ndarray) -> torch.Tensor:
    """sailor_hip_csm_caster_masks: world AABBs [n, 6] (the ECS sweep's output) x cascade frusta [k, 6, 4] -> int64 [k, ceil(n / 64)] bitmasks"""
    pl = np.ascontiguousarray(cascade_planes, np.float32).reshape(-1, 24)
    n = world_aabb.shape[0]
    out = torch.empty((len(pl), (n + 63) // 64), dtype=torch.int64, device=ctx.device)  # every word is written
    _lib.check(ctx._lib.sailor_hip_csm_caster_masks(ctx.handle, n, _ptr(world_aabb), pl.ctypes.data_as(C.POINTER(C.c_float)), len(pl), _ptr(out)),
               "sailor_hip_csm_caster_masks", ctx.handle)
    return out


def hiz_build(ctx: "HipContext", depth: torch.Tensor, width: int, height: int, levels: int) -> torch.Tensor:
    """DepthHighZNode's loop on the GPU: raw depth [H, W] float32 -> flat level-major min pyramid"""
    total = sum(max(width >> l, 1) * max(height >> l, 1) for l in range(levels))
    out = torch.empty(total, dtype=torch.float32, device=ctx.device)
    _lib.check(ctx._lib.sailor_hip_hiz_build(ctx.handle, _ptr(depth), depth.shape[1], depth.shape[0], _ptr(out), width, height, levels),
               "sailor_hip_hiz_build", ctx.handle)
    return out


class MeshCull:
    """ComputeMeshCulling.shader main() (frustum flags + indirect-draw compaction) over resident instance / indirect buffers."""

    def __init__(self, ctx: HipContext, instances: np.ndarray, batches: np.ndarray):
        assert instances.dtype.itemsize == 96
        self.ctx = ctx
        self.n = len(instances)
        self.num_batches = len(batches)
        self.instances = torch.from_numpy(instances.view(np.uint8).reshape(-1).copy()).to(ctx.device)
        self.batches = torch.from_numpy(np.ascontiguousarray(batches, np.uint32).view(np.int32).copy()).to(ctx.device)
        self._dtype = instances.dtype
        self._ws_bytes = int(ctx._lib.sailor_hip_mesh_cull_workspace_bytes(self.n, self.num_batches))
        self.workspace = torch.empty(max(self._ws_bytes, 256), dtype=torch.uint8, device=ctx.device)

    def run(self, frame, num_instances=None, first_instance=0, hiz=None):
        """hiz = (pyramid tensor, width, height, levels) switches the shader's OCCLUSION_CULLING define on"""
        n = self.n - first_instance if num_instances is None else num_instances
        desc = None
        if hiz is not None:
            desc = _lib.HiZDesc(hiz[0].data_ptr(), hiz[1], hiz[2], hiz[3])
        _lib.check(self.ctx._lib.sailor_hip_mesh_cull_compact_ex(self.ctx.handle, C.byref(frame), _ptr(self.instances), n, first_instance,
                                                                  _ptr(self.batches), self.num_batches, _ptr(self.workspace), self._ws_bytes,
                                                                  C.byref(desc) if desc is not None else None),
                   "sailor_hip_mesh_cull_compact_ex", self.ctx.handle)
        return self.instances, self.batches

    def download(self):
        self.ctx.synchronize()
        # no ndarray.copy() of the record view: it would not carry the records' padding bytes
        return (self.instances.cpu().numpy().view(self._dtype),
                self.batches.cpu().numpy().view(np.uint32).reshape(-1, 5).copy())
