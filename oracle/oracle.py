"""ctypes front-end of oracle/liboracle.so (the C restatement in sailor_oracle.c).

TEST INFRASTRUCTURE ONLY -- see the header of sailor_oracle.c.  Importable from tests/, __graft_entry__.smoke() and
bench.py's cpu_baseline leg; never from the sailor_amd package.  PARITY UNPINNED (the reference has no golden vectors).
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess
from pathlib import Path

import numpy as np

_DIR = Path(__file__).resolve().parent
LIB_PATH = Path(os.environ.get("SAILOR_ORACLE_LIB", _DIR / "liboracle.so"))  # (the sanitizer run of the CPU suite points this at liboracle_asan.so)
TILE, CAND, KEEP = 16, 196, 128
_lib = None


def build(force: bool = False) -> Path:
    if force or not LIB_PATH.exists() or LIB_PATH.stat().st_mtime < (_DIR / "sailor_oracle.c").stat().st_mtime:
        subprocess.run(["make", "-C", str(_DIR), "liboracle.so"] + (["-B"] if force else []), check=True, capture_output=True)
    return LIB_PATH


def lib() -> C.CDLL:
    global _lib
    if _lib is None:
        if not LIB_PATH.exists():
            build()
        L = C.CDLL(str(LIB_PATH))
        L.oracle_const_cascade_level_glsl.restype = C.c_float
        L.oracle_const_cascade_level_cpp.restype = C.c_float
        L.oracle_const_poisson.restype = C.c_float
        L.oracle_canonical_expf.restype = C.c_float
        L.oracle_canonical_expf.argtypes = [C.c_float]
        L.oracle_directional_shadow.restype = C.c_float
        L.oracle_perspective_rh_reversed_z.argtypes = [C.c_float] * 4 + [C.c_void_p]
        L.oracle_extract_frustum_planes.argtypes = [C.c_void_p, C.c_float, C.c_float, C.c_float, C.c_float, C.c_void_p, C.c_void_p]
        L.oracle_csm_matrices.argtypes = [C.c_void_p, C.c_void_p, C.c_float, C.c_float, C.c_float, C.c_float, C.c_void_p, C.c_void_p]
        L.oracle_const_evsm_blur_weight.restype = C.c_float
        L.oracle_linearize_depth.argtypes = [C.c_float, C.c_void_p, C.c_size_t, C.c_void_p]
        _lib = L
    return _lib


def _p(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def _frame_bytes(frame) -> np.ndarray:
    """Accepts a ctypes UboFrameData (or raw 232 bytes) and returns a uint8[232] copy."""
    b = np.frombuffer(bytes(frame), np.uint8).copy()
    assert b.size == 232
    return b


def num_tiles(W: int, H: int):
    return (W - 1) // TILE + 1, (H - 1) // TILE + 1


class OracleCsm(C.Structure):
    _fields_ = [("lightsMatrices", (C.c_float * 16) * 4), ("maps", C.c_void_p * 4), ("width", C.c_int32 * 4),
                ("height", C.c_int32 * 4), ("format", C.c_int32 * 4)]


def make_csm(lights_matrices: np.ndarray, maps: list) -> tuple[OracleCsm, list]:
    """maps: 4 numpy arrays (float16[S,S] -> R16F, float32[S,S,4] -> RGBA32F, float32[S,S] -> R32F) or None."""
    d = OracleCsm()
    keep = []
    lm = np.ascontiguousarray(lights_matrices, np.float32).reshape(4, 16)
    for k in range(4):
        for i in range(16):
            d.lightsMatrices[k][i] = float(lm[k, i])
        m = maps[k]
        if m is None:
            d.maps[k] = None
            continue
        m = np.ascontiguousarray(m)
        keep.append(m)
        d.maps[k] = m.ctypes.data
        d.height[k], d.width[k] = m.shape[0], m.shape[1]
        d.format[k] = 0 if m.dtype == np.float16 else (1 if m.ndim == 3 else 2)
    return d, keep


class OracleIbl(C.Structure):
    _fields_ = [("irradiance", C.c_void_p), ("irrSize", C.c_int32), ("env", C.c_void_p), ("envSize", C.c_int32), ("envLevels", C.c_int32),
                ("brdfLut", C.c_void_p), ("lutW", C.c_int32), ("lutH", C.c_int32), ("ao", C.c_void_p)]


def make_ibl(irradiance: np.ndarray, env_chain: np.ndarray, env_size: int, env_levels: int, brdf_lut: np.ndarray, ao: np.ndarray | None):
    """irradiance float32[6,S,S,4]; env_chain float32 flat mip chain (level-major, each level [6,s,s,4]); brdf_lut float32[H,W,2];
    ao float32[H,W] or None.  -> (OracleIbl, keep-alive list)"""
    irr = np.ascontiguousarray(irradiance, np.float32); env = np.ascontiguousarray(env_chain, np.float32).reshape(-1)
    lut = np.ascontiguousarray(brdf_lut, np.float32)
    assert irr.ndim == 4 and irr.shape[0] == 6 and irr.shape[1] == irr.shape[2] and irr.shape[3] == 4 and lut.ndim == 3 and lut.shape[2] == 2
    assert env.size == sum(6 * max(1, env_size >> l) ** 2 * 4 for l in range(env_levels))
    d = OracleIbl()
    d.irradiance, d.irrSize = irr.ctypes.data, irr.shape[1]
    d.env, d.envSize, d.envLevels = env.ctypes.data, env_size, env_levels
    d.brdfLut, d.lutW, d.lutH = lut.ctypes.data, lut.shape[1], lut.shape[0]
    keep = [irr, env, lut]
    if ao is not None:
        a = np.ascontiguousarray(ao, np.float32)
        d.ao = a.ctypes.data
        keep.append(a)
    return d, keep


def cube_level_offsets(size0: int, levels: int):
    """float offsets of the mip levels of a cube in the level-major / face / row / texel RGBA32F layout (+ the total)"""
    offs, o = [], 0
    for l in range(levels):
        offs.append(o)
        sz = max(size0 >> l, 1)
        o += 6 * sz * sz * 4
    return offs, o


def compute_irradiance_map(env: np.ndarray, env_size: int, env_levels: int, size: int) -> np.ndarray:
    """ComputeIrradianceMap.shader: [6, size, size, 4] float32 from the cube `env` (flat RGBA32F mip chain)"""
    env = np.ascontiguousarray(env, np.float32).reshape(-1)
    assert env.size == cube_level_offsets(env_size, env_levels)[1]
    out = np.zeros((6, size, size, 4), np.float32)
    lib().oracle_compute_irradiance_map(_p(env), C.c_int(env_size), C.c_int(env_levels), _p(out), C.c_int(size))
    return out


def prefilter_env_map(raw: np.ndarray, size0: int, levels: int) -> np.ndarray:
    """EnvironmentNode.cpp:196-233: level 0 copied, level l = ComputeEnvMap_IBL.shader at roughness l / (levels - 1)"""
    raw = np.ascontiguousarray(raw, np.float32).reshape(-1)
    offs, total = cube_level_offsets(size0, levels)
    assert raw.size == total
    out = np.zeros(total, np.float32)
    out[:offs[1] if levels > 1 else total] = raw[:offs[1] if levels > 1 else total]
    delta = np.float32(1.0) / np.float32(max(levels - 1, 1))
    for level in range(1, levels):
        lib().oracle_prefilter_env_level(_p(raw), C.c_int(size0), C.c_int(levels), _p(out), C.c_int(level), C.c_float(np.float32(level) * delta))
    return out


def equirect_to_cube(equirect: np.ndarray, size: int, repeat: bool = True, cover=None, out: np.ndarray | None = None) -> np.ndarray:
    """ComputeEquirect2Cube.shader: [6, size, size, 4] float32 from the [H, W, 4] equirect image; `cover` = (w, h) written extent"""
    equirect = np.ascontiguousarray(equirect, np.float32)
    h, w = equirect.shape[:2]
    out = np.zeros((6, size, size, 4), np.float32) if out is None else out
    cw, ch = (size, size) if cover is None else cover
    lib().oracle_equirect_to_cube(_p(equirect), C.c_int(w), C.c_int(h), C.c_int(1 if repeat else 0), _p(out), C.c_int(size), C.c_int(cw), C.c_int(ch))
    return out


def generate_mipmaps_cube(level0: np.ndarray, size0: int, levels: int) -> np.ndarray:
    """VulkanCommandBuffer::GenerateMipMaps on a cube: the flat level-major RGBA32F chain whose level 0 is `level0`"""
    offs, total = cube_level_offsets(size0, levels)
    chain = np.zeros(total, np.float32)
    chain[:6 * size0 * size0 * 4] = np.ascontiguousarray(level0, np.float32).reshape(-1)
    lib().oracle_generate_mipmaps_cube(_p(chain), C.c_int(size0), C.c_int(levels))
    return chain


def compute_brdf_lut(w: int, h: int) -> np.ndarray:
    """ComputeBrdfLut.shader:26-71 -> float32[h, w, 2] (DFG1, DFG2)"""
    out = np.zeros((h, w, 2), np.float32)
    lib().oracle_compute_brdf_lut(w, h, _p(out))
    return out


def cube_sample_lod(cube_chain: np.ndarray, size0: int, levels: int, direction, lod: float) -> np.ndarray:
    c = np.ascontiguousarray(cube_chain, np.float32).reshape(-1); d = np.ascontiguousarray(direction, np.float32)
    out = np.zeros(4, np.float32)
    lib().oracle_cube_sample_lod(_p(c), size0, levels, _p(d), C.c_float(lod), _p(out))
    return out


def cube_face_st(direction):
    d = np.ascontiguousarray(direction, np.float32); face = C.c_int(0); st = np.zeros(2, np.float32)
    lib().oracle_cube_face_st(_p(d), C.byref(face), _p(st))
    return face.value, st


def host_threads() -> int:
    """the host threads the whole-frame checkers may use (the GPU box has 256; this container 8)"""
    import os
    try:
        return max(1, len(os.sched_getaffinity(0)))
    except AttributeError:
        return max(1, os.cpu_count() or 1)


def light_cull(frame, W: int, H: int, lights: np.ndarray, depth: np.ndarray, tile_rows=None, literal_select: bool = False, want_counts: bool = False,
               threads: int = 1):
    """-> (grid uint32[T,2], indices uint32[1+T*128], counts uint32[T] | None) for the band of tile rows.  threads > 1: the same per-tile code with the
    tile rows spread over host threads (oracle_light_cull_threads) -- the whole-frame checker of the full-size GPU tests."""
    Tx, Ty = num_tiles(W, H)
    r0, r1 = (0, Ty) if tile_rows is None else tile_rows
    T = (r1 - r0) * Tx
    fb = _frame_bytes(frame)
    lights = np.ascontiguousarray(lights)
    depth = np.ascontiguousarray(depth, np.float32)
    assert depth.shape == (H, W) and lights.dtype.itemsize == 112
    grid = np.zeros((max(T, 1), 2), np.uint32)
    indices = np.zeros(1 + max(T, 1) * KEEP, np.uint32)
    counts = np.zeros(max(T, 1), np.uint32) if want_counts else None
    if threads > 1:
        fn = lib().oracle_light_cull_threads
        fn.restype = C.c_uint32
        used = fn(_p(fb), W, H, len(lights), _p(lights), _p(depth), _p(grid), _p(indices), _p(counts), r0, r1, int(literal_select), C.c_uint32(threads))
        if used == 0:
            raise MemoryError("oracle_light_cull_threads")
    else:
        lib().oracle_light_cull(_p(fb), W, H, len(lights), _p(lights), _p(depth), _p(grid), _p(indices), _p(counts), r0, r1, int(literal_select))
    return grid[:T], indices, (counts[:T] if want_counts else None)


def shade(frame, W: int, H: int, surface: np.ndarray, lights: np.ndarray, grid: np.ndarray, indices: np.ndarray, csm=None, rows=None, ibl=None,
          threads: int = 1) -> np.ndarray:
    """surface float32[3,H,W,4]; grid/indices in the global canonical layout -> radiance float32[H,W,4] (rows outside `rows` are 0).
    threads > 1: the same per-pixel code with the rows spread over host threads (oracle_shade_threads)."""
    fb = _frame_bytes(frame)
    surface = np.ascontiguousarray(surface, np.float32)
    assert surface.shape == (3, H, W, 4)
    out = np.zeros((H, W, 4), np.float32)
    r0, r1 = (0, H) if rows is None else rows
    lights = np.ascontiguousarray(lights)
    grid = np.ascontiguousarray(grid, np.uint32)
    indices = np.ascontiguousarray(indices, np.uint32)
    if threads > 1:
        fn = lib().oracle_shade_threads
        fn.restype = C.c_uint32
        fn(_p(fb), W, H, _p(surface), _p(lights), _p(grid), _p(indices), C.byref(csm) if csm is not None else None,
           C.byref(ibl) if ibl is not None else None, _p(out), r0, r1, C.c_uint32(threads))
    elif ibl is not None:
        lib().oracle_shade_ibl(_p(fb), W, H, _p(surface), _p(lights), _p(grid), _p(indices), C.byref(csm) if csm is not None else None,
                               C.byref(ibl), _p(out), r0, r1)
    else:
        lib().oracle_shade(_p(fb), W, H, _p(surface), _p(lights), _p(grid), _p(indices), C.byref(csm) if csm is not None else None, _p(out), r0, r1)
    return out


def ecs_sweep(trs: np.ndarray, parent: np.ndarray, local_aabb: np.ndarray, planes: np.ndarray, begin: int = 0, end: int | None = None,
              world=None, world_aabb=None, visibility=None):
    n = len(parent)
    end = n if end is None else end
    trs = np.ascontiguousarray(trs, np.float32); parent = np.ascontiguousarray(parent, np.uint32)
    local_aabb = np.ascontiguousarray(local_aabb, np.float32); planes = np.ascontiguousarray(planes, np.float32).reshape(24)
    world = np.zeros((n, 16), np.float32) if world is None else world
    world_aabb = np.zeros((n, 6), np.float32) if world_aabb is None else world_aabb
    visibility = np.zeros((n + 63) // 64, np.uint64) if visibility is None else visibility
    lib().oracle_ecs_sweep(C.c_uint32(begin), C.c_uint32(end), _p(trs), _p(parent), _p(local_aabb), _p(planes), _p(world), _p(world_aabb), _p(visibility))
    return world, world_aabb, visibility


def ecs_sweep_threads(trs: np.ndarray, parent: np.ndarray, local_aabb: np.ndarray, planes: np.ndarray, level_offsets: np.ndarray, num_threads: int,
                      world=None, world_aabb=None, visibility=None):
    """The sweep on `num_threads` host threads (1 024-entity chunks, level by level) -> (world, world_aabb, visibility, seconds inside the sweep)."""
    n = len(parent)
    trs = np.ascontiguousarray(trs, np.float32); parent = np.ascontiguousarray(parent, np.uint32)
    local_aabb = np.ascontiguousarray(local_aabb, np.float32); planes = np.ascontiguousarray(planes, np.float32).reshape(24)
    offs = np.ascontiguousarray(level_offsets, np.uint32)
    world = np.zeros((n, 16), np.float32) if world is None else world
    world_aabb = np.zeros((n, 6), np.float32) if world_aabb is None else world_aabb
    visibility = np.zeros((n + 63) // 64, np.uint64) if visibility is None else visibility
    fn = lib().oracle_ecs_sweep_threads
    fn.restype = C.c_double
    secs = fn(C.c_uint32(len(offs) - 1), _p(offs), _p(trs), _p(parent), _p(local_aabb), _p(planes), _p(world), _p(world_aabb), _p(visibility),
              C.c_uint32(num_threads))
    if secs < 0:
        raise RuntimeError("oracle_ecs_sweep_threads: could not start the worker threads")
    return world, world_aabb, visibility, float(secs)


def linearize_depth(z_near: float, raw: np.ndarray) -> np.ndarray:
    """LinearizeDepth.shader:61-73 (REVERSE_Z_INF_FAR_PLANE) over a raw reversed-Z depth image; same shape back."""
    raw = np.ascontiguousarray(raw, np.float32)
    out = np.empty_like(raw)
    lib().oracle_linearize_depth(C.c_float(z_near), _p(raw), C.c_size_t(raw.size), _p(out))
    return out


def evsm_blur(image: np.ndarray, radius_umbra: int, radius_penumbra: int) -> np.ndarray:
    """ShadowPrepassNode.cpp:283-356: horizontal then vertical GaussianBlur_Evsm pass over a float32[H, W, 4] moments map."""
    img = np.ascontiguousarray(image, np.float32)
    H, W = img.shape[:2]
    tmp = np.empty_like(img); out = np.empty_like(img)
    lib().oracle_evsm_blur_pass(_p(img), _p(tmp), W, H, radius_umbra, radius_penumbra, 0)
    lib().oracle_evsm_blur_pass(_p(tmp), _p(out), W, H, radius_umbra, radius_penumbra, 1)
    return out


def _copy_records(a: np.ndarray) -> np.ndarray:
    """byte-exact copy of a structured array (ndarray.copy() does not carry the padding bytes of the 96-byte records)."""
    a = np.ascontiguousarray(a)
    return a.view(np.uint8).reshape(-1).copy().view(a.dtype)


def mesh_frustum_cull(frame, instances: np.ndarray) -> np.ndarray:
    fb = _frame_bytes(frame)
    inst = _copy_records(instances)
    assert inst.dtype.itemsize == 96
    lib().oracle_mesh_frustum_cull(_p(fb), _p(inst), C.c_uint32(len(inst)))
    return inst


def hiz_level_offsets(width: int, height: int, levels: int):
    offs, o = [], 0
    for l in range(levels):
        offs.append(o)
        o += max(width >> l, 1) * max(height >> l, 1)
    return offs, o


def hiz_build(depth: np.ndarray, width: int, height: int, levels: int) -> np.ndarray:
    """DepthHighZNode's loop over ComputeDepthHighZ.shader: flat float32 pyramid (level-major) from a raw depth image [H, W]"""
    depth = np.ascontiguousarray(depth, np.float32)
    out = np.zeros(hiz_level_offsets(width, height, levels)[1], np.float32)
    lib().oracle_hiz_build(_p(depth), C.c_int(depth.shape[1]), C.c_int(depth.shape[0]), _p(out), C.c_int(width), C.c_int(height), C.c_int(levels))
    return out


def mesh_cull_occlusion(frame, instances: np.ndarray, pyramid: np.ndarray, width: int, height: int, levels: int) -> np.ndarray:
    """ComputeMeshCulling.shader step 2 with OCCLUSION_CULLING: isCulled = FrustumCulling || OcclusionCulling"""
    fb = _frame_bytes(frame)
    inst = _copy_records(instances)
    pyr = np.ascontiguousarray(pyramid, np.float32)
    lib().oracle_mesh_cull_occlusion(_p(fb), _p(inst), C.c_uint32(len(inst)), _p(pyr), C.c_int32(width), C.c_int32(height), C.c_int32(levels))
    return inst


def mesh_cull_compact(frame, instances: np.ndarray, num_instances: int, first_instance: int, batches: np.ndarray, hiz=None):
    """ComputeMeshCulling.shader main(): frustum flags over [first, first + num), then per-batch stable compaction.
    `batches` is uint32 [numBatches, 5] (indexCount, instanceCount, firstIndex, vertexOffset, firstInstance)."""
    fb = _frame_bytes(frame)
    inst = _copy_records(instances)
    bt = np.ascontiguousarray(batches, np.uint32).copy()
    assert inst.dtype.itemsize == 96 and bt.ndim == 2 and bt.shape[1] == 5
    if hiz is None:
        lib().oracle_mesh_cull_compact(_p(fb), _p(inst), C.c_uint32(num_instances), C.c_uint32(first_instance), _p(bt), C.c_uint32(len(bt)))
    else:  # hiz = (flat pyramid, width, height, levels): the shader's OCCLUSION_CULLING build
        pyr = np.ascontiguousarray(hiz[0], np.float32)
        lib().oracle_mesh_cull_compact_hiz(_p(fb), _p(inst), C.c_uint32(num_instances), C.c_uint32(first_instance), _p(bt), C.c_uint32(len(bt)),
                                           _p(pyr), C.c_int32(hiz[1]), C.c_int32(hiz[2]), C.c_int32(hiz[3]))
    return inst, bt


def extract_frustum_planes(world_matrix, aspect, fov_y, z_near, z_far):
    wm = np.ascontiguousarray(world_matrix, np.float32).reshape(16)
    planes = np.zeros(24, np.float32); corners = np.zeros(24, np.float32)
    lib().oracle_extract_frustum_planes(_p(wm), aspect, fov_y, z_near, z_far, _p(planes), _p(corners))
    return planes.reshape(6, 4), corners.reshape(8, 3)


def extract_frustum_planes_matrix(matrix):
    """Frustum::ExtractFrustumPlanes(projectionViewMatrix): (planes [6, 4], corners [8, 3])"""
    m = np.ascontiguousarray(matrix, np.float32).reshape(16)
    planes = np.zeros(24, np.float32); corners = np.zeros(24, np.float32)
    lib().oracle_extract_frustum_planes_matrix(_p(m), _p(planes), _p(corners))
    return planes.reshape(6, 4), corners.reshape(8, 3)


def csm_caster_masks(world_aabb: np.ndarray, planes: np.ndarray) -> np.ndarray:
    """per cascade, the entities whose world AABB overlaps its frustum: uint64 [numCascades, ceil(n / 64)]"""
    aabb = np.ascontiguousarray(world_aabb, np.float32).reshape(-1, 6)
    pl = np.ascontiguousarray(planes, np.float32).reshape(-1, 24)
    out = np.zeros((len(pl), (len(aabb) + 63) // 64), np.uint64)
    lib().oracle_csm_caster_masks(C.c_uint32(len(aabb)), _p(aabb), _p(pl), C.c_uint32(len(pl)), _p(out))
    return out


def csm_matrices(light_view, camera_world, aspect, fov_y, near, far):
    lv = np.ascontiguousarray(light_view, np.float32).reshape(16); cw = np.ascontiguousarray(camera_world, np.float32).reshape(16)
    out = np.zeros(64, np.float32)
    lib().oracle_csm_matrices(_p(lv), _p(cw), aspect, fov_y, near, far, _p(out), None)
    return out.reshape(4, 16)


def mat4_inverse(m):
    m = np.ascontiguousarray(m, np.float32).reshape(16); out = np.zeros(16, np.float32)
    lib().oracle_mat4_inverse(_p(m), _p(out))
    return out


def perspective_rh(fov_radians, aspect, z_near, z_far):
    out = np.zeros(16, np.float32)
    lib().oracle_perspective_rh_reversed_z(fov_radians, aspect, z_near, z_far, _p(out))
    return out


def transform_matrix(trs12):
    t = np.ascontiguousarray(trs12, np.float32).reshape(12); out = np.zeros(16, np.float32)
    lib().oracle_transform_matrix(_p(t), _p(out))
    return out


def raster_depth(light_matrix, positions, indices, models, width: int, height: int, instance_ids=None, depth=None, view=None, cull_back=False) -> np.ndarray:
    """The canonical depth rasteriser (ShadowPrepassNode's caster draws): float32 [H, W] depth, GREATER test, 0 = nothing drawn."""
    lm = np.ascontiguousarray(light_matrix, np.float32).reshape(16)
    pos = np.ascontiguousarray(positions, np.float32).reshape(-1, 3)
    idx = np.ascontiguousarray(indices, np.uint32).reshape(-1, 3)
    mdl = np.ascontiguousarray(models, np.float32).reshape(-1, 16)
    out = np.zeros((height, width), np.float32) if depth is None else np.ascontiguousarray(depth, np.float32).copy()
    ids = None if instance_ids is None else np.ascontiguousarray(instance_ids, np.uint32)
    n = len(mdl) if ids is None else len(ids)
    if view is None:
        lib().oracle_raster_depth(_p(lm), _p(pos), _p(idx), C.c_uint32(len(idx)), _p(mdl), _p(ids) if ids is not None else None, C.c_uint32(n),
                                  C.c_int32(width), C.c_int32(height), _p(out), C.c_int32(1 if cull_back else 0))
    else:  # the depth prepass: light_matrix is the camera projection, view its view matrix
        vm = np.ascontiguousarray(view, np.float32).reshape(16)
        lib().oracle_raster_depth_camera(_p(lm), _p(vm), _p(pos), _p(idx), C.c_uint32(len(idx)), _p(mdl), _p(ids) if ids is not None else None, C.c_uint32(n),
                                         C.c_int32(width), C.c_int32(height), _p(out), C.c_int32(1 if cull_back else 0))
    return out


def shadow_resolve_evsm(depth: np.ndarray) -> np.ndarray:
    d = np.ascontiguousarray(depth, np.float32)
    out = np.zeros(d.shape + (4,), np.float32)
    lib().oracle_shadow_resolve_evsm(_p(d), C.c_int32(d.shape[1]), C.c_int32(d.shape[0]), _p(out))
    return out


def lighting_tick_runs(dirty, active, mobility, frame_last_change, owner_frame_last_change, skip_list):
    """LightingECS::Tick's loop (Runtime/ECS/LightingECS.cpp:93-192) over per-slot flags, in plain Python: returns the copies it issues as
    (first record slot, [component indices whose records the copy carries]) in issue order; `dirty`, `frame_last_change` (lists) and
    `skip_list` (list of [first, count]) are updated in place as the reference updates them.  Literal, including what follows from the
    position of `continue` (:148-149) and of the flush (:182): an inactive slot neither ends nor flushes a run, and a run still open when the loop
    ends is never copied."""
    runs, batch = [], []
    should_write, start, skip_index, n = True, 0, 0, len(dirty)
    index = 0
    while index < n:
        if skip_index < len(skip_list) and index == skip_list[skip_index][0]:          # :95-103
            index += skip_list[skip_index][1]
            if index >= n:
                break
            skip_index += 1
        if mobility[index] == 0:                                                        # EMobilityType::Static, :108-146
            placed = False
            if skip_index > 0 and index == skip_list[skip_index - 1][0] + skip_list[skip_index - 1][1]:
                skip_list[skip_index - 1][1] += 1
                placed = True
            if not placed and skip_index > 0:
                for i in range(skip_index - 1, len(skip_list) - 1):
                    lo = skip_list[i][0] + skip_list[skip_index - 1][1]
                    hi = skip_list[i + 1][0]
                    if lo < index < hi:
                        skip_list.insert(i + 1, [index, 1])
                        placed = True
                        skip_index += 1
                        break
            if not placed:
                skip_list.append([index, 1])
                skip_index += 1
        if active[index]:                                                               # :148-149 (`continue` otherwise)
            if dirty[index] or frame_last_change[index] < owner_frame_last_change[index]:   # :152
                if should_write:
                    should_write, start = False, index
                batch.append(index)
                frame_last_change[index] = owner_frame_last_change[index]
                dirty[index] = False
            else:
                should_write = True
            if (should_write or index == n - 1) and batch:                              # :182-191
                runs.append((start, batch))
                batch = []
        index += 1
    return runs


OCTREE_ROOT_SIZE, OCTREE_MIN_SIZE = 16536 * 16, 4   # RHI/SceneView.h:91-92


def trace_scene_octree_boxes(world_aabb: np.ndarray, planes: np.ndarray, root_size: int = OCTREE_ROOT_SIZE, min_size: int = OCTREE_MIN_SIZE):
    """E4 as the reference runs it (Containers/Octree.h:239-274 over the integer-truncated boxes of ECS/StaticMeshRendererECS.cpp:81,96,132): the world
    boxes go into a TOctree restated literally, the frustum is traced through it.  -> (visible words, inserted words, int boxes [n, 6] = position + extents,
    stats {nodes, visited, not_inserted, smallest_node})"""
    world_aabb = np.ascontiguousarray(world_aabb, np.float32).reshape(-1, 6)
    planes = np.ascontiguousarray(planes, np.float32).reshape(24)
    n = len(world_aabb)
    words = (n + 63) // 64
    vis, ins = np.zeros(words, np.uint64), np.zeros(words, np.uint64)
    boxes = np.zeros((n, 6), np.int32)
    stats = np.zeros(4, np.uint64)
    rc = lib().oracle_trace_scene_octree_boxes(C.c_uint32(n), _p(world_aabb), _p(planes), C.c_uint32(root_size), C.c_uint32(min_size), _p(vis), _p(ins), _p(boxes), _p(stats))
    if rc != 0:
        raise MemoryError("oracle_trace_scene_octree_boxes")
    return vis, ins, boxes, {"nodes": int(stats[0]), "visited": int(stats[1]), "not_inserted": int(stats[2]), "smallest_node": int(stats[3])}
