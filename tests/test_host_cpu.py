"""CPU suite for the product's host side (no GPU): the native host math of libsailor_hip.so (sailor_host_*) against the
oracle's independent restatement of the same reference functions, the C-ABI's loadability / exported symbols, argument
validation, and the band partition."""
import ctypes as C
import re
from pathlib import Path

import numpy as np
import pytest

from oracle import oracle
from sailor_amd import _lib, host, synth

ROOT = Path(__file__).resolve().parents[1]


def bits(a):
    return np.ascontiguousarray(a, np.float32).view(np.uint32)


# ---------------------------------------------------------------------------------------------------------------
# the drop-in boundary: the library loads here (no GPU) and exports every symbol include/sailor_hip.h declares
# ---------------------------------------------------------------------------------------------------------------
def test_library_loads_and_exports_every_declared_symbol():
    header = (ROOT / "include" / "sailor_hip.h").read_text()
    declared = set(re.findall(r"SAILOR_HIP_API\s+[\w\s\*]+?\b(sailor_\w+)\s*\(", header))
    assert len(declared) >= 30
    lib = C.CDLL(str(_lib.LIB_PATH))
    missing = [n for n in sorted(declared) if not hasattr(lib, n)]
    assert not missing, f"declared but not exported: {missing}"
    assert declared == set(_lib.SIGNATURES), (declared ^ set(_lib.SIGNATURES))
    assert _lib.load().sailor_hip_version() >= 1


def test_binding_flag_constants_are_the_headers():
    """the ctypes binding's cull flags are the header's #defines (a flag added on one side only would silently be another flag)"""
    import re
    header = (ROOT / "include" / "sailor_hip.h").read_text()
    defines = {m.group(1): int(m.group(2)) for m in re.finditer(r"#define SAILOR_(CULL_[A-Z_]+) (\d+)u", header)}
    assert {"CULL_DEFAULT", "CULL_BRUTE_FORCE", "CULL_RAW_DEPTH", "CULL_INTERVAL_MASKS", "CULL_DEFER_PACK", "CULL_PREPARE_LIGHTS", "CULL_BAND_SELECT",
            "CULL_NO_BAND_SELECT", "CULL_PREPARE_SELECTED"} <= set(defines)
    for name, value in defines.items():
        assert getattr(_lib, name) == value, name
    bits = [v for v in defines.values() if v]
    assert len(set(bits)) == len(bits) and all(v & (v - 1) == 0 for v in bits), "one bit each"


def test_struct_sizes_match_the_reference_layouts():
    assert C.sizeof(_lib.UboFrameData) == 232          # RHI/Types.h:751-761
    assert C.sizeof(_lib.LightCullPushConstants) == 88  # FrameGraph/LightCullingNode.h:25-31
    assert C.sizeof(_lib.LightShaderData) == 112        # ECS/LightingECS.h:71-81
    assert _lib.LightShaderData.worldPosition.offset == 16 and _lib.LightShaderData.bounds.offset == 96
    assert _lib.UboFrameData.cameraPosition.offset == 192 and _lib.UboFrameData.viewportSize.offset == 208
    assert _lib.UboFrameData.cameraZNearZFar.offset == 216 and _lib.LightCullPushConstants.lightsNum.offset == 80


def test_device_entry_points_fail_loudly_without_a_gpu_or_with_bad_arguments():
    import torch
    lib = _lib.load()
    handle = C.c_void_p()
    if not torch.cuda.is_available():
        assert lib.sailor_hip_context_create(0, None, 0, C.byref(handle)) == -2  # SAILOR_HIP_ERR_NO_DEVICE, no CPU fallback
        with pytest.raises(_lib.SailorHipError):
            from sailor_amd.forward_plus import HipContext
            HipContext("cpu")
    assert lib.sailor_hip_context_create(0, None, 0, None) == -1
    assert lib.sailor_hip_light_cull(None, None, None, None, None, None, None, 0, None, 0, None, 0) == -1
    assert lib.sailor_hip_shade(None, None, None, 0, None, 0, None, None, None, None, None) == -1
    assert lib.sailor_hip_ecs_sweep(None, 0, None, None, None, 0, None, None, None, None, None) == -1
    assert b"invalid" in lib.sailor_hip_status_string(-1)


# ---------------------------------------------------------------------------------------------------------------
# bands
# ---------------------------------------------------------------------------------------------------------------
def test_num_tiles_matches_the_node():
    assert host.num_tiles(1920, 1080) == (120, 68)     # LightCullingNode.cpp:56-57: (1080 - 1) / 16 + 1
    assert host.num_tiles(3840, 2160) == (240, 135)
    assert host.num_tiles(7680, 4320) == (480, 270)
    assert host.num_tiles(17, 33) == (2, 3)


def test_band_partition_4k_over_8_gpus():
    rows = [(b.tileRowBegin, b.tileRowEnd) for b in (host.band_for_rank(3840, 2160, r, 8) for r in range(8))]
    assert [r[0] for r in rows] + [rows[-1][1]] == [0, 16, 33, 50, 67, 84, 101, 118, 135]
    covered = np.zeros(2160, int)
    for r in range(8):
        b = host.band_for_rank(3840, 2160, r, 8)
        covered[b.fbRowBegin:b.fbRowBegin + b.fbRowCount] += 1
        assert b.fbRowBegin == 2160 - 16 * b.tileRowEnd  # tile row t <-> framebuffer rows H-1-16t-15 .. H-1-16t
    assert (covered == 1).all()


@pytest.mark.parametrize("size,world", [((1920, 1080), 3), ((131, 77), 2), ((17, 33), 5), ((16, 16), 4)])
def test_bands_cover_every_framebuffer_row_exactly_once(size, world):
    w, h = size
    covered = np.zeros(h, int)
    tiles = 0
    for r in range(world):
        b = host.band_for_rank(w, h, r, world)
        covered[b.fbRowBegin:b.fbRowBegin + b.fbRowCount] += 1
        tiles += b.tileRowEnd - b.tileRowBegin
    assert (covered == 1).all() and tiles == host.num_tiles(w, h)[1]
    whole = host.band_whole_frame(w, h)
    assert (whole.tileRowBegin, whole.tileRowEnd, whole.fbRowBegin, whole.fbRowCount) == (0, host.num_tiles(w, h)[1], 0, h)


def test_workspace_size_is_reported_without_a_device():
    lib = _lib.load()
    band = host.band_whole_frame(3840, 2160)
    ws = lib.sailor_hip_light_cull_workspace_size(3840, 2160, 65536, C.byref(band))
    assert 16 * 65536 < ws < 64 << 20, ws   # SoA lights + masks + lists: tens of MB, not numTiles x numLights
    assert lib.sailor_hip_light_cull_workspace_size(0, 2160, 65536, C.byref(band)) == 0


# ---------------------------------------------------------------------------------------------------------------
# host math: product (C++, sailor_amd/csrc/host_math.cpp) vs oracle (C) -- two restatements of the same glm-based code
# ---------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("seed", range(5))
def test_inverse_perspective_and_frame_data(seed):
    rng = np.random.default_rng(seed)
    m = rng.normal(size=16).astype(np.float32)
    np.testing.assert_array_equal(bits(host.mat4_inverse(m)), bits(oracle.mat4_inverse(m)))
    fov, w, h, zn, zf = float(rng.uniform(30, 120)), int(rng.integers(64, 4000)), int(rng.integers(64, 2200)), 1.0, 20000.0
    trs = np.concatenate([rng.uniform(-500, 500, 3), [1.0], rng.normal(size=4), [1, 1, 1, 1]]).astype(np.float32)
    trs[4:8] /= np.linalg.norm(trs[4:8])
    world = host.transform_matrix(trs[0:4], trs[4:8], trs[8:12])
    np.testing.assert_array_equal(bits(world), bits(oracle.transform_matrix(trs)))
    frame = host.fill_frame_data(world, fov, zn, zf, w, h, 1.5, 0.016)
    aspect = float(np.float32(w) / np.float32(h))
    proj = oracle.perspective_rh(float(np.float32(fov) * np.float32(0.01745329251994329576923690768489)), aspect, zn, zf)
    np.testing.assert_array_equal(bits(np.frombuffer(bytes(frame.projection), np.float32)), bits(proj))
    np.testing.assert_array_equal(bits(np.frombuffer(bytes(frame.invProjection), np.float32)), bits(oracle.mat4_inverse(proj)))
    # ECS/CameraECS.cpp:19-20: view = origin (identity) * inverse(world) -- the product matters for the sign of zeros
    inv = oracle.mat4_inverse(world)
    ident = np.eye(4, dtype=np.float32).reshape(16)
    view = np.zeros(16, np.float32)
    oracle.lib().oracle_mat4_mul_mat4(ident.ctypes.data_as(C.c_void_p), inv.ctypes.data_as(C.c_void_p), view.ctypes.data_as(C.c_void_p))
    np.testing.assert_array_equal(bits(np.frombuffer(bytes(frame.view), np.float32)), bits(view))
    assert tuple(frame.viewportSize) == (w, h) and tuple(frame.cameraZNearZFar) == (zn, zf)
    np.testing.assert_array_equal(np.frombuffer(bytes(frame.cameraPosition), np.float32)[:3], world[12:15])


def test_reversed_z_projection_maps_near_to_one_and_far_to_zero():
    cam = synth.make_camera(1920, 1080)
    p = np.frombuffer(bytes(cam.frame.projection), np.float32).reshape(4, 4).T.astype(np.float64)
    for z, expect in ((-1.0, 1.0), (-20000.0, 0.0)):
        clip = p @ np.array([0, 0, z, 1.0])
        assert abs(clip[2] / clip[3] - expect) < 1e-6   # Math/Math.cpp:18-21 reversed Z


@pytest.mark.parametrize("seed", range(4))
def test_frustum_planes_and_csm_matrices(seed):
    rng = np.random.default_rng(100 + seed)
    q = rng.normal(size=4); q /= np.linalg.norm(q)
    world = host.transform_matrix(np.append(rng.uniform(-300, 300, 3), 1), q, [1, 1, 1, 1])
    aspect, fov, zn, zf = float(rng.uniform(1, 2.4)), float(rng.uniform(40, 110)), 1.0, 20000.0
    hp, hc = host.extract_frustum_planes(world, aspect, fov, zn, zf)
    op, oc = oracle.extract_frustum_planes(world, aspect, fov, zn, zf)
    np.testing.assert_array_equal(bits(hp), bits(op))
    np.testing.assert_array_equal(bits(hc), bits(oc))
    # planes are unit length and the camera's forward axis is inside
    np.testing.assert_allclose(np.linalg.norm(hp[:, :3], axis=1), 1.0, atol=1e-6)
    inside = world[12:15] - 100.0 * world[8:11]
    assert (hp[:, :3] @ inside + hp[:, 3] > 0).all()
    lq = rng.normal(size=4); lq /= np.linalg.norm(lq)
    light_view = host.mat4_inverse(host.transform_matrix([0, 0, 0, 0], lq, [1, 1, 1, 1]))
    np.testing.assert_array_equal(bits(host.csm_matrices(light_view, world, aspect, fov, zn, zf)),
                                  bits(oracle.csm_matrices(light_view, world, aspect, fov, zn, zf)))


def test_light_packing_layout_and_cutoff_cosines():
    c_in, c_out = host.cutoff_cosines(30.0, 45.0)   # ECS/LightingECS.h:26 defaults, stored as cosines (LightingECS.cpp:171)
    assert abs(float(c_in) - np.cos(np.radians(30.0))) < 1e-6 and abs(float(c_out) - np.cos(np.radians(45.0))) < 1e-6
    cam = synth.make_camera(320, 200)
    depth = synth.make_linear_depth(320, 200)
    lights = synth.make_lights(cam, depth, synth.LightSetConfig(count=64, spot_fraction=0.5))
    raw = lights.view(np.uint8).reshape(64, 112)
    assert (raw[:, 8:16] == 0).all() and (raw[:, 28:32] == 0).all() and (raw[:, 88:96] == 0).all()  # std430 padding stays zero
    np.testing.assert_array_equal(lights["attenuation"][0], np.array([1.0, 0.022, 0.0019], np.float32))  # LightingECS.h:24


def test_generator_is_deterministic_and_streams_are_independent():
    a = synth.uniforms(synth.STREAM_LIGHTS, 1000)
    b = synth.uniforms(synth.STREAM_LIGHTS, 1000)
    c = synth.uniforms(synth.STREAM_DEPTH, 1000)
    np.testing.assert_array_equal(a, b)
    assert (a != c).mean() > 0.99 and 0.45 < a.mean() < 0.55 and a.min() >= 0 and a.max() < 1
    np.testing.assert_array_equal(synth.uniforms(synth.STREAM_LIGHTS, 10, offset=500), a[500:510])
    f1, f2 = synth.make_frame("tiny"), synth.make_frame("tiny")
    assert f1.lights.tobytes() == f2.lights.tobytes() and (f1.depth == f2.depth).all() and (f1.surface == f2.surface).all()
    # surface rows can be generated band by band
    np.testing.assert_array_equal(synth.make_surface(f1.cam, f1.depth, row_begin=32, row_end=64), f1.surface[:, 32:64])


def test_entities_are_level_sorted_with_the_editor_world_first():
    e = synth.make_entities(1024)
    assert e.level_offsets[0] == 0 and e.level_offsets[-1] == 1024
    roots = e.parent == 0xFFFFFFFF
    assert 0.65 < roots.mean() < 0.75
    child = np.nonzero(~roots)[0]
    assert (e.parent[child] < child).all()
    np.testing.assert_array_equal(e.transforms[0, :3], [0, 150, 0])   # Content/Editor.world:5-9 camera


# ---------------------------------------------------------------------------------------------------------------
# the C++ host mirror (sailor_amd/runtime): loadable without a GPU, nodes registered under the reference's names
# ---------------------------------------------------------------------------------------------------------------
def test_cpp_runtime_registers_the_reference_node_names_and_refuses_to_run_without_a_gpu():
    import torch
    from sailor_amd import runtime_binding
    rt = runtime_binding.load()
    assert rt.sailor_rt_node_registered(b"LightCulling") == 1   # FrameGraph/LightCullingNode.cpp:17
    assert rt.sailor_rt_node_registered(b"RenderScene") == 1    # FrameGraph/RenderSceneNode.cpp:19
    assert rt.sailor_rt_node_registered(b"LinearizeDepth") == 1  # FrameGraph/LinearizeDepthNode.cpp:19
    assert rt.sailor_rt_node_registered(b"Environment") == 1     # FrameGraph/EnvironmentNode.cpp:19
    assert rt.sailor_rt_node_registered(b"DepthHighZ") == 1      # FrameGraph/DepthHighZNode.cpp:18
    assert rt.sailor_rt_node_registered(b"Bloom") == 0          # out of scope
    if not torch.cuda.is_available():
        with pytest.raises(_lib.SailorHipError):
            runtime_binding.Runtime(0, 0)


def test_cost_balanced_tile_rows():
    """sailor_amd/dist.py:balanced_tile_rows -- contiguous, covering, monotone, and balanced to within one row's cost."""
    from sailor_amd import dist as sdist
    rng = np.random.default_rng(3)
    rows = np.concatenate([rng.integers(500, 2000, 30), rng.integers(8000, 20000, 60), rng.integers(500, 3000, 45)])
    for world in (1, 2, 3, 8):
        b = sdist.balanced_tile_rows(rows, 240, world)
        assert b[0] == 0 and b[-1] == len(rows) and len(b) == world + 1 and all(x <= y for x, y in zip(b, b[1:]))
        cost = 240 * sdist.TILE_COST + rows * sdist.ENTRY_COST
        per = [cost[b[i]:b[i + 1]].sum() for i in range(world)]
        assert max(per) - min(per) <= 2 * cost.max()
    assert sdist.balanced_tile_rows(np.zeros(135), 240, 8) == [0, 17, 34, 51, 67, 84, 101, 118, 135]  # uniform cost -> equal rows
    b = sdist.balanced_tile_rows(np.ones(3), 2, 8)                                                  # more ranks than rows: empty bands allowed
    assert b[0] == 0 and b[-1] == 3 and all(x <= y for x, y in zip(b, b[1:]))
    band = host.band_from_tile_rows(3840, 2160, 17, 34)
    assert (band.tileRowBegin, band.tileRowEnd, band.fbRowBegin, band.fbRowCount) == (17, 34, 2160 - 16 * 34, 16 * 17)
    with pytest.raises(_lib.SailorHipError):
        host.band_from_tile_rows(3840, 2160, 100, 136)


# ---------------------------------------------------------------------------------------------------------------
# the `.renderer` frame-graph description (FrameGraphAsset::Deserialize mirror)
# ---------------------------------------------------------------------------------------------------------------
RENDERER_TEXT = """---
samplers:
- name: g_noiseSampler
  fileId: ''
  path: Textures/Noise.png

float:
- PI: 3.1415926

############################
renderTargets:
- name: LinearDepth
  format: R32_SFLOAT
  filtration: Nearest
  width: ViewportWidth
  height: ViewportHeight
  bIsSurface: false

- name: DepthHighZ   # the pyramid of the occlusion test
  format: R32_SFLOAT
  width: ViewportWidth/2
  height: ViewportWidth/2
  bIsCompatibleWithComputeShaders: true
  bGenerateMips: true
  reduction: Min

- name: Main
  format: R16G16B16A16_SFLOAT
  width: 320
  height: 200
  bGenerateMips: true
  maxMipLevel: 4

############################
frame:
############################
- name: Clear
  float:
  - clearDepth: 0
  vec4:
  - clearColor: [0.1, 0.2, 0.3, 1]
  renderTargets:
  - target: DepthBuffer

- name: LinearizeDepth
  renderTargets:
  - depthStencil: DepthBuffer
  - target: LinearDepth

############################
- name: LightCulling
############################
  renderTargets:
  - depthStencil: LinearDepth

- name: Environment

- name: RenderScene
  string:
  - Tag: Opaque
  - GPUCulling: true
  renderTargets:
  - color: Main
  - depthStencil: DepthBuffer
  - depthHighZ: DepthHighZ

- name: Bloom
  tag: PostFx
  string:
  - defines: ~
  float:
  - data.threshold: 1.5
"""


def test_renderer_description_parses_like_the_reference_importer():
    """FrameGraphParser.h:64-213 / FrameGraphParser.cpp:23-78 restated: sections, `ViewportWidth/2`, mip counts (min(maxMipLevel,
    floor(log2(max extent)) + 1)), node dictionaries, `~`, comments."""
    from sailor_amd import runtime_binding
    n, summary = runtime_binding.parse_renderer(RENDERER_TEXT, 1280, 720)
    assert n == 6
    parts = dict(p.split("=", 1) for p in summary.split(";") if "=" in p and p.split("=", 1)[0] in ("targets", "values", "samplers"))
    assert parts["targets"] == "LinearDepth:1280x720:R32_SFLOAT:1,DepthHighZ:640x640:R32_SFLOAT:10:Min,Main:320x200:R16G16B16A16_SFLOAT:4"
    assert parts["values"].startswith("PI=3.14159") and parts["samplers"] == "g_noiseSampler"
    nodes = summary[summary.index("nodes="):summary.index(";values=")]
    assert "Clear[]{float clearDepth=0;vec4 clearColor=0.1 0.2 0.3 1;rt target=DepthBuffer;}" in nodes
    assert "LinearizeDepth[]{rt depthStencil=DepthBuffer;rt target=LinearDepth;}" in nodes
    assert "LightCulling[]{rt depthStencil=LinearDepth;}" in nodes and "Environment[]{}" in nodes
    assert "RenderScene[]{string GPUCulling=true;string Tag=Opaque;rt color=Main;rt depthStencil=DepthBuffer;rt depthHighZ=DepthHighZ;}" in nodes
    assert "Bloom[PostFx]{string defines=;float data.threshold=1.5;}" in nodes
    with pytest.raises(ValueError):
        runtime_binding.parse_renderer("frame:\n\t- name: Clear\n", 16, 16)


def test_the_reference_renderer_file_parses_when_mounted():
    """Content/DefaultRenderer.renderer itself (read only when /root/reference is mounted): every node of the path is found in frame order, the
    pyramid target carries `reduction: Min` and a full mip chain."""
    from pathlib import Path
    from sailor_amd import runtime_binding
    f = Path("/root/reference/Content/DefaultRenderer.renderer")
    if not f.exists():
        pytest.skip("reference not mounted")
    n, summary = runtime_binding.parse_renderer(f.read_text(), 3840, 2160)
    assert n > 20
    nodes = summary[summary.index("nodes="):summary.index(";values=")]
    order = [nodes.index(k) for k in ("LinearizeDepth[", "LightCulling[", "Environment[", "DepthHighZ[", "RenderScene[")]
    assert order == sorted(order)
    assert "LightCulling[]{rt depthStencil=LinearDepth;}" in nodes
    assert "DepthHighZ[]{rt src=HalfDepth;rt dst=DepthHighZ;}" in nodes
    assert "DepthHighZ:1920x1920:R32_SFLOAT:11:Min" in summary
    assert "LinearDepth:3840x2160:R32_SFLOAT:1" in summary


# ---------------------------------------------------------------------------------------------------------------
# shadow-pass planning (ECS/LightingECS.cpp:264-366): cascade frusta from matrices, cascade sets, change tracking
# ---------------------------------------------------------------------------------------------------------------
def camera_planes_for(cam):
    planes, _ = host.extract_frustum_planes(cam.world, cam.aspect, cam.fov, cam.z_near, cam.z_far)
    return planes


def _cascade_planes(cam):
    """the four cascade frusta of a directional light, as PrepareCSMPasses builds them (:272-292)"""
    sh = synth.make_shadow_set(cam, 16)
    planes = [host.extract_frustum_planes_matrix(sh.lights_matrices[k])[0] for k in range(4)]
    return sh, np.stack(planes)


def test_frustum_from_matrix_host_equals_oracle_and_contains_its_own_corners():
    cam = synth.make_camera(1280, 720)
    sh, planes = _cascade_planes(cam)
    for k in range(4):
        op, oc = oracle.extract_frustum_planes_matrix(sh.lights_matrices[k])
        hp, hc = host.extract_frustum_planes_matrix(sh.lights_matrices[k])
        np.testing.assert_array_equal(hp.view(np.uint32), op.view(np.uint32))
        np.testing.assert_array_equal(hc.view(np.uint32), oc.view(np.uint32))
        np.testing.assert_allclose(np.linalg.norm(hp[:, :3], axis=1), 1.0, rtol=1e-6)
        centre = hc.mean(0)
        assert (hp[:, :3] @ centre + hp[:, 3] > 0).all(), "the planes face inwards"
        # the clip-space cube maps back onto the corners
        m = np.asarray(sh.lights_matrices[k], np.float64).reshape(4, 4).T
        clip = (m @ np.c_[hc.astype(np.float64), np.ones(8)].T).T
        np.testing.assert_allclose(np.abs(clip[:, :3] / clip[:, 3:4]), 1.0, rtol=2e-3, atol=2e-3)


def test_csm_pass_planning_on_cascade_sets():
    """oracle_csm_caster_masks + the host bookkeeping, literally as the reference does it: cascades of one shadow type that are rendered in the same
    frame do not draw a mesh twice; an unchanged scene settles to "nothing to render" after two more frames; a moved mesh re-renders its cascade."""
    cam = synth.make_camera(1280, 720)
    ents = synth.make_entities(20000)
    world, aabb, _ = oracle.ecs_sweep(ents.transforms, ents.parent, ents.local_aabb, camera_planes_for(cam))
    _, planes = _cascade_planes(cam)
    masks = oracle.csm_caster_masks(aabb, planes)
    counts = [int(np.unpackbits(m.view(np.uint8)).sum()) for m in masks]
    assert all(c > 0 for c in counts) and counts[0] < counts[3]
    types = [host.SHADOW_EVSM, host.SHADOW_PCF, host.SHADOW_PCF, host.SHADOW_PCF]   # "for EVSM only the 1st cascade is EVSM" (:297)
    frames = np.zeros(20000, np.int64)
    render, final, snaps = host.plan_csm_passes(masks, types, frames, None)
    assert render == [0, 1, 2, 3]
    assert (final[0] == masks[0]).all() and (final[1] == masks[1]).all()          # cascade 1 is PCF, cascade 0 EVSM: nothing removed
    assert (final[2] & masks[1]).sum() == 0 and (final[3] & (masks[1] | masks[2])).sum() == 0
    # The removal only looks at earlier cascades that are re-rendered THIS frame (bCascadeAdded stays None for an unchanged cascade, :306-314),
    # so an unchanged scene settles over the next frames: cascade 2 comes back with its full set, then cascade 3, then nothing.
    render2, final2, snaps2 = host.plan_csm_passes(masks, types, frames, snaps)
    assert render2 == [2] and (final2[2] == masks[2]).all()
    render3, final3, snaps3 = host.plan_csm_passes(masks, types, frames, snaps2)
    assert render3 == [3] and (final3[3] == masks[3]).all()
    render4, _, snaps4 = host.plan_csm_passes(masks, types, frames, snaps3)
    assert render4 == []
    moved = int(np.nonzero(np.unpackbits(masks[2].view(np.uint8), bitorder="little"))[0][0])
    frames2 = frames.copy(); frames2[moved] = 7
    render5, _, _ = host.plan_csm_passes(masks, types, frames2, snaps4)
    holds = [bool((int(masks[k][moved >> 6]) >> (moved & 63)) & 1) for k in range(4)]
    assert render5 and min(render5) == holds.index(True)   # the first cascade that holds the moved mesh is re-rendered ...
    for k in render5:                                       # ... and every other one either holds it or follows a re-rendered cascade of its type
        assert holds[k] or any(z in render5 and types[z] == types[k] for z in range(k))


# ---- the `.world` scene description -------------------------------------------------------------------------------
def _summary_fields(summary, prefix):
    return [e for e in summary.split(";") if e.startswith(prefix)]


def _world_objects():
    cam = {"name": "Camera", "position": [0, 150, 0, 1], "rotation": [0, 0, 0, 0], "scale": [1, 1, 1, 1],
           "components": [{"typename": "Sailor::CameraComponent", "properties": {"fov": 75, "zNear": 0.5, "zFar": 9000}},
                          {"typename": "Sailor::EditorComponent", "properties": {}}]}
    rig = [{"name": "Rig", "position": [10, 20, 30, 1], "rotation": [0, 0.38268343, 0, 0.92387953], "scale": [2, 2, 2, 1], "components": []},
           {"name": "Arm", "position": [1.5, 2.25, -3.1, 1], "rotation": [0.1, 0.2, 0.3, 0.9273618], "scale": [1, 0.5, 1, 1], "parent": 0,
            "components": [{"typename": "Sailor::MeshRendererComponent", "properties": {"model": {"fileId": '"{0123-AB}"', "instanceId": "NullInstanceId"}}}]},
           {"name": "Lamp", "position": [0, 4, 0, 1], "rotation": [0.70710677, 0, 0, 0.70710677], "scale": [1, 1, 1, 1], "parent": 1,
            "components": [{"typename": "Sailor::LightComponent",
                            "properties": {"intensity": [255, 128, 7], "lightType": "Spot", "bounds": [12, 12, 12], "cutOff": [20, 35]}}]}]
    sun = {"name": "Sun", "position": [0, 0, 0, 0], "rotation": [0.0918623805, 0.858316064, 0.477002949, 0.477002949], "scale": [1, 1, 1, 1],
           "components": [{"typename": "Sailor::LightComponent", "properties": {"intensity": [17, 17, 17], "lightType": "Directional"}}]}
    return [[cam], rig, [sun]]


def test_world_description_instantiates_like_the_reference():
    """WorldPrefab::Deserialize + the game-object half of World::Instantiate: objects in file order, parents by index inside their prefab, world
    matrices through the parent chain, the camera's properties, LightComponents -> LightData with LightingECS's direction / position rule
    (world * (0, 0, -1, 0), world[3]); a zero quaternion (as Editor.world stores for unrotated objects) is the identity rotation."""
    from sailor_amd import runtime_binding, synth
    prefabs = _world_objects()
    text = synth.make_world_text("WorldEditor", prefabs)
    n, summary = runtime_binding.parse_world("# written by a test\n" + text)
    assert n == 5 and summary.startswith("name=WorldEditor;objects=5;")
    assert "Arm[parent=1 " in summary and "Lamp[parent=2 " in summary and "Rig[parent=-1 " in summary and "Sun[parent=-1 " in summary
    assert _summary_fields(summary, "camera{") == ["camera{owner=0 fov=75 zNear=0.5 zFar=9000}"]
    assert _summary_fields(summary, "mesh{") == ["mesh{owner=2 model={0123-AB}}"] and summary.endswith("other=1")
    # the lamp hangs two levels down: its world matrix is Rig * Arm * Lamp
    rig, arm, lamp = prefabs[1]
    world = host.mat4_mul(host.mat4_mul(host.transform_matrix(rig["position"], rig["rotation"], rig["scale"]),
                                        host.transform_matrix(arm["position"], arm["rotation"], arm["scale"])),
                          host.transform_matrix(lamp["position"], lamp["rotation"], lamp["scale"]))
    lights = _summary_fields(summary, "light{")
    assert len(lights) == 2 and lights[0].startswith("light{owner=3 type=2 intensity=255 128 7 attenuation=1 0.0219999999 0.00190000003 bounds=12 12 12 cutOff=20 35 ")
    got_dir = np.float32(lights[0].split("dir=")[1].split(" pos=")[0].split())
    got_pos = np.float32(lights[0].split("pos=")[1].rstrip("}").split())
    np.testing.assert_array_equal(got_dir, -world[8:11])
    np.testing.assert_allclose(got_pos, world[12:15], rtol=1e-5)
    # the directional light keeps LightData's defaults for what the file leaves out (ECS/LightingECS.h:23-28) and the file's unnormalised quaternion
    assert lights[1].startswith("light{owner=4 type=0 intensity=17 17 17 attenuation=1 0.0219999999 0.00190000003 bounds=100 100 100 cutOff=30 45 ")
    sun_world = host.transform_matrix([0, 0, 0, 0], prefabs[2][0]["rotation"], [1, 1, 1, 1])
    np.testing.assert_array_equal(np.float32(lights[1].split("dir=")[1].split(" pos=")[0].split()), -sun_world[8:11])
    # the camera's zero quaternion: the identity rotation, like glm::mat4_cast
    np.testing.assert_array_equal(host.transform_matrix([0, 150, 0, 1], [0, 0, 0, 0], [1, 1, 1, 1]), host.transform_matrix([0, 150, 0, 1], [0, 0, 0, 1], [1, 1, 1, 1]))


def test_world_description_rejects_what_it_cannot_read():
    from sailor_amd import runtime_binding, synth
    text = synth.make_world_text("W", _world_objects())
    with pytest.raises(ValueError, match="parentIndex out of range"):
        runtime_binding.parse_world(text.replace("parentIndex: 1\n", "parentIndex: 7\n"))
    with pytest.raises(ValueError, match="component index out of range"):
        runtime_binding.parse_world(text.replace("          - 1\n    components:", "          - 9\n    components:", 1))
    with pytest.raises(ValueError, match="not a sequence"):
        runtime_binding.parse_world("name: W\nprefabs: none\n")
    with pytest.raises(ValueError, match="line 3"):
        runtime_binding.parse_world("name: W\nprefabs:\n\t- gameObjects:\n")
    with pytest.raises(ValueError, match="indentation"):
        runtime_binding.parse_world("name: W\nprefabs:\n  - gameObjects:\n      - name: A\n       position: 3\n")
    assert runtime_binding.parse_world("name: Empty\nprefabs: []\n") == (0, "name=Empty;objects=0;other=0")


def test_the_reference_world_file_parses_when_mounted():
    """Content/Editor.world itself (read only when /root/reference is mounted): the four objects, the camera the synthetic frames copy
    (position (0, 150, 0), fov 90, zNear 1, zFar 20 000 -- SURVEY.md 8c), the directional light's intensity 17 and the two mesh renderers."""
    from sailor_amd import runtime_binding
    f = Path("/root/reference/Content/Editor.world")
    if not f.exists():
        pytest.skip("reference not mounted")
    n, summary = runtime_binding.parse_world(f.read_text())
    assert n == 4 and summary.startswith("name=WorldEditor;objects=4;Camera[parent=-1 pos=0 150 0 ")
    assert _summary_fields(summary, "camera{") == ["camera{owner=0 fov=90 zNear=1 zFar=20000}"]
    lights = _summary_fields(summary, "light{")
    assert len(lights) == 1 and lights[0].startswith("light{owner=3 type=0 intensity=17 17 17 attenuation=1 0.0219999999 0.00190000003 bounds=100 100 100 cutOff=30 45 ")
    assert len(_summary_fields(summary, "mesh{")) == 2 and summary.endswith("other=2")
    cam = synth_camera = __import__("sailor_amd.synth", fromlist=["make_camera"]).make_camera(64, 64)
    assert (cam.fov, cam.z_near, cam.z_far) == (90.0, 1.0, 20000.0) and tuple(cam.world[12:15]) == (0.0, 150.0, 0.0)


def test_csm_passes_follow_the_camera_and_light_thresholds():
    """CSMLightState::Equals (ECS/LightingECS.cpp:14-38): with unchanged caster sets a cascade is re-rendered when the camera has moved more than 15 units
    or turned past dot(forward, forward') = 0.9995 since its snapshot was TAKEN (kept snapshots are not refreshed, so drift accumulates), or when the light's
    transform differs at all."""
    masks = np.zeros((4, 1), np.uint64)
    masks[:, 0] = [0b0011, 0b0110, 0b1100, 0b1000]
    types = [host.SHADOW_EVSM, host.SHADOW_PCF, host.SHADOW_PCF, host.SHADOW_PCF]
    frames = np.zeros(4, np.int64)
    ident = np.float32([0, 0, 0, 1])

    def view(cam_pos, cam_rot=ident, light_rot=ident, index=0):
        return (index, np.float32(cam_pos + [1]), np.float32(cam_rot), np.float32([0, 0, 0, 1]), np.float32(light_rot))

    render, _, snaps = host.plan_csm_passes(masks, types, frames, None, view([0, 150, 0]))
    assert render == [0, 1, 2, 3]
    for _ in range(4):  # settle: later cascades re-render once the earlier ones stop claiming their meshes
        render, _, snaps = host.plan_csm_passes(masks, types, frames, snaps, view([0, 150, 0]))
    assert render == []
    # 10 units: inside the threshold, nothing re-rendered, the old snapshot (old camera) is kept ...
    render, _, snaps = host.plan_csm_passes(masks, types, frames, snaps, view([10, 150, 0]))
    assert render == [] and float(snaps[0][2][1][0]) == 0.0
    # ... so another 10 units is 20 from the snapshot: every cascade goes again
    render, _, snaps = host.plan_csm_passes(masks, types, frames, snaps, view([20, 150, 0]))
    assert render == [0, 1, 2, 3]
    for _ in range(4):
        render, _, snaps = host.plan_csm_passes(masks, types, frames, snaps, view([20, 150, 0]))
    assert render == []
    # a turn of 1.5 degrees keeps dot = 0.99966, 2 degrees gives 0.99939 < 0.9995
    def yaw(deg):
        h = np.radians(deg) / 2
        return np.float32([0, np.sin(h), 0, np.cos(h)])
    np.testing.assert_allclose(host.quat_rotate(yaw(90), [0, 0, -1]), [-1, 0, 0], atol=1e-6)
    render, _, snaps = host.plan_csm_passes(masks, types, frames, snaps, view([20, 150, 0], cam_rot=yaw(1.5)))
    assert render == []
    render, _, snaps = host.plan_csm_passes(masks, types, frames, snaps, view([20, 150, 0], cam_rot=yaw(2.0)))
    assert render == [0, 1, 2, 3]
    for _ in range(4):
        render, _, snaps = host.plan_csm_passes(masks, types, frames, snaps, view([20, 150, 0], cam_rot=yaw(2.0)))
    assert render == []
    # the light: any change at all, and a different light component
    tiny = np.float32([1e-7, 0, 0, 1])
    render, _, snaps = host.plan_csm_passes(masks, types, frames, snaps, view([20, 150, 0], cam_rot=yaw(2.0), light_rot=tiny))
    assert render == [0, 1, 2, 3]
    for _ in range(4):
        render, _, snaps = host.plan_csm_passes(masks, types, frames, snaps, view([20, 150, 0], cam_rot=yaw(2.0), light_rot=tiny))
    render, _, _ = host.plan_csm_passes(masks, types, frames, snaps, view([20, 150, 0], cam_rot=yaw(2.0), light_rot=tiny, index=3))
    assert render == [0, 1, 2, 3]


def test_row_costs_charge_long_tiles():
    """sailor_amd/dist.py:row_cost_entries -- a row's cost is its list volume plus LONG_TILE_ENTRIES per tile in a light cluster, in numpy and torch alike."""
    import torch
    from sailor_amd import dist as sdist
    num = np.array([[3, 95, 0, 7], [96, 128, 1, 0], [0, 0, 0, 0]], np.int64)
    want = np.array([105, 225 + 2 * sdist.LONG_TILE_ENTRIES, 0])
    np.testing.assert_array_equal(sdist.row_cost_entries(num.reshape(-1), 4), want)
    np.testing.assert_array_equal(sdist.row_cost_entries(torch.from_numpy(num.reshape(-1)), 4).numpy(), want)
    # a cluster row pulls the boundary towards itself: the band that holds it gets fewer rows
    flat = np.full((40, 10), 20, np.int64)
    clustered = flat.copy(); clustered[5, :6] = 128
    b0 = sdist.balanced_tile_rows(sdist.row_cost_entries(flat.reshape(-1), 10), 10, 4)
    b1 = sdist.balanced_tile_rows(sdist.row_cost_entries(clustered.reshape(-1), 10), 10, 4)
    assert b0 == [0, 10, 20, 30, 40] and b1[1] < b0[1]


# ---- L7: LightingECS::Tick's dirty-run streaming (Runtime/ECS/LightingECS.cpp:93-192) -- the C++ mirror's loop against the Python restatement ----
def _plan_light_uploads(dirty, active, mobility, frame_last, owner_frame, ticks=1, redirty=None):
    """sailor_rt_plan_light_uploads: (runs per tick as [(start, count)], records per tick as structured arrays, dirty / frameLastChange afterwards)"""
    from sailor_amd import runtime_binding
    rt = runtime_binding.load()
    n = len(dirty)
    d = np.array(dirty, np.uint8); a = np.array(active, np.uint8); m = np.array(mobility, np.uint8)
    fl = np.array(frame_last, np.uint64); of = np.array(owner_frame, np.uint64)
    rd = np.ascontiguousarray(redirty, np.uint8) if redirty is not None else None
    counts = np.zeros(ticks, np.int32)
    max_runs, max_records = ticks * (n + 1), ticks * (n + 1)
    sc = np.zeros(2 * max_runs, np.uint32)
    recs = np.zeros(max_records, host.LIGHT_DTYPE)
    P = C.c_void_p
    rt.sailor_rt_plan_light_uploads.argtypes = [C.c_int, P, P, P, P, P, C.c_int, P, P, P, C.c_int, P, C.c_int]
    total = rt.sailor_rt_plan_light_uploads(n, d.ctypes.data, a.ctypes.data, m.ctypes.data, fl.ctypes.data, of.ctypes.data, ticks,
                                            rd.ctypes.data if rd is not None else None, counts.ctypes.data, sc.ctypes.data, max_runs,
                                            recs.ctypes.data, max_records)
    assert total == counts.sum() <= max_runs
    runs, records, at, rat = [], [], 0, 0
    for t in range(ticks):
        r = [(int(sc[2 * (at + i)]), int(sc[2 * (at + i) + 1])) for i in range(counts[t])]
        at += counts[t]
        k = sum(c for _s, c in r)
        runs.append(r); records.append(recs[rat:rat + k]); rat += k
    return runs, records, d.astype(bool), fl


def test_lighting_tick_issues_one_copy_per_contiguous_dirty_run():
    n = 12
    clean = [False] * n
    ones, zeros, stationary = [True] * n, [0] * n, [1] * n
    # two lights that are not neighbours -> exactly two copies, at their own record slots
    dirty = list(clean); dirty[3] = dirty[7] = True
    runs, records, after, _ = _plan_light_uploads(dirty, ones, stationary, zeros, zeros)
    assert runs == [[(3, 1), (7, 1)]] and not after.any()
    assert [float(r["worldPosition"][0]) for r in records[0]] == [3.0, 7.0]
    # neighbours share one copy; a run that reaches the last slot is closed by the `index == Num() - 1` clause
    dirty = list(clean); dirty[4] = dirty[5] = dirty[6] = dirty[10] = dirty[11] = True
    runs, records, _, _ = _plan_light_uploads(dirty, ones, stationary, zeros, zeros)
    assert runs == [[(4, 3), (10, 2)]]
    assert [float(r["worldPosition"][0]) for r in records[0]] == [4.0, 5.0, 6.0, 10.0, 11.0]
    # nothing dirty -> nothing copied; everything dirty -> one copy of all
    assert _plan_light_uploads(clean, ones, stationary, zeros, zeros)[0] == [[]]
    assert _plan_light_uploads(ones, ones, stationary, zeros, zeros)[0] == [[(0, n)]]
    # dirtiness also comes from the owner's transform: frameLastChange < owner->GetFrameLastChange() (:152), and the component catches up (:174)
    owner = list(zeros); owner[2] = 5; owner[9] = 3
    runs, _, _, fl = _plan_light_uploads(clean, ones, stationary, zeros, owner)
    assert runs == [[(2, 1), (9, 1)]] and fl[2] == 5 and fl[9] == 3


def test_lighting_tick_inactive_slots_follow_the_reference_literally():
    n = 8
    ones, zeros, stationary = [True] * n, [0] * n, [1] * n
    # an inactive light is not packed even when dirty, and stays dirty
    dirty = [False] * n; dirty[2] = True
    active = list(ones); active[2] = False
    runs, _, after, _ = _plan_light_uploads(dirty, active, stationary, zeros, zeros)
    assert runs == [[]] and after[2]
    # `continue` sits in front of the flush: an inactive slot inside a dirty run does not end it -- one copy of two records from slot 3 on
    # (the record of light 5 lands in slot 4: the reference's behaviour, reproduced and pinned here)
    dirty = [False] * n; dirty[3] = dirty[5] = True
    active = list(ones); active[4] = False
    runs, records, _, _ = _plan_light_uploads(dirty, active, stationary, zeros, zeros)
    assert runs == [[(3, 2)]] and [float(r["worldPosition"][0]) for r in records[0]] == [3.0, 5.0]
    # ... and a run that is still open when the LAST slot is inactive is never copied, although its lights were marked clean
    dirty = [False] * n; dirty[6] = True
    active = list(ones); active[7] = False
    runs, records, after, _ = _plan_light_uploads(dirty, active, stationary, zeros, zeros)
    assert runs == [[]] and len(records[0]) == 0 and not after[6]


def test_lighting_tick_steps_over_static_lights_from_the_second_pass_on():
    n = 10
    ones, zeros = [True] * n, [0] * n
    mobility = [1] * n
    mobility[4] = mobility[5] = mobility[8] = 0 # EMobilityType::Static
    redirty = np.ones((2, n), np.uint8)         # everything marked dirty again before pass 2 and pass 3
    runs, records, _, _ = _plan_light_uploads(ones, ones, mobility, zeros, zeros, ticks=3, redirty=redirty)
    assert runs[0] == [(0, n)]                  # the first pass packs the static lights too (and lists them)
    # afterwards slots 4, 5 and 8 are stepped over, dirty or not.  Stepping over does not close the open run either (the jump happens before
    # the body, :95-103), so the reference issues ONE copy of the seven remaining records from slot 0 on -- reproduced literally
    assert runs[1] == [(0, 7)] and runs[2] == runs[1]
    assert [int(r["worldPosition"][0]) for r in records[1]] == [0, 1, 2, 3, 6, 7, 9]
    # with the dynamic lights clean, a static light that is dirtied again is never looked at
    redirty = np.zeros((1, n), np.uint8); redirty[0, 5] = 1
    runs, _, after, _ = _plan_light_uploads(ones, ones, mobility, zeros, zeros, ticks=2, redirty=redirty)
    assert runs[1] == [] and after[5]


@pytest.mark.parametrize("seed", range(8))
def test_lighting_tick_mirror_equals_the_python_restatement(seed):
    rng = np.random.default_rng(1000 + seed)
    n, ticks = int(rng.integers(1, 70)), 4
    dirty = rng.random(n) < 0.4
    active = rng.random(n) < 0.85
    mobility = rng.choice([0, 1, 2], size=n, p=[0.2, 0.5, 0.3])
    frame_last = rng.integers(0, 3, n)
    owner = rng.integers(0, 4, n)
    redirty = rng.random((ticks - 1, n)) < 0.3
    runs, records, after, fl = _plan_light_uploads(dirty, active, mobility, frame_last, owner, ticks=ticks, redirty=redirty)
    d, f, skip = [bool(x) for x in dirty], [int(x) for x in frame_last], []
    for t in range(ticks):
        if t > 0:
            d = [a or bool(b) for a, b in zip(d, redirty[t - 1])]
        want = oracle.lighting_tick_runs(d, list(active), list(mobility), f, list(owner), skip)
        assert runs[t] == [(s, len(b)) for s, b in want]
        carried = [i for _s, b in want for i in b]
        assert [int(r["worldPosition"][0]) for r in records[t]] == carried and all(int(r["worldPosition"][1]) == t for r in records[t])
    assert list(after) == d and [int(x) for x in fl] == f


def test_bands_recut_on_measured_times_move_rows_from_the_slow_bands_to_the_fast_ones():
    """sailor_amd.dist.rebalance_on_measured_times: a band measured slower than the rest gives rows away, one measured faster takes rows, equal times leave
    the boundaries alone, and the conventions of balanced_tile_rows hold (non-decreasing, first 0, last = rows, no empty band while rows last)."""
    from sailor_amd import dist as sdist
    rows = np.full(135, 240 * 20.0)
    rows[30:60] *= 3
    b = sdist.balanced_tile_rows(rows, 240, 8)
    same = sdist.rebalance_on_measured_times(b, [40.0] * 8, rows, 240)
    assert max(abs(x - y) for x, y in zip(same, b)) <= 1, (b, same)
    ms = [40, 44, 60, 47, 41, 41, 40, 30]
    nb = sdist.rebalance_on_measured_times(b, ms, rows, 240)
    assert nb[0] == 0 and nb[-1] == 135 and all(x < y for x, y in zip(nb, nb[1:]))
    width = lambda bb, r: bb[r + 1] - bb[r]
    assert width(nb, 2) < width(b, 2) and width(nb, 7) > width(b, 7)
    assert sdist.rebalance_on_measured_times([0, 1, 2], [5.0, 1.0], np.ones(2), 8) == [0, 1, 2], "two rows, two bands: nothing to move"
