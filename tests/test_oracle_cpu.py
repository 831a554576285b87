"""CPU suite for the oracle itself (no GPU): the C restatement against
  * every constant / layout the reference lets us assert -- read from the reference files at test
    time when /root/reference is mounted (this container), from their recorded values otherwise (the GPU box);
  * the independent NumPy restatement, bit for bit (the reference has no golden vectors: parity is otherwise unpinned);
  * the literal shader bubble sort vs the closed-form selection;
  * the committed golden fixtures under tests/golden/.
"""
import ctypes as C
import os
import sys
import re
from pathlib import Path

import numpy as np
import pytest

from oracle import oracle, oracle_np
from sailor_amd import host, synth

REF = Path("/root/reference")
GOLDEN = Path(__file__).resolve().parent / "golden"
ROOT = Path(__file__).resolve().parents[1]


# ---------------------------------------------------------------------------------------------------------------
# Appendix B: constants and layouts
# ---------------------------------------------------------------------------------------------------------------
def test_tile_constants():
    L = oracle.lib()
    assert (L.oracle_const_tile_size(), L.oracle_const_candidates_per_tile(), L.oracle_const_lights_per_tile()) == (16, 196, 128)
    assert L.oracle_const_num_cascades() == 4
    if REF.exists():
        txt = (REF / "Content/Shaders/Constants.glsl").read_text()
        assert int(re.search(r"LIGHTS_CULLING_TILE_SIZE (\d+)", txt).group(1)) == L.oracle_const_tile_size()
        assert int(re.search(r"LIGHTS_CANDIDATES_PER_TILE (\d+)", txt).group(1)) == L.oracle_const_candidates_per_tile()
        assert int(re.search(r"LIGHTS_PER_TILE (\d+)", txt).group(1)) == L.oracle_const_lights_per_tile()
        assert int(re.search(r"NUM_CSM_CASCADES (\d+)", txt).group(1)) == L.oracle_const_num_cascades()
        levels = [float(x) for x in re.search(r"ShadowCascadeLevels\[4\] = \{([^}]*)\}", txt).group(1).split(",")]
        assert [np.float32(v) for v in levels] == [np.float32(L.oracle_const_cascade_level_glsl(i)) for i in range(4)]
        hdr = (REF / "Runtime/FrameGraph/LightCullingNode.h").read_text()
        assert "LightsPerTile = 128" in hdr and "TileSize = 16" in hdr


def test_cascade_levels_glsl_literal_vs_cpp_fraction():
    L = oracle.lib()
    assert [L.oracle_const_cascade_level_glsl(i) for i in range(4)] == [np.float32(v) for v in (0.05, 0.1, 0.333333, 0.5)]
    assert [L.oracle_const_cascade_level_cpp(i) for i in range(4)] == [np.float32(1) / np.float32(d) for d in (20, 10, 3, 2)]
    assert L.oracle_const_cascade_level_glsl(2) != L.oracle_const_cascade_level_cpp(2)  # 0.333333 is not 1/3


def test_light_and_frame_layouts():
    L = oracle.lib()
    assert L.oracle_sizeof_light() == 112 and L.oracle_sizeof_ubo() == 232
    assert [L.oracle_offsetof_light(i) for i in range(8)] == [0, 4, 16, 32, 48, 64, 80, 96]
    assert host.LIGHT_DTYPE.itemsize == 112
    assert [host.LIGHT_DTYPE.fields[n][1] for n in host.LIGHT_DTYPE.names] == [0, 4, 16, 32, 48, 64, 80, 96]
    assert host.INSTANCE_DTYPE.itemsize == 96
    if REF.exists():
        glsl = (REF / "Content/Shaders/Lighting.glsl").read_text()
        order = re.findall(r"^\s*(?:uint|vec3|vec2)\s+(\w+);", glsl.split("struct LightData")[1].split("};")[0], re.M)
        assert order == ["type", "shadowType", "worldPosition", "direction", "intensity", "attenuation", "cutOff", "bounds"]


def test_poisson_disk_and_evsm_constants():
    L = oracle.lib()
    disk = np.array([[L.oracle_const_poisson(i, c) for c in range(2)] for i in range(16)], np.float32)
    assert disk.shape == (16, 2) and np.float32(disk[0, 0]) == np.float32(-0.94201624) and np.float32(disk[15, 1]) == np.float32(-0.14100790)
    if REF.exists():
        glsl = (REF / "Content/Shaders/Lighting.glsl").read_text()
        block = glsl.split("vec2 poissonDisk[16] = vec2[](")[1].split(");")[0]
        vals = np.array([float(v) for v in re.findall(r"-?\d+\.\d+", block)], np.float32).reshape(16, 2)
        np.testing.assert_array_equal(vals, disk)
        assert "EVSM_C1 = 40.0f" in glsl and "EVSM_C2 = 40.0f" in glsl


def test_canonical_exp_is_an_accurate_exp():
    L = oracle.lib()
    xs = np.linspace(-45.0, 45.0, 20001).astype(np.float32)
    got = np.array([L.oracle_canonical_expf(float(x)) for x in xs], np.float64)
    ref = np.exp(xs.astype(np.float64))
    assert np.max(np.abs(got - ref) / ref) < 2.5e-7  # ~2 ulp


# ---------------------------------------------------------------------------------------------------------------
# C restatement vs the independent NumPy restatement: bit for bit
# ---------------------------------------------------------------------------------------------------------------
def _frame(w, h, n, seed, **kw):
    cam = synth.make_camera(w, h)
    depth = synth.make_linear_depth(w, h, seed)
    lights = synth.make_lights(cam, depth, synth.LightSetConfig(count=n, **kw), seed)
    return cam, depth, lights


@pytest.mark.parametrize("case", [
    dict(w=128, h=96, n=512, seed=synth.SEED, radius_scale=6.0, spot_fraction=0.25, cluster_lights=300),
    dict(w=131, h=77, n=700, seed=5, radius_scale=8.0, spot_fraction=0.5),
    dict(w=320, h=200, n=3000, seed=9, radius_scale=4.0, cluster_lights=500, cluster_count=2),
    dict(w=16, h=16, n=300, seed=2, radius_scale=20.0),
    dict(w=40, h=24, n=0, seed=2),
])
def test_c_and_numpy_restatements_agree_bitwise(case):
    w, h, n, seed = case.pop("w"), case.pop("h"), case.pop("n"), case.pop("seed")
    cam, depth, lights = _frame(w, h, n, seed, **case)
    if n >= 300:
        lights["type"][7] = host.LIGHT_DIRECTIONAL
    cg, ci, cnt = oracle.light_cull(cam.frame, w, h, lights, depth, want_counts=True)
    ng, ni = oracle_np.light_cull(bytes(cam.frame), w, h, lights, depth)
    np.testing.assert_array_equal(cg, ng)
    np.testing.assert_array_equal(ci[: 1 + int(ci[0])], ni)
    lg, li, _ = oracle.light_cull(cam.frame, w, h, lights, depth, literal_select=True)
    np.testing.assert_array_equal(cg, lg)
    np.testing.assert_array_equal(ci, li)


@pytest.mark.parametrize("case", [
    dict(w=128, h=96, n=512, seed=synth.SEED, radius_scale=6.0, spot_fraction=0.25, cluster_lights=300),
    dict(w=131, h=77, n=700, seed=5, radius_scale=8.0, spot_fraction=0.5),
    dict(w=320, h=200, n=3000, seed=9, radius_scale=4.0, cluster_lights=500, cluster_count=2),
])
def test_fp32_overlap_decisions_against_float64(case):
    """The oracle is unpinned by the reference, so the float32 sphere / tile-frustum decisions (Math.glsl:224-239) are held against the same
    decisions in float64: of the ~10^5 .. 10^6 (tile, light) pairs of a frame the two may only differ where float64 itself puts the sphere within
    rounding of a plane or a depth bound -- and the lists the C oracle emits are exactly the float32 table's (first 196 in index order)."""
    w, h, n, seed = case.pop("w"), case.pop("h"), case.pop("n"), case.pop("seed")
    cam, depth, lights = _frame(w, h, n, seed, **case)
    lights["type"][7] = host.LIGHT_DIRECTIONAL
    fb = bytes(cam.frame)
    ok32, _ = oracle_np.overlap_table(fb, w, h, lights, depth, np.float32)
    ok64, slack = oracle_np.overlap_table(fb, w, h, lights, depth, np.float64)
    differ = ok32 != ok64
    print(f"{w}x{h}, {n} lights: {ok64.sum()} overlaps of {ok64.size} pairs; float32 differs on {differ.sum()}, largest slack there "
          f"{slack[differ].max() if differ.any() else 0.0:.2e}; smallest slack anywhere {slack.min():.2e}")
    assert differ.sum() <= 1e-4 * ok64.size
    assert (slack[differ] < 1e-5).all()
    # the C oracle's lists are the float32 table: per tile, the first 196 set bits (before the nearest-128 selection reorders them)
    g, idx, cnt = oracle.light_cull(cam.frame, w, h, lights, depth, want_counts=True)
    np.testing.assert_array_equal(np.minimum(ok32.sum(1), oracle.CAND), np.minimum(cnt, oracle.CAND))
    for t in np.nonzero(ok32.sum(1) <= oracle.KEEP)[0][:200]:
        np.testing.assert_array_equal(np.sort(idx[g[t, 0]: g[t, 0] + g[t, 1]]), np.nonzero(ok32[t])[0])


def test_bands_of_the_oracle_compose():
    cam, depth, lights = _frame(320, 200, 2000, 4, radius_scale=5.0, cluster_lights=300)
    g, idx, _ = oracle.light_cull(cam.frame, 320, 200, lights, depth)
    Ty = 13
    base, grids, segs = 0, [], []
    for r0, r1 in [(0, 4), (4, 9), (9, Ty)]:
        bg, bi, _ = oracle.light_cull(cam.frame, 320, 200, lights, depth, tile_rows=(r0, r1))
        ng, ni = oracle_np.light_cull(bytes(cam.frame), 320, 200, lights, depth, tile_rows=(r0, r1))
        np.testing.assert_array_equal(bg, ng)
        np.testing.assert_array_equal(bi[: 1 + int(bi[0])], ni)
        bg = bg.copy(); bg[:, 0] += base
        grids.append(bg); segs.append(bi[1: 1 + int(bi[0])]); base += int(bi[0])
    np.testing.assert_array_equal(np.concatenate(grids), g)
    np.testing.assert_array_equal(np.concatenate(segs), idx[1: 1 + int(idx[0])])


def test_selection_closed_form_equals_the_shaders_bubble_sort():
    """Appendix A step 4 / Appendix E: heavy ties included."""
    rng = np.random.default_rng(42)
    L = oracle.lib()
    for trial in range(400):
        n = int(rng.integers(1, 197))
        idx = rng.permutation(100000)[:n].astype(np.uint32)
        imp = rng.choice([0.0, 1.0, 2.5, 2.5, 7.0, rng.random()], n).astype(np.float32) if trial % 2 else rng.random(n).astype(np.float32)
        outs = []
        for literal in (0, 1):
            lst = np.zeros(128, np.uint32); num = C.c_uint32()
            L.oracle_select_emit(idx.ctypes.data_as(C.c_void_p), imp.ctypes.data_as(C.c_void_p), C.c_uint32(n), literal, lst.ctypes.data_as(C.c_void_p), C.byref(num))
            outs.append(lst[: num.value].copy())
        np.testing.assert_array_equal(outs[0], outs[1])
        assert len(outs[0]) == min(n, 128)
        if n <= 128:
            np.testing.assert_array_equal(outs[0], idx[::-1])       # descending candidate position
        else:
            kept = imp[np.searchsorted(idx[np.argsort(idx)], outs[0], sorter=None)] if False else None
            order = np.argsort(-imp, kind="stable")
            np.testing.assert_array_equal(outs[0], idx[order][::-1][:128])


def test_tile_and_pixel_conventions_agree():
    """Appendix D invariant: pixel (px, py) lies inside the frustum of tile (px/16, (H-1-py)/16)."""
    cam = synth.make_camera(640, 360)
    depth = synth.make_linear_depth(640, 360)
    surf = synth.make_surface(cam, depth)
    view = np.frombuffer(bytes(cam.frame.view), np.float32).reshape(4, 4).astype(np.float64)
    fb = np.frombuffer(bytes(cam.frame), np.uint8).copy()
    rng = np.random.default_rng(0)
    planes = np.zeros(16, np.float32); center = np.zeros(2, np.float32)
    for _ in range(300):
        px, py = int(rng.integers(0, 640)), int(rng.integers(0, 360))
        wp = np.append(surf[0, py, px, :3].astype(np.float64), 1.0)
        pv = wp @ view
        pv[2] = -pv[2]
        oracle.lib().oracle_tile_frustum(fb.ctypes.data_as(C.c_void_p), px // 16, (360 - 1 - py) // 16, planes.ctypes.data_as(C.c_void_p), center.ctypes.data_as(C.c_void_p))
        d = planes.reshape(4, 4)[:, :3].astype(np.float64) @ pv[:3]
        assert (d > -1e-3 * depth[py, px]).all(), (px, py, d)


# ---------------------------------------------------------------------------------------------------------------
# golden fixtures (inputs regenerated from the frozen generator, outputs committed)
# ---------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("name", ["tiny", "tiny_csm"])
def test_golden_fixture(name):
    path = GOLDEN / f"{name}.npz"
    assert path.exists(), "run tests/golden/make_golden.py"
    z = np.load(path)
    f = synth.make_frame(name)
    W, H = f.cam.width, f.cam.height
    # the generator is frozen: inputs must regenerate byte-identically
    assert z["frame_ubo"].tobytes() == bytes(f.cam.frame)
    np.testing.assert_array_equal(z["depth"], f.depth)
    assert z["lights"].tobytes() == f.lights.tobytes()
    np.testing.assert_array_equal(z["surface"], f.surface)
    g, idx, cnt = oracle.light_cull(f.cam.frame, W, H, f.lights, f.depth, want_counts=True)
    np.testing.assert_array_equal(g, z["grid"])
    np.testing.assert_array_equal(idx[: 1 + int(idx[0])], z["indices"])
    np.testing.assert_array_equal(cnt, z["passing"])
    csm = None
    if f.shadows is not None:
        np.testing.assert_array_equal(z["lights_matrices"], f.shadows.lights_matrices)
        for k in range(4):
            np.testing.assert_array_equal(z[f"shadow_map{k}"], f.shadows.maps[k])
        csm, _keep = oracle.make_csm(f.shadows.lights_matrices, f.shadows.maps)
    rad = oracle.shade(f.cam.frame, W, H, f.surface, f.lights, g, idx, csm)
    np.testing.assert_allclose(rad, z["radiance"], rtol=2e-6, atol=1e-6)  # libm powf may differ by an ulp across hosts


def test_linearize_depth_golden_and_numpy():
    """LinearizeDepth.shader:70,74: the C restatement against the committed fixture and against the
    same two IEEE operations in NumPy float32; raw 0 (nothing drawn) -> +inf, denormal raw -> finite or +inf, never NaN."""
    g = np.load(GOLDEN / "tiny_depth.npz")
    zn = np.float32(g["z_near"])
    lin = oracle.linearize_depth(float(zn), g["raw"])
    np.testing.assert_array_equal(lin.view(np.uint32), g["linear"].view(np.uint32))
    with np.errstate(divide="ignore", over="ignore"):
        ref = -((-zn) / g["raw"].astype(np.float32))
    np.testing.assert_array_equal(lin.view(np.uint32), ref.astype(np.float32).view(np.uint32))
    assert not np.isnan(lin).any() and np.isposinf(lin[g["raw"] == 0]).all()
    # regenerating the fixture's input from the frozen generator gives the same bytes
    f = synth.make_frame("tiny", with_surface=False)
    raw = synth.make_raw_depth(f.depth, 1.0, sky_fraction=0.06)
    np.testing.assert_array_equal(raw[1:], g["raw"][1:])


def test_nan_impacts_c_numpy_and_literal_agree():
    """Sky tiles (linear depth +inf -> NaN frustum centre -> NaN impacts) and NaN light positions: the closed-form selection
    has no meaning there; the C oracle defers to the literal bubble sort, the NumPy restatement replays it independently."""
    f = synth.make_frame("tiny", with_surface=False)
    W, H = f.cam.width, f.cam.height
    depth = f.depth.copy()
    depth[synth.make_raw_depth(depth, 1.0, sky_fraction=0.5) == 0] = np.inf
    lights = f.lights.copy()
    lights["worldPosition"][::7] = np.nan
    fb = np.frombuffer(bytes(f.cam.frame), np.uint8)
    for L, D in ((f.lights, depth), (lights, f.depth)):
        g, i, cnt = oracle.light_cull(f.cam.frame, W, H, L, D, want_counts=True)
        assert (cnt > oracle.KEEP).any()
        g2, i2, _ = oracle.light_cull(f.cam.frame, W, H, L, D, literal_select=True)
        with np.errstate(invalid="ignore"):
            g3, i3 = oracle_np.light_cull(fb, W, H, L, D)
        n = 1 + int(i[0])
        np.testing.assert_array_equal(g, g2); np.testing.assert_array_equal(i[:n], i2[:n])
        np.testing.assert_array_equal(g, g3); np.testing.assert_array_equal(i[:n], i3[:n])


def test_ambient_term_golden_and_sampler_conventions():
    """SURVEY.md 8f rank 2.  (a) the BRDF table and the ambient-lit tiny frame against the committed fixture; (b) the canonical cube
    sampler: face / (s, t) by the Vulkan major-axis table, texel centres reproduce the texel, lod interpolates between levels."""
    g = np.load(GOLDEN / "tiny_ibl.npz")
    lut = oracle.compute_brdf_lut(32, 32)
    np.testing.assert_allclose(lut, g["brdf_lut"], rtol=0, atol=1e-7)
    f = synth.make_frame("tiny")
    W, H = f.cam.width, f.cam.height
    ibl = synth.make_ibl_set(W, H, lut)
    assert abs(ibl.env_chain.astype(np.float64).sum() - float(g["env_checksum"])) < 1e-6 * float(g["env_checksum"])
    assert abs(ibl.ao.astype(np.float64).sum() - float(g["ao_checksum"])) < 1e-6 * float(g["ao_checksum"])
    gr, idx, _ = oracle.light_cull(f.cam.frame, W, H, f.lights, f.depth)
    oibl, _k = oracle.make_ibl(ibl.irradiance, ibl.env_chain, ibl.env_size, ibl.env_levels, ibl.brdf_lut, ibl.ao)
    rad = oracle.shade(f.cam.frame, W, H, f.surface, f.lights, gr, idx, ibl=oibl)
    np.testing.assert_allclose(rad, g["radiance"], rtol=1e-6, atol=1e-6)
    # (b)
    for d, face in (([1, .2, .3], 0), ([-1, .2, .3], 1), ([.1, 1, .3], 2), ([.1, -1, .3], 3), ([.1, .2, 1], 4), ([.1, .2, -1], 5), ([1, 1, 1], 4), ([1, 1, -1], 5)):
        assert oracle.cube_face_st(d)[0] == face
    dirs = synth._cube_dirs(ibl.env_size)
    lvl0 = ibl.env_chain[: 6 * ibl.env_size ** 2 * 4].reshape(6, ibl.env_size, ibl.env_size, 4)
    for fc, y, x in ((0, 3, 5), (1, 0, 0), (2, 63, 63), (3, 10, 40), (4, 31, 32), (5, 7, 7)):
        np.testing.assert_allclose(oracle.cube_sample_lod(ibl.env_chain, ibl.env_size, ibl.env_levels, dirs[fc, y, x], 0.0), lvl0[fc, y, x], rtol=2e-5)
    a = oracle.cube_sample_lod(ibl.env_chain, ibl.env_size, ibl.env_levels, [0.2, 0.9, -0.1], 2.0)
    b = oracle.cube_sample_lod(ibl.env_chain, ibl.env_size, ibl.env_levels, [0.2, 0.9, -0.1], 3.0)
    m = oracle.cube_sample_lod(ibl.env_chain, ibl.env_size, ibl.env_levels, [0.2, 0.9, -0.1], 2.25)
    np.testing.assert_allclose(m, 0.75 * a + 0.25 * b, rtol=1e-6)
    top = oracle.cube_sample_lod(ibl.env_chain, ibl.env_size, ibl.env_levels, [0.2, 0.9, -0.1], 99.0)  # clamped to the 1x1 level
    np.testing.assert_allclose(top, ibl.env_chain[-6 * 4:].reshape(6, 4)[2], rtol=1e-6)


def test_evsm_blur_weights_golden_and_numpy():
    """SURVEY.md 8f rank 3.  Lighting.glsl:87-99: every weight row sums to 0.5 (centre tap counted twice -> 1); the blurred cascade-0 map of the
    tiny_csm fixture; and an independent NumPy restatement of the two passes (same op order) bit for bit."""
    L = oracle.lib()
    W = np.array([[L.oracle_const_evsm_blur_weight(r, i) for i in range(12)] for r in range(12)], np.float64)
    assert np.abs(W.sum(1) - 0.5).max() < 2e-6 and (np.triu(W, 1) == 0).all() and (np.diff(W, axis=1)[np.tril_indices(12, -1)] <= 0).all() is not None
    g = np.load(GOLDEN / "tiny_blur.npz")
    f = synth.make_frame("tiny_csm", with_surface=False)
    m = np.ascontiguousarray(f.shadows.maps[0])
    out = oracle.evsm_blur(m, int(g["radii"][0]), int(g["radii"][1]))
    np.testing.assert_array_equal(out.view(np.uint32), g["blurred"].view(np.uint32))

    def np_pass(img, rx, ry, vertical):
        H, Wd = img.shape[:2]
        acc = np.zeros_like(img)
        w1 = W[min(rx, 12) - 1].astype(np.float32) if rx > 0 else None
        w2 = W[min(ry, 12) - 1].astype(np.float32) if ry > 0 else None
        idx = np.arange(H if vertical else Wd)
        for i in range(min(max(rx, ry), 12)):
            hi, lo = np.minimum(idx + i, idx[-1]), np.maximum(idx - i, 0)
            a, b = (img[hi], img[lo]) if vertical else (img[:, hi], img[:, lo])
            if i < rx:
                acc[..., 2:] = acc[..., 2:] + (a[..., 2:] + b[..., 2:]) * w1[i]
            if i < ry:
                acc[..., :2] = acc[..., :2] + (a[..., :2] + b[..., :2]) * w2[i]
        return acc

    ref = np_pass(np_pass(m, 2, 5, False), 2, 5, True)
    np.testing.assert_array_equal(out.view(np.uint32), ref.astype(np.float32).view(np.uint32))


def test_mesh_cull_compaction_is_a_stable_per_batch_partition():
    """oracle_mesh_cull_compact (ComputeMeshCulling.shader:146-177 restated literally) against an independent NumPy statement: per batch the
    kept records in order at the front, instanceCount = their number, everything behind the kept prefix untouched except for the flag."""
    cam = synth.make_camera(1920, 1080)
    s = synth.make_instance_set(20000, 100, first_instance=37)
    inst, bt = oracle.mesh_cull_compact(cam.frame, s.instances, 20000, 37, s.batches)
    flags = oracle.mesh_frustum_cull(cam.frame, s.instances[37:])["isCulled"]
    expect = s.instances.view(np.uint32).reshape(-1, 24).copy()
    expect[37:, 21] = flags
    after_flags = expect.copy()
    for b in range(100):
        f, c = int(s.batches[b, 4]), int(s.batches[b, 1])
        keep = np.nonzero(after_flags[f:f + c, 21] == 0)[0] + f
        expect[f:f + len(keep)] = after_flags[keep]
        assert bt[b, 1] == len(keep)
    np.testing.assert_array_equal(inst.view(np.uint32).reshape(-1, 24), expect)
    assert (bt[:, [0, 2, 3, 4]] == s.batches[:, [0, 2, 3, 4]]).all()
    assert (s.batches[:, 1] == 0).sum() > 0 and 0 < int(bt[:, 1].sum()) < 20000
    assert (inst["isCulled"][:37] == 7).all()


def test_ibl_prefilters_of_a_constant_sky():
    """ComputeIrradianceMap / ComputeEnvMap_IBL restated: a constant sky stays the constant (irradiance: 2 L mean(cos) over the uniform
    hemisphere set; env: a cosine-weighted mean of constant texels, at any mip level), alpha = 1, level 0 copied."""
    levels = 4
    offs, total = oracle.cube_level_offsets(8, levels)
    sky = np.tile(np.float32([0.5, 1.25, 2.0, 1.0]), total // 4)
    irr = oracle.compute_irradiance_map(sky, 8, levels, 2)
    np.testing.assert_allclose(irr[..., :3], np.broadcast_to(np.float32([0.5, 1.25, 2.0]), irr[..., :3].shape), rtol=2e-4)
    assert (irr[..., 3] == 1.0).all()
    env = oracle.prefilter_env_map(sky, 8, levels)
    np.testing.assert_allclose(env.reshape(-1, 4)[:, :3], np.broadcast_to(np.float32([0.5, 1.25, 2.0]), (total // 4, 3)), rtol=1e-5)
    np.testing.assert_array_equal(env[:offs[1]], sky[:offs[1]])


def test_env_prefilter_mip_selection_and_sampling_vector():
    """The texel direction comes from the texel CORNER (ComputeEnvMap_IBL.shader:42-43 has no + 0.5): texel (0, 0) of the 1 x 1 top mip of
    face +X looks along normalize(1, 1, 1)-ish (1, uv.y = 1, -uv.x = 1); and a sharper sky loses contrast level by level."""
    ibl = synth.make_ibl_set(16, 16, np.zeros((2, 2, 2), np.float32), env_size=16, with_ao=False)
    env = oracle.prefilter_env_map(ibl.env_chain, 16, ibl.env_levels)
    offs, total = oracle.cube_level_offsets(16, ibl.env_levels)
    top = env[offs[-1]:].reshape(6, 4)
    # roughness 1 at the 1 x 1 level: nearly the cosine-weighted hemisphere mean around (1, 1, 1) / sqrt(3) for face 0
    d = np.float32([1.0, 1.0, 1.0]) / np.sqrt(np.float32(3.0))
    out = np.zeros(4, np.float32)
    oracle.lib().oracle_cube_sample_lod(oracle._p(np.ascontiguousarray(ibl.env_chain)), 16, ibl.env_levels, oracle._p(np.ascontiguousarray(d)),
                                        C.c_float(ibl.env_levels - 1.0), oracle._p(out))
    assert np.all(np.abs(top[0, :3] - out[:3]) < 0.5 * np.abs(out[:3]) + 0.2)
    spread = [np.ptp(env[offs[l]:offs[l + 1]].reshape(-1, 4)[:, 0]) for l in range(1, ibl.env_levels - 1)]
    assert all(a >= b for a, b in zip(spread, spread[1:]))


def test_hiz_min_pyramid_and_occlusion_semantics():
    """ComputeDepthHighZ with the min-reduction sampler: 2:1 steps are the min of each 2 x 2 quad, every level's minimum is the image's; a texel
    centre hit (weight exactly 0 on the neighbours) returns that texel alone; OcclusionCulling only ever adds to the frustum result, and a
    pyramid of zeros (nothing drawn: reversed-Z far plane) occludes nothing."""
    rng = np.random.default_rng(3)
    img = rng.random((64, 64), dtype=np.float32) + np.float32(0.1)
    pyr = oracle.hiz_build(img, 64, 64, 7)
    offs, total = oracle.hiz_level_offsets(64, 64, 7)
    assert total == sum((64 >> l) ** 2 for l in range(7))
    np.testing.assert_array_equal(pyr[:64 * 64].reshape(64, 64), img)  # same size: every sample sits on a texel centre
    l1 = pyr[offs[1]:offs[2]].reshape(32, 32)
    np.testing.assert_array_equal(l1, img.reshape(32, 2, 32, 2).min(axis=(1, 3)))
    for l in range(7):
        assert pyr[offs[l]:(offs[l + 1] if l < 6 else total)].min() == img.min()
    # upsampling step (the reference's first step: W/2 x H/2 depth into a W/2 x W/2 target)
    tall = oracle.hiz_build(img[:32], 64, 64, 1).reshape(64, 64)
    assert tall.min() == img[:32].min() and tall.max() <= img[:32].max()
    cam = synth.make_camera(1280, 720)
    s = synth.make_instance_set(5000, 10)
    frustum = oracle.mesh_frustum_cull(cam.frame, s.instances)["isCulled"]
    zeros = oracle.mesh_cull_occlusion(cam.frame, s.instances, np.zeros(total, np.float32), 64, 64, 7)["isCulled"]
    np.testing.assert_array_equal(zeros, frustum)
    near = oracle.mesh_cull_occlusion(cam.frame, s.instances, np.full(total, 1.0, np.float32), 64, 64, 7)["isCulled"]  # an occluder on the near plane
    assert ((frustum == 1) <= (near == 1)).all() and near.sum() > frustum.sum()


def test_mesh_cull_and_prefilter_goldens():
    """tests/golden/tiny_mesh_cull.npz, tiny_prefilter.npz: the oracle reproduces its committed outputs from the regenerated inputs."""
    g = np.load(GOLDEN / "tiny_mesh_cull.npz")
    cam = synth.make_camera(640, 360)
    s = synth.make_instance_set(3000, 24, first_instance=9)
    raw = synth.make_raw_depth(synth.make_linear_depth(96, 54, 5, d_min=200.0, d_max=2500.0), cam.frame.cameraZNearZFar[0])
    pyr = oracle.hiz_build(raw, 96, 96, 7)
    np.testing.assert_array_equal(pyr.view(np.uint32), g["pyramid"].view(np.uint32))
    fi, fb = oracle.mesh_cull_compact(cam.frame, s.instances, 3000, 9, s.batches)
    np.testing.assert_array_equal(fi.view(np.uint32).reshape(-1, 24), g["frustum_instances"])
    np.testing.assert_array_equal(fb, g["frustum_batches"])
    oi, ob = oracle.mesh_cull_compact(cam.frame, s.instances, 3000, 9, s.batches, hiz=(pyr, 96, 96, 7))
    np.testing.assert_array_equal(oi.view(np.uint32).reshape(-1, 24), g["occlusion_instances"])
    np.testing.assert_array_equal(ob, g["occlusion_batches"])
    p = np.load(GOLDEN / "tiny_prefilter.npz")
    sky = synth.make_ibl_set(16, 16, np.zeros((2, 2, 2), np.float32), env_size=16, with_ao=False)
    assert float(sky.env_chain.astype(np.float64).sum()) == float(p["raw_checksum"])
    env = oracle.prefilter_env_map(sky.env_chain, 16, sky.env_levels)
    np.testing.assert_allclose(env, p["env"], rtol=2e-6, atol=1e-7)   # libm's cosf / sinf / log2f may differ in the last bit between hosts
    np.testing.assert_allclose(oracle.compute_irradiance_map(env, 16, sky.env_levels, 2), p["irradiance"], rtol=2e-6, atol=1e-7)


def test_depth_rasteriser_rules():
    """The canonical rasteriser of the shadow casters: a quad of two triangles covers its 8 x 8 texels exactly once each (top-left rule), the map's
    row 0 is clip-space +y, the larger (nearer, reversed-Z) depth wins, depths outside [0, 1] are clipped, a box seen along an axis is a square."""
    ident = np.eye(4, dtype=np.float32).reshape(16)
    quad = np.float32([[-0.5, -0.5, 0.3], [0.5, -0.5, 0.3], [0.5, 0.5, 0.3], [-0.5, 0.5, 0.3]])
    idx = np.uint32([[0, 1, 2], [0, 2, 3]])
    one = ident.reshape(1, 16)
    d = oracle.raster_depth(ident, quad, idx, one, 16, 16)
    assert (d > 0).sum() == 64 and (d[4:12, 4:12] == np.float32(0.3)).all()
    a, b = oracle.raster_depth(ident, quad, idx[:1], one, 16, 16), oracle.raster_depth(ident, quad, idx[1:, ::-1], one, 16, 16)
    assert ((a > 0) & (b > 0)).sum() == 0 and ((a > 0) | (b > 0)).sum() == 64
    top = oracle.raster_depth(ident, np.float32([[-1, 0.5, 0.2], [1, 0.5, 0.2], [0, 1.0, 0.2]]), np.uint32([[0, 1, 2]]), one, 16, 16)
    assert top[:4].max() > 0 and top[4:].max() == 0
    pos, tris = synth.unit_cube_mesh()
    scale = np.diag(np.float32([0.5, 0.25, 0.1, 1.0])).T.reshape(1, 16)
    scale[0, 14] = 0.5                                           # centre at z = 0.5: the box spans z in [0.4, 0.6]
    box = oracle.raster_depth(ident, pos, tris, scale, 32, 32)
    assert (box > 0).sum() == 16 * 8 and np.allclose(box[box > 0], 0.6)
    # back-face culling (frontFace counter-clockwise in Vulkan's framebuffer convention): the cube's outward faces are the front faces
    np.testing.assert_array_equal(oracle.raster_depth(ident, pos, tris, scale, 32, 32, cull_back=True), box)
    inside_out = oracle.raster_depth(ident, pos, tris[:, ::-1], scale, 32, 32, cull_back=True)
    assert (inside_out > 0).sum() == 16 * 8 and np.allclose(inside_out[inside_out > 0], 0.4)   # only the far faces survive
    far = scale.copy(); far[0, 14] = 1.2                         # spans [1.1, 1.3]: clipped away
    assert oracle.raster_depth(ident, pos, tris, far, 32, 32).max() == 0
    m = oracle.shadow_resolve_evsm(box)
    assert (m[box == 0] == 0).all() and np.allclose(m[box > 0][:, 0], np.exp(40 * 0.6), rtol=1e-5)


def test_equirect_to_cube_looks_where_the_direction_points():
    """ComputeEquirect2Cube.shader: a panorama whose colour IS its own (u, v) must come out as (phi / TwoPI, theta / PI) of each texel's
    direction; phi < 0 wraps with the Repeat sampler and sticks to column 0 with Clamp."""
    w, h, size = 512, 256, 16
    u = (np.arange(w, dtype=np.float32) + 0.5) / w
    v = (np.arange(h, dtype=np.float32) + 0.5) / h
    eq = np.zeros((h, w, 4), np.float32)
    eq[..., 0], eq[..., 1] = np.meshgrid(u, v)
    eq[..., 3] = 1.0
    cube = oracle.equirect_to_cube(eq, size, repeat=True)
    # face 4 (+Z): ret = (uv.x, uv.y, 1); texel (x, y) from its corner
    x, y = 11, 5
    d = np.array([2 * x / size - 1, 2 * (1 - y / size) - 1, 1.0])
    d /= np.linalg.norm(d)
    phi, theta = np.arctan2(d[2], d[0]), np.arccos(d[1])
    np.testing.assert_allclose(cube[4, y, x, :2], [phi / (2 * 3.141592), theta / 3.141592], atol=2e-3)
    # face 5 (-Z): phi < 0 -> u in (-0.5, 0): Repeat reads column u + 1, Clamp reads column 0
    d = np.array([-(2 * x / size - 1), 2 * (1 - y / size) - 1, -1.0])
    d /= np.linalg.norm(d)
    phi = np.arctan2(d[2], d[0])
    assert phi < 0
    np.testing.assert_allclose(cube[5, y, x, 0], phi / (2 * 3.141592) + 1.0, atol=2e-3)
    clamp = oracle.equirect_to_cube(eq, size, repeat=False)
    np.testing.assert_allclose(clamp[5, y, x, 0], u[0], atol=1e-6)
    # partial dispatch: untouched texels keep their contents
    part = oracle.equirect_to_cube(eq, size, cover=(8, 4), out=np.full((6, size, size, 4), 7.0, np.float32))
    assert (part[:, 4:, :, :] == 7.0).all() and (part[:, :, 8:, :] == 7.0).all()
    np.testing.assert_array_equal(part[:, :4, :8], cube[:, :4, :8])


def test_generate_mipmaps_is_a_chain_of_2x2_means():
    rng = np.random.default_rng(9)
    size, levels = 8, 4
    l0 = rng.uniform(0, 4, (6, size, size, 4)).astype(np.float32)
    chain = oracle.generate_mipmaps_cube(l0, size, levels)
    offs, total = oracle.cube_level_offsets(size, levels)
    np.testing.assert_array_equal(chain[:offs[1]], l0.reshape(-1))
    l1 = chain[offs[1]:offs[2]].reshape(6, 4, 4, 4)
    want = ((l0[:, 0::2, 0::2] + l0[:, 0::2, 1::2]) + (l0[:, 1::2, 0::2] + l0[:, 1::2, 1::2])) * np.float32(0.25)
    np.testing.assert_array_equal(l1, want)
    l3 = chain[offs[3]:].reshape(6, 1, 1, 4)
    np.testing.assert_allclose(l3[:, 0, 0], l0.reshape(6, -1, 4).mean(axis=1), rtol=1e-6)
    # a chain longer than log2(size) + 1 repeats the 1 x 1 level (mip extents stop at 1)
    longer = oracle.generate_mipmaps_cube(l0, size, 6)
    np.testing.assert_array_equal(longer[-24:], longer[-48:-24])


def test_rasteriser_cuts_triangles_at_the_near_plane():
    """A floor that runs from far in front of the eye to far behind it (two triangles, both with vertices at w <= 0): every pixel below the horizon
    that looks at the floor gets the analytic depth near / distance -- no holes where whole triangles used to be dropped; a floor so close that part
    of it is nearer than the near plane is cut there (ndc z > 1 is clipped)."""
    W, H, near = 160, 120, np.float32(0.1)
    P = np.zeros(16, np.float32)
    P[0], P[5], P[2 * 4 + 3], P[3 * 4 + 2] = H / W, 1.0, -1.0, near    # reversed Z, infinite far, looking down -Z
    one = np.eye(4, dtype=np.float32).reshape(1, 16)
    idx = np.uint32([[0, 1, 2], [0, 2, 3]])
    ny = 1.0 - 2.0 * (np.arange(H) + 0.5) / H
    nx = 2.0 * (np.arange(W) + 0.5) / W - 1.0
    for y_floor, x_half in ((-1.0, 40.0), (-0.05, 40.0)):
        pos = np.float32([[-x_half, y_floor, -200.0], [x_half, y_floor, -200.0], [x_half, y_floor, 30.0], [-x_half, y_floor, 30.0]])
        for flip in (False, True):
            tri = idx[:, ::-1] if flip else idx
            depth = oracle.raster_depth(P, pos, tri, one, W, H)
            with np.errstate(divide="ignore"):
                dist = np.where(ny < 0, y_floor / ny, np.inf)            # -z of the floor point a pixel row looks at
            want = np.where(np.isfinite(dist), near / dist, 0.0)
            x_at = nx[None, :] * dist[:, None] * (W / H)
            seen = (dist[:, None] < 199.0) & (np.abs(x_at) < x_half - 1e-3) & (want[:, None] < 0.999)
            gone = (want[:, None] > 1.001) | (ny[:, None] > 0) | (dist[:, None] > 201.0)
            assert seen.sum() > 1000
            np.testing.assert_allclose(depth[seen], np.broadcast_to(want[:, None], depth.shape)[seen], rtol=2e-4, atol=1e-4)  # vertices snap to 1/256 pixel: up to (depth gradient per pixel) / 256
            assert (depth[gone & np.ones_like(seen)] == 0).all()
        if y_floor == -0.05:
            assert (want > 1.001).sum() > 10, "part of the near floor is nearer than the near plane"
    # winding survives the cut: the floor seen from above is front- or back-facing as a whole
    a = oracle.raster_depth(P, pos, idx, one, W, H, cull_back=True)
    b = oracle.raster_depth(P, pos, idx[:, ::-1], one, W, H, cull_back=True)
    assert ((a > 0).sum() == 0) != ((b > 0).sum() == 0)
    np.testing.assert_array_equal(np.maximum(a, b), depth)


def test_raster_golden():
    """tests/golden/tiny_raster.npz: the canonical rasteriser (fill rule, snapping, near-plane cut, back-face rule) reproduces its committed depth images
    from the regenerated soup -- bit for bit: every operation on the way is a single correctly rounded fp32 / integer operation."""
    g = np.load(GOLDEN / "tiny_raster.npz")
    pos, idx = synth.make_triangle_soup(1500)
    assert float(pos.astype(np.float64).sum()) == float(g["soup_checksum"])
    d = pos.reshape(-1, 3, 3)[..., 2] * -1.0 - 0.1
    assert ((d >= 0).any(axis=1) & (d < 0).any(axis=1)).sum() > 50, "triangles across the near plane"
    P = synth.perspective_reversed_z(96, 64)
    one = np.eye(4, dtype=np.float32).reshape(1, 16)
    np.testing.assert_array_equal(oracle.raster_depth(P, pos, idx, one, 96, 64).view(np.uint32), g["both"].view(np.uint32))
    np.testing.assert_array_equal(oracle.raster_depth(P, pos, idx, one, 96, 64, cull_back=True).view(np.uint32), g["front"].view(np.uint32))
    assert (g["both"] != g["front"]).any() and float((g["both"] > 0).mean()) > 0.5


# ---------------------------------------------------------------------------------------------------------------
# hardening of the parity-unpinned oracle: an independent float64 restatement (oracle/oracle_f64.py, written from the
# reference's shader / C++ text only) against the fp32 C oracle's committed outputs
# ---------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("name", ["tiny", "tiny_csm"])
def test_c_oracle_radiance_agrees_with_the_float64_restatement(name):
    """K2 / K3: north_star's tolerance (1e-4 relative, floor 1e-7 of the frame's largest value) holds between the fp32 oracle and the float64
    restatement wherever the expression is well conditioned.  The only exceptions allowed are pixels at the specular peak of a smooth surface,
    where NdfGGX's denominator cosLh^2 (a^2 - 1) + 1 cancels to < 1e-3 and amplifies the fp32 rounding of the INPUTS (normal, half vector) by
    1 / denominator -- no fp32 evaluation of the shader can do better there; they must be few and still within 2 %."""
    from oracle import oracle_f64
    z = np.load(GOLDEN / f"{name}.npz")
    f = synth.make_frame(name)
    W, H = f.cam.width, f.cam.height
    csm = (z["lights_matrices"], [z[f"shadow_map{k}"] for k in range(4)]) if name == "tiny_csm" else None
    ref, min_denom = oracle_f64.shade(bytes(f.cam.frame), W, H, f.surface, f.lights, z["grid"], z["indices"], csm, want_conditioning=True)
    got = z["radiance"].astype(np.float64)
    np.testing.assert_array_equal(got[..., 3], ref[..., 3])  # outColor.a = albedo.a
    err = np.abs(got[..., :3] - ref[..., :3])
    tol = 1e-4 * np.abs(ref[..., :3]) + 1e-7 * np.abs(ref[..., :3]).max()
    bad = (err > tol).any(-1)
    assert bad.sum() <= 0.001 * bad.size, f"{bad.sum()} pixels beyond 1e-4"
    assert (min_denom[bad] < 1e-3).all(), "a well-conditioned pixel differs between the fp32 oracle and the float64 restatement"
    assert (err[bad] <= 2e-2 * np.abs(ref[..., :3][bad]) + 1e-7 * np.abs(ref).max()).all()
    assert np.median(min_denom[np.isfinite(min_denom)]) > 0.05  # (the frame as a whole is well conditioned)


def test_c_oracle_ecs_sweep_agrees_with_the_float64_restatement():
    """K4: world matrices and boxes to fp32 rounding of their largest terms, visibility bits wherever no plane distance is within rounding of 0."""
    from oracle import oracle_f64
    ents = synth.make_entities(4096)
    cam = synth.make_camera(128, 96)
    planes, _ = host.extract_frustum_planes(cam.world, cam.aspect, cam.fov, cam.z_near, cam.z_far)
    ow, oa, ov = oracle.ecs_sweep(ents.transforms, ents.parent, ents.local_aabb, planes)
    w64, a64, vis, margin = oracle_f64.ecs_sweep(ents.transforms, ents.parent, ents.local_aabb, planes)
    w32 = ow.reshape(-1, 4, 4).transpose(0, 2, 1).astype(np.float64)  # column-major storage -> M @ v
    scale = np.abs(w64).max(axis=(1, 2), keepdims=True)
    assert (np.abs(w32 - w64) <= 4e-6 * scale).all()
    bscale = np.abs(a64).max(axis=1, keepdims=True)
    assert (np.abs(oa - a64) <= 4e-6 * bscale).all()
    bits = np.unpackbits(ov.view(np.uint8), bitorder="little")[: len(vis)].astype(bool)
    differ = bits != vis
    assert (margin[differ] < 1e-2).all() and differ.sum() <= 2
    assert 0.1 < vis.mean() < 0.9
    # the FLT_MIN quirk of AABB::Apply (Math/Bounds.cpp:484) is part of both
    trs = np.array([[0, 0, 0, 1, 0, 0, 0, 1, 1, 1, 1, 1]], np.float32); trs[0, :3] = [-100, -100, -100]
    box = np.array([[-1, -1, -1, 1, 1, 1]], np.float32)
    _, a, _, _ = oracle_f64.ecs_sweep(trs, np.array([0xFFFFFFFF], np.uint32), box, planes)
    _, a32, _ = oracle.ecs_sweep(trs, np.array([0xFFFFFFFF], np.uint32), box, planes)
    assert (a[0, 3:] > 0).all() and (a[0, 3:] < 1e-37).all() and np.array_equal(a32[0, 3:], a[0, 3:].astype(np.float32))


def test_sse_batch_overlaps_aabb_against_the_scalar_form():
    """Frustum::OverlapsAABB(AABB*, n, int32*) (Math/Bounds.cpp:264-325), restated literally.  Its row loads (float offsets
    0,4,8 / 12,16,20, 24 floats per step) and the (row0, row1, row2, zero) transposes only make sense for THREE boxes per step laid out as
    vec4 min0, min1, min2, vec4 max0, max1, max2 with zero padding (then `zero`, which the transposes overwrite with the rows' fourth
    components, stays zero).  On such input lanes 0..2 must say what the scalar form (:245-260) says, with the inverted output convention
    (0x80000000 = culled); lane 3 is a box of zeros.  On the reference's own 24-byte {vec3 min, vec3 max} array the lanes do not correspond to
    boxes -- checked too, as the documented reason why no caller uses the function."""
    rng = np.random.default_rng(5)
    cam = synth.make_camera(1280, 720)
    planes, _ = host.extract_frustum_planes(cam.world, cam.aspect, cam.fov, cam.z_near, cam.z_far)
    pl = np.ascontiguousarray(planes, np.float32).reshape(24)
    steps = 4096
    centre = rng.uniform(-3000, 3000, (steps, 3, 3)).astype(np.float32); centre[..., 1] += 150
    ext = rng.uniform(0.5, 400, (steps, 3, 3)).astype(np.float32)
    mn, mx = centre - ext, centre + ext
    buf = np.zeros(steps * 24 + 4, np.float32)
    off = (-buf.ctypes.data % 16) // 4
    data = buf[off:off + steps * 24].reshape(steps, 6, 4)
    data[:, 0:3, :3] = mn; data[:, 3:6, :3] = mx           # .w padding stays 0
    res = np.zeros(steps * 4, np.int32)
    L = oracle.lib()
    L.oracle_overlaps_aabb_sse(pl.ctypes.data_as(C.c_void_p), data.ctypes.data_as(C.c_void_p), C.c_uint32(steps * 4), res.ctypes.data_as(C.c_void_p))
    res = res.reshape(steps, 4)
    assert set(np.unique(res)) <= {0, -0x80000000}
    # scalar form on the same boxes; the two add their four terms in a different order, so boxes within rounding of a plane are left out
    boxes = np.concatenate([mn, mx], -1).reshape(-1, 6).astype(np.float32)
    scalar = np.array([L.oracle_overlaps_aabb(pl.ctypes.data_as(C.c_void_p), np.ascontiguousarray(b).ctypes.data_as(C.c_void_p)) for b in boxes]).reshape(steps, 3)
    p4 = planes.reshape(6, 4).astype(np.float64)
    d = np.maximum(boxes[:, None, :3] * p4[None, :, :3], boxes[:, None, 3:] * p4[None, :, :3]).sum(-1) + p4[None, :, 3]
    clear = (np.abs(d).min(1) > 1e-2).reshape(steps, 3)
    assert clear.mean() > 0.99 and 0.05 < scalar.mean() < 0.95
    np.testing.assert_array_equal((res[:, :3] == 0)[clear], scalar.astype(bool)[clear])
    zero_box = (p4[:, 3] > 0).all()                          # lane 3: min = max = 0 -> distance = d of every plane
    assert ((res[:, 3] == 0) == zero_box).all()
    # the reference's own layout (24-byte boxes back to back): lane k is NOT box 4 i + k
    packed = np.zeros(steps * 24 + 4, np.float32)
    po = (-packed.ctypes.data % 16) // 4
    pk = packed[po:po + steps * 24]
    pk[:] = boxes[: steps * 4].reshape(-1)[: steps * 24] if boxes.shape[0] >= steps * 4 else np.resize(boxes.reshape(-1), steps * 24)
    res2 = np.zeros(steps * 4, np.int32)
    L.oracle_overlaps_aabb_sse(pl.ctypes.data_as(C.c_void_p), pk.ctypes.data_as(C.c_void_p), C.c_uint32(steps * 4), res2.ctypes.data_as(C.c_void_p))
    flat = np.resize(boxes, (steps * 4, 6))
    scalar2 = np.array([L.oracle_overlaps_aabb(pl.ctypes.data_as(C.c_void_p), np.ascontiguousarray(b).ctypes.data_as(C.c_void_p)) for b in flat])
    assert ((res2 == 0) != scalar2.astype(bool)).mean() > 0.05


def test_sphere_tests_and_the_light_pre_sort():
    """Frustum::OverlapsSphere / ContainsSphere (Math/Bounds.cpp:211-243) -- the product's host functions against the oracle's --
    and LightingECS::GetLightsInFrustum (ECS/LightingECS.cpp:209-260): only shadow-casting, active lights; directional ones in component order;
    point / spot lights whose sphere is CONTAINED, by distance, equal distances in reverse component order (std::lower_bound insertion)."""
    rng = np.random.default_rng(9)
    cam = synth.make_camera(1280, 720)
    planes, _ = host.extract_frustum_planes(cam.world, cam.aspect, cam.fov, cam.z_near, cam.z_far)
    pl = np.ascontiguousarray(planes, np.float32).reshape(24)
    L = oracle.lib()
    n = 3000
    pos = rng.uniform(-2500, 2500, (n, 3)).astype(np.float32); pos[:, 1] += 150; pos[:, 2] -= 1500
    rad = rng.uniform(1, 600, n).astype(np.float32)
    ov = ct = 0
    for i in range(n):
        sp = np.float32([*pos[i], rad[i]])
        o_ref = L.oracle_overlaps_sphere(pl.ctypes.data_as(C.c_void_p), sp.ctypes.data_as(C.c_void_p))
        c_ref = L.oracle_contains_sphere(pl.ctypes.data_as(C.c_void_p), sp.ctypes.data_as(C.c_void_p))
        assert host.overlaps_sphere(pl, pos[i], float(rad[i])) == bool(o_ref) and host.contains_sphere(pl, pos[i], float(rad[i])) == bool(c_ref)
        assert not c_ref or o_ref  # contained implies overlapping
        ov += o_ref; ct += c_ref
    assert 0 < ct < ov < n
    # the pre-sort
    types = rng.choice([0, 1, 2], n, p=[0.02, 0.6, 0.38]).astype(np.uint32)
    shadow = rng.choice([0, 1, 2], n, p=[0.3, 0.5, 0.2]).astype(np.uint32)
    active = (rng.uniform(size=n) > 0.1).astype(np.uint8)
    bounds = np.stack([rad * rng.uniform(0.2, 1, n), rad, rad * rng.uniform(0.2, 1, n)], 1).astype(np.float32)
    pos[10] = pos[20]; pos[30] = pos[20]; types[[10, 20, 30]] = 1; shadow[[10, 20, 30]] = 1; active[[10, 20, 30]] = 1; bounds[[10, 20, 30]] = 5.0
    pos[[10, 20, 30]] = np.float32([0, 150, -500])            # three point lights at the same distance, inside the frustum
    cam_pos = np.float32([0, 150, 0])
    dirs, (pi, pd), (si, sd) = host.lights_in_frustum(pl, cam_pos, types, shadow, pos, bounds, active)
    # reference semantics in plain Python
    e_dir, e_pt, e_sp = [], [], []
    for i in range(n):
        if shadow[i] == 0 or not active[i]:
            continue
        if types[i] == 0:
            e_dir.append(i); continue
        sp = np.float32([*pos[i], bounds[i].max()])
        if not L.oracle_contains_sphere(pl.ctypes.data_as(C.c_void_p), sp.ctypes.data_as(C.c_void_p)):
            continue
        dv = pos[i] - cam_pos
        dist = np.sqrt(np.float32(np.float32(dv[0] * dv[0] + dv[1] * dv[1]) + dv[2] * dv[2]), dtype=np.float32)
        lst = e_sp if types[i] == 2 else e_pt
        k = 0
        while k < len(lst) and lst[k][1] < dist:             # std::lower_bound under operator< on the distance
            k += 1
        lst.insert(k, (i, dist))
    np.testing.assert_array_equal(dirs, e_dir)
    np.testing.assert_array_equal(pi, [i for i, _ in e_pt]); np.testing.assert_array_equal(pd, np.float32([d for _, d in e_pt]))
    np.testing.assert_array_equal(si, [i for i, _ in e_sp]); np.testing.assert_array_equal(sd, np.float32([d for _, d in e_sp]))
    assert len(e_dir) > 3 and len(e_pt) > 20 and len(e_sp) > 10 and (np.diff(pd) >= 0).all()
    where = [int(np.nonzero(pi == k)[0][0]) for k in (30, 20, 10)]
    assert where[1] == where[0] + 1 and where[2] == where[0] + 2  # ties: the later component in front
    # all-active default, nothing to report
    d0, (p0, _), (s0, _) = host.lights_in_frustum(pl, cam_pos, types[:0], shadow[:0], pos[:0], bounds[:0])
    assert len(d0) == len(p0) == len(s0) == 0


def test_cpu_suite_of_the_oracle_under_address_and_ub_sanitizers():
    """a sanitizer pass over the C oracle on the CPU build (GPU sanitizers are not available on the pool): the tiny cull,
    shade with shadow maps, the ECS sweep on one and on several threads, the SSE batch form -- in a child interpreter with libasan preloaded."""
    import shutil
    import subprocess
    gcc = shutil.which("gcc")
    if gcc is None:
        pytest.skip("no gcc")
    asan = subprocess.run([gcc, "-print-file-name=libasan.so"], capture_output=True, text=True).stdout.strip()
    if not asan or not Path(asan).exists():
        pytest.skip("no libasan.so")
    subprocess.run(["make", "-C", str(ROOT / "oracle"), "liboracle_asan.so"], check=True, capture_output=True)
    code = r"""
import numpy as np
from oracle import oracle
from sailor_amd import synth, host
f = synth.make_frame("tiny_csm")
W, H = f.cam.width, f.cam.height
g, idx, cnt = oracle.light_cull(f.cam.frame, W, H, f.lights, f.depth, want_counts=True)
csm, keep = oracle.make_csm(f.shadows.lights_matrices, f.shadows.maps)
rad = oracle.shade(f.cam.frame, W, H, f.surface, f.lights, g, idx, csm)
ents = synth.make_entities(5000)
planes, _ = host.extract_frustum_planes(f.cam.world, f.cam.aspect, f.cam.fov, f.cam.z_near, f.cam.z_far)
w, a, v = oracle.ecs_sweep(ents.transforms, ents.parent, ents.local_aabb, planes)
w2, a2, v2, _ = oracle.ecs_sweep_threads(ents.transforms, ents.parent, ents.local_aabb, planes, ents.level_offsets, 3)
assert np.array_equal(w, w2) and np.array_equal(a, a2) and np.array_equal(v, v2)
masks = oracle.csm_caster_masks(a, np.stack([planes] * 4))
print("sanitized pass ok", int(idx[0]), float(rad.max()), int(np.unpackbits(v.view(np.uint8)).sum()))
"""
    env = dict(os.environ, LD_PRELOAD=asan, SAILOR_ORACLE_LIB=str(ROOT / "oracle" / "liboracle_asan.so"), ASAN_OPTIONS="detect_leaks=0:abort_on_error=0",
               PYTHONPATH=str(ROOT))
    p = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, env=env, cwd=ROOT, timeout=600)
    assert p.returncode == 0 and "sanitized pass ok" in p.stdout, (p.stdout[-2000:], p.stderr[-4000:])
    assert "ERROR: AddressSanitizer" not in p.stderr and "runtime error" not in p.stderr, p.stderr[-4000:]


def _exact_coverage(tri, W, H):
    """Which pixel centres does a triangle with vertices on the 1/256-pixel grid cover, in exact integer arithmetic, by the fill rule's DEFINITION
    (Vulkan 1.3 spec 27.7.1 / the D3D top-left rule in a y-down framebuffer): strictly inside, or on an edge that is a top edge (horizontal, the
    interior below it) or a left edge (not horizontal, the interior to its right) -- written orientation-free, from the third vertex's side, not
    from a normalised winding as the oracle's raster_top_left does."""
    cov = np.zeros((H, W), bool)
    (ax, ay), (bx, by), (cx, cy) = tri
    if (bx - ax) * (cy - ay) - (cx - ax) * (by - ay) == 0:
        return cov
    for j in range(H):
        for i in range(W):
            px, py = 256 * i + 128, 256 * j + 128
            inside = True
            for (x0, y0), (x1, y1), (x2, y2) in (((ax, ay), (bx, by), (cx, cy)), ((bx, by), (cx, cy), (ax, ay)), ((cx, cy), (ax, ay), (bx, by))):
                side_p = (x1 - x0) * (py - y0) - (y1 - y0) * (px - x0)
                side_c = (x1 - x0) * (y2 - y0) - (y1 - y0) * (x2 - x0)   # the interior's side of this edge
                if side_p == 0:
                    if y0 == y1:
                        ok = y2 > y0                       # top edge: horizontal, interior below (larger y)
                    else:
                        # left edge: the interior lies towards larger x.  x of the edge line at the third vertex's height, compared exactly:
                        # (x2 - x_edge(y2)) has the sign of ((x2 - x0) (y1 - y0) - (x1 - x0) (y2 - y0)) / (y1 - y0)
                        num = (x2 - x0) * (y1 - y0) - (x1 - x0) * (y2 - y0)
                        ok = (num > 0) == ((y1 - y0) > 0)
                    if not ok:
                        inside = False; break
                elif (side_p > 0) != (side_c > 0):
                    inside = False; break
            cov[j, i] = inside
    return cov


def test_rasteriser_fill_rule_against_exact_arithmetic():
    """The oracle's depth rasteriser (the thing the GPU's is bit-compared with) against the fill rule evaluated from its definition in exact integers:
    vertices placed exactly on the 1/256-pixel grid through an identity light matrix, random and degenerate-prone triangles (axis-aligned edges
    through pixel centres, shared edges, slivers), both windings.  Coverage must be identical; a mesh of two triangles sharing an edge covers every
    pixel of the quad exactly once; the depth of a covered pixel is the plane's depth at its centre to float rounding."""
    from fractions import Fraction
    W = H = 32
    rng = np.random.default_rng(17)
    ident = np.eye(4, dtype=np.float32)

    def to_position(X, Y, z):   # snapped window (X, Y) / 256 -> the ndc that the viewport transform maps back onto it, exactly in float32
        return [X / (256.0 * W / 2) - 1.0, 1.0 - Y / (256.0 * H / 2), z]

    def draw(tris, zs):
        pos = np.array([to_position(X, Y, z) for tri, zz in zip(tris, zs) for (X, Y), z in zip(tri, zz)], np.float32)
        idx = np.arange(len(pos), dtype=np.uint32).reshape(-1, 3)
        return oracle.raster_depth(ident.T.copy(), pos, idx, ident[None].copy(), W, H)

    cases = []
    for _ in range(150):
        kind = rng.integers(0, 4)
        if kind == 0:    # anywhere on the fine grid
            tri = [(int(rng.integers(0, 256 * W)), int(rng.integers(0, 256 * H))) for _ in range(3)]
        elif kind == 1:  # vertices on pixel centres: edges run through many centres
            tri = [(256 * int(rng.integers(0, W)) + 128, 256 * int(rng.integers(0, H)) + 128) for _ in range(3)]
        elif kind == 2:  # an axis-aligned edge through pixel centres
            y = 256 * int(rng.integers(1, H - 1)) + 128
            tri = [(256 * int(rng.integers(0, W // 2)) + 128, y), (256 * int(rng.integers(W // 2, W)) + 128, y), (int(rng.integers(0, 256 * W)), int(rng.integers(0, 256 * H)))]
        else:            # a vertical edge through pixel centres
            x = 256 * int(rng.integers(1, W - 1)) + 128
            tri = [(x, 256 * int(rng.integers(0, H // 2)) + 128), (x, 256 * int(rng.integers(H // 2, H)) + 128), (int(rng.integers(0, 256 * W)), int(rng.integers(0, 256 * H)))]
        if rng.integers(0, 2):
            tri = [tri[0], tri[2], tri[1]]
        cases.append(tri)
    for tri in cases:
        zs = [float(v) for v in rng.integers(1, 255, 3) / 256.0]
        got = draw([tri], [zs])
        want = _exact_coverage(tri, W, H)
        np.testing.assert_array_equal(got > 0, want, err_msg=str(tri))
        # depth = the plane through the three vertices at the pixel centre (exact rational barycentrics), to float rounding
        (ax, ay), (bx, by), (cx, cy) = tri
        area = (bx - ax) * (cy - ay) - (cx - ax) * (by - ay)
        for j, i in zip(*np.nonzero(want)):
            px, py = 256 * int(i) + 128, 256 * int(j) + 128
            w1 = Fraction((px - ax) * (cy - ay) - (py - ay) * (cx - ax), area)   # weight of vertex b: cross(p - a, c - a) / cross(b - a, c - a)
            w2 = Fraction((bx - ax) * (py - ay) - (by - ay) * (px - ax), area)   # weight of vertex c: cross(b - a, p - a) / cross(b - a, c - a)
            z = Fraction(zs[0]) + (Fraction(zs[1]) - Fraction(zs[0])) * w1 + (Fraction(zs[2]) - Fraction(zs[0])) * w2
            assert abs(float(z) - float(got[j, i])) <= 4e-7 * max(1.0, abs(float(z))), (tri, i, j)
    # two triangles sharing an edge: every pixel of the quad exactly once (drawn separately, coverage counted)
    for _ in range(40):
        q = [(int(rng.integers(0, 256 * W)), int(rng.integers(0, 256 * H))) for _ in range(4)]
        a, b, c, d = q
        if ((b[0] - a[0]) * (c[1] - a[1]) - (c[0] - a[0]) * (b[1] - a[1])) * ((c[0] - a[0]) * (d[1] - a[1]) - (d[0] - a[0]) * (c[1] - a[1])) <= 0:
            continue   # not convex along the diagonal a-c: the halves would overlap or be degenerate
        n1 = (draw([[a, b, c]], [[0.5] * 3]) > 0).astype(int)
        n2 = (draw([[a, c, d]], [[0.5] * 3]) > 0).astype(int)
        assert (n1 + n2).max() <= 1
        np.testing.assert_array_equal(n1 + n2, (_exact_coverage([a, b, c], W, H) | _exact_coverage([a, c, d], W, H)).astype(int))


def test_threaded_whole_frame_checkers_equal_the_single_thread_oracle():
    """oracle_light_cull_threads / oracle_shade_threads (the whole-frame checkers of the full-size GPU tests) are the same per-tile / per-pixel
    code spread over host threads: same bits, whatever the thread count, with and without a band, with the counts, with shadow maps."""
    f = synth.make_frame("tiny_csm")
    W, H = f.cam.width, f.cam.height
    g1, i1, c1 = oracle.light_cull(f.cam.frame, W, H, f.lights, f.depth, want_counts=True)
    for threads in (2, 3, 8):
        g, i, c = oracle.light_cull(f.cam.frame, W, H, f.lights, f.depth, want_counts=True, threads=threads)
        assert np.array_equal(g, g1) and np.array_equal(i, i1) and np.array_equal(c, c1)
    Ty = oracle.num_tiles(W, H)[1]
    gb1, ib1, _ = oracle.light_cull(f.cam.frame, W, H, f.lights, f.depth, tile_rows=(1, Ty - 1), literal_select=True)
    gb, ib, _ = oracle.light_cull(f.cam.frame, W, H, f.lights, f.depth, tile_rows=(1, Ty - 1), literal_select=True, threads=4)
    assert np.array_equal(gb, gb1) and np.array_equal(ib, ib1)
    csm, _keep = oracle.make_csm(f.shadows.lights_matrices, f.shadows.maps)
    r1 = oracle.shade(f.cam.frame, W, H, f.surface, f.lights, g1, i1, csm)
    for threads in (2, 5):
        r = oracle.shade(f.cam.frame, W, H, f.surface, f.lights, g1, i1, csm, threads=threads)
        assert np.array_equal(r.view(np.uint32), r1.view(np.uint32))
    rb1 = oracle.shade(f.cam.frame, W, H, f.surface, f.lights, g1, i1, csm, rows=(5, H - 9))
    rb = oracle.shade(f.cam.frame, W, H, f.surface, f.lights, g1, i1, csm, rows=(5, H - 9), threads=3)
    assert np.array_equal(rb.view(np.uint32), rb1.view(np.uint32))


# ---- round 3: the "next" rows a second time (oracle/oracle_f64.py, written from the shader text): EVSM blur, irradiance cube, pre-filtered env cube ----
def _smooth_cube(size0, levels):
    """a cube map whose texels are a smooth function of their direction -- continuous across the faces -- every level evaluated at its own texel
    centres.  (On random texels a sample direction that lands exactly on a face boundary -- the Hammersley set is full of them for texels on the
    cube's symmetry planes -- picks one face or the other by the last bit, and the two restatements then read unrelated values.)"""
    out = []
    for l in range(levels):
        s = max(size0 >> l, 1)
        c = (np.arange(s) + 0.5) / s * 2 - 1
        u, v = np.meshgrid(c, c)
        one = np.ones_like(u)
        faces = []
        for d in ((one, -v, -u), (-one, -v, u), (u, one, v), (u, -one, -v), (u, -v, one), (-u, -v, -one)):  # the inverse of the face-selection table
            d = np.stack(d, -1)
            d /= np.linalg.norm(d, axis=-1, keepdims=True)
            x, y, z = d[..., 0], d[..., 1], d[..., 2]
            faces.append(np.stack([1.5 + np.sin(2 * x + y), 1.2 + 0.8 * y * z + 0.3 * x, 2.0 + np.cos(3 * z - x) * 0.7, one], -1))
        out.append(np.stack(faces).astype(np.float32).reshape(-1))
    return np.concatenate(out)


@pytest.mark.parametrize("radii", [(2, 5), (1, 4), (1, 3), (1, 2), (12, 7), (20, 3)])
def test_evsm_blur_against_the_float64_restatement(radii):
    """GaussianBlur_Evsm (Lighting.glsl:83-127) twice: the C oracle's pass against oracle_f64.evsm_blur_pass, both directions, the reference's four
    radius pairs (ShadowCascadeBlur, ECS/LightingECS.h:60) and radii at / beyond the shader's cap of 12"""
    from oracle import oracle_f64
    rng = np.random.default_rng(31)
    img = (rng.random((41, 29, 4)) * 6 - 1).astype(np.float32)
    for vertical in (0, 1):
        dst = np.zeros_like(img)
        oracle.lib().oracle_evsm_blur_pass(oracle._p(img), oracle._p(dst), 29, 41, radii[0], radii[1], vertical)
        ref = oracle_f64.evsm_blur_pass(img, radii[0], radii[1], bool(vertical))
        assert np.abs(dst - ref).max() <= 2e-6 * np.abs(ref).max()
    both = oracle.evsm_blur(img, radii[0], radii[1])
    ref = oracle_f64.evsm_blur_pass(oracle_f64.evsm_blur_pass(img, radii[0], radii[1], False), radii[0], radii[1], True)
    assert np.abs(both - ref).max() <= 4e-6 * np.abs(ref).max()


def test_ibl_bakes_against_the_float64_restatement():
    """ComputeIrradianceMap.shader and ComputeEnvMap_IBL.shader: the C oracle against oracle_f64 (written from the shader text and the Vulkan cube
    sampling rules) on a smooth cube.  The irradiance cube (65 536 samples a texel) agrees to 1e-5; the pre-filtered levels to 2e-4 where the
    samples stay on the finer mips, and to 2e-3 at roughness 1, whose wide lobe reads the coarsest mip (8 x 8 faces: a direction on a face
    boundary is a tie that the two precisions break differently, and neighbouring faces' edge texels are a texel apart there)."""
    from oracle import oracle_f64
    S, L = 32, 3
    env = _smooth_cube(S, L)
    offs, total = oracle.cube_level_offsets(S, L)
    assert env.size == total
    a = oracle.compute_irradiance_map(env, S, L, 2)
    b = oracle_f64.compute_irradiance_map(env, S, L, 2)
    assert (np.abs(a - b) <= 5e-5 * np.abs(b)).all(), (np.abs(a - b) / np.abs(b)).max()
    p = oracle.prefilter_env_map(env, S, L)
    np.testing.assert_array_equal(p[: offs[1]], env[: offs[1]])   # level 0 is the raw cube (EnvironmentNode.cpp:196-206: a blit)
    for level, tol in ((1, 2e-4), (2, 2e-3)):
        q = oracle_f64.prefilter_env_level(env, S, L, level, level / (L - 1.0))
        sz = S >> level
        got = p[offs[level]: offs[level] + 6 * sz * sz * 4].reshape(6, sz, sz, 4)
        rel = np.abs(got - q) / np.abs(q)
        assert rel.max() <= tol, (level, rel.max())
        assert np.percentile(rel, 50) <= tol / 20, (level, np.percentile(rel, 50))


# ---- known answers worked out by hand from the shader's formulas (Standard.shader:286-340): independent of both restatements ----
from known_answers import one_light_frame as _one_light_frame, point_light_at_normal_incidence  # noqa: E402


@pytest.mark.parametrize("roughness, metallic", [(1.0, 0.0), (0.5, 0.0), (0.7, 1.0), (0.35, 0.6)])
def test_known_answer_point_light_at_normal_incidence(roughness, metallic):
    """n = Lo = Li = Lh: F = F0 (Schlick at cos = 1), NdfGGX = 1 / (pi a^2) with a = roughness^2 (:316-322: a2 / (pi (cos^2 (a2 - 1) + 1)^2) at
    cos = 1), both Schlick-GGX G1 terms are 1 (x / (x (1 - k) + k) at x = 1), so specular = F0 / (4 pi a^2); kd = (1 - F0)(1 - metallic); the
    point falloff is (1 - (d / r)^2) / (a.x + a.y d + a.z d^2) (:289-290).  radiance = (kd albedo + specular) intensity falloff."""
    got, want = point_light_at_normal_incidence(roughness, metallic)
    np.testing.assert_allclose(got[:3], want, rtol=3e-5)


def test_known_answer_point_light_window_and_attenuation():
    """On the sphere of radius bounds.x the window 1 - (d / r)^2 is exactly 0 and beyond it clamps to 0 (:290); at d = r / 2 it is 3 / 4."""
    albedo, intensity, att = (1.0, 1.0, 1.0), (1.0, 1.0, 1.0), (1.0, 0.0, 0.0)
    inside, _, d_in = _one_light_frame(host.LIGHT_POINT, 1.0, 0.0, albedo, 50.0, 100.0, att, intensity)
    beyond, _, _ = _one_light_frame(host.LIGHT_POINT, 1.0, 0.0, albedo, 150.0, 100.0, att, intensity)
    base = 0.96 + 0.04 / (4 * np.pi)   # kd albedo + F0 / (4 pi a^2) at roughness 1
    np.testing.assert_allclose(inside[:3], base * (1 - (d_in / 100.0) ** 2), rtol=3e-5)
    assert (beyond[:3] == 0).all()


def test_known_answer_spot_light_cone():
    """theta = cos of the angle between the light's axis and the direction to the surface point; intensity = clamp((theta - cutOff.y) / (cutOff.x -
    cutOff.y), 0, 1) times 1 / (a.x + a.y d + a.z d^2), and exactly 0 outside the outer cone (:297-306).  On the axis: 1; half way between the
    cones' cosines: 1 / 2; outside: 0."""
    albedo, intensity, att, d = (0.5, 0.5, 0.5), (2.0, 2.0, 2.0), (1.0, 0.01, 0.0), 30.0
    c_in, c_out = np.cos(np.radians(20.0)), np.cos(np.radians(40.0))
    base = (0.96 * 0.5 + 0.04 / (4 * np.pi)) * 2.0

    def want(theta, d32):
        return base * min(max((theta - c_out) / (c_in - c_out), 0.0), 1.0) / (att[0] + att[1] * d32)

    on_axis, _, d32 = _one_light_frame(host.LIGHT_SPOT, 1.0, 0.0, albedo, d, 1e9, att, intensity, cut_off=(c_in, c_out), off_axis=0.0)
    np.testing.assert_allclose(on_axis[:3], want(1.0, d32), rtol=3e-5)
    mid_angle = np.arccos((c_in + c_out) / 2)
    mid, _, d32 = _one_light_frame(host.LIGHT_SPOT, 1.0, 0.0, albedo, d, 1e9, att, intensity, cut_off=(c_in, c_out), off_axis=mid_angle)
    # (tilting the axis tilts Li = -direction with it: cosLi = cosLh-chain changes, so compare the CONE factor through the ratio to a wide cone)
    wide, _, _ = _one_light_frame(host.LIGHT_SPOT, 1.0, 0.0, albedo, d, 1e9, att, intensity, cut_off=(np.cos(mid_angle) - 1e-3, -1.0), off_axis=mid_angle)
    np.testing.assert_allclose(mid[:3] / wide[:3], 0.5, rtol=2e-4)
    outside, _, _ = _one_light_frame(host.LIGHT_SPOT, 1.0, 0.0, albedo, d, 1e9, att, intensity, cut_off=(c_in, c_out), off_axis=np.radians(50.0))
    assert (outside[:3] == 0).all()


def test_known_answer_light_cull_membership():
    """K1 from its definition: a small sphere in the middle of one tile's frustum slab is in that tile's list and in no other; a sphere behind the
    eye is in none; a sphere that contains the whole view volume is in every tile that has depth.  (Positions through the camera's own matrices:
    view space looks down -z, Appendix D.)"""
    f = synth.make_frame("tiny")
    W, H = f.cam.width, f.cam.height
    Tx, Ty = oracle.num_tiles(W, H)
    depth = np.full((H, W), 100.0, np.float32)           # a wall 100 units in front of the eye
    world = np.asarray(f.cam.world, np.float64).reshape(4, 4)   # column-major: rows of this array are the matrix' columns
    tan_half = np.tan(np.radians(f.cam.fov) / 2)
    aspect = W / H

    def view_to_world(x, y, z):
        return (world[0, :3] * x + world[1, :3] * y + world[2, :3] * z + world[3, :3]).astype(np.float32)

    def centre_of_tile(tx, ty, dist):   # the view-space point `dist` in front of the eye on the ray through the tile's centre (tile row 0 = the BOTTOM 16 rows: Appendix D)
        ndc_x = (tx * 16 + 8) / W * 2 - 1
        ndc_y = (ty * 16 + 8) / H * 2 - 1
        return view_to_world(ndc_x * tan_half * aspect * dist, ndc_y * tan_half * dist, -dist)

    lights = np.zeros(3, host.LIGHT_DTYPE)
    lights["type"] = host.LIGHT_POINT
    tx, ty = Tx // 2, Ty // 3
    lights["worldPosition"][0] = centre_of_tile(tx, ty, 100.0); lights["bounds"][0] = 0.05    # 0.05 units: far inside one tile (a tile is ~ 100 * 2 tan / H * 16 wide)
    lights["worldPosition"][1] = view_to_world(0.0, 0.0, +50.0); lights["bounds"][1] = 10.0   # behind the eye
    lights["worldPosition"][2] = view_to_world(0.0, 0.0, -100.0); lights["bounds"][2] = 1e6   # contains everything
    g, idx, _ = oracle.light_cull(f.cam.frame, W, H, lights, depth)
    lists = [set(idx[o: o + n].tolist()) for o, n in g]
    for t, l in enumerate(lists):
        assert 1 not in l and 2 in l
        assert (0 in l) == (t == ty * Tx + tx), (t, ty * Tx + tx)


def test_known_answer_directional_light_behind_a_constant_shadow_map():
    """K3's PCF from its definition (Lighting.glsl:168-197, :242-261): a light matrix that sends every point to the middle of the map at clip depth z0,
    maps holding the constant v.  Stored depth = v / 2 + 1 / 2, the fragment's = z0 / 2 + 1 / 2: all sixteen taps pass (factor 1) when the fragment is
    nearer than the map (reversed Z: larger), none (factor 0, radiance exactly 0) when it is farther, and 1 without a look-up below depth 1 / 2.  The
    unshadowed radiance is the normal-incidence closed form with falloff 1 (:287)."""
    albedo, intensity = (0.8, 0.5, 0.25), (3.0, 2.0, 5.0)
    F0 = 0.04
    want = ((1 - F0) * np.array(albedo) + F0 / (4 * np.pi)) * np.array(intensity)   # roughness 1, metallic 0
    v = 0.2
    maps = [np.zeros((64, 64, 4), np.float32)] + [np.full((64, 64), v, np.float16) for _ in range(3)]
    maps[0][..., 0] = v   # a PCF light reads cascade 0's red channel
    for z0, factor in ((0.6, 1.0), (0.1, 0.0), (-0.5, 1.0)):
        lm = np.zeros((4, 16), np.float32)
        lm[:, 14] = z0; lm[:, 15] = 1.0   # column-major: the last column = (0, 0, z0, 1)
        csm, _keep = oracle.make_csm(lm, maps)
        got, _, _ = _one_light_frame(host.LIGHT_DIRECTIONAL, 1.0, 0.0, albedo, 1.0, 1.0, (1.0, 0.0, 0.0), intensity, shadow_type=host.SHADOW_PCF, csm=csm)
        np.testing.assert_allclose(got[:3], want * factor, rtol=3e-5, atol=0.0)


@pytest.mark.parametrize("n, flips_expected", [(1024, (0, 0)), (1 << 20, (75, 2))])
def test_e4_octree_trace_over_integer_boxes_against_the_flat_float_sweep(n, flips_expected):
    """VERDICT r05 item 5c -- the SIZE of the sanctioned divergence of row E4, counted, not hidden.  The reference's TraceScene walks a TOctree whose elements
    are the world boxes TRUNCATED to integers (ECS/StaticMeshRendererECS.cpp:81,96,132 -> Containers/Octree.h:183-200,239-274); the product (and
    oracle.ecs_sweep) tests the float boxes flat.  oracle.trace_scene_octree_boxes restates the octree literally (insert, subdivide at eight elements,
    trace through the node boxes).  Held here: (i) the hierarchical walk visits exactly the elements whose own integer box passes -- a flat NumPy
    evaluation of Frustum::OverlapsAABB on the truncated boxes gives the same set, bit for bit; (ii) every entity is inserted (the root, 264 576 wide,
    strictly contains the scene); (iii) the entities whose visibility differs from the float sweep: none of C1's 1 024, 77 of C5's 1 048 576 (75 the
    float sweep sees and the octree does not: truncation shrinks a box by up to one unit per side; 2 the other way: truncation towards zero moves a
    centre).  DESIGN.md section 2 quotes these counts; a change in either the generator or the restatement shows up here."""
    cam = synth.make_camera(3840, 2160)
    planes, _ = host.extract_frustum_planes(cam.world, cam.aspect, cam.fov, cam.z_near, cam.z_far)
    ents = synth.make_entities(n)
    _, aabb, vis = oracle.ecs_sweep(ents.transforms, ents.parent, ents.local_aabb, planes)
    ov, ins, boxes, st = oracle.trace_scene_octree_boxes(aabb, planes)
    bits = lambda w: np.unpackbits(w.view(np.uint8), bitorder="little")[:n].astype(bool)
    fv, tv, inserted = bits(vis), bits(ov), bits(ins)
    assert inserted.all() and st["not_inserted"] == 0 and st["nodes"] % 8 == 1 and st["visited"] <= n
    # the truncation itself, from its definition
    c, e = (aabb[:, :3] + aabb[:, 3:]) * np.float32(0.5), (aabb[:, 3:] - aabb[:, :3]) * np.float32(0.5)
    np.testing.assert_array_equal(boxes[:, :3], np.trunc(c).astype(np.int32))
    np.testing.assert_array_equal(boxes[:, 3:], np.trunc(e).astype(np.int32))
    # (i) flat evaluation on the integer boxes, written independently of the C code
    p, x = boxes[:, :3].astype(np.float32), boxes[:, 3:].astype(np.float32)
    mn, mx = p - x, p + x
    pl = np.asarray(planes, np.float32).reshape(6, 4)
    flat = np.ones(n, bool)
    for i in range(6):
        d = ((np.maximum(mn[:, 0] * pl[i, 0], mx[:, 0] * pl[i, 0]) + np.maximum(mn[:, 1] * pl[i, 1], mx[:, 1] * pl[i, 1])) +
             np.maximum(mn[:, 2] * pl[i, 2], mx[:, 2] * pl[i, 2])) + pl[i, 3]
        flat &= d > 0
    np.testing.assert_array_equal(flat, tv)
    # (iii) the divergence, counted
    float_only, octree_only = int((fv & ~tv).sum()), int((tv & ~fv).sum())
    print(f"E4, {n} entities: float sweep sees {int(fv.sum())}, octree trace {int(tv.sum())}; {float_only} only the float sweep, {octree_only} only the octree "
          f"({st['nodes']} nodes, {st['visited']} elements visited)")
    assert (float_only, octree_only) == flips_expected
    if n == 1024:   # the four Editor.world objects among them
        assert np.array_equal(fv[:4], tv[:4])
