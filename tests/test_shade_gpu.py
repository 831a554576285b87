"""K2+K3 parity (GPU): PBR shade over per-tile lists with CSM sampling, through the C-ABI, against the CPU oracle.
Tolerance (BASELINE.json north_star): radiance within 1e-4 relative fp32 -- here exactly that, |gpu - ref| <= 1e-4*|ref| with NO absolute
floor: exact zeros must be exact zeros and a dim pixel is held to the same relative bound as a bright one (measured: 6e-7 on the fixtures)."""
import numpy as np
import pytest
import torch

from oracle import oracle
from sailor_amd import _lib, host, synth
from conftest import oracle_tile_rows
from sailor_amd.forward_plus import ForwardPlus, PreparedLights, upload_lights, upload_shadow_maps

pytestmark = pytest.mark.gpu
RTOL = 1e-4


def atol_of(ref):
    return 0.0


def gpu_frame(ctx, f, band=None, csm=True, flags=_lib.CULL_DEFAULT):
    cam = f.cam
    fp = ForwardPlus(ctx, cam.width, cam.height, max(len(f.lights), 1), band=band)
    b = fp.band
    rows = slice(b.fbRowBegin, b.fbRowBegin + b.fbRowCount)
    d = torch.from_numpy(np.ascontiguousarray(f.depth[rows])).to(ctx.device)
    s = torch.from_numpy(np.ascontiguousarray(f.surface[:, rows])).to(ctx.device)
    l = upload_lights(f.lights, ctx.device)
    fp.cull(cam.frame, l, len(f.lights), d, flags)
    desc, keep = (upload_shadow_maps(f.shadows, ctx.device) if (csm and f.shadows is not None) else (None, None))
    out = fp.shade(cam.frame, s, l, len(f.lights), desc)
    ctx.synchronize()
    return out.cpu().numpy(), fp


def oracle_frame(f, csm=True):
    W, H = f.cam.width, f.cam.height
    g, idx, _ = oracle.light_cull(f.cam.frame, W, H, f.lights, f.depth)
    desc = keep = None
    if csm and f.shadows is not None:
        desc, keep = oracle.make_csm(f.shadows.lights_matrices, f.shadows.maps)
    return oracle.shade(f.cam.frame, W, H, f.surface, f.lights, g, idx, desc)


def assert_oracle_rows(f, got, spans, csm_desc=None, gpu_lists=None):
    """`got` (the GPU's radiance, whole frame) against the oracle on the tile-row spans [(first tile row, count)]: the oracle culls the span itself
    (all host threads) and shades its framebuffer rows from its OWN lists; with gpu_lists = (grid, indices) the GPU's lists of the span must be the
    oracle's too."""
    W, H = f.cam.width, f.cam.height
    Tx, Ty = host.num_tiles(W, H)
    threads = oracle.host_threads()
    for tr0, n in spans:
        og, oi, _ = oracle.light_cull(f.cam.frame, W, H, f.lights, f.depth, tile_rows=(tr0, tr0 + n), threads=threads)
        if gpu_lists is not None:
            g, idx = gpu_lists
            np.testing.assert_array_equal(g[tr0 * Tx:(tr0 + n) * Tx, 1], og[:, 1])
            if n == Ty:
                np.testing.assert_array_equal(idx[: 1 + int(oi[0])], oi[: 1 + int(oi[0])])
        grid = np.zeros((Tx * Ty, 2), np.uint32); grid[:, 0] = 1
        grid[tr0 * Tx:(tr0 + n) * Tx] = og
        r0, r1 = max(H - 16 * (tr0 + n), 0), H - 16 * tr0
        ref = oracle.shade(f.cam.frame, W, H, f.surface, f.lights, grid, oi, csm_desc, rows=(r0, r1), threads=threads)
        assert_radiance_close(got[r0:r1], ref[r0:r1])


def assert_radiance_close(got, ref):
    assert got.shape == ref.shape
    assert np.isfinite(got).all()
    err = np.abs(got.astype(np.float64) - ref.astype(np.float64))
    tol = RTOL * np.abs(ref.astype(np.float64)) + atol_of(ref[..., :3])
    bad = err > tol
    assert not bad.any(), f"{bad.sum()} of {bad.size} values out of tolerance; worst rel {np.max(err / (np.abs(ref) + 1e-30)):.3e} abs {err.max():.3e}"
    np.testing.assert_array_equal(got[..., 3], ref[..., 3])  # outColor.a = albedo.a, passed through


def assert_bands_match_whole(parts, whole, whole_fp, H):
    """bands (bottom rows first in `parts`) against the whole frame's radiance: bit for bit on every tile whose list is shorter than the split blocks' 40
    entries; a band's longer tiles go to the split blocks (four waves share a quadrant's list, shadow maps or not since round 4) and differ by the order
    of four partial sums -- within the radiance tolerance"""
    got = np.concatenate(parts, 0)
    assert_radiance_close(got, whole)
    g, _ = whole_fp.lists_to_host()
    Tx, Ty = whole_fp.Tx, whole_fp.Ty
    short = (g[:, 1].reshape(Ty, Tx) < 40)
    rows = H - 1 - np.arange(H)                      # framebuffer row -> shader row (tile rows count from the bottom)
    mask = np.repeat(np.repeat(short, 16, 0)[rows], 16, 1)[:, : got.shape[1]]
    np.testing.assert_array_equal(got[mask], whole[mask])


def test_tiny_point_and_spot(ctx):
    f = synth.make_frame("tiny")
    got, _ = gpu_frame(ctx, f)
    ref = oracle_frame(f)
    assert (ref[..., :3].sum(-1) > 0).mean() > 0.2
    assert_radiance_close(got, ref)


def test_bright_metal_without_a_specular_term(ctx):
    """F0 -> 1 (metallic and albedo near 1) makes the diffuse factor 1 - F a difference of neighbours: a half-ulp difference in F -- a fused
    multiply-add where the reference rounds twice -- is 1e-4 of it.  Roughness 0 on every other pixel removes the specular term (NdfGGX = 0),
    so nothing hides it; directional lights so that every pixel is lit.  (Found by scripts/fuzz_parity.py, seed 11 case 222.)"""
    f = synth.make_frame("tiny")
    rng = np.random.default_rng(5)
    H, W = f.surface.shape[1:3]
    f.surface[2, :, :, :3] = (1.0 - rng.random((H, W, 3)) * 5e-4).astype(np.float32)   # albedo
    f.surface[2, :, :, 3] = (1.0 - rng.random((H, W)) * 5e-4).astype(np.float32)        # metallic
    f.surface[1, :, ::2, 3] = 0.0                                                        # roughness
    f.lights["type"][:4] = host.LIGHT_DIRECTIONAL
    got, _ = gpu_frame(ctx, f)
    ref = oracle_frame(f)
    lit = ref[..., :3][np.isfinite(ref[..., :3])]
    assert (lit > 0).mean() > 0.2 and np.median(lit[lit > 0]) < 1.0, "the diffuse term alone is small"
    fin = np.isfinite(ref)
    np.testing.assert_array_equal(np.isnan(got), np.isnan(ref))
    np.testing.assert_array_equal(np.isposinf(got), np.isposinf(ref))
    np.testing.assert_array_equal(np.isneginf(got), np.isneginf(ref))
    err = np.abs(got[fin].astype(np.float64) - ref[fin].astype(np.float64))
    assert (err <= RTOL * np.abs(ref.astype(np.float64))[fin]).all(), f"worst rel {np.max(err / (np.abs(ref.astype(np.float64))[fin] + 1e-300)):.3e}"


def test_tiny_with_csm_evsm_and_pcf(ctx):
    """configs[3] shape at fixture size: light 0 directional + EVSM, 4 cascades (64x64 maps)."""
    f = synth.make_frame("tiny_csm")
    got, _ = gpu_frame(ctx, f)
    ref = oracle_frame(f)
    no_shadow = oracle_frame(f, csm=False)
    assert np.abs(ref - no_shadow).max() > 1.0, "shadowing must change the picture"
    assert_radiance_close(got, ref)


def test_directional_with_pcf_shadow_type_reads_cascade0_red_channel(ctx):
    """Appendix C: shadowType != EVSM takes the PCF branch even on cascade 0 (the RGBA32F map's .r)."""
    f = synth.make_frame("tiny_csm")
    f.lights["shadowType"][0] = host.SHADOW_NONE
    got, _ = gpu_frame(ctx, f)
    assert_radiance_close(got, oracle_frame(f))


@pytest.mark.parametrize("size", [(131, 77), (320, 200), (1000, 562)])
def test_ragged_viewports_with_csm(ctx, size):
    w, h = size
    f = synth.make_frame("tiny_csm", width=w, height=h,
                         lights=synth.LightSetConfig(count=3000, spot_fraction=0.3, radius_scale=5.0, cluster_lights=300, directional_first=True),
                         shadow_size=96)
    got, _ = gpu_frame(ctx, f)
    assert_radiance_close(got, oracle_frame(f))


def test_deep_scene_reaches_all_cascades(ctx):
    """Depths up to 12 000 so that cascades 1..3 (PCF, R16F maps) are sampled, not only the EVSM cascade."""
    w, h = 256, 144
    cam = synth.make_camera(w, h)
    ramp = 60.0 * 300.0 ** ((np.arange(w) + 0.5) / w)  # 60 .. 18 000 across the frame
    depth = (ramp[None, :] * (0.8 + 0.4 * synth.uniforms(synth.STREAM_DEPTH, w * h, 77).reshape(h, w))).astype(np.float32)
    cfg = synth.LightSetConfig(count=2000, spot_fraction=0.25, radius_scale=6.0, directional_first=True, d_min=50.0, d_max=12000.0)
    f = synth.Frame("deep", cam, depth, synth.make_lights(cam, depth, cfg, 21), synth.make_surface(cam, depth, 21), synth.make_shadow_set(cam, 128, 21))
    fb = np.frombuffer(bytes(cam.frame.view), np.float32).reshape(4, 4)
    wp = f.surface[0, :, :, :3].reshape(-1, 3)
    zv = np.abs(wp @ fb[:3, 2] + fb[3, 2])
    levels = np.array([0.05, 0.1, 0.333333, 0.5]) * 20000.0
    hist = np.bincount(np.minimum(np.searchsorted(levels, zv, side="right"), 3), minlength=4)
    assert (hist > 200).all(), hist
    got, _ = gpu_frame(ctx, f)
    assert_radiance_close(got, oracle_frame(f))


def _deep_frame(w=256, h=144, seed=21):
    cam = synth.make_camera(w, h)
    ramp = 60.0 * 300.0 ** ((np.arange(w) + 0.5) / w)  # 60 .. 18 000 across the frame: every cascade is reached
    depth = (ramp[None, :] * (0.8 + 0.4 * synth.uniforms(synth.STREAM_DEPTH, w * h, 77).reshape(h, w))).astype(np.float32)
    cfg = synth.LightSetConfig(count=2000, spot_fraction=0.25, radius_scale=6.0, directional_first=True, d_min=50.0, d_max=12000.0)
    return synth.Frame("deep", cam, depth, synth.make_lights(cam, depth, cfg, seed), synth.make_surface(cam, depth, seed), synth.make_shadow_set(cam, 128, seed))


def _hostile_map(kind: str, shape, seed: int) -> np.ndarray:
    rng = np.random.default_rng(seed)
    if kind == "noise":          # no window is flat: every pixel's sixteen taps are computed from the window
        return rng.uniform(0.2, 0.9, shape).astype(np.float16)
    if kind == "flat":           # every window is flat: the reference depth against one value, both outcomes and the margin between them
        return np.full(shape, 0.43, np.float16)
    if kind == "steps":          # flat patches of a few texels: windows on, beside and across their edges
        coarse = rng.uniform(-0.2, 0.6, (-(-shape[0] // 5), -(-shape[1] // 7)))
        return np.kron(coarse, np.ones((5, 7)))[: shape[0], : shape[1]].astype(np.float16)
    if kind == "signed_denormal":  # negative texels, both zeros, half denormals: the halves are picked by their bits and converted, sign and all
        vals = np.array([-0.5, -6e-8, -0.0, 0.0, 6e-8, 3e-5, -3e-5, 0.43, 0.47], np.float16)
        return vals[rng.integers(0, len(vals), shape)]
    if kind == "nonfinite":      # an infinity or a NaN in the window: the taps propagate them as the reference does
        m = rng.uniform(0.0, 0.3, shape).astype(np.float16)
        bad = rng.uniform(size=shape)
        m[bad < 0.02] = np.float16(np.inf); m[(bad >= 0.02) & (bad < 0.04)] = np.float16(-np.inf); m[(bad >= 0.04) & (bad < 0.06)] = np.float16(np.nan)
        return m
    raise ValueError(kind)


@pytest.mark.parametrize("kind", ["noise", "flat", "steps", "signed_denormal", "nonfinite"])
@pytest.mark.parametrize("shape", [(128, 128), (96, 160), (40, 5), (12, 8192), (8, 8200)])
def test_pcf_window_against_the_oracle_on_hostile_maps(ctx, kind, shape):
    """Round 5: the sixteen PCF taps of an R16F cascade come out of one 6 x 6-texel window (shade_body.h: shadow_pcf_window; the form in which the window's
    extremes decided all sixteen compares at once was measured and dropped -- every tap is computed).  Maps built for every branch of the window's texel
    selection and of the compares -- not flat anywhere, flat everywhere, small flat patches, signed / zero /
    denormal texels, infinities and NaNs -- in shapes that are square, ragged, narrower than a window (every lane takes the tap-by-tap path) and wider
    than PCF_WINDOW_MAX_SIZE (likewise; 8192 itself is the widest map the window takes): the picture is the oracle's."""
    f = _deep_frame()
    for k in (1, 2, 3):
        f.shadows.maps[k] = _hostile_map(kind, shape, 1000 + k)
    got, _ = gpu_frame(ctx, f)
    ref = oracle_frame(f)
    unshadowed = oracle_frame(f, csm=False)
    if kind != "flat":
        assert np.abs(ref - unshadowed).max() > 0.0
    assert_radiance_close(got, ref)


@pytest.mark.parametrize("seed", range(12))
def test_pcf_window_on_noise_maps_of_odd_sizes(ctx, seed):
    """The window's two-origins rule rests on 1 / W being close enough to exact and on the disk's offsets staying clear of the integers (shade_body.h):
    noise maps (every tap computed) of sizes that are no powers of two, three different ones per frame, against the oracle."""
    rng = np.random.default_rng(4000 + seed)
    f = _deep_frame(seed=21 + seed)
    for k in (1, 2, 3):
        shape = (int(rng.integers(6, 700)), int(rng.integers(6, 700)))
        f.shadows.maps[k] = _hostile_map("noise" if k != 2 else "steps", shape, 5000 + 10 * seed + k)
    got, _ = gpu_frame(ctx, f)
    assert_radiance_close(got, oracle_frame(f))


def test_pcf_partial_shadow_counts_occur_on_the_noise_map(ctx):
    """(the test above would pass vacuously if every pixel's count were 0 or 16: on the noise map the directional light's shadow factor takes many of
    the seventeen values)"""
    f = _deep_frame()
    for k in (1, 2, 3):
        f.shadows.maps[k] = _hostile_map("noise", (128, 128), 1000 + k)
    f.lights = f.lights[:1].copy()     # the directional light alone: radiance = shadow * its unshadowed term
    lit = oracle_frame(f, csm=False)
    ref = oracle_frame(f)
    with np.errstate(divide="ignore", invalid="ignore"):
        ratio = np.where(lit[..., 0] > 0, ref[..., 0] / lit[..., 0], np.nan)
    sixteenths = np.unique(np.round(ratio[np.isfinite(ratio)] * 16).astype(int))
    assert len(sixteenths) >= 8, sixteenths
    got, _ = gpu_frame(ctx, f)
    assert_radiance_close(got, ref)


@pytest.mark.parametrize("world_size", [2, 3])
def test_bands_shade_identically(ctx, world_size):
    f = synth.make_frame("tiny_csm", width=320, height=200,
                         lights=synth.LightSetConfig(count=2000, spot_fraction=0.3, radius_scale=5.0, directional_first=True), shadow_size=64)
    whole, wfp = gpu_frame(ctx, f)
    parts = []
    for r in reversed(range(world_size)):  # band 0 = bottom rows of the framebuffer
        band = host.band_for_rank(320, 200, r, world_size)
        got, _ = gpu_frame(ctx, f, band=band)
        parts.append(got)
    assert_bands_match_whole(parts, whole, wfp, 200)
    g, _ = wfp.lists_to_host()
    assert (g[:, 1] >= 40).any(), "the bands have tiles for the split blocks (shadowed: k2_shade_band_csm*)"
    assert_radiance_close(np.concatenate(parts, 0), oracle_frame(f))


@pytest.mark.parametrize("from_tile_lists", [False, True])
def test_sentinel_index_stops_the_light_loop(ctx, from_tile_lists):
    """Standard.shader:430-433: an index of 0xFFFFFFFF ends the tile's loop -- in culledLights (the reference's buffers) and likewise in the cull's
    per-tile slot when the shade reads the lists there."""
    f = synth.make_frame("tiny")
    W, H = f.cam.width, f.cam.height
    got, fp = gpu_frame(ctx, f)
    g, idx = fp.lists_to_host()
    t = int(np.argmax(g[:, 1]))
    cut = int(g[t, 0]) + 2
    if from_tile_lists:   # the third entry of tile t's 128-entry slot
        at = fp.tile_lists - fp.workspace.data_ptr() + 4 * (128 * t + 2)
        fp.workspace[at: at + 4] = 255
    else:
        culled = fp.culled.clone()
        culled[cut] = -1
        fp.culled = culled
    fp.shade_from_tile_lists = from_tile_lists
    s = torch.from_numpy(f.surface).to(ctx.device)
    out = fp.shade(f.cam.frame, s, upload_lights(f.lights, ctx.device), len(f.lights), None).cpu().numpy()
    ref_idx = np.zeros(1 + len(g) * 128, np.uint32); ref_idx[: len(idx)] = idx; ref_idx[cut] = 0xFFFFFFFF
    ref = oracle.shade(f.cam.frame, W, H, f.surface, f.lights, g, ref_idx, None)
    assert_radiance_close(out, ref)
    assert np.abs(out - got).max() > 0


def test_linearity_in_light_intensity_at_4k(ctx):
    """Full-size property (configs[2]): radiance is linear in the light intensities -- doubling every intensity doubles
    the picture exactly (power-of-two scaling commutes with every rounding), and the ENTIRE 4K frame is within tolerance of the oracle."""
    f = synth.make_frame("C3")
    a, fp_a = gpu_frame(ctx, f)
    f2 = synth.Frame(f.name, f.cam, f.depth, f.lights.copy(), f.surface, None)
    f2.lights["intensity"] *= 2.0
    b, _ = gpu_frame(ctx, f2)
    np.testing.assert_array_equal(b[..., :3], 2.0 * a[..., :3])
    # the WHOLE frame against the oracle (its own lists, which the GPU's must equal; all host threads)
    assert_oracle_rows(f, a, oracle_tile_rows(135, [(40, 3), (101, 1)]), gpu_lists=fp_a.lists_to_host())


def test_c4_4k_with_cascaded_shadow_maps(ctx):
    """BASELINE.json configs[3] at full size (C3 + a directional EVSM light over four 4096^2 cascades): shadowing only ever removes light
    (factor in [0, 1], every term non-negative), two bands reproduce the whole frame (the split of configs[3]: bit for bit but for the long tiles the
    band kernel splits), and the
    ENTIRE frame -- lists and shadowed radiance -- is held against the oracle."""
    f = synth.make_frame("C4")
    W, H = f.cam.width, f.cam.height
    whole, fp = gpu_frame(ctx, f)
    unshadowed, _ = gpu_frame(ctx, f, csm=False)
    assert np.isfinite(whole).all()
    assert (whole[..., :3] <= unshadowed[..., :3] * (1 + 1e-5) + 1e-6).all()
    assert (whole[..., :3] < unshadowed[..., :3] * 0.99).mean() > 0.01, "the directional light is shadowed somewhere"
    parts = [gpu_frame(ctx, f, band=host.band_for_rank(W, H, r, 2))[0] for r in (1, 0)]  # band 0 = bottom rows
    assert_bands_match_whole(parts, whole, fp, H)
    desc, keep = oracle.make_csm(f.shadows.lights_matrices, f.shadows.maps)
    assert_oracle_rows(f, whole, oracle_tile_rows(135, [(77, 1), (20, 1)]), csm_desc=desc, gpu_lists=fp.lists_to_host())


def test_tile_order_hint_lists_the_long_tiles_and_they_are_split(ctx):
    """sailor_hip_light_cull_tile_order (split frames only): the band's per-tile list lengths as bytes, in tile order (written by k1_tile_cull beside
    tileNum: also after a cull with a deferred pack; round 4's form was a list of the long tiles appended through two device-scope counters).  Shading a
    band with it hands the tiles of >= 40 lights to the split blocks (four waves share one quadrant's list), which find them in these bytes: tiles below
    40 lights keep their bits, the split ones differ from the one-block form by the order of four partial sums only -- both within the radiance
    tolerance of the oracle."""
    import ctypes as C
    f = synth.make_frame("tiny")
    W, H, N = f.cam.width, f.cam.height, len(f.lights)
    whole = ForwardPlus(ctx, W, H, N)
    assert not whole.tile_order, "no hint for the whole frame"
    band = host.band_for_rank(W, H, 1, 2)
    fp = ForwardPlus(ctx, W, H, N, band=band)
    assert fp.tile_order
    rows = slice(band.fbRowBegin, band.fbRowBegin + band.fbRowCount)
    lights = upload_lights(f.lights, ctx.device)
    fp.cull(f.cam.frame, lights, N, torch.from_numpy(np.ascontiguousarray(f.depth[rows])).to(ctx.device))
    g, _ = fp.lists_to_host()
    T = fp.band_tiles
    lengths = np.empty(T, np.uint8)
    lib = _lib.load()
    _lib.check(lib.sailor_hip_buffer_download(ctx.handle, lengths.ctypes.data, C.c_void_p(fp.tile_order), 0, T), "download", ctx.handle)
    num = g[:, 1].astype(np.int64)
    np.testing.assert_array_equal(lengths, num)
    assert (num >= 96).any() and ((num >= 40) & (num < 96)).any() and (num < 40).any()
    s = torch.from_numpy(np.ascontiguousarray(f.surface[:, rows])).to(ctx.device)
    with_hint = fp.shade(f.cam.frame, s, lights, N).cpu().numpy()
    fp.use_tile_order = False
    without = fp.shade(f.cam.frame, s, lights, N).cpu().numpy()
    ref = oracle_frame(f)[rows]
    assert_radiance_close(with_hint, ref)
    assert_radiance_close(without, ref)
    # per tile: identical bits unless the tile went to the split blocks
    Hb = band.fbRowCount
    for t in range(T):
        ty, tx = divmod(t, fp.Tx)
        gy0 = (band.tileRowBegin + ty) * 16   # shader rows count from the bottom of the frame
        r1, r0 = H - gy0 - band.fbRowBegin, max(H - gy0 - 16, band.fbRowBegin) - band.fbRowBegin
        a, b = with_hint[r0:r1, tx * 16:tx * 16 + 16], without[r0:r1, tx * 16:tx * 16 + 16]
        if num[t] < 40:
            np.testing.assert_array_equal(a, b)
    assert np.abs(with_hint - without).max() > 0, "the long tiles took the split path"


def test_a_band_is_shaded_from_a_cull_whose_pack_is_still_deferred(ctx):
    """Round 4: the hint a band's split blocks take their long tiles from is written by k1_tile_cull, so a band's shade needs nothing of k1_pack --
    shading right behind cull(defer_pack=True), with the canonical buffers not written yet, gives the bits of the ordinary sequence."""
    f = synth.make_frame("tiny")
    W, H, N = f.cam.width, f.cam.height, len(f.lights)
    band = host.band_for_rank(W, H, 1, 2)
    rows = slice(band.fbRowBegin, band.fbRowBegin + band.fbRowCount)
    lights = upload_lights(f.lights, ctx.device)
    d = torch.from_numpy(np.ascontiguousarray(f.depth[rows])).to(ctx.device)
    s = torch.from_numpy(np.ascontiguousarray(f.surface[:, rows])).to(ctx.device)
    fp = ForwardPlus(ctx, W, H, N, band=band)
    fp.cull(f.cam.frame, lights, N, d)
    g, _ = fp.lists_to_host()
    assert (g[:, 1] >= 40).any(), "the band has tiles for the split blocks"
    usual = fp.shade(f.cam.frame, s, lights, N).cpu().numpy()
    fp2 = ForwardPlus(ctx, W, H, N, band=band)
    fp2.grid.fill_(-1); fp2.culled.fill_(-1)           # what k1_pack would write: untouched until pack()
    fp2.cull(f.cam.frame, lights, N, d, defer_pack=True)
    assert fp2.tile_order and fp2.use_tile_order
    deferred = fp2.shade(f.cam.frame, s, lights, N).cpu().numpy()
    assert int(fp2.grid[1].item()) == -1, "the pack has not run"
    np.testing.assert_array_equal(deferred, usual)
    assert_radiance_close(deferred, oracle_frame(f)[rows])
    fp2.pack()
    g2, i2 = fp2.lists_to_host()
    g1, i1 = fp.lists_to_host()
    np.testing.assert_array_equal(g2, g1)
    np.testing.assert_array_equal(i2[: 1 + int(i2[0])], i1[: 1 + int(i1[0])])


def test_tile_order_hint_with_fewer_lights_than_the_workspace_capacity(ctx):
    """the hint's address must not depend on the light count.  A ForwardPlus sized for three times the lights it culls
    (the pointer is looked up once, from the capacity) shades a band through the split blocks exactly as one sized for the count itself."""
    f = synth.make_frame("tiny")
    W, H, N = f.cam.width, f.cam.height, len(f.lights)
    band = host.band_for_rank(W, H, 1, 2)
    rows = slice(band.fbRowBegin, band.fbRowBegin + band.fbRowCount)
    lights = upload_lights(f.lights, ctx.device)
    d = torch.from_numpy(np.ascontiguousarray(f.depth[rows])).to(ctx.device)
    s = torch.from_numpy(np.ascontiguousarray(f.surface[:, rows])).to(ctx.device)
    outs = []
    for cap in (N, 3 * N + 17):
        fp = ForwardPlus(ctx, W, H, cap, band=band)
        assert fp.tile_order
        fp.cull(f.cam.frame, lights, N, d)
        g, idx = fp.lists_to_host()
        outs.append((fp.shade(f.cam.frame, s, lights, N).cpu().numpy(), g, idx))
    assert (outs[0][1][:, 1] >= 40).any(), "the band has tiles for the split blocks"
    np.testing.assert_array_equal(outs[0][1], outs[1][1])
    np.testing.assert_array_equal(outs[0][2][: 1 + int(outs[0][2][0])], outs[1][2][: 1 + int(outs[1][2][0])])
    np.testing.assert_array_equal(outs[0][0], outs[1][0])
    assert_radiance_close(outs[1][0], oracle_frame(f)[rows])


def test_split_tiles_on_a_band_of_the_4k_frame(ctx):
    """configs[2], band 3 of 8 (it crosses a light cluster: hundreds of tiles with 40..128 lights go to the split blocks): the split form and
    the one-block-per-tile form agree within the oracle tolerance everywhere, bit for bit on the short tiles, and an oracle-checked strip
    of the band is within tolerance."""
    f = synth.make_frame("C3")
    W, H, N = f.cam.width, f.cam.height, len(f.lights)
    band = host.band_for_rank(W, H, 3, 8)
    fp = ForwardPlus(ctx, W, H, N, band=band)
    rows = slice(band.fbRowBegin, band.fbRowBegin + band.fbRowCount)
    lights = upload_lights(f.lights, ctx.device)
    fp.cull(f.cam.frame, lights, N, torch.from_numpy(np.ascontiguousarray(f.depth[rows])).to(ctx.device))
    g, idx = fp.lists_to_host()
    s = torch.from_numpy(np.ascontiguousarray(f.surface[:, rows])).to(ctx.device)
    split = fp.shade(f.cam.frame, s, lights, N).cpu().numpy()
    fp.use_tile_order = False
    plain = fp.shade(f.cam.frame, s, lights, N).cpu().numpy()
    num = g[:, 1].reshape(-1, fp.Tx)
    assert (num >= 40).sum() > 100 and (num == 128).any()
    err = np.abs(split.astype(np.float64) - plain)
    assert (err <= RTOL * np.abs(plain)).all()   # (measured: 2e-6; the two forms differ by the association of one sum of non-negative terms)
    long_px = np.repeat(np.repeat(num[::-1] >= 40, 16, 0), 16, 1)[-band.fbRowCount:, :W]  # tile row 0 of the band = its bottom rows
    assert (split[~long_px] == plain[~long_px]).all() and err[long_px].max() > 0
    # the oracle on the band's first tile row (16 framebuffer rows)
    tr0 = band.tileRowBegin
    og, oi, _ = oracle.light_cull(f.cam.frame, W, H, f.lights, f.depth, tile_rows=(tr0, tr0 + 1))
    grid = np.zeros((240 * 135, 2), np.uint32); grid[:, 0] = 1
    grid[tr0 * 240:(tr0 + 1) * 240] = og
    r0, r1 = H - 16 * (tr0 + 1), H - 16 * tr0
    ref = oracle.shade(f.cam.frame, W, H, f.surface, f.lights, grid, oi, None, rows=(r0, r1))
    assert_radiance_close(split[r0 - band.fbRowBegin:r1 - band.fbRowBegin], ref[r0:r1])


def test_all_eight_bands_of_the_4k_frame_through_the_band_kernel_against_the_oracle(ctx):
    """The kernel every rank of an N > 1 run executes, held against the ORACLE over the ENTIRE frame at north_star's 1e-4: the
    eight bands of configs[2], each culled with the tile-order hint and shaded by k2_shade_band_p (prepared lights, long tiles through the split
    blocks whose four waves add their partial sums in a fixed order), stitched and compared with oracle_shade_threads' whole frame -- not with the
    one-block form and not at a relaxed tolerance.  (Standard.shader:427-436 sums the lights in list order; every term is >= 0 here, so a different
    association moves the sum by at most n * 2^-24 of itself.)"""
    f = synth.make_frame("C3")
    W, H, N = f.cam.width, f.cam.height, len(f.lights)
    lights = upload_lights(f.lights, ctx.device)
    prep = PreparedLights(ctx, lights, N)
    got = np.empty((H, W, 4), np.float32)
    grids, segs, long_tiles, base = [], [], 0, 0
    for r in range(8):
        band = host.band_for_rank(W, H, r, 8)
        rows = slice(band.fbRowBegin, band.fbRowBegin + band.fbRowCount)
        fp = ForwardPlus(ctx, W, H, N, band=band, prepared=prep)
        assert fp.tile_order and fp.use_tile_order
        fp.cull(f.cam.frame, lights, N, torch.from_numpy(np.ascontiguousarray(f.depth[rows])).to(ctx.device))
        got[rows] = fp.shade(f.cam.frame, torch.from_numpy(np.ascontiguousarray(f.surface[:, rows])).to(ctx.device), lights, N).cpu().numpy()
        g, idx = fp.lists_to_host()
        long_tiles += int((g[:, 1] >= 40).sum())
        g = g.copy(); g[:, 0] += base
        base += int(idx[0])
        grids.append(g); segs.append(idx[1:])
    assert long_tiles > 1000, "the split blocks had work in this frame"
    g_all = np.concatenate(grids)
    idx_all = np.concatenate([np.uint32([base])] + segs)
    assert_oracle_rows(f, got, oracle_tile_rows(135, [(16, 1), (50, 2), (84, 1), (118, 1)]), gpu_lists=(g_all, idx_all))


def test_shading_from_the_per_tile_lists_is_shading_from_the_canonical_buffers(ctx):
    """sailor_hip_shade_tile_lists (the default of ForwardPlus.shade since round 4) reads tile t's list at tileLists[128 t ..] / tileNum[t], the entry
    points of rounds 1-3 through lightsGrid / culledLights: the same entries in the same order, so the radiance is the same BITS -- plain and prepared
    lights, whole frame and a band with its split tiles."""
    f = synth.make_frame("tiny")
    W, H, N = f.cam.width, f.cam.height, len(f.lights)
    lights = upload_lights(f.lights, ctx.device)
    for prep in (None, PreparedLights(ctx, lights, N)):
        for band in (None, host.band_for_rank(W, H, 1, 2)):
            fp = ForwardPlus(ctx, W, H, N, band=band, prepared=prep)
            rows = slice(fp.band.fbRowBegin, fp.band.fbRowBegin + fp.band.fbRowCount)
            fp.cull(f.cam.frame, lights, N, torch.from_numpy(np.ascontiguousarray(f.depth[rows])).to(ctx.device))
            s = torch.from_numpy(np.ascontiguousarray(f.surface[:, rows])).to(ctx.device)
            assert fp.shade_from_tile_lists
            a = fp.shade(f.cam.frame, s, lights, N).cpu().numpy().copy()
            fp.shade_from_tile_lists = False
            b = fp.shade(f.cam.frame, s, lights, N).cpu().numpy()
            np.testing.assert_array_equal(a.view(np.uint32), b.view(np.uint32))
            assert_radiance_close(a, oracle_frame(f)[rows])


def test_every_light_reaches_every_pixel(ctx):
    """128 large lights over a small viewport: every tile list is full and every (pixel, light) is a lit pair -- 8 192 pairs per quadrant,
    i.e. dozens of pair windows per wave, each closed by the 128-pair / 4-per-pixel limits of the queue."""
    w, h = 64, 48
    cam = synth.make_camera(w, h)
    depth = synth.make_linear_depth(w, h, 9, d_min=40.0, d_max=60.0)
    lights = synth.make_lights(cam, depth, synth.LightSetConfig(count=160, radius_scale=400.0, spot_fraction=0.2), 9)
    surface = synth.make_surface(cam, depth, 9)
    g, idx, cnt = oracle.light_cull(cam.frame, w, h, lights, depth, want_counts=True)
    assert (g[:, 1] == 128).all() and (cnt > 128).all()
    ref = oracle.shade(cam.frame, w, h, surface, lights, g, idx)
    fp = ForwardPlus(ctx, w, h, len(lights))
    l = upload_lights(lights, ctx.device)
    fp.cull(cam.frame, l, len(lights), torch.from_numpy(depth).to(ctx.device))
    got = fp.shade(cam.frame, torch.from_numpy(surface).to(ctx.device), l, len(lights)).cpu().numpy()
    assert_radiance_close(got, ref)
    assert (ref[..., :3] > 0).all()


def test_short_forms_of_the_exact_square_root_and_reciprocal_on_this_device(ctx):
    """normalize() is v * (1 / sqrt(dot(v, v))) with a correctly rounded root and reciprocal; the kernels compute both in a handful of
    instructions (one Newton step on v_rsq_f32 / v_rcp_f32) instead of the compiler's expansions.  That these are sqrtf's and the IEEE
    division's bits is a property of the chip's approximation instructions, so it is checked on the chip the tests run on: every float of
    [2^-96, inf) for the root, every float with 2^-126 <= |x| <= 2^126 for the reciprocal (the library's own self-check, about a second)."""
    import ctypes
    scratch = torch.zeros(6, dtype=torch.int32, device=ctx.device)
    out = (ctypes.c_uint64 * 3)(7, 7, 7)
    rc = ctx._lib.sailor_hip_self_check_exact_math(ctx.handle, ctypes.c_void_p(scratch.data_ptr()), out)
    assert rc == 0
    assert (out[0], out[1], out[2]) == (0, 0, 0), f"square root differs on {out[0]} inputs, reciprocal on {out[1]}, staged-reciprocal quotient on {out[2]} of 2^30 pairs"


def test_lengths_outside_the_fast_square_root(ctx):
    """The kernel's square roots take a short form that is sqrtf's bits on [2^-96, inf) and fall back to sqrtf for anything else in the wave:
    a point light exactly AT a surface point (squared distance 0), a spot light whose direction is the zero vector (length 0, normalised to
    NaN) and one with a direction of 1e-25 (squared length a denormal), a light 1e-25 away from a surface point."""
    f = synth.make_frame("tiny")
    W, H, N = f.cam.width, f.cam.height, len(f.lights)
    lights = f.lights.copy()
    py, px = H // 2, W // 2
    p0 = f.surface[0, py, px, :3].copy(); p1 = f.surface[0, py + 3, px + 5, :3].copy()
    lights["type"][0] = host.LIGHT_POINT; lights["worldPosition"][0] = p0; lights["bounds"][0, 0] = 60.0
    lights["type"][1] = host.LIGHT_SPOT; lights["direction"][1] = 0.0
    lights["type"][2] = host.LIGHT_SPOT; lights["direction"][2] = (1e-25, 0.0, 0.0)
    lights["type"][3] = host.LIGHT_POINT; lights["worldPosition"][3] = p1 + np.float32(1e-25); lights["bounds"][3, 0] = 60.0
    assert np.array_equal(lights["worldPosition"][0], p0)
    g, idx, _ = oracle.light_cull(f.cam.frame, W, H, lights, f.depth)
    t = (H - 1 - py) // 16 * ((W + 15) // 16) + px // 16
    assert 0 in idx[g[t, 0]: g[t, 0] + g[t, 1]], "the light at the pixel is in the pixel's list"
    with np.errstate(all="ignore"):
        ref = oracle.shade(f.cam.frame, W, H, f.surface, lights, g, idx)
    fp = ForwardPlus(ctx, W, H, N)
    l = upload_lights(lights, ctx.device)
    fp.cull(f.cam.frame, l, N, torch.from_numpy(f.depth).to(ctx.device))
    got = fp.shade(f.cam.frame, torch.from_numpy(f.surface).to(ctx.device), l, N).cpu().numpy()
    np.testing.assert_array_equal(np.isnan(got), np.isnan(ref))
    np.testing.assert_array_equal(np.isposinf(got), np.isposinf(ref))
    np.testing.assert_array_equal(np.isneginf(got), np.isneginf(ref))
    fin = np.isfinite(ref)
    err = np.abs(got[fin].astype(np.float64) - ref[fin])
    assert (err <= RTOL * np.abs(ref[fin])).all(), (err / (np.abs(ref[fin]) + 1e-300)).max()


@pytest.mark.parametrize("prepared", [False, True])
def test_divisors_outside_the_staged_reciprocal(ctx, prepared):
    """dist / bounds.x and (theta - cutOff.y) / epsilon run on a reciprocal staged with the light when the divisor lies in [2^-40, 2^40], and
    on the IEEE division otherwise (such a light is shaded one lane per pixel, behind the pair queue): a point light with a radius of 1e15 (reaches everything, window ~1), one with 1e-15 at a
    surface point, a spot light whose inner and outer cone coincide (epsilon = 0: x / 0), one with epsilon < 0."""
    f = synth.make_frame("tiny")
    W, H, N = f.cam.width, f.cam.height, len(f.lights)
    lights = f.lights.copy()
    py, px = H // 2, W // 2
    lights["type"][0] = host.LIGHT_POINT; lights["bounds"][0, 0] = 1e15
    lights["type"][1] = host.LIGHT_POINT; lights["bounds"][1, 0] = 1e-15; lights["worldPosition"][1] = f.surface[0, py, px, :3]
    spots = np.nonzero(lights["type"] == host.LIGHT_SPOT)[0]
    assert len(spots) >= 2
    lights["cutOff"][spots[0], 0] = lights["cutOff"][spots[0], 1]
    lights["cutOff"][spots[1], 0] = lights["cutOff"][spots[1], 1] - np.float32(0.05)
    g, idx, _ = oracle.light_cull(f.cam.frame, W, H, lights, f.depth)
    assert (idx[1: 1 + int(g[:, 1].sum())] == 0).sum() >= len(g) // 2, "the 1e15 light is in most lists"
    with np.errstate(all="ignore"):
        ref = oracle.shade(f.cam.frame, W, H, f.surface, lights, g, idx)
    l = upload_lights(lights, ctx.device)
    fp = ForwardPlus(ctx, W, H, N, prepared=PreparedLights(ctx, l, N) if prepared else None)
    fp.cull(f.cam.frame, l, N, torch.from_numpy(f.depth).to(ctx.device))
    got = fp.shade(f.cam.frame, torch.from_numpy(f.surface).to(ctx.device), l, N).cpu().numpy()
    np.testing.assert_array_equal(np.isnan(got), np.isnan(ref))
    np.testing.assert_array_equal(np.isposinf(got), np.isposinf(ref))
    np.testing.assert_array_equal(np.isneginf(got), np.isneginf(ref))
    fin = np.isfinite(ref)
    err = np.abs(got[fin].astype(np.float64) - ref[fin])
    assert (err <= RTOL * np.abs(ref[fin])).all(), (err / (np.abs(ref[fin]) + 1e-300)).max()


def test_non_finite_terms_propagate_like_the_reference(ctx):
    """0 * inf must stay NaN: a zero window / facing factor only annihilates a FINITE product.  Pixels with roughness 0 (NdfGGX = 0/0) and
    lights with an infinite or NaN intensity may not be skipped by any of the conservative tests; the NaN / inf pattern of the radiance must be
    the oracle's, the finite rest within tolerance."""
    f = synth.make_frame("tiny")
    W, H, N = f.cam.width, f.cam.height, len(f.lights)
    surface = f.surface.copy()
    surface[1, ::7, ::5, 3] = 0.0           # roughness 0 on a lattice of pixels
    lights = f.lights.copy()
    lights["intensity"][3] = [np.inf, 1.0, 2.0]
    lights["intensity"][40] = [np.nan, 5.0, 5.0]
    lights["intensity"][77] = [-np.inf, 0.0, 1.0]
    g, idx, _ = oracle.light_cull(f.cam.frame, W, H, lights, f.depth)
    with np.errstate(all="ignore"):
        ref = oracle.shade(f.cam.frame, W, H, surface, lights, g, idx)
    assert np.isnan(ref).any() and np.isfinite(ref).any()
    fp = ForwardPlus(ctx, W, H, N)
    l = upload_lights(lights, ctx.device)
    fp.cull(f.cam.frame, l, N, torch.from_numpy(f.depth).to(ctx.device))
    got = fp.shade(f.cam.frame, torch.from_numpy(surface).to(ctx.device), l, N).cpu().numpy()
    np.testing.assert_array_equal(np.isnan(got), np.isnan(ref))
    np.testing.assert_array_equal(np.isposinf(got), np.isposinf(ref))
    np.testing.assert_array_equal(np.isneginf(got), np.isneginf(ref))
    fin = np.isfinite(ref)
    err = np.abs(got[fin].astype(np.float64) - ref[fin])
    assert (err <= RTOL * np.abs(ref[fin]) + atol_of(ref[..., :3])).all(), err.max()


@pytest.mark.parametrize("field", ["worldPosition", "direction", "attenuation", "cutOff", "bounds"])
def test_non_finite_light_parameters_other_than_the_intensity(ctx, field):
    """Found by scripts/fuzz_parity.py: a light whose POSITION is NaN makes every pixel of the tiles that list it NaN in the reference (the falloff
    is NaN, and NaN times the zero facing factor stays NaN) -- the conservative skips may only be taken for lights that are finite in every
    parameter that reaches the product, not just in the intensity."""
    f = synth.make_frame("tiny")
    W, H, N = f.cam.width, f.cam.height, len(f.lights)
    lights = f.lights.copy()
    point, spot = np.nonzero(lights["type"] == host.LIGHT_POINT)[0][:2], np.nonzero(lights["type"] == host.LIGHT_SPOT)[0][:2]
    for k, bad in zip((point[0], spot[0], point[1], spot[1]), (np.nan, np.nan, np.inf, -np.inf)):
        lights[field][k, 0] = bad
    g, idx, _ = oracle.light_cull(f.cam.frame, W, H, lights, f.depth)
    with np.errstate(all="ignore"):
        ref = oracle.shade(f.cam.frame, W, H, f.surface, lights, g, idx)
    fp = ForwardPlus(ctx, W, H, N)
    l = upload_lights(lights, ctx.device)
    fp.cull(f.cam.frame, l, N, torch.from_numpy(f.depth).to(ctx.device))
    assert_lists = fp.lists_to_host()
    np.testing.assert_array_equal(assert_lists[0], g)
    got = fp.shade(f.cam.frame, torch.from_numpy(f.surface).to(ctx.device), l, N).cpu().numpy()
    np.testing.assert_array_equal(np.isnan(got), np.isnan(ref))
    np.testing.assert_array_equal(np.isposinf(got), np.isposinf(ref))
    np.testing.assert_array_equal(np.isneginf(got), np.isneginf(ref))
    fin = np.isfinite(ref)
    err = np.abs(got[fin].astype(np.float64) - ref[fin])
    assert (err <= RTOL * np.abs(ref[fin])).all(), err.max()


def test_c5_shade_8k(ctx):
    """BASELINE.json configs[4] on one GPU: 7680 x 4320, 1 048 576 lights.  Full-size properties -- finite everywhere, alpha passed through, exact
    doubling under doubled intensities (power-of-two scaling commutes with every rounding) -- and the oracle on its own lists over twelve tile rows from top to bottom
    (two on a small host; the 8K cull is what costs: 480 tiles x 1 M lights per row)."""
    f = synth.make_frame("C5")
    W, H, N = f.cam.width, f.cam.height, len(f.lights)
    assert (W, H, N) == (7680, 4320, 1 << 20)
    a, fp = gpu_frame(ctx, f)
    assert np.isfinite(a).all()
    np.testing.assert_array_equal(a[..., 3], f.surface[0, ..., 3])
    g, idx = fp.lists_to_host()
    lit = a[..., :3].max(-1) > 0
    assert 0.2 < lit.mean() <= 1.0
    f2 = synth.Frame(f.name, f.cam, f.depth, f.lights.copy(), f.surface, None)
    f2.lights["intensity"] *= 2.0
    b, _ = gpu_frame(ctx, f2)
    np.testing.assert_array_equal(b[..., :3], 2.0 * a[..., :3])
    del b, f2
    spans = [(r, 1) for r in range(7, 270, 24)] if oracle.host_threads() >= 32 else [(131, 1), (17, 1)]
    assert_oracle_rows(f, a, oracle_tile_rows(270, spans, whole_from_threads=1 << 30), gpu_lists=(g, idx))


@pytest.mark.parametrize("prepared", [False, True])
@pytest.mark.parametrize("roughness, metallic", [(1.0, 0.0), (0.5, 0.0), (0.35, 0.6)])
def test_known_answer_point_light_at_normal_incidence_on_the_hip_path(ctx, roughness, metallic, prepared):
    """The closed form of tests/known_answers.py (worked out by hand from Standard.shader:286-340, no oracle involved) against the HIP path itself."""
    from known_answers import point_light_at_normal_incidence

    def hip_shade(frame, W, H, surface, lights, grid, idx):
        N = len(lights)
        l = upload_lights(lights, ctx.device)
        fp = ForwardPlus(ctx, W, H, N, prepared=PreparedLights(ctx, l, N) if prepared else None)
        fp.grid.copy_(torch.from_numpy(grid.astype(np.int32).reshape(-1)).to(ctx.device).view(fp.grid.dtype)[: fp.grid.numel()])
        fp.culled[: len(idx)].copy_(torch.from_numpy(idx.astype(np.int32)).to(ctx.device).view(fp.culled.dtype))
        return fp.shade(frame, torch.from_numpy(surface).to(ctx.device), l, N).cpu().numpy()

    got, want = point_light_at_normal_incidence(roughness, metallic, shade=hip_shade)
    np.testing.assert_allclose(got[:3], want, rtol=3e-5)
