"""K0+K1 parity (GPU): the HIP light cull, called through the C-ABI, against the CPU oracle -- bit-exact
`lightsGrid` + `culledLights` bytes, on seeded synthetic frames, edge cases and bands."""
import numpy as np
import pytest
import torch

from oracle import oracle
from sailor_amd import _lib, host, synth
from sailor_amd.forward_plus import ForwardPlus, upload_lights
from conftest import oracle_tile_rows

pytestmark = pytest.mark.gpu


def gpu_cull(ctx, cam, lights, depth, flags=_lib.CULL_DEFAULT, band=None):
    fp = ForwardPlus(ctx, cam.width, cam.height, max(len(lights), 1), band=band)
    b = fp.band
    d = torch.from_numpy(np.ascontiguousarray(depth[b.fbRowBegin:b.fbRowBegin + b.fbRowCount])).to(ctx.device)
    fp.cull(cam.frame, upload_lights(lights, ctx.device), len(lights), d, flags)
    return fp.lists_to_host()


def assert_lists_equal(got, ref_grid, ref_idx):
    g, idx = got
    total = int(ref_idx[0])
    assert int(idx[0]) == total
    np.testing.assert_array_equal(g, ref_grid)
    np.testing.assert_array_equal(idx[: 1 + total], ref_idx[: 1 + total])


# the three ways to the same lists: band masks by plane tests (default below 262 144 lights), no pre-filter at all, band masks from per-light
# band intervals found by bisection (the default above; forced here on the small frames)
ALL_PATHS = [_lib.CULL_DEFAULT, _lib.CULL_BRUTE_FORCE, _lib.CULL_INTERVAL_MASKS]


def frame(width, height, n_lights, seed=synth.SEED, **kw):
    cam = synth.make_camera(width, height)
    depth = synth.make_linear_depth(width, height, seed)
    lights = synth.make_lights(cam, depth, synth.LightSetConfig(count=n_lights, **kw), seed)
    return cam, depth, lights


@pytest.mark.parametrize("flags", ALL_PATHS)
def test_tiny_fixture_bit_exact(ctx, flags):
    f = synth.make_frame("tiny", with_surface=False)
    ref_g, ref_i, cnt = oracle.light_cull(f.cam.frame, f.cam.width, f.cam.height, f.lights, f.depth, want_counts=True)
    assert (cnt > oracle.CAND).any() and (cnt > oracle.KEEP).any(), "fixture must exercise the >128 and >196 regimes"
    assert_lists_equal(gpu_cull(ctx, f.cam, f.lights, f.depth, flags), ref_g, ref_i)


@pytest.mark.parametrize("flags", ALL_PATHS)
def test_c2_1080p_4096_point_lights_bit_exact(ctx, flags):
    """BASELINE.json configs[1]: 1080p, 16x16 tiles (last tile row partial), 4 096 point lights."""
    f = synth.make_frame("C2", with_surface=False)
    assert host.num_tiles(1920, 1080) == (120, 68)
    ref_g, ref_i, _ = oracle.light_cull(f.cam.frame, 1920, 1080, f.lights, f.depth)
    assert_lists_equal(gpu_cull(ctx, f.cam, f.lights, f.depth, flags), ref_g, ref_i)


@pytest.mark.parametrize("size", [(131, 77), (1000, 562), (16, 16), (17, 33), (640, 360)])
def test_ragged_viewports(ctx, size):
    """W, H not multiples of 16 (nor of 4): clamp-to-edge border replication, scalar depth path."""
    w, h = size
    cam, depth, lights = frame(w, h, 1500, radius_scale=6.0, spot_fraction=0.3, seed=7)
    ref_g, ref_i, _ = oracle.light_cull(cam.frame, w, h, lights, depth)
    for flags in ALL_PATHS:
        assert_lists_equal(gpu_cull(ctx, cam, lights, depth, flags), ref_g, ref_i)


def test_hierarchical_path_matches_oracle_and_brute_force(ctx):
    """16 384 lights at 960x540: the macro-tile pre-filter is active (several 2048-light chunks)."""
    cam, depth, lights = frame(960, 540, 16384, radius_scale=2.5, spot_fraction=0.25, cluster_lights=600, cluster_count=2, seed=11)
    ref_g, ref_i, cnt = oracle.light_cull(cam.frame, 960, 540, lights, depth, want_counts=True)
    assert (cnt > oracle.CAND).any()
    a = gpu_cull(ctx, cam, lights, depth, _lib.CULL_DEFAULT)
    b = gpu_cull(ctx, cam, lights, depth, _lib.CULL_BRUTE_FORCE)
    assert_lists_equal(a, ref_g, ref_i)
    assert_lists_equal(b, ref_g, ref_i)


def test_no_lights_and_single_light(ctx):
    cam, depth, lights = frame(128, 96, 4, radius_scale=4.0)
    g, idx = gpu_cull(ctx, cam, lights[:0], depth)
    assert int(idx[0]) == 0 and (g[:, 1] == 0).all() and (g[:, 0] == 1).all()
    one = lights[:1].copy()
    one["type"] = host.LIGHT_DIRECTIONAL
    ref_g, ref_i, _ = oracle.light_cull(cam.frame, 128, 96, one, depth)
    assert (ref_g[:, 1] == 1).all()
    assert_lists_equal(gpu_cull(ctx, cam, one, depth), ref_g, ref_i)


def test_directional_lights_fill_every_tile(ctx):
    """300 directional lights: every tile sees > 196 candidates, all with impact 0 (ties): selection keeps the stable order."""
    cam, depth, lights = frame(128, 96, 5000, radius_scale=3.0, seed=3)
    lights["type"][100:400] = host.LIGHT_DIRECTIONAL
    ref_g, ref_i, _ = oracle.light_cull(cam.frame, 128, 96, lights, depth)
    assert (ref_g[:, 1] == 128).all()
    for flags in ALL_PATHS:
        assert_lists_equal(gpu_cull(ctx, cam, lights, depth, flags), ref_g, ref_i)


def test_lights_behind_and_around_the_camera(ctx):
    """Spheres behind the eye or containing it: the macro pre-filter must not use side planes for them."""
    cam, depth, lights = frame(640, 360, 6000, radius_scale=2.0, seed=5)
    view = np.frombuffer(bytes(cam.frame.view), np.float32).reshape(4, 4)
    rng = np.random.default_rng(1)
    k = 3000  # > 2048 always-kept lights per 4x4-tile group: exercises the per-tile mask-walk fallback
    # world positions scattered in a box around the camera (0,150,0), big radii
    lights["worldPosition"][:k] = (rng.uniform(-400, 400, (k, 3)) + np.array([0, 150, 0])).astype(np.float32)
    lights["bounds"][:k] = rng.uniform(5, 600, (k, 1)).astype(np.float32)
    del view
    ref_g, ref_i, _ = oracle.light_cull(cam.frame, 640, 360, lights, depth)
    for flags in ALL_PATHS:
        assert_lists_equal(gpu_cull(ctx, cam, lights, depth, flags), ref_g, ref_i)


def test_deterministic_over_repeated_launches(ctx):
    """The reference kernel is racy by design; ours must return identical bytes every time."""
    f = synth.make_frame("tiny", with_surface=False)
    fp = ForwardPlus(ctx, f.cam.width, f.cam.height, len(f.lights))
    d = torch.from_numpy(f.depth).to(ctx.device)
    l = upload_lights(f.lights, ctx.device)
    first = None
    for _ in range(50):
        fp.cull(f.cam.frame, l, len(f.lights), d)
        got = fp.lists_to_host()
        if first is None:
            first = got
        else:
            np.testing.assert_array_equal(got[0], first[0])
            np.testing.assert_array_equal(got[1], first[1])


@pytest.mark.parametrize("world_size", [2, 3, 8])
def test_tile_row_bands_stitch_to_the_whole_frame(ctx, world_size):
    """per-band lists + prefix of band totals == the whole-frame canonical buffers."""
    cam, depth, lights = frame(1000, 562, 5000, radius_scale=3.0, spot_fraction=0.25, cluster_lights=300, seed=13)
    ref_g, ref_i, _ = oracle.light_cull(cam.frame, 1000, 562, lights, depth)
    grids, segs, base = [], [], 0
    for r in range(world_size):
        band = host.band_for_rank(1000, 562, r, world_size)
        g, idx = gpu_cull(ctx, cam, lights, depth, band=band)
        og, oi, _ = oracle.light_cull(cam.frame, 1000, 562, lights, depth, tile_rows=(band.tileRowBegin, band.tileRowEnd))
        assert_lists_equal((g, idx), og, oi)
        g = g.copy(); g[:, 0] += base
        grids.append(g); segs.append(idx[1:1 + int(idx[0])]); base += int(idx[0])
    np.testing.assert_array_equal(np.concatenate(grids), ref_g)
    np.testing.assert_array_equal(np.concatenate(segs), ref_i[1:1 + int(ref_i[0])])
    assert base == int(ref_i[0])


@pytest.mark.parametrize("n_lights, radius_scale, extra", [(5000, 3.0, 0), (4000, 40.0, 0), (4000, 0.5, _lib.CULL_INTERVAL_MASKS), (1100, 6.0, 0)])
@pytest.mark.parametrize("world_size", [2, 5])
def test_band_local_light_selection_gives_the_same_lists(ctx, world_size, n_lights, radius_scale, extra):
    """The band selection (k0_band_count + k0_band_scatter; the default on bands from 131 072 lights on, forced here by SAILOR_CULL_BAND_SELECT): the lights that can reach a band are
    compacted -- in ascending index, view space -- in front of the chain, which then runs on compact indices and translates them when a list leaves.
    Same lists bit for bit as without it and as the oracle's: few lights kept (small radii), ALL kept with an odd number of mask words (4 000 lights,
    huge radii: no pad word to zero), directional / NaN / behind-the-eye lights (always kept), several selection blocks, both mask forms."""
    cam, depth, lights = frame(1000, 562, n_lights, radius_scale=radius_scale, spot_fraction=0.25, cluster_lights=300, seed=17)
    lights["type"][[3, n_lights // 2, n_lights - 1]] = host.LIGHT_DIRECTIONAL
    lights["worldPosition"][7] = np.nan
    lights["worldPosition"][11] = (0.0, 150.0, 50.0); lights["bounds"][11, 0] = 400.0
    for r in range(world_size):
        band = host.band_for_rank(1000, 562, r, world_size)
        og, oi, _ = oracle.light_cull(cam.frame, 1000, 562, lights, depth, tile_rows=(band.tileRowBegin, band.tileRowEnd))
        with_sel = gpu_cull(ctx, cam, lights, depth, _lib.CULL_BAND_SELECT | extra, band=band)
        without = gpu_cull(ctx, cam, lights, depth, _lib.CULL_NO_BAND_SELECT | extra, band=band)
        assert_lists_equal(with_sel, og, oi)
        assert_lists_equal(without, og, oi)


def test_band_selection_with_the_preparation_folded_in(ctx):
    """... and with SAILOR_CULL_PREPARE_LIGHTS: the band selection derives the prepared views of ALL lights (they outlive the band), bit for bit
    sailor_hip_prepare_lights' own."""
    from sailor_amd.forward_plus import PreparedLights
    cam, depth, lights = frame(640, 400, 3000, radius_scale=2.0, spot_fraction=0.3, cluster_lights=400, seed=5)
    N = len(lights)
    dev = upload_lights(lights, ctx.device)
    ref = PreparedLights(ctx, dev, N)
    ctx.synchronize()
    want = [t.cpu().numpy().copy() for t in ref.views()]
    band = host.band_for_rank(640, 400, 1, 3)
    d = torch.from_numpy(np.ascontiguousarray(depth[band.fbRowBegin:band.fbRowBegin + band.fbRowCount])).to(ctx.device)
    mine = PreparedLights(ctx, dev, 0, capacity=N)
    mine.buffer.fill_(0x5A)
    fp = ForwardPlus(ctx, 640, 400, N, band=band, prepared=mine)
    fp.cull(cam.frame, dev, N, d, _lib.CULL_BAND_SELECT, prepare_lights=True)
    og, oi, _ = oracle.light_cull(cam.frame, 640, 400, lights, depth, tile_rows=(band.tileRowBegin, band.tileRowEnd))
    assert_lists_equal(fp.lists_to_host(), og, oi)
    for a, b in zip((t.cpu().numpy() for t in mine.views()), want):
        np.testing.assert_array_equal(a[:N].view(np.uint32), b[:N].view(np.uint32))


def test_band_selection_staging_only_the_selected_lights(ctx):
    """SAILOR_CULL_PREPARE_SELECTED (hosts that re-prepare every frame): the staged shade records only of the lights the band's selection keeps -- those are
    the bits sailor_hip_prepare_lights writes, the others are left alone, the 20-byte cull views are there for all -- and the band's shade is the ordinary one."""
    from sailor_amd.forward_plus import PreparedLights
    cam, depth, lights = frame(640, 400, 3000, radius_scale=0.6, spot_fraction=0.3, cluster_lights=400, seed=5)
    N = len(lights)
    surface = synth.make_surface(cam, depth, 5)
    dev = upload_lights(lights, ctx.device)
    ref = PreparedLights(ctx, dev, N)
    ctx.synchronize()
    want = [t.cpu().numpy().copy() for t in ref.views()]
    band = host.band_for_rank(640, 400, 1, 3)
    rows = slice(band.fbRowBegin, band.fbRowBegin + band.fbRowCount)
    d = torch.from_numpy(np.ascontiguousarray(depth[rows])).to(ctx.device)
    s = torch.from_numpy(np.ascontiguousarray(surface[:, rows])).to(ctx.device)
    fp0 = ForwardPlus(ctx, 640, 400, N, band=band, prepared=ref)
    fp0.cull(cam.frame, dev, N, d)
    usual = fp0.shade(cam.frame, s, dev, N).cpu().numpy().copy()
    mine = PreparedLights(ctx, dev, 0, capacity=N)
    mine.buffer.fill_(0x5A)
    fp = ForwardPlus(ctx, 640, 400, N, band=band, prepared=mine)
    fp.cull(cam.frame, dev, N, d, _lib.CULL_BAND_SELECT | _lib.CULL_PREPARE_SELECTED, prepare_lights=True)
    got = fp.shade(cam.frame, s, dev, N).cpu().numpy()
    np.testing.assert_array_equal(got.view(np.uint32), usual.view(np.uint32))
    g, idx = fp.lists_to_host()
    og, oi, _ = oracle.light_cull(cam.frame, 640, 400, lights, depth, tile_rows=(band.tileRowBegin, band.tileRowEnd))
    assert_lists_equal((g, idx), og, oi)
    listed = np.unique(idx[1:1 + int(idx[0])])
    pr, ty, st = (t.cpu().numpy() for t in mine.views())
    np.testing.assert_array_equal(pr[:N].view(np.uint32), want[0][:N].view(np.uint32))
    np.testing.assert_array_equal(ty[:N], want[1][:N])
    np.testing.assert_array_equal(st[listed].view(np.uint32), want[2][listed].view(np.uint32))          # every light a tile of the band lists is staged
    untouched = (st[:N].view(np.uint32).reshape(N, -1) == 0x5A5A5A5A).all(axis=1)
    assert 0.2 < untouched.mean() < 0.95, "most lights cannot reach a third of the frame with these radii: their staged records were not written"
    assert not untouched[listed].any()


def test_c3_4k_65536_lights_default_equals_brute_force_and_invariants(ctx):
    """BASELINE.json configs[2] at full size: too big for the scalar oracle in seconds, so the hierarchical path is checked
    against the brute-force HIP walk (itself oracle-checked above) plus size-independent invariants, and -- round 3 -- the ENTIRE frame
    against the oracle running on all host threads (on a host with fewer than the default threads it is still the whole frame, only slower)."""
    f = synth.make_frame("C3", with_surface=False)
    W, H, N = 3840, 2160, 65536
    a = gpu_cull(ctx, f.cam, f.lights, f.depth, _lib.CULL_DEFAULT)
    b = gpu_cull(ctx, f.cam, f.lights, f.depth, _lib.CULL_BRUTE_FORCE)
    np.testing.assert_array_equal(a[0], b[0])
    np.testing.assert_array_equal(a[1], b[1])
    c = gpu_cull(ctx, f.cam, f.lights, f.depth, _lib.CULL_INTERVAL_MASKS)  # 94 bands in one block: the large-set mask builder on the 4K frame
    np.testing.assert_array_equal(a[0], c[0])
    np.testing.assert_array_equal(a[1], c[1])
    g, idx = a
    assert g.shape == (240 * 135, 2)
    num = g[:, 1].astype(np.int64)
    assert num.max() <= 128 and int(idx[0]) == num.sum()
    np.testing.assert_array_equal(g[:, 0].astype(np.int64), 1 + np.concatenate([[0], np.cumsum(num)[:-1]]))  # Appendix A step 6
    assert idx[1:].max() < N
    # no light appears twice in a tile
    for t in np.random.default_rng(0).choice(len(g), 500, replace=False):
        seg = idx[g[t, 0]: g[t, 0] + g[t, 1]]
        assert len(np.unique(seg)) == len(seg)
    mean = num.mean()
    assert 16 <= mean <= 32, f"frozen generator: mean list length {mean}"
    # the WHOLE frame against the oracle (32 400 tiles x 65 536 lights = 2.1 G sphere tests on all host threads: seconds)
    for r0, rows in oracle_tile_rows(135, [(60, 3), (7, 2)]):
        og, oi, cnt = oracle.light_cull(f.cam.frame, W, H, f.lights, f.depth, tile_rows=(r0, r0 + rows), want_counts=True, threads=oracle.host_threads())
        t0 = r0 * 240
        np.testing.assert_array_equal(g[t0:t0 + rows * 240, 1], og[:, 1])
        if rows == 135:   # the whole frame: offsets and the compact index array are the oracle's, word for word
            np.testing.assert_array_equal(g, og)
            np.testing.assert_array_equal(idx[: 1 + int(oi[0])], oi[: 1 + int(oi[0])])
            assert (cnt > 196).any() and (cnt > 128).sum() > 100, "the frame exercises the candidate cap and the nearest-128 selection"
        else:
            for t in range(rows * 240):
                np.testing.assert_array_equal(idx[g[t0 + t, 0]: g[t0 + t, 0] + g[t0 + t, 1]], oi[og[t, 0]: og[t, 0] + og[t, 1]])


@pytest.mark.parametrize("flags", ALL_PATHS)
def test_nan_impacts_follow_the_literal_bubble_sort(ctx, flags):
    """Tiles with nothing drawn (linear depth +inf) get a NaN frustum centre, i.e. NaN impacts: no total order, the shader's
    compare-and-swap (ComputeLightCulling.shader:207) is false next to a NaN.  (a) sky tiles: every impact is NaN or a
    directional 0 -> the sort is a no-op; (b) lights with a NaN position among finite ones -> NaNs act as walls, finite runs
    are bubbled between them.  Both must equal the oracle, whose closed form defers to the literal sort when a NaN is present."""
    f = synth.make_frame("tiny", with_surface=False)
    W, H = f.cam.width, f.cam.height
    depth = f.depth.copy()
    sky = synth.make_raw_depth(depth, 1.0, sky_fraction=0.5) == 0
    depth[sky] = np.inf
    ref_g, ref_i, cnt = oracle.light_cull(f.cam.frame, W, H, f.lights, depth, want_counts=True)
    sky_tiles = sky.reshape(H // 16, 16, W // 16, 16).any(axis=(1, 3)).reshape(-1)
    assert ((cnt > oracle.KEEP) & sky_tiles).any(), "need a sky tile that runs the selection"
    assert_lists_equal(gpu_cull(ctx, f.cam, f.lights, depth, flags), ref_g, ref_i)

    lights = f.lights.copy()
    lights["worldPosition"][::7] = np.nan
    ref_g, ref_i, cnt = oracle.light_cull(f.cam.frame, W, H, lights, f.depth, want_counts=True)
    assert (cnt > oracle.KEEP).any()
    lit_g, lit_i, _ = oracle.light_cull(f.cam.frame, W, H, lights, f.depth, literal_select=True)
    np.testing.assert_array_equal(ref_i[: 1 + int(ref_i[0])], lit_i[: 1 + int(lit_i[0])])
    assert_lists_equal(gpu_cull(ctx, f.cam, lights, f.depth, flags), ref_g, ref_i)


def test_c5_8k_1m_lights_invariants_and_an_oracle_tile_row(ctx):
    """BASELINE.json configs[4] at full size (7680 x 4320, 1 048 576 lights -- sixteen times the reference's 65 535-light cap): list
    invariants over all 129 600 tiles, and the lists of an eighth of the frame (seventeen two-row spans, top to bottom) against the oracle."""
    f = synth.make_frame("C5", with_surface=False)
    W, H, N = 7680, 4320, 1 << 20
    g, idx = gpu_cull(ctx, f.cam, f.lights, f.depth, _lib.CULL_DEFAULT)
    assert g.shape == (480 * 270, 2)
    num = g[:, 1].astype(np.int64)
    assert num.max() <= 128 and int(idx[0]) == num.sum() and num.sum() > 0
    np.testing.assert_array_equal(g[:, 0].astype(np.int64), 1 + np.concatenate([[0], np.cumsum(num)[:-1]]))
    assert idx[1:1 + int(idx[0])].max() < N
    for t in np.random.default_rng(1).choice(len(g), 500, replace=False):
        seg = idx[g[t, 0]: g[t, 0] + g[t, 1]]
        assert len(np.unique(seg)) == len(seg)
    # the oracle: an eighth of the frame in seventeen two-row spans from top to bottom (the whole frame is 136 G sphere tests: minutes even on the
    # GPU box's 256 threads; SAILOR_ORACLE_ROWS=0:270 runs it), four rows on a small host
    spans = [(r, 2) for r in range(3, 270, 16)] if oracle.host_threads() >= 32 else [(131, 1), (5, 1), (200, 1), (268, 1)]
    for r0, rows in oracle_tile_rows(270, spans, whole_from_threads=1 << 30):
        og, oi, _ = oracle.light_cull(f.cam.frame, W, H, f.lights, f.depth, tile_rows=(r0, r0 + rows), threads=oracle.host_threads())
        t0 = r0 * 480
        np.testing.assert_array_equal(g[t0:t0 + rows * 480, 1], og[:, 1])
        if rows == 270:
            np.testing.assert_array_equal(g, og)
            np.testing.assert_array_equal(idx[: 1 + int(oi[0])], oi[: 1 + int(oi[0])])
        else:
            for t in range(rows * 480):
                np.testing.assert_array_equal(idx[g[t0 + t, 0]: g[t0 + t, 0] + g[t0 + t, 1]], oi[og[t, 0]: og[t, 0] + og[t, 1]])


def test_large_light_set_with_directional_lights_on_a_small_frame(ctx):
    """300 000 lights on a 640 x 360 frame: the kernels of the large-set path (band intervals + bit transpose in k01_prepare, k1_group_lists_wide
    in its bounds-checked form: 4 688 mask words are not a multiple of 512) against the oracle over the WHOLE frame, with and without
    directional lights on one workspace (the "some light is directional" flag that lets the list builder skip the directional words is set by
    one cull and cleared at its end)."""
    W, H, N = 640, 360, 300_000
    cam, depth, lights = frame(W, H, N, radius_scale=0.35, spot_fraction=0.3, seed=11)
    with_dir = lights.copy()
    with_dir["type"][[5, 70_001, 299_999]] = host.LIGHT_DIRECTIONAL
    fp = ForwardPlus(ctx, W, H, N)
    d = torch.from_numpy(np.ascontiguousarray(depth)).to(ctx.device)
    refs = {}
    for name, ls in (("directional", with_dir), ("none", lights), ("directional", with_dir)):
        if name not in refs:
            refs[name] = oracle.light_cull(cam.frame, W, H, ls, depth)
        fp.cull(cam.frame, upload_lights(ls, ctx.device), N, d)
        assert_lists_equal(fp.lists_to_host(), refs[name][0], refs[name][1])
    assert refs["directional"][0][:, 1].min() >= 3 and refs["none"][0][:, 1].mean() > 4


def _cluster_frame():
    """a small frame with light clusters: groups of more than 512 candidates (their tiles take k1_tile_cull's block-per-tile path) and tiles whose
    196 -> 128 selection runs"""
    f = synth.make_frame("tiny", with_surface=False, width=320, height=208,
                         lights=synth.LightSetConfig(count=6000, spot_fraction=0.3, radius_scale=4.0, cluster_lights=2400, cluster_count=2, cluster_spread=1.5,
                                                     cluster_radius=(0.5, 1.2)))
    return f


@pytest.mark.parametrize("band_of", [None, (1, 2)])
def test_cluster_tiles_take_a_block_each_and_give_the_same_lists(ctx, band_of):
    """Round 4: the tiles of a light cluster (a 4x4-tile group with more than 512 candidates) are culled by a whole block each -- four waves share the
    candidates and the selection's rank.  Same candidate order, same impacts, same rank rule: the lists are the brute-force walk's and the oracle's."""
    f = _cluster_frame()
    W, H = f.cam.width, f.cam.height
    band = None if band_of is None else host.band_for_rank(W, H, *band_of)
    fp = ForwardPlus(ctx, W, H, len(f.lights), band=band)
    b = fp.band
    d = torch.from_numpy(np.ascontiguousarray(f.depth[b.fbRowBegin:b.fbRowBegin + b.fbRowCount])).to(ctx.device)
    l = upload_lights(f.lights, ctx.device)
    fp.cull(f.cam.frame, l, len(f.lights), d)
    diag = fp.cull_diagnostics(len(f.lights))
    assert diag["group_list_max"] > 512, "the frame has listed light clusters"
    got = fp.lists_to_host()
    ref_g, ref_i, cnt = oracle.light_cull(f.cam.frame, W, H, f.lights, f.depth, tile_rows=(b.tileRowBegin, b.tileRowEnd), want_counts=True)
    assert (cnt > oracle.CAND).any() and (ref_g[:, 1] == 128).sum() > 8, "selections ran"
    assert_lists_equal(got, ref_g, ref_i)
    fp.cull(f.cam.frame, l, len(f.lights), d, _lib.CULL_BRUTE_FORCE)
    assert_lists_equal(fp.lists_to_host(), ref_g, ref_i)


def test_per_tile_lists_and_the_deferred_pack(ctx):
    """sailor_hip_light_cull_tile_lists / SAILOR_CULL_DEFER_PACK / sailor_hip_light_cull_pack: after a deferred cull the canonical buffers are untouched
    and every tile's list sits in its own 128-entry slot of the workspace, the same entries in the same order; the pack -- recorded on ANOTHER stream,
    as the frame pipeline does -- then writes lightsGrid / culledLights bit for bit what the undeferred call writes (the oracle's)."""
    from sailor_amd.forward_plus import HipContext
    f = _cluster_frame()
    W, H, N = f.cam.width, f.cam.height, len(f.lights)
    ref_g, ref_i, _ = oracle.light_cull(f.cam.frame, W, H, f.lights, f.depth)
    fp = ForwardPlus(ctx, W, H, N)
    d = torch.from_numpy(f.depth).to(ctx.device)
    l = upload_lights(f.lights, ctx.device)
    fp.grid.fill_(-7); fp.culled.fill_(-7)
    fp.cull(f.cam.frame, l, N, d, defer_pack=True)
    ctx.synchronize()
    assert (fp.grid == -7).all() and (fp.culled == -7).all(), "a deferred cull does not write the canonical buffers"
    base = fp.workspace.data_ptr()
    T = fp.band_tiles
    num = fp.workspace[fp.tile_num - base: fp.tile_num - base + 4 * T].view(torch.int32).cpu().numpy().view(np.uint32)
    lists = fp.workspace[fp.tile_lists - base: fp.tile_lists - base + 4 * 128 * T].view(torch.int32).cpu().numpy().view(np.uint32).reshape(T, 128)
    np.testing.assert_array_equal(num, ref_g[:, 1])
    for t in np.random.default_rng(3).choice(T, 64, replace=False).tolist() + [int(np.argmax(num))]:
        np.testing.assert_array_equal(lists[t, : num[t]], ref_i[ref_g[t, 0]: ref_g[t, 0] + num[t]])
    side = torch.cuda.Stream(device=ctx.device)
    side.wait_stream(torch.cuda.current_stream())
    ctx2 = HipContext(ctx.device, stream=side)
    fp.pack(ctx2)
    ctx2.synchronize()
    assert_lists_equal(fp.lists_to_host(), ref_g, ref_i)


def test_dispatch_packet_timing_of_the_cull_chain(ctx):
    """sailor_hip_context_time_launches: the four kernels of one cull each carry an event pair on their own dispatch packet; the durations are those
    of the kernels (positive, and together no longer than the call as seen by events around it), and an unarmed call is launched the ordinary way."""
    f = synth.make_frame("C2", with_surface=False)
    W, H, N = f.cam.width, f.cam.height, len(f.lights)
    fp = ForwardPlus(ctx, W, H, N)
    d = torch.from_numpy(f.depth).to(ctx.device)
    l = upload_lights(f.lights, ctx.device)
    fp.cull(f.cam.frame, l, N, d)
    ref = fp.lists_to_host()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    ctx.time_launches(0, 4)
    a.record(); fp.cull(f.cam.frame, l, N, d); b.record()
    ctx.synchronize()
    ms = [ctx.timed_launch_ms(i) for i in range(4)]
    assert all(0.0005 < m < 5.0 for m in ms), ms
    assert sum(ms) <= a.elapsed_time(b) * 1.05
    got = fp.lists_to_host()
    np.testing.assert_array_equal(got[0], ref[0]); np.testing.assert_array_equal(got[1], ref[1])
    fp.cull(f.cam.frame, l, N, d)   # all four slots are used up: ordinary launches again
    ctx.synchronize()


def test_the_4x4_patch_form_of_the_wide_list_builder_gives_the_same_lists():
    """Round 6: k1_group_lists_wide16 (SAILOR_CULL_WIDE16=1: a block per 4 x 4 patch of groups, every mask row fetched once for sixteen groups) is off by
    default -- it halves the kernel's traffic and shortens nothing -- and stays under the oracle: the large-light-set tests of this file and the band tests
    of test_split_paths_gpu.py (the form behind the band selection: a selected count that is no multiple of 256 words) in a child process with the switch on
    (the library reads it once per process)."""
    import os
    import subprocess
    import sys
    env = dict(os.environ, SAILOR_CULL_WIDE16="1")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    p = subprocess.run([sys.executable, "-m", "pytest", "tests/test_light_cull_gpu.py", "tests/test_split_paths_gpu.py", "-m", "gpu", "-q", "-x",
                        "-k", "large_light_set or 300_000_lights"], cwd=root, env=env, capture_output=True, text=True, timeout=1500)
    assert p.returncode == 0, p.stdout[-3000:] + p.stderr[-1000:]
    assert " passed" in p.stdout and "no tests ran" not in p.stdout, p.stdout[-500:]
