"""The product paths a rank of a split frame runs at the sizes they exist for ("configs untested"), through the C-ABI, against the oracle:

  * C5 (8K, 1 048 576 lights), a band of an 8-way split with DEFAULT flags: the band selection (k0_band_count + k0_band_scatter, on by itself from 131 072 lights) -> k01_prepare on the
    selected lights -> k1_group_lists_wide<false>(..., selCount) -> k1_tile_cull<*, HINT, SEL>; static lights, SAILOR_CULL_PREPARE_LIGHTS, and
    SAILOR_CULL_PREPARE_SELECTED followed by the band's shade;
  * the same chain on a small frame under 300 000 lights (seconds: the wide list builder with a selected count and a word count that is no multiple
    of 512);
  * C4 (4K + four shadow cascades), all eight bands through k2_shade_band_csm* with prepared lights, stitched, the ENTIRE frame at 1e-4;
  * the band selection beside a resident shade (the two-frames-in-flight pipeline it runs in).

Which kernels a call launched is the library's own record (sailor_hip_context_launch_log), not a guess from the flags."""
import os

import numpy as np
import pytest
import torch

from oracle import oracle
from sailor_amd import _lib, host, synth
from sailor_amd.forward_plus import ForwardPlus, HipContext, PreparedLights, upload_lights, upload_shadow_maps
from test_light_cull_gpu import assert_lists_equal, frame
from test_shade_gpu import assert_oracle_rows, assert_radiance_close
from conftest import oracle_tile_rows

pytestmark = pytest.mark.gpu

SELECT_CHAIN_WIDE = ["k0_band_count", "k0_band_scatter", "k01_prepare", "k1_group_lists_wide", "k1_tile_cull", "k1_pack"]


def band_rows(band):
    return slice(band.fbRowBegin, band.fbRowBegin + band.fbRowCount)


def assert_band_rows_equal_oracle(got, band, Tx, local_rows, refs):
    """the lists of the band's tile rows `local_rows` (band-local indices) in got = (grid, indices) against refs[row] = the oracle's (grid, indices) of
    that ONE tile row: lengths and entries word for word"""
    g, idx = got
    for r in local_rows:
        og, oi = refs[r]
        t0 = r * Tx
        np.testing.assert_array_equal(g[t0:t0 + Tx, 1], og[:, 1])
        for t in range(Tx):
            np.testing.assert_array_equal(idx[g[t0 + t, 0]: g[t0 + t, 0] + g[t0 + t, 1]], oi[og[t, 0]: og[t, 0] + og[t, 1]])


@pytest.mark.parametrize("band_index", [3, 0])
def test_c5_band_of_an_eight_way_split_runs_the_selection_chain_and_gives_the_oracles_lists(ctx, band_index):
    """BASELINE.json configs[4] as ONE RANK of its 8-way split sees it (a middle band and the edge band 0), default flags."""
    f = synth.make_frame("C5", with_surface=False)
    W, H, N = f.cam.width, f.cam.height, len(f.lights)
    assert (W, H, N) == (7680, 4320, 1 << 20)
    Tx, Ty = host.num_tiles(W, H)
    band = host.band_for_rank(W, H, band_index, 8)
    nrows = band.tileRowEnd - band.tileRowBegin
    rows = band_rows(band)
    lights = upload_lights(f.lights, ctx.device)
    d = torch.from_numpy(np.ascontiguousarray(f.depth[rows])).to(ctx.device)
    # the oracle: six tile rows of the band (first two, two in the middle, last two), one row at a time on all host threads
    local = sorted({0, 1, nrows // 2, nrows // 2 + 1, nrows - 2, nrows - 1}) if oracle.host_threads() >= 32 else [0, nrows // 2, nrows - 1]
    if os.environ.get("SAILOR_ORACLE_ROWS"):   # names frame rows (conftest.oracle_tile_rows): those of them that lie in the band
        local = sorted({r - band.tileRowBegin for r0, n in oracle_tile_rows(Ty, []) for r in range(r0, r0 + n) if band.tileRowBegin <= r < band.tileRowEnd}) or local
    refs = {}
    for r in local:
        og, oi, _ = oracle.light_cull(f.cam.frame, W, H, f.lights, f.depth, tile_rows=(band.tileRowBegin + r, band.tileRowBegin + r + 1), threads=oracle.host_threads())
        refs[r] = (og, oi)
    assert sum(int(refs[r][1][0]) for r in local) > 0

    # (i) static lights: the prepared views exist, the chain reads them
    prep = PreparedLights(ctx, lights, N)
    fp = ForwardPlus(ctx, W, H, N, band=band, prepared=prep)
    assert not fp.tile_order, "16 320 tiles under a million lights: short lists, no long tiles to split -- the band takes the whole frame's launch form"
    names = ctx.launches_of(lambda: fp.cull(f.cam.frame, lights, N, d))
    assert names == SELECT_CHAIN_WIDE, names
    M, lm = fp.band_selection(N)
    assert 0 < M < N // 2 and len(lm) == M and (np.diff(lm.astype(np.int64)) > 0).all() and lm[-1] < N, "a band's own light set: fewer lights, ascending"
    static = fp.lists_to_host()
    listed = np.unique(static[1][1:1 + int(static[1][0])])
    assert np.isin(listed, lm).all(), "every listed light was selected"
    assert_band_rows_equal_oracle(static, band, Tx, local, refs)
    want_views = [t.cpu().numpy().copy() for t in prep.views()]

    # (ii) every light dirty: the selection kernel derives the prepared views of ALL lights on the way (they outlive the band)
    mine = PreparedLights(ctx, lights, 0, capacity=N)
    mine.buffer.fill_(0x5A)
    fp2 = ForwardPlus(ctx, W, H, N, band=band, prepared=mine)
    names = ctx.launches_of(lambda: fp2.cull(f.cam.frame, lights, N, d, prepare_lights=True))
    assert names == SELECT_CHAIN_WIDE, names
    dyn = fp2.lists_to_host()
    np.testing.assert_array_equal(dyn[0], static[0]); np.testing.assert_array_equal(dyn[1], static[1])
    for a, b in zip((t.cpu().numpy() for t in mine.views()), want_views):
        np.testing.assert_array_equal(a[:N].view(np.uint32), b[:N].view(np.uint32))

    # (iii) ... staging only the selected lights' shade records (hosts that re-prepare every frame), then the band's shade on the checked rows
    del mine, fp2
    sel = PreparedLights(ctx, lights, 0, capacity=N)
    sel.buffer.fill_(0x5A)
    fp3 = ForwardPlus(ctx, W, H, N, band=band, prepared=sel)
    names = ctx.launches_of(lambda: fp3.cull(f.cam.frame, lights, N, d, _lib.CULL_PREPARE_SELECTED, prepare_lights=True))
    assert names == SELECT_CHAIN_WIDE, names
    got = fp3.lists_to_host()
    np.testing.assert_array_equal(got[0], static[0]); np.testing.assert_array_equal(got[1], static[1])
    st = sel.views()[2].cpu().numpy()
    np.testing.assert_array_equal(st[listed].view(np.uint32), want_views[2][listed].view(np.uint32))
    untouched = (st[:N].view(np.uint32).reshape(N, -1) == 0x5A5A5A5A).all(axis=1)
    assert untouched.mean() > 0.5 and not untouched[lm].any(), "the staged records of the selected lights, and of those only"
    surface = synth.make_surface(f.cam, f.depth, row_begin=rows.start, row_end=rows.stop)
    ds = torch.from_numpy(surface).to(ctx.device)
    names = ctx.launches_of(lambda: fp3.shade(f.cam.frame, ds, lights, N))
    assert len(names) == 1 and names[0].startswith("k2_shade") and "band" not in names[0] and names[0].endswith("_pt"), names
    rad = fp3.radiance.cpu().numpy()
    assert np.isfinite(rad).all()
    planes = np.zeros((3, H, W, 4), np.float32)   # (the oracle addresses rows of full-frame planes; untouched pages stay virtual)
    planes[:, rows] = surface
    for r in local:
        og, oi = refs[r]
        tr = band.tileRowBegin + r
        grid = np.zeros((Tx * Ty, 2), np.uint32); grid[:, 0] = 1
        grid[tr * Tx:(tr + 1) * Tx] = og
        r0, r1 = H - 16 * (tr + 1), H - 16 * tr
        ref = oracle.shade(f.cam.frame, W, H, planes, f.lights, grid, oi, None, rows=(r0, r1), threads=oracle.host_threads())
        assert_radiance_close(rad[r0 - rows.start:r1 - rows.start], ref[r0:r1])


@pytest.mark.parametrize("tile_rows", [(5, 13), (0, 4), (19, 23)])
def test_bands_under_300_000_lights_take_the_wide_list_builder_behind_the_selection(ctx, tile_rows):
    """The C5 rank chain in seconds: 300 000 lights on 640 x 360 (4 688 mask words: the wide list builder, in its bounds-checked form; from 131 072
    lights on a band selects its own light set by default), a middle band, the edge bands, with and without directional lights -- the WHOLE band
    against the oracle, and against the same band culled without the selection."""
    W, H, N = 640, 360, 300_000
    cam, depth, lights = frame(W, H, N, radius_scale=0.35, spot_fraction=0.3, seed=11)
    with_dir = lights.copy()
    with_dir["type"][[5, 70_001, 299_999]] = host.LIGHT_DIRECTIONAL
    band = host.band_from_tile_rows(W, H, *tile_rows)
    d = torch.from_numpy(np.ascontiguousarray(depth[band_rows(band)])).to(ctx.device)
    fp = ForwardPlus(ctx, W, H, N, band=band)
    for ls in (with_dir, lights, with_dir):   # (one workspace: the "some light is directional" flag is set by one cull and cleared at its end)
        dl = upload_lights(ls, ctx.device)
        names = ctx.launches_of(lambda: fp.cull(cam.frame, dl, N, d))
        assert names == SELECT_CHAIN_WIDE, names
        M, lm = fp.band_selection(N)
        assert 0 < M < N and M % 512 != 0
        og, oi, _ = oracle.light_cull(cam.frame, W, H, ls, depth, tile_rows=tile_rows)
        assert_lists_equal(fp.lists_to_host(), og, oi)
        names = ctx.launches_of(lambda: fp.cull(cam.frame, dl, N, d, _lib.CULL_NO_BAND_SELECT))
        assert names == SELECT_CHAIN_WIDE[2:], names
        assert_lists_equal(fp.lists_to_host(), og, oi)
    assert oi[0] > 0


def test_all_eight_bands_of_c4_through_the_shadowed_band_kernels_against_the_oracle(ctx):
    """The kernel every rank of BASELINE.json configs[3]'s 8-way split executes -- k2_shade_band_csm_pt: the band kernel's shadowed twin, prepared
    lights, the lists from the cull's per-tile slots, long tiles through the split blocks -- over all eight bands, stitched, the ENTIRE frame at
    1e-4 against oracle_shade_threads with the four shadow maps (the twin of test_all_eight_bands_of_the_4k_frame_... for the plain kernel)."""
    f = synth.make_frame("C4")
    W, H, N = f.cam.width, f.cam.height, len(f.lights)
    lights = upload_lights(f.lights, ctx.device)
    prep = PreparedLights(ctx, lights, N)
    csm, keep = upload_shadow_maps(f.shadows, ctx.device)
    got = np.empty((H, W, 4), np.float32)
    grids, segs, long_tiles, base = [], [], 0, 0
    for r in range(8):
        band = host.band_for_rank(W, H, r, 8)
        rows = band_rows(band)
        fp = ForwardPlus(ctx, W, H, N, band=band, prepared=prep)
        assert fp.tile_order and fp.use_tile_order and fp.shade_from_tile_lists
        fp.cull(f.cam.frame, lights, N, torch.from_numpy(np.ascontiguousarray(f.depth[rows])).to(ctx.device))
        s = torch.from_numpy(np.ascontiguousarray(f.surface[:, rows])).to(ctx.device)
        names = ctx.launches_of(lambda: fp.shade(f.cam.frame, s, lights, N, csm))
        assert names == ["k2_shade_band_csm_pt"], names
        got[rows] = fp.radiance.cpu().numpy()
        g, idx = fp.lists_to_host()
        long_tiles += int((g[:, 1] >= 40).sum())
        g = g.copy(); g[:, 0] += base
        base += int(idx[0])
        grids.append(g); segs.append(idx[1:])
    assert long_tiles > 1000, "the split blocks had work in this frame"
    g_all = np.concatenate(grids)
    idx_all = np.concatenate([np.uint32([base])] + segs)
    desc, keep2 = oracle.make_csm(f.shadows.lights_matrices, f.shadows.maps)
    assert_oracle_rows(f, got, oracle_tile_rows(135, [(16, 1), (50, 1), (84, 1), (118, 1)]), csm_desc=desc, gpu_lists=(g_all, idx_all))


def test_band_selection_keeps_its_order_beside_a_resident_shade(ctx):
    """The band selection is two launches (count; scatter) and no block of either waits for another: the compaction's order cannot depend on how the
    hardware starts blocks or on how many fit beside another kernel (round 4's one-launch form polled the blocks in front of it).  The situation it
    runs in -- the next frame's chain on a second stream beside the previous frame's band shade, which holds most wave slots -- 200 times on a C5
    band: the same selection, the same lightMap, the same lists every time."""
    f = synth.make_frame("C5", with_surface=False)
    W, H, N = f.cam.width, f.cam.height, len(f.lights)
    band = host.band_for_rank(W, H, 3, 8)
    rows = band_rows(band)
    lights = upload_lights(f.lights, ctx.device)
    prep = PreparedLights(ctx, lights, N)
    d = torch.from_numpy(np.ascontiguousarray(f.depth[rows])).to(ctx.device)
    s = torch.from_numpy(synth.make_surface(f.cam, f.depth, row_begin=rows.start, row_end=rows.stop)).to(ctx.device)
    shader = ForwardPlus(ctx, W, H, N, band=band, prepared=prep)          # the previous frame: its lists exist, its shade runs on the test's stream
    shader.cull(f.cam.frame, lights, N, d)
    side = torch.cuda.Stream(device=ctx.device)
    ctx2 = HipContext(ctx.device, stream=side)
    culler = ForwardPlus(ctx, W, H, N, band=band, prepared=prep)          # the next frame: its chain is recorded on the second stream
    names = ctx2.launches_of(lambda: culler.cull(f.cam.frame, lights, N, d, ctx=ctx2))
    assert names == SELECT_CHAIN_WIDE, names
    ctx2.synchronize()
    M0, lm0 = culler.band_selection(N)
    ref = culler.lists_to_host()
    ws0 = culler.workspace.clone()
    a, b = _lightmap_extent(culler, N, M0)
    torch.cuda.synchronize()
    for it in range(200):
        culler.workspace[a:b].fill_(0)                                     # (a stale lightMap must not pass for a fresh one)
        side.wait_stream(torch.cuda.current_stream())
        shader.shade(f.cam.frame, s, lights, N)                           # holds the chip's wave slots ...
        culler.cull(f.cam.frame, lights, N, d, ctx=ctx2)                  # ... while the selection's blocks trickle in beside it
        torch.cuda.current_stream().wait_stream(side)
        assert torch.equal(culler.workspace[a:b], ws0[a:b]), f"iteration {it}: the selection's lightMap differs"
        if it % 50 == 49:
            M, lm = culler.band_selection(N)
            assert M == M0 and np.array_equal(lm, lm0)
            got = culler.lists_to_host()
            np.testing.assert_array_equal(got[0], ref[0]); np.testing.assert_array_equal(got[1], ref[1])
    ctx2.close()


def _lightmap_extent(fp, n, m):
    """byte range of lightMap[0, m) inside fp.workspace"""
    import ctypes as C
    a, b = C.c_void_p(), C.c_void_p()
    _lib.check(fp.ctx._lib.sailor_hip_light_cull_band_selection(fp.W, fp.H, n, C.byref(fp.band), fp.workspace.data_ptr(), C.byref(a), C.byref(b)), "band_selection")
    lo = b.value - fp.workspace.data_ptr()
    return lo, lo + 4 * m


def test_launch_log_names_the_kernels_of_a_call(ctx):
    """sailor_hip_context_launch_log: what a cull chain consists of is read from the library -- the brute-force chain of a tiny light set, the four
    kernels of the whole frame, the selection in front of a band's chain when forced -- and timing slots cannot be armed inside a capture."""
    f = synth.make_frame("tiny", with_surface=False)
    W, H, N = f.cam.width, f.cam.height, len(f.lights)
    d = torch.from_numpy(f.depth).to(ctx.device)
    l = upload_lights(f.lights, ctx.device)
    fp = ForwardPlus(ctx, W, H, N)
    assert ctx.launches_of(lambda: fp.cull(f.cam.frame, l, N, d)) == ["k01_prepare", "k1_group_lists", "k1_tile_cull", "k1_pack"]
    assert ctx.launches_of(lambda: fp.cull(f.cam.frame, l, N, d, defer_pack=True)) == ["k01_prepare", "k1_group_lists", "k1_tile_cull"]
    assert ctx.launches_of(lambda: fp.pack()) == ["k1_pack"]
    assert ctx.launches_of(lambda: fp.cull(f.cam.frame, l, 100, d)) == ["k01_prepare", "k1_tile_cull<brute>", "k1_pack"]
    band = host.band_for_rank(W, H, 0, 2)
    fb = ForwardPlus(ctx, W, H, N, band=band)
    db = torch.from_numpy(np.ascontiguousarray(f.depth[band_rows(band)])).to(ctx.device)
    assert ctx.launches_of(lambda: fb.cull(f.cam.frame, l, N, db)) == ["k01_prepare", "k1_group_lists", "k1_tile_cull", "k1_pack"]
    assert ctx.launches_of(lambda: fb.cull(f.cam.frame, l, N, db, _lib.CULL_BAND_SELECT)) == ["k0_band_count", "k0_band_scatter", "k01_prepare", "k1_group_lists", "k1_tile_cull", "k1_pack"]
    count0, _ = ctx.launch_log(0)
    assert ctx.launches_of(lambda: None) == [] and ctx.launch_log(0)[0] == count0
    # arming timing slots while the stream is captured is refused (a launch with events on its packet cannot be a graph node)
    side = torch.cuda.Stream(device=ctx.device)
    c2 = HipContext(ctx.device, stream=side)
    g = torch.cuda.CUDAGraph()
    refused = []
    with torch.cuda.graph(g, stream=side):
        try:
            c2.time_launches(0, 4)
        except _lib.SailorHipError as e:
            refused.append(e)
        fp.cull(f.cam.frame, l, N, d, ctx=c2)
    assert refused and refused[0].status == -7, "SAILOR_HIP_ERR_UNSUPPORTED inside a capture"
    g.replay()
    torch.cuda.synchronize()
    c2.close()
    # the two measurement kernels are launches of the path like any other: named in the log, timed by their own dispatch packets
    import ctypes as C
    n = 64 << 20
    src = torch.full((n,), 7, dtype=torch.uint8, device=ctx.device)
    dst = torch.zeros_like(src)

    def probe():
        _lib.check(ctx._lib.sailor_hip_marker(ctx.handle), "sailor_hip_marker", ctx.handle)
        _lib.check(ctx._lib.sailor_hip_copy_probe(ctx.handle, src.data_ptr(), dst.data_ptr(), n), "sailor_hip_copy_probe", ctx.handle)
    assert ctx.launches_of(probe) == ["k_marker", "k_copy_probe"]
    ctx.time_launches(0, 2)
    probe()
    ctx.synchronize()
    assert torch.equal(dst, src)
    marker_ms, copy_ms = ctx.timed_launch_ms(0), ctx.timed_launch_ms(1)
    assert 0.0 < marker_ms < 0.05 and 2 * n / (copy_ms * 1e-3) / 1e9 > 1000.0, (marker_ms, copy_ms)   # an empty kernel; a 64 MB copy at more than 1 TB/s
    assert ctx._lib.sailor_hip_copy_probe(ctx.handle, src.data_ptr() + 4, dst.data_ptr(), n - 16) == -1, "unaligned pointers are refused"
