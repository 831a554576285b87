"""Prepared lights (include/sailor_hip.h: sailor_hip_prepare_lights): the per-light half of the path run where the `light` SSBO is written.
The prepared entry points must give the bits of the plain ones -- same lists, same radiance -- on every kind of frame, after partial updates, for every
cull path and on bands; and the views must be what the header says they are."""
import numpy as np
import pytest
import torch

from oracle import oracle
from sailor_amd import _lib, host, synth
from sailor_amd.forward_plus import ForwardPlus, PreparedLights, upload_lights, upload_shadow_maps

pytestmark = pytest.mark.gpu


def run(ctx, f, lights_dev, prepared, band=None, flags=_lib.CULL_DEFAULT, csm=None, capacity=None):
    W, H, N = f.cam.width, f.cam.height, len(f.lights)
    fp = ForwardPlus(ctx, W, H, max(capacity or N, 1), band=band)
    b = fp.band
    rows = slice(b.fbRowBegin, b.fbRowBegin + b.fbRowCount)
    d = torch.from_numpy(np.ascontiguousarray(f.depth[rows])).to(ctx.device)
    s = torch.from_numpy(np.ascontiguousarray(f.surface[:, rows])).to(ctx.device)
    fp.cull(f.cam.frame, lights_dev, N, d, flags, prepared=prepared)
    g, idx = fp.lists_to_host()
    rad = fp.shade(f.cam.frame, s, lights_dev, N, csm, prepared=prepared).cpu().numpy()
    return g, idx, rad


def test_views_hold_the_cull_record_and_the_staged_record(ctx):
    f = synth.make_frame("tiny")
    N = len(f.lights)
    dev = upload_lights(f.lights, ctx.device)
    p = PreparedLights(ctx, dev, N, capacity=N + 37)
    ctx.synchronize()
    pos_radius, kind, staged = (t.cpu().numpy() for t in p.views())
    np.testing.assert_array_equal(pos_radius[:N, :3], f.lights["worldPosition"][:, :3])
    np.testing.assert_array_equal(pos_radius[:N, 3], f.lights["bounds"][:, 0])
    np.testing.assert_array_equal(kind[:N].view(np.uint32), f.lights["type"])
    # the staged record (shade_body.h): rec0 = (worldPosition, reach threshold), rec3 = (-direction, cutOff.y), rec4 = (intensity, 1 / B)
    np.testing.assert_array_equal(staged[:N, 0, :3], f.lights["worldPosition"][:, :3])
    np.testing.assert_array_equal(staged[:N, 3, :3], -f.lights["direction"][:, :3])
    np.testing.assert_array_equal(staged[:N, 3, 3], f.lights["cutOff"][:, 1])
    np.testing.assert_array_equal(staged[:N, 4, :3], f.lights["intensity"][:, :3])
    point = f.lights["type"] == host.LIGHT_POINT
    r = f.lights["bounds"][point, 0]
    np.testing.assert_array_equal(staged[:N][point, 0, 3], (r * r) * np.float32(1.00001))
    np.testing.assert_array_equal(staged[:N][point, 4, 3], np.float32(1.0) / r)          # RN(1 / bounds.x): the IEEE quotient
    axis = staged[:N, 1, :3].astype(np.float64)
    assert np.abs(np.linalg.norm(axis, axis=1) - 1.0).max() < 1e-6
    bits = staged[:N, 1, 3].view(np.uint32)
    np.testing.assert_array_equal(bits & 0xFF, f.lights["type"])
    assert ((bits >> 16) & 1).all(), "every parameter of the tiny frame's lights is finite"


@pytest.mark.parametrize("name", ["tiny", "tiny_csm"])
def test_prepared_entry_points_give_the_bits_of_the_plain_ones(ctx, name):
    f = synth.make_frame(name)
    N = len(f.lights)
    W, H = f.cam.width, f.cam.height
    dev = upload_lights(f.lights, ctx.device)
    csm = keep = None
    if f.shadows is not None:
        csm, keep = upload_shadow_maps(f.shadows, ctx.device)
    p = PreparedLights(ctx, dev, N)
    ref = run(ctx, f, dev, None, csm=csm)
    og, oi, _ = oracle.light_cull(f.cam.frame, W, H, f.lights, f.depth)
    np.testing.assert_array_equal(ref[0], og)
    for flags in (_lib.CULL_DEFAULT, _lib.CULL_BRUTE_FORCE, _lib.CULL_INTERVAL_MASKS):
        got = run(ctx, f, dev, p, flags=flags, csm=csm)
        for a, b in zip(got, ref):
            np.testing.assert_array_equal(a, b)
    # bands (the band shade goes through the split blocks of k2_shade_band_p), and a prepared buffer larger than the light count
    big = PreparedLights(ctx, dev, N, capacity=2 * N + 5)
    for r in range(2):
        band = host.band_for_rank(W, H, r, 2)
        a = run(ctx, f, dev, None, band=band, csm=csm, capacity=2 * N + 5)
        b = run(ctx, f, dev, big, band=band, csm=csm, capacity=2 * N + 5)
        for x, y in zip(a, b):
            np.testing.assert_array_equal(x, y)


def test_a_dirty_run_is_prepared_again_and_nothing_else_moves(ctx):
    """LightingECS uploads dirty runs; the backend prepares exactly those slots.  After changing lights [a, b) and preparing that run only, the
    prepared path equals the plain path on the new records -- and differs from the old frame, so the run really was picked up."""
    f = synth.make_frame("tiny")
    N = len(f.lights)
    dev = upload_lights(f.lights, ctx.device)
    p = PreparedLights(ctx, dev, N)
    before = run(ctx, f, dev, p)
    a, b = N // 3, N // 3 + 40
    changed = f.lights.copy()
    changed["worldPosition"][a:b, 1] += np.float32(3.0)
    changed["intensity"][a:b] *= np.float32(4.0)
    changed["bounds"][a:b, 0] *= np.float32(1.5)
    raw = torch.from_numpy(np.ascontiguousarray(changed[a:b]).view(np.uint8).reshape(-1).copy()).to(ctx.device)
    dev[a * 112:b * 112] = raw
    p.prepare(a, b - a)
    f2 = synth.Frame(f.name, f.cam, f.depth, changed, f.surface, None)
    got = run(ctx, f2, dev, p)
    ref = run(ctx, f2, dev, None)
    for x, y in zip(got, ref):
        np.testing.assert_array_equal(x, y)
    assert not np.array_equal(got[2], before[2])
    og, oi, _ = oracle.light_cull(f.cam.frame, f.cam.width, f.cam.height, changed, f.depth)
    np.testing.assert_array_equal(got[0], og)


def test_prepared_path_on_the_4k_frame(ctx):
    """BASELINE.json configs[2] through the prepared entry points: lists and radiance bit for bit those of the plain ones (which the whole-frame
    oracle tests hold against the oracle)."""
    f = synth.make_frame("C3")
    N = len(f.lights)
    dev = upload_lights(f.lights, ctx.device)
    p = PreparedLights(ctx, dev, N)
    ref = run(ctx, f, dev, None)
    got = run(ctx, f, dev, p)
    for x, y in zip(got, ref):
        np.testing.assert_array_equal(x, y)
    assert int(ref[1][0]) > 500_000


@pytest.mark.parametrize("flags", [_lib.CULL_DEFAULT, _lib.CULL_INTERVAL_MASKS, _lib.CULL_BRUTE_FORCE])
def test_preparation_folded_into_the_cull_writes_the_same_views_and_lists(ctx, flags):
    """SAILOR_CULL_PREPARE_LIGHTS (dynamic lights: every record dirty every frame): the cull's per-light pass reads the 112-byte records and writes the
    prepared views on the way.  The views are sailor_hip_prepare_lights' bit for bit, the lists are the same, and a shade from the views it wrote is the
    shade from prepared lights -- for the plane-test masks, the interval masks (C5's form) and the brute-force walk."""
    f = synth.make_frame("tiny", width=320, height=208, lights=synth.LightSetConfig(count=3000, spot_fraction=0.3, radius_scale=5.0, cluster_lights=600))
    W, H, N = f.cam.width, f.cam.height, len(f.lights)
    dev = upload_lights(f.lights, ctx.device)
    ref = PreparedLights(ctx, dev, N, capacity=N + 5)
    ctx.synchronize()
    want = [t.cpu().numpy().copy() for t in ref.views()]
    d = torch.from_numpy(f.depth).to(ctx.device)
    s = torch.from_numpy(f.surface).to(ctx.device)
    fp0 = ForwardPlus(ctx, W, H, N, prepared=ref)
    fp0.cull(f.cam.frame, dev, N, d, flags)
    g0, i0 = fp0.lists_to_host()
    r0 = fp0.shade(f.cam.frame, s, dev, N).cpu().numpy().copy()
    mine = PreparedLights(ctx, dev, 0, capacity=N + 5)      # nothing prepared: the buffer holds whatever the allocator left
    mine.buffer.fill_(0x5A)
    fp = ForwardPlus(ctx, W, H, N, prepared=mine)
    fp.cull(f.cam.frame, dev, N, d, flags, prepare_lights=True)
    g, i = fp.lists_to_host()
    np.testing.assert_array_equal(g, g0)
    np.testing.assert_array_equal(i, i0)
    for a, b in zip((t.cpu().numpy() for t in mine.views()), want):
        np.testing.assert_array_equal(a[:N].view(np.uint32), b[:N].view(np.uint32))
    r = fp.shade(f.cam.frame, s, dev, N).cpu().numpy()
    np.testing.assert_array_equal(r.view(np.uint32), r0.view(np.uint32))
    og, oi, _ = oracle.light_cull(f.cam.frame, W, H, f.lights, f.depth)
    np.testing.assert_array_equal(g, og)
    np.testing.assert_array_equal(i, oi[: 1 + int(oi[0])])
