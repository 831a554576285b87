"""sailor_hip_frame_pipelined: the cull of frame k+1 launched together with slices of the shade of frame k.  Same device code as the two plain
calls, so the next frame's lists and this frame's radiance must come out bit for bit the same (on a band with the tile-order hint: the same as the
band kernel with its split blocks)."""
import numpy as np
import pytest
import torch

from sailor_amd import _lib, host, synth
from sailor_amd.forward_plus import ForwardPlus, upload_lights

pytestmark = pytest.mark.gpu


def _frames(name, **kw):
    f = synth.make_frame(name, **kw)
    lights2 = f.lights.copy()
    lights2["worldPosition"][:, :3] += np.float32([3.0, -2.0, 5.0])   # the next frame: every light has moved
    return f, lights2


@pytest.mark.parametrize("case", ["tiny", "ragged", "band"])
def test_pipelined_equals_cull_then_shade(ctx, case):
    if case == "tiny":
        f, lights2 = _frames("tiny")
        band = None
    elif case == "ragged":
        f, lights2 = _frames("tiny", width=333, height=201, lights=synth.LightSetConfig(count=3000, spot_fraction=0.3, radius_scale=5.0, cluster_lights=400))
        band = None
    else:
        f, lights2 = _frames("tiny", width=640, height=480, lights=synth.LightSetConfig(count=6000, spot_fraction=0.3, radius_scale=6.0, cluster_lights=1500))
        band = host.band_for_rank(640, 480, 1, 3)
    cam, W, H, N = f.cam, f.cam.width, f.cam.height, len(f.lights)
    a, b = ForwardPlus(ctx, W, H, N, band=band), ForwardPlus(ctx, W, H, N, band=band)
    rows = slice(a.band.fbRowBegin, a.band.fbRowBegin + a.band.fbRowCount)
    depth = torch.from_numpy(np.ascontiguousarray(f.depth[rows])).to(ctx.device)
    surface = torch.from_numpy(np.ascontiguousarray(f.surface[:, rows])).to(ctx.device)
    l1, l2 = upload_lights(f.lights, ctx.device), upload_lights(lights2, ctx.device)
    # reference: plain calls -- frame k culled into `a`, shaded; frame k+1 culled into `b`
    a.cull(cam.frame, l1, N, depth)
    ref_rad = a.shade(cam.frame, surface, l1, N).clone()
    b.cull(cam.frame, l2, N, depth)
    ref_grid, ref_culled = b.grid.clone(), b.culled.clone()
    if case == "band":
        g, _ = a.lists_to_host()
        assert (g[:, 1] >= 40).sum() > 10, "the band has long tiles for the split blocks"
    b.grid.zero_(); b.culled.zero_()
    out = a.shade_while_culling(b, cam.frame, l2, N, depth, cam.frame, surface, l1, N, out=torch.zeros_like(ref_rad))
    ctx.synchronize()
    assert torch.equal(out, ref_rad)
    assert torch.equal(b.grid, ref_grid)
    n = int(ref_culled[0].item())
    assert torch.equal(b.culled[: 1 + n], ref_culled[: 1 + n])
    # and the roles swap for the next step: shade `b` (frame k+1) while `a` culls frame k+2 (= frame k's lights again)
    ref2 = b.shade(cam.frame, surface, l2, N).clone()
    a.grid.zero_()
    out2 = b.shade_while_culling(a, cam.frame, l1, N, depth, cam.frame, surface, l2, N, out=torch.zeros_like(ref2))
    ctx.synchronize()
    assert torch.equal(out2, ref2)
    a2 = ForwardPlus(ctx, W, H, N, band=band)
    a2.cull(cam.frame, l1, N, depth)
    assert torch.equal(a.grid, a2.grid)


def test_pipelined_c3_full_frame(ctx):
    f = synth.make_frame("C3")
    cam, W, H, N = f.cam, f.cam.width, f.cam.height, len(f.lights)
    a, b = ForwardPlus(ctx, W, H, N), ForwardPlus(ctx, W, H, N)
    depth = torch.from_numpy(f.depth).to(ctx.device)
    surface = torch.from_numpy(f.surface).to(ctx.device)
    lights = upload_lights(f.lights, ctx.device)
    a.cull(cam.frame, lights, N, depth)
    ref = a.shade(cam.frame, surface, lights, N).clone()
    out = a.shade_while_culling(b, cam.frame, lights, N, depth, cam.frame, surface, lights, N, out=torch.zeros_like(ref))
    ctx.synchronize()
    assert torch.equal(out, ref) and torch.equal(a.grid, b.grid)
    n = int(a.culled[0].item())
    assert torch.equal(a.culled[: 1 + n], b.culled[: 1 + n])
