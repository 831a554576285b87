"""LinearizeDepth on the GPU, through the C-ABI: the standalone pass is bit-exact against the oracle, and
the light cull fed with the RAW attachment (SAILOR_CULL_RAW_DEPTH) produces the same bytes as linearise-then-cull."""
from pathlib import Path

import numpy as np
import pytest
import torch

from oracle import oracle
from sailor_amd import _lib, synth
from sailor_amd.forward_plus import ForwardPlus, linearize_depth, upload_lights

pytestmark = pytest.mark.gpu
GOLDEN = Path(__file__).resolve().parent / "golden"


def bits(a):
    return np.ascontiguousarray(a, np.float32).view(np.uint32)


def test_golden_depth_fixture_bit_exact(ctx):
    g = np.load(GOLDEN / "tiny_depth.npz")
    cam = synth.make_camera(g["raw"].shape[1], g["raw"].shape[0])
    assert cam.frame.cameraZNearZFar[0] == float(g["z_near"])
    out = linearize_depth(ctx, cam.frame, torch.from_numpy(g["raw"]).to(ctx.device)).cpu().numpy()
    np.testing.assert_array_equal(bits(out), bits(g["linear"]))
    assert np.isinf(out[g["raw"] == 0]).all() and (out[g["raw"] == 0] > 0).all()  # nothing drawn -> +inf, as in the shader


@pytest.mark.parametrize("size", [(1920, 1080), (131, 77), (17, 33), (1, 1)])
def test_sizes_and_unaligned_rows(ctx, size):
    """Whole images (float4 path), ragged sizes (scalar tail) and a view that starts 4 bytes into an allocation (scalar path)."""
    w, h = size
    cam = synth.make_camera(w, h)
    raw = synth.make_raw_depth(synth.make_linear_depth(w, h, 11), cam.frame.cameraZNearZFar[0], sky_fraction=0.05, seed=11)
    ref = oracle.linearize_depth(cam.frame.cameraZNearZFar[0], raw)
    d = torch.from_numpy(raw).to(ctx.device)
    np.testing.assert_array_equal(bits(linearize_depth(ctx, cam.frame, d).cpu().numpy()), bits(ref))
    flat = torch.empty(w * h + 1, dtype=torch.float32, device=ctx.device)
    flat[1:] = d.reshape(-1)
    shifted = flat[1:].reshape(h, w)  # same values, base address not 16-byte aligned
    out = torch.empty(w * h + 1, dtype=torch.float32, device=ctx.device)[1:].reshape(h, w)
    np.testing.assert_array_equal(bits(linearize_depth(ctx, cam.frame, shifted, out).cpu().numpy()), bits(ref))


def test_full_size_properties(ctx):
    """8K: the oracle finishes this in well under a second too, so compare everything, plus the size-independent facts:
    monotone (larger raw -> not larger linear) and in-place operation allowed."""
    w, h = 7680, 4320
    cam = synth.make_camera(w, h)
    lin0 = synth.make_linear_depth(w, h)
    raw = synth.make_raw_depth(lin0, cam.frame.cameraZNearZFar[0])
    d = torch.from_numpy(raw).to(ctx.device)
    out = linearize_depth(ctx, cam.frame, d, d).cpu().numpy()  # in place
    np.testing.assert_array_equal(bits(out), bits(oracle.linearize_depth(cam.frame.cameraZNearZFar[0], raw)))
    order = np.argsort(raw.reshape(-1)[::997], kind="stable")
    assert (np.diff(out.reshape(-1)[::997][order]) <= 0).all()
    assert np.abs(out / lin0 - 1).max() < 3e-7  # two roundings away from the distance it was generated from


@pytest.mark.parametrize("name,flags", [("tiny", _lib.CULL_DEFAULT), ("tiny", _lib.CULL_BRUTE_FORCE), ("C2", _lib.CULL_DEFAULT)])
def test_cull_on_raw_depth_equals_linearise_then_cull(ctx, name, flags):
    f = synth.make_frame(name, with_surface=False)
    W, H, N = f.cam.width, f.cam.height, len(f.lights)
    zn = f.cam.frame.cameraZNearZFar[0]
    raw = synth.make_raw_depth(f.depth, zn, sky_fraction=0.04)
    lin = oracle.linearize_depth(zn, raw)
    ref_g, ref_i, _ = oracle.light_cull(f.cam.frame, W, H, f.lights, lin)
    lights = upload_lights(f.lights, ctx.device)
    d_raw = torch.from_numpy(raw).to(ctx.device)
    fused = ForwardPlus(ctx, W, H, N)
    fused.cull(f.cam.frame, lights, N, d_raw, flags | _lib.CULL_RAW_DEPTH)
    g1, i1 = fused.lists_to_host()
    two_pass = ForwardPlus(ctx, W, H, N)
    two_pass.cull(f.cam.frame, lights, N, linearize_depth(ctx, f.cam.frame, d_raw), flags)
    g2, i2 = two_pass.lists_to_host()
    total = int(ref_i[0])
    for g, i in ((g1, i1), (g2, i2)):
        assert int(i[0]) == total
        np.testing.assert_array_equal(g, ref_g)
        np.testing.assert_array_equal(i[: 1 + total], ref_i[: 1 + total])
