"""ComputeMeshCulling.shader main() without OCCLUSION_CULLING (frustum flags + indirect-draw compaction) through the C-ABI:
every byte of the instance buffer and of the indirect buffer equals the oracle's (oracle_mesh_cull_compact)."""
import ctypes as C

import numpy as np
import pytest
import torch

from sailor_amd import _lib, host, synth
from sailor_amd.forward_plus import MeshCull
from oracle import oracle

pytestmark = pytest.mark.gpu


def _check(ctx, cam, inst, batches, n, first):
    mc = MeshCull(ctx, inst, batches)
    mc.run(cam.frame, n, first)
    got_i, got_b = mc.download()
    ref_i, ref_b = oracle.mesh_cull_compact(cam.frame, inst, n, first, batches)
    np.testing.assert_array_equal(got_b, ref_b)
    np.testing.assert_array_equal(got_i.view(np.uint32).reshape(-1, 24), ref_i.view(np.uint32).reshape(-1, 24))
    return mc, ref_i, ref_b


def test_ragged_batches_with_untouched_head(ctx):
    cam = synth.make_camera(1920, 1080)
    s = synth.make_instance_set(20000, 100, first_instance=37)
    _, ref_i, ref_b = _check(ctx, cam, s.instances, s.batches, 20000, 37)
    assert (ref_i["isCulled"][:37] == 7).all()
    assert (s.batches[:, 1] == 0).sum() > 0 and 0 < int(ref_b[:, 1].sum()) < 20000
    assert (ref_b[:, [0, 2, 3, 4]] == s.batches[:, [0, 2, 3, 4]]).all()


def test_one_long_batch_spans_more_chunks_than_blocks(ctx):
    """300 000 records in one draw: 1 172 chunks on the 1 024-block persistent grid, look-back over > 64 predecessors."""
    cam = synth.make_camera(3840, 2160)
    s = synth.make_instance_set(300000, 1)
    assert s.batches[0, 1] == 300000
    _check(ctx, cam, s.instances, s.batches, 300000, 0)


def test_nothing_culled_and_everything_culled(ctx):
    cam = synth.make_camera(1920, 1080)
    s = synth.make_instance_set(5000, 7)
    near = oracle._copy_records(s.instances)
    near["model"][:, 12:15] = np.float32([0.0, 150.0, -500.0])  # in front of the camera: all visible
    mc, ref_i, ref_b = _check(ctx, cam, near, s.batches, 5000, 0)
    np.testing.assert_array_equal(ref_b, s.batches)
    assert (ref_i["materialInstance"] == np.arange(5000)).all()
    behind = oracle._copy_records(s.instances)
    behind["model"][:, 12:15] = np.float32([0.0, 150.0, 5000.0])  # behind it: all culled
    _, ref_i, ref_b = _check(ctx, cam, behind, s.batches, 5000, 0)
    assert (ref_b[:, 1] == 0).all() and (ref_i["materialInstance"] == np.arange(5000)).all()


def test_many_tiny_batches_and_no_batches(ctx):
    cam = synth.make_camera(1920, 1080)
    s = synth.make_instance_set(6000, 5000)
    _check(ctx, cam, s.instances, s.batches, 6000, 0)
    # numBatches = 0: only the flags
    mc = MeshCull(ctx, s.instances, s.batches[:0])
    mc.run(cam.frame)
    got_i, _ = mc.download()
    np.testing.assert_array_equal(got_i["isCulled"], oracle.mesh_frustum_cull(cam.frame, s.instances)["isCulled"])
    np.testing.assert_array_equal(got_i["materialInstance"], np.arange(6000))


def test_second_frame_over_the_compacted_buffers(ctx):
    """The buffers persist across frames in the reference (Batch.hpp re-uploads them only when the draw list changes)."""
    cam = synth.make_camera(1920, 1080)
    s = synth.make_instance_set(30000, 64)
    mc = MeshCull(ctx, s.instances, s.batches)
    mc.run(cam.frame)
    i1, b1 = oracle.mesh_cull_compact(cam.frame, s.instances, 30000, 0, s.batches)
    cam2 = synth.make_camera(1280, 720, fov=50.0)
    mc.run(cam2.frame)
    i2, b2 = oracle.mesh_cull_compact(cam2.frame, i1, 30000, 0, b1)
    got_i, got_b = mc.download()
    np.testing.assert_array_equal(got_b, b2)
    np.testing.assert_array_equal(got_i.view(np.uint32).reshape(-1, 24), i2.view(np.uint32).reshape(-1, 24))
    assert int(b2[:, 1].sum()) < int(b1[:, 1].sum())


def test_1m_instances_4096_batches(ctx):
    cam = synth.make_camera(7680, 4320)
    s = synth.make_instance_set(1 << 20, 4096)
    _check(ctx, cam, s.instances, s.batches, 1 << 20, 0)


def test_argument_errors(ctx):
    lib = _lib.load()
    cam = synth.make_camera(640, 480)
    s = synth.make_instance_set(100, 3)
    mc = MeshCull(ctx, s.instances, s.batches)
    rc = lib.sailor_hip_mesh_cull_compact(ctx.handle, C.byref(cam.frame), mc.instances.data_ptr(), 100, 0, mc.batches.data_ptr(), 3,
                                          mc.workspace.data_ptr(), 16)
    assert rc == -1
    rc = lib.sailor_hip_mesh_cull_compact(ctx.handle, C.byref(cam.frame), mc.instances.data_ptr(), 100, 0, None, 3,
                                          mc.workspace.data_ptr(), mc._ws_bytes)
    assert rc == -1


# ---- the shader as shipped: OCCLUSION_CULLING against the Hi-Z pyramid ------------------------------------------------------------
def _depth_and_pyramid(cam, w, h, levels, seed=5):
    lin = synth.make_linear_depth(w, h, seed, d_min=200.0, d_max=2500.0)
    raw = synth.make_raw_depth(lin, cam.frame.cameraZNearZFar[0])
    return raw


@pytest.mark.parametrize("shape", [((540, 960), 960, 960, 10), ((67, 131), 131, 131, 8), ((300, 500), 250, 250, 8), ((64, 64), 64, 64, 7)])
def test_hiz_pyramid_matches_the_oracle_bit_for_bit(ctx, shape):
    """DepthHighZNode's loop: mip 0 from the half-resolution depth (DefaultRenderer.renderer: DepthHighZ is ViewportWidth/2 squared, so the first
    step is not 2:1), every further mip from the previous; ragged sizes exercise the zero-weight and clamp rules of the min sampler."""
    from sailor_amd.forward_plus import hiz_build
    (dh, dw), w, h, levels = shape
    cam = synth.make_camera(1920, 1080)
    raw = _depth_and_pyramid(cam, dw, dh, levels)
    got = hiz_build(ctx, torch.from_numpy(raw).to(ctx.device), w, h, levels).cpu().numpy()
    ref = oracle.hiz_build(raw, w, h, levels)
    np.testing.assert_array_equal(got.view(np.uint32), ref.view(np.uint32))


def test_occlusion_flags_and_compaction_with_the_pyramid(ctx):
    from sailor_amd.forward_plus import hiz_build
    W, H, levels = 1920, 1080, 10
    cam = synth.make_camera(W, H)
    raw = _depth_and_pyramid(cam, W // 2, H // 2, levels)
    pyr = hiz_build(ctx, torch.from_numpy(raw).to(ctx.device), W // 2, W // 2, levels)
    ref_pyr = oracle.hiz_build(raw, W // 2, W // 2, levels)
    s = synth.make_instance_set(50000, 300, first_instance=5)
    mc = MeshCull(ctx, s.instances, s.batches)
    mc.run(cam.frame, 50000, 5, hiz=(pyr, W // 2, W // 2, levels))
    got_i, got_b = mc.download()
    ref_i, ref_b = oracle.mesh_cull_compact(cam.frame, s.instances, 50000, 5, s.batches, hiz=(ref_pyr, W // 2, W // 2, levels))
    np.testing.assert_array_equal(got_b, ref_b)
    np.testing.assert_array_equal(got_i.view(np.uint32).reshape(-1, 24), ref_i.view(np.uint32).reshape(-1, 24))
    frustum_only = oracle.mesh_frustum_cull(cam.frame, s.instances[5:])["isCulled"]
    with_hiz = oracle.mesh_cull_occlusion(cam.frame, s.instances[5:], ref_pyr, W // 2, W // 2, levels)["isCulled"]
    assert ((frustum_only == 1) <= (with_hiz == 1)).all() and int(with_hiz.sum()) > int(frustum_only.sum()) + 1000
    assert 0 < int(ref_b[:, 1].sum()) < 50000 - int(frustum_only.sum())


def test_occlusion_1m_instances(ctx):
    from sailor_amd.forward_plus import hiz_build
    W, H, levels = 3840, 2160, 11
    cam = synth.make_camera(W, H)
    raw = _depth_and_pyramid(cam, W // 2, H // 2, levels, seed=9)
    pyr = hiz_build(ctx, torch.from_numpy(raw).to(ctx.device), W // 2, W // 2, levels)
    s = synth.make_instance_set(1 << 20, 4096)
    mc = MeshCull(ctx, s.instances, s.batches)
    mc.run(cam.frame, hiz=(pyr, W // 2, W // 2, levels))
    got_i, got_b = mc.download()
    ref_i, ref_b = oracle.mesh_cull_compact(cam.frame, s.instances, 1 << 20, 0, s.batches, hiz=(pyr.cpu().numpy(), W // 2, W // 2, levels))
    np.testing.assert_array_equal(got_b, ref_b)
    np.testing.assert_array_equal(got_i.view(np.uint32).reshape(-1, 24), ref_i.view(np.uint32).reshape(-1, 24))


def test_golden_fixture_through_the_c_abi(ctx):
    """tests/golden/tiny_mesh_cull.npz: pyramid, frustum-only and occlusion results of the GPU equal the committed vectors."""
    from pathlib import Path
    from sailor_amd.forward_plus import hiz_build
    g = np.load(Path(__file__).resolve().parent / "golden" / "tiny_mesh_cull.npz")
    cam = synth.make_camera(640, 360)
    s = synth.make_instance_set(3000, 24, first_instance=9)
    raw = synth.make_raw_depth(synth.make_linear_depth(96, 54, 5, d_min=200.0, d_max=2500.0), cam.frame.cameraZNearZFar[0])
    pyr = hiz_build(ctx, torch.from_numpy(raw).to(ctx.device), 96, 96, 7)
    np.testing.assert_array_equal(pyr.cpu().numpy().view(np.uint32), g["pyramid"].view(np.uint32))
    mc = MeshCull(ctx, s.instances, s.batches)
    mc.run(cam.frame, 3000, 9)
    gi, gb = mc.download()
    np.testing.assert_array_equal(gi.view(np.uint32).reshape(-1, 24), g["frustum_instances"])
    np.testing.assert_array_equal(gb, g["frustum_batches"])
    mc = MeshCull(ctx, s.instances, s.batches)
    mc.run(cam.frame, 3000, 9, hiz=(pyr, 96, 96, 7))
    gi, gb = mc.download()
    np.testing.assert_array_equal(gi.view(np.uint32).reshape(-1, 24), g["occlusion_instances"])
    np.testing.assert_array_equal(gb, g["occlusion_batches"])
