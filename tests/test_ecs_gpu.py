"""K4 parity (GPU): ECS transform + bounds + frustum cull sweep and the sphere-vs-view-frustum instance cull, through the
C-ABI, against the CPU oracle -- bit-exact world matrices, world AABBs and visibility words."""
import ctypes as C

import numpy as np
import pytest
import torch

from oracle import oracle
from sailor_amd import _lib, host, synth
from sailor_amd.forward_plus import EcsSweep

pytestmark = pytest.mark.gpu


def camera_planes(cam):
    planes, _ = host.extract_frustum_planes(cam.world, cam.aspect, cam.fov, cam.z_near, cam.z_far)
    return planes


@pytest.mark.parametrize("count", [1024, 1000, 65, 7, 100000])
def test_sweep_bit_exact(ctx, count):
    """configs[0] (1 024 entities, the Editor.world objects first) and ragged sizes."""
    ents = synth.make_entities(count)
    cam = synth.make_camera(1920, 1080)
    planes = camera_planes(cam)
    sweep = EcsSweep(ctx, ents)
    world, aabb, vis = sweep.run(planes)
    ctx.synchronize()
    ow, oa, ov = oracle.ecs_sweep(ents.transforms, ents.parent, ents.local_aabb, planes)
    np.testing.assert_array_equal(world.cpu().numpy().view(np.uint32), ow.view(np.uint32))
    np.testing.assert_array_equal(aabb.cpu().numpy().view(np.uint32), oa.view(np.uint32))
    np.testing.assert_array_equal(vis.cpu().numpy().view(np.uint64), ov)
    nvis = int(np.unpackbits(ov.view(np.uint8)).sum())
    if count >= 1000:
        assert 0 < nvis < count


@pytest.mark.parametrize("levels", [4, 5, 9])
def test_deep_hierarchy(ctx, levels):
    """<= 4 levels run as ONE launch (ancestors' relative matrices recomputed, same left-to-right product); deeper trees take
    one launch per level with level boundaries inside 64-entity words.  Both must give the oracle's bits."""
    per, n = 37 + 64 * 3, 0
    ents = synth.make_entities(per * levels, editor_world=False)
    offs = [0]
    for lvl in range(levels):
        n += per + 5 * lvl  # uneven level sizes
        offs.append(min(n, per * levels))
    offs[-1] = per * levels
    ents.level_offsets = np.array(offs, np.uint32)
    ents.parent[:] = 0xFFFFFFFF
    u = synth.uniforms(synth.STREAM_ENTITIES, per * levels, 1 << 26)
    for lvl in range(1, levels):
        lo, hi, plo, phi = offs[lvl], offs[lvl + 1], offs[lvl - 1], offs[lvl]
        ents.parent[lo:hi] = (plo + np.floor(u[lo:hi] * (phi - plo))).astype(np.uint32)
        ents.transforms[lo:hi, 0:3] *= np.float32(0.05)
        ents.transforms[lo:hi, 8:11] = np.float32(0.9) + np.float32(0.2) * ents.transforms[lo:hi, 8:11] / np.float32(4.0)  # keep the scale chain bounded
    cam = synth.make_camera(1920, 1080)
    planes = camera_planes(cam)
    world, aabb, vis = EcsSweep(ctx, ents).run(planes)
    ctx.synchronize()
    ow, oa, ov = oracle.ecs_sweep(ents.transforms, ents.parent, ents.local_aabb, planes)
    np.testing.assert_array_equal(world.cpu().numpy().view(np.uint32), ow.view(np.uint32))
    np.testing.assert_array_equal(aabb.cpu().numpy().view(np.uint32), oa.view(np.uint32))
    np.testing.assert_array_equal(vis.cpu().numpy().view(np.uint64), ov)


def test_flt_min_quirk_all_negative_box(ctx):
    """Math/Bounds.cpp:484: max is seeded with the smallest POSITIVE float, so an all-negative box keeps max ~ 1.18e-38."""
    ents = synth.make_entities(64, editor_world=False)
    ents.transforms[:, 0:3] = -np.abs(ents.transforms[:, 0:3]) - 1000.0
    ents.transforms[:, 4:8] = [0, 0, 0, 1]
    ents.parent[:] = 0xFFFFFFFF
    ents.level_offsets = np.array([0, 64], np.uint32)
    cam = synth.make_camera(640, 360)
    planes = camera_planes(cam)
    sweep = EcsSweep(ctx, ents)
    _, aabb, _ = sweep.run(planes)
    ctx.synchronize()
    _, oa, _ = oracle.ecs_sweep(ents.transforms, ents.parent, ents.local_aabb, planes)
    got = aabb.cpu().numpy()
    np.testing.assert_array_equal(got.view(np.uint32), oa.view(np.uint32))
    assert (got[:, 3:] == np.finfo(np.float32).tiny).all()


def test_mesh_frustum_cull_matches_shader_semantics(ctx):
    cam = synth.make_camera(1920, 1080)
    ents = synth.make_entities(5000)
    planes = camera_planes(cam)
    ow, _, _ = oracle.ecs_sweep(ents.transforms, ents.parent, ents.local_aabb, planes)
    inst = np.zeros(5000, host.INSTANCE_DTYPE)
    inst["model"] = ow
    u = synth.uniforms(synth.STREAM_ENTITIES, 5000 * 4, 1 << 24).reshape(5000, 4)
    inst["sphereBounds"][:, :3] = (u[:, :3] - 0.5) * 10
    inst["sphereBounds"][:, 3] = 1 + 60 * u[:, 3]
    inst["isCulled"] = 7
    ref = oracle.mesh_frustum_cull(cam.frame, inst)
    t = torch.from_numpy(inst.view(np.uint8).reshape(-1).copy()).to(ctx.device)
    lib = _lib.load()
    _lib.check(lib.sailor_hip_mesh_frustum_cull(ctx.handle, C.byref(cam.frame), t.data_ptr(), 5000, 0), "mesh_frustum_cull", ctx.handle)
    ctx.synchronize()
    got = t.cpu().numpy().view(host.INSTANCE_DTYPE)
    np.testing.assert_array_equal(got["isCulled"], ref["isCulled"])
    assert 0 < int(ref["isCulled"].sum()) < 5000
    # firstInstanceIndex / numInstances window: only [100, 300) is touched
    inst["isCulled"] = 7
    t = torch.from_numpy(inst.view(np.uint8).reshape(-1).copy()).to(ctx.device)
    _lib.check(lib.sailor_hip_mesh_frustum_cull(ctx.handle, C.byref(cam.frame), t.data_ptr(), 200, 100), "mesh_frustum_cull", ctx.handle)
    ctx.synchronize()
    got = t.cpu().numpy().view(host.INSTANCE_DTYPE)
    assert (got["isCulled"][:100] == 7).all() and (got["isCulled"][300:] == 7).all()
    np.testing.assert_array_equal(got["isCulled"][100:300], ref["isCulled"][100:300])


def test_sweep_1m_entities_visibility_checksum(ctx):
    """configs[4] entity count: 1 048 576 entities, compared word for word with the oracle (a second of CPU time)."""
    ents = synth.make_entities(1 << 20)
    cam = synth.make_camera(7680, 4320)
    planes = camera_planes(cam)
    sweep = EcsSweep(ctx, ents)
    world, aabb, vis = sweep.run(planes)
    ctx.synchronize()
    ow, oa, ov = oracle.ecs_sweep(ents.transforms, ents.parent, ents.local_aabb, planes)
    np.testing.assert_array_equal(vis.cpu().numpy().view(np.uint64), ov)
    np.testing.assert_array_equal(aabb.cpu().numpy().view(np.uint32), oa.view(np.uint32))
    np.testing.assert_array_equal(world.cpu().numpy().view(np.uint32), ow.view(np.uint32))


def test_cascade_caster_sets_from_the_sweeps_world_boxes(ctx):
    """sailor_hip_csm_caster_masks over the world AABBs the sweep just produced (1 048 576 entities, four cascade frusta extracted from the
    light matrices as LightingECS.cpp:287-292 does): every mask word equals the oracle's."""
    from sailor_amd.forward_plus import csm_caster_masks
    ents = synth.make_entities(1 << 20)
    cam = synth.make_camera(3840, 2160)
    sweep = EcsSweep(ctx, ents)
    _, aabb, _ = sweep.run(camera_planes(cam))
    sh = synth.make_shadow_set(cam, 16)
    planes = np.stack([host.extract_frustum_planes_matrix(sh.lights_matrices[k])[0] for k in range(4)])
    got = csm_caster_masks(ctx, aabb, planes)
    ctx.synchronize()
    ref = oracle.csm_caster_masks(aabb.cpu().numpy(), planes)
    np.testing.assert_array_equal(got.cpu().numpy().view(np.uint64), ref)
    counts = [int(np.unpackbits(m.view(np.uint8)).sum()) for m in ref]
    assert 0 < counts[0] < counts[3] < (1 << 20)
    # ragged count, fewer cascades
    got = csm_caster_masks(ctx, aabb[:1000], planes[:2])
    ctx.synchronize()
    np.testing.assert_array_equal(got.cpu().numpy().view(np.uint64), oracle.csm_caster_masks(aabb[:1000].cpu().numpy(), planes[:2]))


@pytest.mark.parametrize("world, count", [(2, 100000), (3, 1000), (8, 1 << 20), (5, 130)])
def test_entity_slices_of_the_sweep_are_the_whole_sweeps_bits(ctx, world, count):
    """K4 split across the ranks of a node (sailor_hip_ecs_range_for_rank / sailor_hip_ecs_sweep_range): every rank sweeps a slice of
    whole visibility words -- children rebuild their ancestors' relative matrices, so a slice needs nothing of another rank's -- and the slices'
    matrices, boxes and visibility words, laid side by side as the all-gather lays them, are the whole sweep's and the oracle's, bit for bit.
    (Ragged ends: ranks with fewer entities than the others, ranks with none.)"""
    ents = synth.make_entities(count)
    cam = synth.make_camera(1920, 1080)
    planes = camera_planes(cam)
    ow, oa, ov = oracle.ecs_sweep(ents.transforms, ents.parent, ents.local_aabb, planes)
    words = (count + 63) // 64
    gathered, covered = [], 0
    for r in range(world):
        sw = EcsSweep(ctx, ents, rank=r, world=world)
        assert (sw.begin % 64 == 0 or sw.begin == count) and (sw.end % 64 == 0 or sw.end == count) and sw.begin == min(r * sw.words_per_rank * 64, count)
        sw.world.fill_(-7.0); sw.world_aabb.fill_(-7.0)
        w, a, v = sw.run(planes)
        ctx.synchronize()
        w, a, v = w.cpu().numpy(), a.cpu().numpy(), v.cpu().numpy().view(np.uint64)
        lo, hi = sw.begin, sw.end
        np.testing.assert_array_equal(w[lo:hi].view(np.uint32), ow[lo:hi].view(np.uint32))
        np.testing.assert_array_equal(a[lo:hi].view(np.uint32), oa[lo:hi].view(np.uint32))
        assert (w[:lo] == -7.0).all() and (w[hi:] == -7.0).all() and (a[:lo] == -7.0).all() and (a[hi:] == -7.0).all(), "nothing outside the slice is written"
        assert v.shape[0] >= world * sw.words_per_rank
        gathered.append(v[r * sw.words_per_rank:(r + 1) * sw.words_per_rank])   # this rank's slot of the in-place all-gather
        covered += hi - lo
    assert covered == count
    np.testing.assert_array_equal(np.concatenate(gathered)[:words], ov)


def test_a_slice_of_a_deep_hierarchy_is_refused(ctx):
    """More than four levels: an entity reads its parent's WORLD matrix, which only the sweep of the whole set has on this rank -- a proper slice is
    SAILOR_HIP_ERR_UNSUPPORTED (sweep it replicated), the whole range is the ordinary sweep."""
    levels, per = 6, 101
    ents = synth.make_entities(per * levels, editor_world=False)
    ents.level_offsets = np.arange(levels + 1, dtype=np.uint32) * per
    ents.parent[:] = 0xFFFFFFFF
    for lvl in range(1, levels):
        ents.parent[lvl * per:(lvl + 1) * per] = np.arange((lvl - 1) * per, lvl * per, dtype=np.uint32)
        ents.transforms[lvl * per:(lvl + 1) * per, 8:11] = 1.0
    planes = camera_planes(synth.make_camera(1920, 1080))
    sw = EcsSweep(ctx, ents, rank=1, world=2)
    with pytest.raises(_lib.SailorHipError) as e:
        sw.run(planes)
    assert e.value.status == -7
    world, aabb, vis = EcsSweep(ctx, ents).run(planes)
    ctx.synchronize()
    ow, oa, ov = oracle.ecs_sweep(ents.transforms, ents.parent, ents.local_aabb, planes)
    np.testing.assert_array_equal(world.cpu().numpy().view(np.uint32), ow.view(np.uint32))
    np.testing.assert_array_equal(vis.cpu().numpy().view(np.uint64), ov)
