"""The shadow-map producer (compute rasteriser for ShadowPrepassNode's caster draws + ShadowCaster.shader's fragment stage) through the C-ABI:
depth buffers and resolved shadow maps equal the oracle's bit for bit -- the winning depth of a texel does not depend on fragment order."""
import numpy as np
import pytest
import torch

from oracle import oracle
from sailor_amd import _lib, host, synth
from sailor_amd.forward_plus import EcsSweep, csm_caster_masks, raster_depth, shadow_resolve

pytestmark = pytest.mark.gpu
IDENTITY = np.eye(4, dtype=np.float32).reshape(16)


def _gpu(ctx, lm, pos, idx, models, w, h, ids=None, depth=None):
    d = raster_depth(ctx, lm, torch.from_numpy(np.ascontiguousarray(pos, np.float32)).to(ctx.device),
                     torch.from_numpy(np.ascontiguousarray(idx, np.uint32).view(np.int32)).to(ctx.device),
                     torch.from_numpy(np.ascontiguousarray(models, np.float32)).to(ctx.device), w, h,
                     None if ids is None else torch.from_numpy(np.ascontiguousarray(ids, np.uint32).view(np.int32)).to(ctx.device), depth)
    ctx.synchronize()
    return d


def test_fill_rule_viewport_flip_depth_test_and_clipping(ctx):
    W = H = 16
    quad = np.float32([[-0.5, -0.5, 0.3], [0.5, -0.5, 0.3], [0.5, 0.5, 0.3], [-0.5, 0.5, 0.3]])
    idx = np.uint32([[0, 1, 2], [0, 2, 3]])
    one = IDENTITY.reshape(1, 16)
    d = _gpu(ctx, IDENTITY, quad, idx, one, W, H).cpu().numpy()
    np.testing.assert_array_equal(d, oracle.raster_depth(IDENTITY, quad, idx, one, W, H))
    assert (d > 0).sum() == 64 and (d[4:12, 4:12] == np.float32(0.3)).all()
    # the two triangles share the diagonal: no texel belongs to both, none is missed (top-left rule); either winding is drawn
    a = _gpu(ctx, IDENTITY, quad, idx[:1], one, W, H).cpu().numpy()
    b = _gpu(ctx, IDENTITY, quad, idx[1:, ::-1], one, W, H).cpu().numpy()
    assert ((a > 0) & (b > 0)).sum() == 0 and ((a > 0) | (b > 0)).sum() == 64
    # viewport (0, H, W, -H): +y of clip space is the TOP of the map; reversed Z: the larger depth wins; z outside [0, 1] is clipped
    tri = np.float32([[-1, 0.5, 0.2], [1, 0.5, 0.2], [0, 1.0, 0.2], [-1, -1, 0.6], [1, -1, 0.6], [0, 1, 0.6], [-1, -1, 1.5], [1, -1, 1.5], [0, 1, 1.5]])
    idx3 = np.uint32([[0, 1, 2], [3, 4, 5], [6, 7, 8]])
    d = _gpu(ctx, IDENTITY, tri, idx3, one, W, H).cpu().numpy()
    np.testing.assert_array_equal(d, oracle.raster_depth(IDENTITY, tri, idx3, one, W, H))
    assert d[:4].max() > 0 and d.max() == np.float32(0.6) and set(np.unique(d)) == {np.float32(0.0), np.float32(0.2), np.float32(0.6)}
    assert (d[:3][d[:3] > 0] >= np.float32(0.2)).all() and d[12, 8] == np.float32(0.6)
    # a sloped triangle: interpolated depths, drawn on top of an existing buffer (a dependent pass)
    slope = np.float32([[-0.9, -0.8, 0.1], [0.8, -0.6, 0.9], [-0.2, 0.9, 0.5]])
    base = _gpu(ctx, IDENTITY, quad, idx, one, W, H)
    d = _gpu(ctx, IDENTITY, slope, np.uint32([[0, 1, 2]]), one, W, H, depth=base).cpu().numpy()
    ref = oracle.raster_depth(IDENTITY, slope, np.uint32([[0, 1, 2]]), one, W, H, depth=oracle.raster_depth(IDENTITY, quad, idx, one, W, H))
    np.testing.assert_array_equal(d.view(np.uint32), ref.view(np.uint32))
    assert len(np.unique(d)) > 20


@pytest.mark.parametrize("cascade,size", [(0, 512), (1, 256), (3, 256)])
def test_box_casters_of_a_cascade_and_the_resolved_maps(ctx, cascade, size):
    """3 000 entities drawn as their bounding boxes into a cascade of the directional light (the light matrices of LightingECS.cpp:276-298), only the
    entities the cascade's frustum overlaps (sailor_hip_csm_caster_masks); then ShadowCaster's fragment stage in all three map formats."""
    cam = synth.make_camera(1280, 720)
    ents = synth.make_entities(3000)
    sweep = EcsSweep(ctx, ents)
    planes, _ = host.extract_frustum_planes(cam.world, cam.aspect, cam.fov, cam.z_near, cam.z_far)
    world, aabb, _ = sweep.run(planes)
    sh = synth.make_shadow_set(cam, 16)
    cplanes = np.stack([host.extract_frustum_planes_matrix(sh.lights_matrices[k])[0] for k in range(4)])
    masks = csm_caster_masks(ctx, aabb, cplanes).cpu().numpy().view(np.uint64)
    ids = np.nonzero(np.unpackbits(masks[cascade].view(np.uint8), bitorder="little")[:3000])[0].astype(np.uint32)
    assert len(ids) > 10
    pos, tris = synth.unit_cube_mesh()
    models = synth.caster_models(world.cpu().numpy(), ents.local_aabb)
    lm = sh.lights_matrices[cascade]
    d = _gpu(ctx, lm, pos, tris, models, size, size, ids=ids)
    ref = oracle.raster_depth(lm, pos, tris, models, size, size, instance_ids=ids)
    np.testing.assert_array_equal(d.cpu().numpy().view(np.uint32), ref.view(np.uint32))
    # ECullMode::Back, as the reference's shadow material: closed boxes inside the depth slab show the same front faces; in general the oracle decides
    dcull = raster_depth(ctx, lm, torch.from_numpy(pos).to(ctx.device), torch.from_numpy(tris.view(np.int32)).to(ctx.device), torch.from_numpy(models).to(ctx.device),
                         size, size, torch.from_numpy(ids.view(np.int32)).to(ctx.device), cull_back=True)
    ctx.synchronize()
    np.testing.assert_array_equal(dcull.cpu().numpy().view(np.uint32), oracle.raster_depth(lm, pos, tris, models, size, size, instance_ids=ids, cull_back=True).view(np.uint32))
    # with the coarse-depth workspace: fewer texels touched, the same buffer; every coarse word is a lower bound of its block
    coarse = torch.empty(int(_lib.load().sailor_hip_raster_coarse_words(size, size)), dtype=torch.int32, device=ctx.device)
    dc = raster_depth(ctx, lm, torch.from_numpy(pos).to(ctx.device), torch.from_numpy(tris.view(np.int32)).to(ctx.device), torch.from_numpy(models).to(ctx.device),
                      size, size, torch.from_numpy(ids.view(np.int32)).to(ctx.device), coarse=coarse)
    ctx.synchronize()
    np.testing.assert_array_equal(dc.cpu().numpy().view(np.uint32), ref.view(np.uint32))
    words = coarse.cpu().numpy().view(np.float32)
    n1, n2 = (size // 8) ** 2, (size // 64) ** 2   # (behind the two levels: eight words for the box of the mesh of a many-instance draw -- round 6)
    lower, lower2 = words[:n1].reshape(size // 8, size // 8), words[n1:n1 + n2].reshape(size // 64, size // 64)
    assert (lower <= ref.reshape(size // 8, 8, size // 8, 8).min(axis=(1, 3))).all() and (lower > 0).any()
    assert (lower2 <= ref.reshape(size // 64, 64, size // 64, 64).min(axis=(1, 3))).all()
    cover = float((ref > 0).mean())
    assert 0.001 < cover < 1.0, cover
    evsm = shadow_resolve(ctx, d, _lib.SHADOWMAP_RGBA32F).cpu().numpy()
    np.testing.assert_array_equal(evsm.view(np.uint32), oracle.shadow_resolve_evsm(ref).view(np.uint32))
    assert (evsm[ref == 0] == 0).all() and (evsm[ref > 0][:, 0] >= 1.0).all()
    np.testing.assert_array_equal(shadow_resolve(ctx, d, _lib.SHADOWMAP_R16F).cpu().numpy().view(np.uint16), ref.astype(np.float16).view(np.uint16))
    np.testing.assert_array_equal(shadow_resolve(ctx, d, _lib.SHADOWMAP_R32F).cpu().numpy(), ref)


def test_entities_to_shadow_maps_to_shaded_frame(ctx):
    """The whole producer chain in front of K3, on the GPU and on the oracle: ECS sweep -> cascade caster sets -> caster draws of the four cascades
    -> ShadowCaster's fragment stage (EVSM moments for cascade 0, R16F depth for 1-3) -> EVSM blur of cascade 0 -> the frame shaded with THESE maps.
    Every stage up to the maps is bit-exact, so both shaders see the same maps; the radiance agrees to the shade tolerance."""
    from sailor_amd.forward_plus import ForwardPlus, evsm_blur, upload_lights, upload_shadow_maps
    f = synth.make_frame("tiny_csm")
    cam, W, H, S = f.cam, f.cam.width, f.cam.height, 128
    ents = synth.make_entities(4000)
    ents.transforms[:, 0:3] *= np.float32(0.05)           # pull the cloud into the first cascades so that the boxes shadow the visible surface
    planes, _ = host.extract_frustum_planes(cam.world, cam.aspect, cam.fov, cam.z_near, cam.z_far)
    world, aabb, _ = EcsSweep(ctx, ents).run(planes)
    lms = f.shadows.lights_matrices
    cplanes = np.stack([host.extract_frustum_planes_matrix(lms[k])[0] for k in range(4)])
    masks = csm_caster_masks(ctx, aabb, cplanes).cpu().numpy().view(np.uint64)
    pos, tris = synth.unit_cube_mesh()
    models = synth.caster_models(world.cpu().numpy(), ents.local_aabb)
    d_pos, d_tris, d_models = (torch.from_numpy(pos).to(ctx.device), torch.from_numpy(tris.view(np.int32)).to(ctx.device), torch.from_numpy(models).to(ctx.device))
    # oracle side of the same chain
    ow, oaabb, _ = oracle.ecs_sweep(ents.transforms, ents.parent, ents.local_aabb, planes)
    np.testing.assert_array_equal(masks, oracle.csm_caster_masks(oaabb, cplanes))
    gpu_maps, ref_maps = [], []
    for k in range(4):
        ids = np.nonzero(np.unpackbits(masks[k].view(np.uint8), bitorder="little")[:4000])[0].astype(np.uint32)
        d = raster_depth(ctx, lms[k], d_pos, d_tris, d_models, S, S, torch.from_numpy(ids.view(np.int32)).to(ctx.device))
        ref = oracle.raster_depth(lms[k], pos, tris, models, S, S, instance_ids=ids)
        if k == 0:
            m = evsm_blur(ctx, shadow_resolve(ctx, d, _lib.SHADOWMAP_RGBA32F), 2, 5)
            gpu_maps.append(m.cpu().numpy())
            ref_maps.append(oracle.evsm_blur(oracle.shadow_resolve_evsm(ref), 2, 5))
        else:
            gpu_maps.append(shadow_resolve(ctx, d, _lib.SHADOWMAP_R16F).cpu().numpy())
            ref_maps.append(ref.astype(np.float16))
        np.testing.assert_array_equal(gpu_maps[k].view(np.uint8), ref_maps[k].view(np.uint8))
    assert sum(float((m != 0).mean()) for m in ref_maps) > 0.05
    # shade with the produced maps
    shadows = synth.ShadowSet(lights_matrices=lms, maps=gpu_maps, size=S)
    fp = ForwardPlus(ctx, W, H, len(f.lights))
    lights = upload_lights(f.lights, ctx.device)
    fp.cull(cam.frame, lights, len(f.lights), torch.from_numpy(f.depth).to(ctx.device))
    desc, keep = upload_shadow_maps(shadows, ctx.device)
    got = fp.shade(cam.frame, torch.from_numpy(f.surface).to(ctx.device), lights, len(f.lights), desc).cpu().numpy()
    g, idx, _ = oracle.light_cull(cam.frame, W, H, f.lights, f.depth)
    odesc, okeep = oracle.make_csm(lms, ref_maps)
    ref = oracle.shade(cam.frame, W, H, f.surface, f.lights, g, idx, odesc)
    unshadowed = oracle.shade(cam.frame, W, H, f.surface, f.lights, g, idx, None)
    assert np.abs(ref - unshadowed).max() > 0.5, "the boxes cast shadows on the visible surface"
    err = np.abs(got.astype(np.float64) - ref)
    assert (err <= 1e-4 * np.abs(ref)).all(), err.max()


def test_depth_prepass_feeds_linearize_and_the_light_cull(ctx):
    """The depth producer in front of the path: the visible entities' boxes drawn with the camera's matrices
    (DepthOnly.shader: projection * (view * (model * position))) give the raw reversed-Z attachment; linearised it drives the tile light cull --
    depth, linear depth and per-tile lists all equal the oracle's."""
    from sailor_amd.forward_plus import ForwardPlus, linearize_depth, raster_depth_camera, upload_lights
    f = synth.make_frame("tiny", with_surface=False)
    cam, W, H = f.cam, f.cam.width, f.cam.height
    ents = synth.make_entities(500)
    ents.transforms[:, 0:3] *= np.float32(0.12)
    planes, _ = host.extract_frustum_planes(cam.world, cam.aspect, cam.fov, cam.z_near, cam.z_far)
    world, aabb, vis = EcsSweep(ctx, ents).run(planes)
    ids = np.nonzero(np.unpackbits(vis.cpu().numpy().view(np.uint8), bitorder="little")[:500])[0].astype(np.uint32)
    pos, tris = synth.unit_cube_mesh()
    models = synth.caster_models(world.cpu().numpy(), ents.local_aabb)
    coarse = torch.empty(int(_lib.load().sailor_hip_raster_coarse_words(W, H)), dtype=torch.int32, device=ctx.device)
    raw = raster_depth_camera(ctx, cam.frame, torch.from_numpy(pos).to(ctx.device), torch.from_numpy(tris.view(np.int32)).to(ctx.device),
                              torch.from_numpy(models).to(ctx.device), W, H, torch.from_numpy(ids.view(np.int32)).to(ctx.device), coarse)
    ctx.synchronize()
    fb = np.frombuffer(bytes(cam.frame), np.float32)
    ref_raw = oracle.raster_depth(fb[16:32], pos, tris, models, W, H, instance_ids=ids, view=fb[0:16])
    np.testing.assert_array_equal(raw.cpu().numpy().view(np.uint32), ref_raw.view(np.uint32))
    assert float((ref_raw > 0).mean()) > 0.05 and len(np.unique(ref_raw)) > 1000
    zn = cam.frame.cameraZNearZFar[0]
    lin = linearize_depth(ctx, cam.frame, raw)
    ref_lin = oracle.linearize_depth(zn, ref_raw)
    np.testing.assert_array_equal(lin.cpu().numpy().view(np.uint32), ref_lin.view(np.uint32))
    fp = ForwardPlus(ctx, W, H, len(f.lights))
    fp.cull(cam.frame, upload_lights(f.lights, ctx.device), len(f.lights), raw, _lib.CULL_RAW_DEPTH)
    g, idx = fp.lists_to_host()
    og, oi, _ = oracle.light_cull(cam.frame, W, H, f.lights, ref_lin)
    np.testing.assert_array_equal(g, og)
    np.testing.assert_array_equal(idx[:1 + int(idx[0])], oi[:1 + int(oi[0])])


@pytest.mark.parametrize("size", [(64, 64), (200, 120), (517, 333)])
def test_random_triangle_soup(ctx, size):
    """6 000 random triangles -- slivers, degenerate ones, sub-texel ones, ones larger than the map, partly or wholly outside it and outside the depth
    range, with and without the coarse-depth workspace, in two batches (the second drawn on top of the first): identical to the sequential oracle."""
    W, H = size
    rng = np.random.default_rng(W * 1000 + H)
    n = 6000
    centre = rng.uniform(-1.3, 1.3, (n, 1, 3)).astype(np.float32)
    centre[..., 2] = rng.uniform(-0.2, 1.2, (n, 1)).astype(np.float32)
    extent = (10.0 ** rng.uniform(-3.0, 0.6, (n, 1, 1))).astype(np.float32)
    verts = centre + rng.uniform(-1, 1, (n, 3, 3)).astype(np.float32) * extent * np.float32([1, 1, 0.3])
    verts[::97, 1] = verts[::97, 0]                                   # degenerate: two equal vertices
    verts[5::89, :, 1] = np.float32(0.25)                             # degenerate: zero height
    snap = rng.integers(-W, W, (n // 50, 3, 2)).astype(np.float32)    # vertices exactly on texel centres / edges (fill-rule ties)
    verts[3::50][: len(snap), :, 0] = (snap[: len(verts[3::50]), :, 0] + 0.5) / W * 2 - 1
    pos = np.ascontiguousarray(verts.reshape(-1, 3))
    idx = np.arange(3 * n, dtype=np.uint32).reshape(n, 3)
    one = IDENTITY.reshape(1, 16)
    half = n // 2
    ref = oracle.raster_depth(IDENTITY, pos, idx[:half], one, W, H)
    ref = oracle.raster_depth(IDENTITY, pos, idx[half:], one, W, H, depth=ref)
    assert 0.3 < float((ref > 0).mean()) and len(np.unique(ref)) > 500
    culled = oracle.raster_depth(IDENTITY, pos, idx, one, W, H, cull_back=True)
    got = raster_depth(ctx, IDENTITY, torch.from_numpy(pos).to(ctx.device), torch.from_numpy(idx.view(np.int32).copy()).to(ctx.device),
                       torch.from_numpy(one.copy()).to(ctx.device), W, H, cull_back=True)
    ctx.synchronize()
    np.testing.assert_array_equal(got.cpu().numpy().view(np.uint32), culled.view(np.uint32))
    assert (culled != ref).any(), "half of the random triangles face away"
    for use_coarse in (False, True):
        coarse = torch.empty(int(_lib.load().sailor_hip_raster_coarse_words(W, H)), dtype=torch.int32, device=ctx.device) if use_coarse else None
        d_pos = torch.from_numpy(pos).to(ctx.device)
        d_one = torch.from_numpy(one.copy()).to(ctx.device)
        d = raster_depth(ctx, IDENTITY, d_pos, torch.from_numpy(idx[:half].view(np.int32).copy()).to(ctx.device), d_one, W, H, coarse=coarse)
        d = raster_depth(ctx, IDENTITY, d_pos, torch.from_numpy(idx[half:].view(np.int32).copy()).to(ctx.device), d_one, W, H, depth=d, coarse=coarse)
        ctx.synchronize()
        np.testing.assert_array_equal(d.cpu().numpy().view(np.uint32), ref.view(np.uint32))


def _perspective(width, height, near=0.1, f=1.0):
    """reversed-Z, infinite far plane, looking down -Z: w = -z, ndc z = near / -z (1 at the near plane, 0 at infinity); column-major"""
    m = np.zeros(16, np.float32)
    m[0] = f * height / width
    m[5] = f
    m[2 * 4 + 3] = -1.0
    m[3 * 4 + 2] = near
    return m


@pytest.mark.parametrize("size", [(96, 64), (333, 200)])
def test_triangles_through_the_near_plane_and_behind_the_eye(ctx, size):
    """Perspective soup around the eye: several hundred triangles cross the near plane (z_clip > w_clip), many reach behind the eye (w <= 0) -- cut
    in clip space into one or two triangles, on the GPU exactly as in the oracle; with and without back-face culling and the coarse-depth workspace,
    through the light-matrix entry point and the camera (projection, view) one."""
    from sailor_amd.forward_plus import raster_depth_camera
    W, H = size
    rng = np.random.default_rng(77 + W)
    n = 4000
    centre = rng.uniform(-3, 3, (n, 1, 3)).astype(np.float32)
    centre[..., 2] = rng.uniform(-8.0, 1.5, (n, 1)).astype(np.float32)
    extent = (10.0 ** rng.uniform(-2.0, 0.7, (n, 1, 1))).astype(np.float32)
    verts = centre + rng.uniform(-1, 1, (n, 3, 3)).astype(np.float32) * extent
    verts[::41, :, 2] = np.float32(-0.1)       # exactly on the near plane: d = w - z = 0 counts as inside
    verts[7::53, 0, 2] = np.float32(0.0)       # a vertex at w = 0
    pos = np.ascontiguousarray(verts.reshape(-1, 3))
    idx = np.arange(3 * n, dtype=np.uint32).reshape(n, 3)
    one = IDENTITY.reshape(1, 16)
    P = _perspective(W, H)
    d = verts[..., 2] * -1.0 - 0.1             # w - z_clip per vertex
    crossing = ((d >= 0).any(axis=1) & (d < 0).any(axis=1)).mean()
    assert crossing > 0.05 and (verts[..., 2] > 0).any(axis=1).mean() > 0.1
    d_pos, d_idx, d_one = torch.from_numpy(pos).to(ctx.device), torch.from_numpy(idx.view(np.int32).copy()).to(ctx.device), torch.from_numpy(one.copy()).to(ctx.device)
    for cull in (False, True):
        ref = oracle.raster_depth(P, pos, idx, one, W, H, cull_back=cull)
        assert float((ref > 0).mean()) > 0.5 and float(ref.max()) <= 1.0
        for use_coarse in (False, True):
            coarse = torch.empty(int(_lib.load().sailor_hip_raster_coarse_words(W, H)), dtype=torch.int32, device=ctx.device) if use_coarse else None
            got = raster_depth(ctx, P, d_pos, d_idx, d_one, W, H, coarse=coarse, cull_back=cull)
            ctx.synchronize()
            np.testing.assert_array_equal(got.cpu().numpy().view(np.uint32), ref.view(np.uint32))
    # the same soup through DepthOnly's projection * (view * (model * position)): the tiny frame's camera, the soup moved in front of it
    f = synth.make_frame("tiny", with_surface=False)
    cam = f.cam
    fb = np.frombuffer(bytes(cam.frame), np.float32)
    inv_view = np.linalg.inv(fb[0:16].reshape(4, 4).T.astype(np.float64))
    world = (np.c_[pos.astype(np.float64), np.ones(len(pos))] @ inv_view.T)[:, :3].astype(np.float32)
    ref = oracle.raster_depth(fb[16:32], world, idx, one, cam.width, cam.height, view=fb[0:16])
    assert float((ref > 0).mean()) > 0.3
    got = raster_depth_camera(ctx, cam.frame, torch.from_numpy(world).to(ctx.device), d_idx, d_one, cam.width, cam.height)
    ctx.synchronize()
    np.testing.assert_array_equal(got.cpu().numpy().view(np.uint32), ref.view(np.uint32))


def test_golden_fixture_through_the_c_abi(ctx):
    """tests/golden/tiny_raster.npz: the committed oracle depth images of the near-plane soup, from the GPU, bit for bit."""
    from pathlib import Path
    g = np.load(Path(__file__).resolve().parent / "golden" / "tiny_raster.npz")
    pos, idx = synth.make_triangle_soup(1500)
    P = synth.perspective_reversed_z(96, 64)
    d_pos, d_idx = torch.from_numpy(pos).to(ctx.device), torch.from_numpy(idx.view(np.int32).copy()).to(ctx.device)
    d_one = torch.from_numpy(IDENTITY.reshape(1, 16).copy()).to(ctx.device)
    for key, cull in (("both", False), ("front", True)):
        got = raster_depth(ctx, P, d_pos, d_idx, d_one, 96, 64, cull_back=cull)
        ctx.synchronize()
        np.testing.assert_array_equal(got.cpu().numpy().view(np.uint32), g[key].view(np.uint32))


@pytest.mark.parametrize("cascade, size, order", [(0, 512, "front_to_back"), (0, 512, "back_to_front"), (1, 256, "unsorted"), (3, 256, "front_to_back")])
def test_many_instances_go_out_in_chunks_with_the_instance_test(ctx, cascade, size, order):
    """Round 6: a draw of >= 4 096 instances with a coarse depth goes out in chunks of growing size, and every instance's box is held against the coarse depth
    before any of its triangles is set up (raster.hip: raster_instance_hidden, k_mesh_bounds).  Both are bounds-only: the depth buffer equals the oracle's bit
    for bit in any drawing order -- front to back (where they pay), back to front (where nothing is ever hidden), unsorted -- and equals the buffer of the same
    draw without a coarse depth.  The coarse words stay lower bounds."""
    cam = synth.make_camera(640, 360)
    ents = synth.make_entities(40000)
    planes, _ = host.extract_frustum_planes(cam.world, cam.aspect, cam.fov, cam.z_near, cam.z_far)
    world, aabb, _ = EcsSweep(ctx, ents).run(planes)
    sh = synth.make_shadow_set(cam, 16)
    cplanes = np.stack([host.extract_frustum_planes_matrix(sh.lights_matrices[k])[0] for k in range(4)])
    masks = csm_caster_masks(ctx, aabb, cplanes).cpu().numpy().view(np.uint64)
    ids = np.nonzero(np.unpackbits(masks[cascade].view(np.uint8), bitorder="little")[:40000])[0].astype(np.uint32)
    assert len(ids) >= 8192, len(ids)     # at least three chunks (2 048 + 8 192 + ...)
    pos, tris = synth.unit_cube_mesh()
    models = synth.caster_models(world.cpu().numpy(), ents.local_aabb)
    lm = sh.lights_matrices[cascade]
    if order != "unsorted":
        m = np.asarray(lm, np.float64).reshape(4, 4).T
        z = models[ids].reshape(-1, 4, 4)[:, 3, :3].astype(np.float64) @ m[2, :3] + m[2, 3]
        ids = np.ascontiguousarray(ids[np.argsort(-z if order == "front_to_back" else z, kind="stable")])
    ref = oracle.raster_depth(lm, pos, tris, models, size, size, instance_ids=ids)
    dev = lambda a, t=np.float32: torch.from_numpy(np.ascontiguousarray(a, t)).to(ctx.device)
    d_pos, d_tris, d_models, d_ids = dev(pos), dev(tris.view(np.int32), np.int32), dev(models), dev(ids.view(np.int32), np.int32)
    coarse = torch.empty(int(_lib.load().sailor_hip_raster_coarse_words(size, size)), dtype=torch.int32, device=ctx.device)
    got = raster_depth(ctx, lm, d_pos, d_tris, d_models, size, size, d_ids, coarse=coarse)
    plain = raster_depth(ctx, lm, d_pos, d_tris, d_models, size, size, d_ids)
    ctx.synchronize()
    np.testing.assert_array_equal(got.cpu().numpy().view(np.uint32), ref.view(np.uint32))
    np.testing.assert_array_equal(plain.cpu().numpy().view(np.uint32), ref.view(np.uint32))
    words = coarse.cpu().numpy().view(np.float32)
    n1, n2 = (size // 8) ** 2, (size // 64) ** 2
    assert len(words) == n1 + n2 + 8 + 4 + max(64, 2 * n2) * 24   # two coarse levels, the mesh's box, the giant triangles' queue (count + entries)
    assert (words[:n1].reshape(size // 8, size // 8) <= ref.reshape(size // 8, 8, size // 8, 8).min(axis=(1, 3))).all()
    assert (words[n1:n1 + n2].reshape(size // 64, size // 64) <= ref.reshape(size // 64, 64, size // 64, 64).min(axis=(1, 3))).all()
    np.testing.assert_array_equal(words[n1 + n2:n1 + n2 + 6], np.float32([-1, -1, -1, 1, 1, 1]))   # the unit cube's box, where k_mesh_bounds left it
    queued = int(coarse.cpu().numpy().view(np.uint32)[n1 + n2 + 8])
    print(f"cascade {cascade}, {size}^2, {order}: {len(ids)} instances, {queued} giant triangles queued (capacity {max(64, 2 * n2)})")
    # a dependent pass on top of the finished buffer (no clear): nothing changes, with the coarse depth telling most instances so
    again = raster_depth(ctx, lm, d_pos, d_tris, d_models, size, size, d_ids, depth=got, coarse=coarse)
    ctx.synchronize()
    np.testing.assert_array_equal(again.cpu().numpy().view(np.uint32), ref.view(np.uint32))


def test_many_instances_of_a_general_mesh_through_the_camera(ctx):
    """The instance test with the camera's matrices (projective: the box's corners bound x / w, y / w, z / w only while every corner lies in front of the eye and
    inside the near plane -- instances around the eye are never skipped) and a mesh that is no box, with a vertex no triangle references lying far outside."""
    from sailor_amd.forward_plus import raster_depth_camera
    f = synth.make_frame("tiny", with_surface=False)
    cam, W, H = f.cam, f.cam.width, f.cam.height
    rng = np.random.default_rng(77)
    n = 6000
    pos = np.float32([[0, 1, 0], [-1, -1, 1], [1, -1, 1], [0, -1, -1], [50, 50, 50]])                # a tetrahedron + an unreferenced vertex
    tris = np.uint32([[0, 1, 2], [0, 2, 3], [0, 3, 1], [1, 3, 2]])
    centre = np.stack([rng.uniform(-400, 400, n), rng.uniform(-50, 350, n), rng.uniform(-900, 100, n)], 1)   # the camera sits at (0, 150, 0) looking down -z: some are around it
    models = np.zeros((n, 4, 4), np.float32)
    s = rng.uniform(2, 30, n)
    models[:, 0, 0] = s; models[:, 1, 1] = s; models[:, 2, 2] = s; models[:, 3, 3] = 1; models[:, 3, :3] = centre
    order = np.argsort(np.linalg.norm(centre - np.float64([0, 150, 0]), axis=1))                          # front to back
    ids = order.astype(np.uint32)
    models = models.reshape(n, 16)
    fb = np.frombuffer(bytes(cam.frame), np.float32)
    ref = oracle.raster_depth(fb[16:32], pos, tris, models, W, H, instance_ids=ids, view=fb[0:16])
    dev = lambda a, t=np.float32: torch.from_numpy(np.ascontiguousarray(a, t)).to(ctx.device)
    coarse = torch.empty(int(_lib.load().sailor_hip_raster_coarse_words(W, H)), dtype=torch.int32, device=ctx.device)
    got = raster_depth_camera(ctx, cam.frame, dev(pos), dev(tris.view(np.int32), np.int32), dev(models), W, H, dev(ids.view(np.int32), np.int32), coarse)
    ctx.synchronize()
    np.testing.assert_array_equal(got.cpu().numpy().view(np.uint32), ref.view(np.uint32))
    assert float((ref > 0).mean()) > 0.3
    words = coarse.cpu().numpy().view(np.float32)
    nlev = ((W + 7) // 8) * ((H + 7) // 8) + ((W + 63) // 64) * ((H + 63) // 64)
    np.testing.assert_array_equal(words[nlev:nlev + 6], np.float32([-1, -1, -1, 1, 1, 1]))             # the referenced vertices only
