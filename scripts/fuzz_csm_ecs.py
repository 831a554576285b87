"""Randomised parity sweep, part 2 (run through gpurun): K3 -- random viewports, shadow-map sizes, shadow types, several directional lights, depth
ranges that reach all cascades -- radiance within 1e-4 relative; K4 -- random hierarchies (depth 1..6, ragged level sizes, degenerate scales,
zero-size boxes) -- world matrices / boxes / visibility bit for bit.   usage: fuzz_csm_ecs.py [cases] [seed]"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from oracle import oracle
from sailor_amd import _lib, host, synth
from sailor_amd.forward_plus import HipContext, ForwardPlus, EcsSweep, upload_lights, upload_shadow_maps

cases = int(sys.argv[1]) if len(sys.argv) > 1 else 40
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
ctx = HipContext("cuda:0")
worst = 0.0
for c in range(cases):
    W, H = int(rng.integers(16, 400)), int(rng.integers(16, 260))
    N = int(rng.choice([1, 3, 64, 300, 2000]))
    seed = int(rng.integers(1, 1 << 20))
    f = synth.make_frame("tiny_csm", width=W, height=H, seed=seed, shadow_size=int(rng.choice([2, 3, 17, 64, 96])),
                         lights=synth.LightSetConfig(count=N, spot_fraction=float(rng.choice([0.0, 0.4])), radius_scale=float(rng.choice([1.0, 6.0])),
                                                     cluster_lights=int(rng.choice([0, min(N, 200)])), directional_first=True))
    lights = f.lights
    lights["shadowType"][0] = int(rng.choice([host.SHADOW_NONE, host.SHADOW_PCF, host.SHADOW_EVSM]))
    if N >= 64:
        k = rng.integers(1, N, 3)
        lights["type"][k] = host.LIGHT_DIRECTIONAL
        lights["shadowType"][k] = rng.choice([host.SHADOW_NONE, host.SHADOW_PCF, host.SHADOW_EVSM], 3)
        lights["direction"][k, :3] = rng.normal(size=(3, 3)).astype(np.float32)
    depth = f.depth
    if rng.random() < 0.5:  # stretch the depth so that the far cascades are selected too
        depth = (depth * np.float32(rng.choice([3.0, 8.0]))).astype(np.float32)
        f.surface = synth.make_surface(f.cam, depth, seed)
    og, oi, _ = oracle.light_cull(f.cam.frame, W, H, lights, depth)
    desc, keep = oracle.make_csm(f.shadows.lights_matrices, f.shadows.maps)
    ref = oracle.shade(f.cam.frame, W, H, f.surface, lights, og, oi, desc)
    fp = ForwardPlus(ctx, W, H, N)
    l = upload_lights(lights, ctx.device)
    fp.cull(f.cam.frame, l, N, torch.from_numpy(np.ascontiguousarray(depth)).to(ctx.device))
    gdesc, gkeep = upload_shadow_maps(f.shadows, ctx.device)
    got = fp.shade(f.cam.frame, torch.from_numpy(np.ascontiguousarray(f.surface)).to(ctx.device), l, N, gdesc).cpu().numpy()
    fin = np.isfinite(ref)
    assert np.array_equal(np.isfinite(got), fin), (c, W, H, N, "finiteness")
    err = np.abs(got.astype(np.float64) - ref.astype(np.float64))[fin]
    tol = 1e-4 * np.abs(ref.astype(np.float64))[fin]
    if not (err <= tol).all():
        bad = np.argwhere(np.abs(got.astype(np.float64) - ref) > 1e-4 * np.abs(ref))
        raise SystemExit(f"K3 case {c}: {W}x{H}, {N} lights, shadow size {f.shadows.size}, type {lights['shadowType'][0]}: {len(bad)} values off, first {bad[0]}, got {got[tuple(bad[0][:2])]} ref {ref[tuple(bad[0][:2])]}")
    m = np.abs(ref[fin]) > 0
    if m.any(): worst = max(worst, float((err[m] / np.abs(ref[fin][m])).max()))
print("K3 fuzz ok:", cases, "cases, worst relative radiance error", worst, flush=True)

cam = synth.make_camera(1920, 1080)
planes, _ = host.extract_frustum_planes(cam.world, cam.aspect, cam.fov, cam.z_near, cam.z_far)
for c in range(cases):
    n = int(rng.choice([1, 63, 64, 65, 1000, 4097, 70000]))
    ents = synth.make_entities(n, seed=int(rng.integers(1, 1 << 20)))
    levels = int(rng.integers(1, 7))
    if n > levels * 2:
        cuts = np.sort(rng.choice(np.arange(1, n), levels - 1, replace=False)) if levels > 1 else np.array([], int)
        off = np.concatenate([[0], cuts, [n]]).astype(np.uint32)
        parent = np.full(n, 0xFFFFFFFF, np.uint32)
        for L in range(1, levels):
            lo, hi, plo, phi = off[L], off[L + 1], off[L - 1], off[L]
            parent[lo:hi] = rng.integers(plo, phi, hi - lo).astype(np.uint32)
        ents.parent = parent; ents.level_offsets = off
    k = rng.integers(0, n, max(1, n // 50))
    ents.transforms[k, 8:11] = rng.choice([0.0, -1.0, 1e-20, 1e10], (len(k), 1)).astype(np.float32)   # degenerate / mirrored / huge scales
    ents.local_aabb[k[: len(k) // 2], 3:] = ents.local_aabb[k[: len(k) // 2], :3]                        # zero-size boxes
    ow, ob, ov = oracle.ecs_sweep(ents.transforms, ents.parent, ents.local_aabb, planes)
    sw = EcsSweep(ctx, ents)
    w_, b_, v_ = sw.run(planes)
    ctx.synchronize()
    gw, gb, gv = w_.cpu().numpy(), b_.cpu().numpy(), v_.cpu().numpy().view(np.uint64)
    assert np.array_equal(gw.view(np.uint32), ow.view(np.uint32)), ("K4 world", c, n, levels)
    assert np.array_equal(gb.view(np.uint32), ob.view(np.uint32)), ("K4 boxes", c, n, levels)
    assert np.array_equal(gv, ov), ("K4 visibility", c, n, levels)
print("K4 fuzz ok:", cases, "cases", flush=True)
