"""Randomised parity sweep, part 3 (run through gpurun): the depth rasteriser -- random box / triangle scenes through orthographic light matrices
and the perspective camera path (triangles crossing the near plane, behind the eye, slivers, huge and sub-texel ones), back-face culling on and
off, dependent passes, with and without the coarse-depth scratch -- depth buffers bit for bit; the indirect-draw compaction -- random instance
sets, batch shapes and windows -- every byte of both buffers.   usage: fuzz_raster_meshcull.py [cases] [seed]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ctypes as C
import numpy as np, torch
from oracle import oracle
from sailor_amd import _lib, host, synth
from sailor_amd.forward_plus import HipContext, MeshCull, raster_depth, raster_depth_camera

cases = int(sys.argv[1]) if len(sys.argv) > 1 else 40
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
ctx = HipContext("cuda:0")
dev = ctx.device
t = lambda a, dt=np.float32: torch.from_numpy(np.ascontiguousarray(a, dt)).to(dev)
ti = lambda a: torch.from_numpy(np.ascontiguousarray(a, np.uint32).view(np.int32)).to(dev)
cube_pos, cube_tris = synth.unit_cube_mesh()
for c in range(cases):
    W, H = int(rng.integers(1, 200)), int(rng.integers(1, 200))
    n = int(rng.choice([1, 2, 30, 400]))
    if c % 5 == 4:   # (round 6) draws of >= 4 096 instances on larger maps: chunked launches, the instance test, the giant triangles' queue and its overflow
        W, H = int(rng.integers(100, 700)), int(rng.integers(100, 700))
        n = int(rng.choice([4096, 5000, 9000]))
    cam = synth.make_camera(max(W, 16), max(H, 16))
    cull_back = bool(rng.integers(0, 2))
    if rng.random() < 0.5:
        pos, tris = cube_pos, cube_tris
    else:   # free triangles: slivers, huge ones, degenerate ones
        m = int(rng.integers(1, 40))
        pos = (rng.normal(size=(3 * m, 3)) * rng.choice([0.01, 1.0, 30.0])).astype(np.float32)
        if m > 1 and rng.random() < 0.3: pos[3:6] = pos[3]   # a degenerate triangle
        tris = np.arange(3 * m, dtype=np.uint32).reshape(m, 3)
    models = np.zeros((n, 16), np.float32)
    sc = rng.choice([0.02, 1.0, 20.0, 300.0], n).astype(np.float32) * (0.5 + rng.random(n)).astype(np.float32)
    models[:, 0] = sc; models[:, 5] = sc * rng.choice([1.0, -1.0], n); models[:, 10] = sc; models[:, 15] = 1.0
    ids = None if rng.random() < 0.5 else rng.integers(0, n, int(rng.integers(1, 2 * n + 1))).astype(np.uint32)
    if rng.random() < 0.5:
        # camera path: instances scattered in front of, around and behind the eye (near-plane clipping, w <= 0)
        models[:, 12:15] = (rng.normal(size=(n, 3)) * 400.0 + np.array([0.0, 150.0, -300.0])).astype(np.float32)
        ref = oracle.raster_depth(np.array(cam.frame.projection, np.float32), pos, tris, models, W, H, instance_ids=ids,
                                  view=np.array(cam.frame.view, np.float32), cull_back=cull_back)
        ccoarse = torch.zeros(int(ctx._lib.sailor_hip_raster_coarse_words(W, H)), dtype=torch.int32, device=dev) if n >= 4096 else None
        got = raster_depth_camera(ctx, cam.frame, t(pos), ti(tris), t(models), W, H, None if ids is None else ti(ids), ccoarse, cull_back=cull_back)
    else:
        sh = synth.make_shadow_set(cam, 2, int(rng.integers(1, 1000)))
        lm = sh.lights_matrices[int(rng.integers(0, 4))]
        models[:, 12:15] = (rng.normal(size=(n, 3)) * 600.0 + np.array([0.0, 150.0, -600.0])).astype(np.float32)
        base = None
        if rng.random() < 0.4:   # a dependent pass on top of an earlier one
            base = oracle.raster_depth(lm, cube_pos, cube_tris, models[:1] * 1.0, W, H)
        ref = oracle.raster_depth(lm, pos, tris, models, W, H, instance_ids=ids, depth=base, cull_back=cull_back)
        coarse = None
        if (rng.random() < 0.5 and base is None) or (n >= 4096 and rng.random() < 0.8):   # (a dependent pass with a coarse depth: the bounds start from zero -- valid, merely loose)
            words = ctx._lib.sailor_hip_raster_coarse_words(W, H)
            coarse = torch.zeros(int(words), dtype=torch.int32, device=dev)
        got = raster_depth(ctx, lm, t(pos), ti(tris), t(models), W, H, None if ids is None else ti(ids), None if base is None else t(base), coarse, cull_back)
    ctx.synchronize()
    g = got.cpu().numpy()
    if not np.array_equal(g.view(np.uint32), ref.view(np.uint32)):
        bad = np.argwhere(g.view(np.uint32) != ref.view(np.uint32))
        raise SystemExit(f"raster case {c}: {W}x{H}, {n} instances, {len(tris)} triangles, cull_back {cull_back}: {len(bad)} texels differ, first {bad[0]}: got {g[tuple(bad[0])]} ref {ref[tuple(bad[0])]}")
print("raster fuzz ok:", cases, "cases", flush=True)

for c in range(cases):
    cam = synth.make_camera(int(rng.integers(64, 2000)), int(rng.integers(64, 1200)))
    n = int(rng.choice([1, 255, 256, 257, 3000, 40000]))
    nb = int(rng.choice([1, 2, 17, 300, min(n, 2000)]))
    first = int(rng.choice([0, 1, 37]))
    s = synth.make_instance_set(n, nb, seed=int(rng.integers(1, 1 << 20)), first_instance=first, spread=float(rng.choice([300.0, 3000.0, 30000.0])))
    mc = MeshCull(ctx, s.instances, s.batches)
    mc.run(cam.frame, n, first)
    gi, gb = mc.download()
    ri, rb = oracle.mesh_cull_compact(cam.frame, s.instances, n, first, s.batches)
    assert np.array_equal(gb, rb), ("mesh cull batches", c, n, nb, first)
    assert np.array_equal(gi.view(np.uint32).reshape(-1, 24), ri.view(np.uint32).reshape(-1, 24)), ("mesh cull instances", c, n, nb, first)
    mc.run(cam.frame, n, first)   # a second frame over the already compacted buffers
    gi, gb = mc.download()
    ri, rb = oracle.mesh_cull_compact(cam.frame, ri, n, first, rb)
    assert np.array_equal(gb, rb) and np.array_equal(gi.view(np.uint32).reshape(-1, 24), ri.view(np.uint32).reshape(-1, 24)), ("mesh cull second frame", c, n, nb, first)
print("mesh cull fuzz ok:", cases, "cases", flush=True)
