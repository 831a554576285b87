"""Randomised parity sweep, part 4 (run through gpurun): the "next" rows -- LinearizeDepth (bit-exact, incl. raw 0 and denormal depths), the EVSM
blur (random sizes and radius pairs, bit-exact), the Hi-Z pyramid (random, ragged sizes, bit-exact) and the occlusion test + compaction against
it (bit-exact), the ambient / IBL term (random cubemap and table sizes, AO on / off, 1e-4 relative).   usage: fuzz_next_rows.py [cases] [seed]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from oracle import oracle
from sailor_amd import _lib, host, synth
from sailor_amd.forward_plus import HipContext, ForwardPlus, MeshCull, evsm_blur, hiz_build, linearize_depth, upload_ibl, upload_lights

cases = int(sys.argv[1]) if len(sys.argv) > 1 else 30
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
ctx = HipContext("cuda:0")
dev = ctx.device
for c in range(cases):
    w, h = int(rng.integers(1, 700)), int(rng.integers(1, 400))
    cam = synth.make_camera(max(w, 16), max(h, 16))
    raw = rng.random((h, w)).astype(np.float32) ** np.float32(rng.choice([1.0, 8.0, 40.0]))
    raw[rng.random((h, w)) < 0.05] = 0.0
    raw[rng.random((h, w)) < 0.01] = np.float32(1e-42)
    got = linearize_depth(ctx, cam.frame, torch.from_numpy(raw).to(dev)).cpu().numpy()
    with np.errstate(all="ignore"):
        ref = oracle.linearize_depth(cam.frame.cameraZNearZFar[0], raw)
    assert np.array_equal(got.view(np.uint32), ref.view(np.uint32)), ("linearize", c, w, h)
print("linearize fuzz ok:", cases, flush=True)

for c in range(cases):
    w, h = int(rng.integers(1, 300)), int(rng.integers(1, 300))
    m = (rng.random((h, w, 4)).astype(np.float32) * np.float32(rng.choice([1.0, 1e6, 1e-6])))
    ru, rp = int(rng.integers(0, 15)), int(rng.integers(0, 15))
    got = evsm_blur(ctx, torch.from_numpy(m.copy()).to(dev), ru, rp).cpu().numpy()
    assert np.array_equal(got.view(np.uint32), oracle.evsm_blur(m, ru, rp).view(np.uint32)), ("blur", c, w, h, ru, rp)
print("blur fuzz ok:", cases, flush=True)

for c in range(cases):
    W, H = int(rng.integers(64, 1500)), int(rng.integers(64, 900))
    cam = synth.make_camera(W, H)
    dw, dh = max(1, W // 2), max(1, H // 2)
    lin = synth.make_linear_depth(dw, dh, int(rng.integers(1, 1000)), d_min=200.0, d_max=2500.0)
    raw = synth.make_raw_depth(lin, cam.frame.cameraZNearZFar[0])
    pw = ph = int(rng.choice([dw, max(dw, dh), max(1, dw // 2) + 1]))
    levels = int(rng.integers(1, int(np.floor(np.log2(max(pw, ph)))) + 2))
    got = hiz_build(ctx, torch.from_numpy(raw).to(dev), pw, ph, levels)
    ref = oracle.hiz_build(raw, pw, ph, levels)
    assert np.array_equal(got.cpu().numpy().view(np.uint32), ref.view(np.uint32)), ("hiz", c, W, H, pw, ph, levels)
    n, nb, first = int(rng.choice([1, 300, 5000, 30000])), int(rng.choice([1, 7, 200])), int(rng.choice([0, 5]))
    s = synth.make_instance_set(n, nb, seed=int(rng.integers(1, 1 << 20)), first_instance=first)
    mc = MeshCull(ctx, s.instances, s.batches)
    mc.run(cam.frame, n, first, hiz=(got, pw, ph, levels))
    gi, gb = mc.download()
    ri, rb = oracle.mesh_cull_compact(cam.frame, s.instances, n, first, s.batches, hiz=(ref, pw, ph, levels))
    assert np.array_equal(gb, rb) and np.array_equal(gi.view(np.uint32).reshape(-1, 24), ri.view(np.uint32).reshape(-1, 24)), ("occlusion", c, W, H, n, nb)
print("hi-z + occlusion fuzz ok:", cases, flush=True)

worst = 0.0
for c in range(cases):
    W, H = int(rng.integers(16, 300)), int(rng.integers(16, 200))
    N = int(rng.choice([0, 5, 400]))
    seed = int(rng.integers(1, 1 << 20))
    cam = synth.make_camera(W, H)
    depth = synth.make_linear_depth(W, H, seed)
    lights = synth.make_lights(cam, depth, synth.LightSetConfig(count=N, radius_scale=5.0, spot_fraction=0.3), seed)
    surface = synth.make_surface(cam, depth, seed)
    lw, lh = int(rng.integers(2, 40)), int(rng.integers(2, 40))
    ibl = synth.make_ibl_set(W, H, oracle.compute_brdf_lut(lw, lh), env_size=int(rng.choice([1, 2, 8, 32, 64])), irr_size=int(rng.choice([1, 4, 16])), seed=seed,
                             with_ao=bool(rng.integers(0, 2)))
    g, idx, _ = oracle.light_cull(cam.frame, W, H, lights, depth)
    oibl, _k = oracle.make_ibl(ibl.irradiance, ibl.env_chain, ibl.env_size, ibl.env_levels, ibl.brdf_lut, ibl.ao)
    ref = oracle.shade(cam.frame, W, H, surface, lights, g, idx, ibl=oibl)
    fp = ForwardPlus(ctx, W, H, max(N, 1))
    l = upload_lights(lights, dev)
    fp.cull(cam.frame, l, N, torch.from_numpy(depth).to(dev))
    desc, keep = upload_ibl(ibl, dev)
    got = fp.shade(cam.frame, torch.from_numpy(surface).to(dev), l, N, None, ibl=desc).cpu().numpy()
    err = np.abs(got.astype(np.float64) - ref.astype(np.float64))
    assert np.isfinite(got).all() and (err <= 1e-4 * np.abs(ref.astype(np.float64))).all(), ("ambient", c, W, H, N, ibl.env_size, float(err.max()))
    worst = max(worst, float((err / (np.abs(ref) + 1e-30)).max()))
print("ambient fuzz ok:", cases, "worst relative", worst, flush=True)
