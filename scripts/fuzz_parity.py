"""Randomised parity sweep (run through gpurun; not part of the test suite -- minutes of oracle time): random small frames, light sets
(radii, spot share, clusters, directional / NaN / behind-the-eye lights, roughness 0), every cull path and random bands, against the C oracle:
lists bit for bit, radiance within 1e-4 relative.   usage: fuzz_parity.py [cases] [seed] [only this case: the others only draw their random numbers]"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from oracle import oracle
from sailor_amd import _lib, host, synth
from sailor_amd.forward_plus import HipContext, ForwardPlus, upload_lights

cases = int(sys.argv[1]) if len(sys.argv) > 1 else 60
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
only = int(sys.argv[3]) if len(sys.argv) > 3 else None
ctx = HipContext("cuda:0")
paths = [_lib.CULL_DEFAULT, _lib.CULL_BRUTE_FORCE, _lib.CULL_INTERVAL_MASKS]
worst = 0.0
worst_case = -1
t0 = time.time()
for c in range(cases):
    W, H = int(rng.integers(16, 500)), int(rng.integers(16, 300))
    N = int(rng.choice([0, 1, 63, 64, 65, 500, 1500, 4000]))
    seed = int(rng.integers(1, 1 << 20))
    cfg = synth.LightSetConfig(count=N, spot_fraction=float(rng.choice([0.0, 0.3, 1.0])), radius_scale=float(rng.choice([0.3, 2.0, 6.0, 20.0])),
                               cluster_lights=int(rng.choice([0, 0, min(N, 300)])), cluster_count=int(rng.integers(1, 3)))
    cam = synth.make_camera(W, H)
    depth = synth.make_linear_depth(W, H, seed)
    lights = synth.make_lights(cam, depth, cfg, seed)
    if N >= 63:
        k = rng.integers(0, N, 6)
        lights["type"][k[:2]] = host.LIGHT_DIRECTIONAL
        if rng.random() < 0.3: lights["worldPosition"][k[2]] = np.nan
        if rng.random() < 0.3: lights["intensity"][k[3], 0] = np.inf
        if rng.random() < 0.5: lights["worldPosition"][k[4]] = (0.0, 150.0, 50.0); lights["bounds"][k[4], 0] = 400.0   # around / behind the eye
        if rng.random() < 0.3: lights["bounds"][k[5], 0] = -5.0
    if rng.random() < 0.2:
        depth = depth.copy(); depth[: H // 3] = np.inf   # sky: NaN frustum centres
    surface = synth.make_surface(cam, np.where(np.isfinite(depth), depth, 1000.0).astype(np.float32), seed)
    if rng.random() < 0.3:
        surface[1, :, ::7, 3] = 0.0   # roughness 0 pixels: 0 / 0 in NdfGGX, must see every light
    Tx, Ty = host.num_tiles(W, H)
    if only is not None and c != only:
        for flags in paths:
            rng.integers(0, Ty + 1); rng.random()
        continue
    og, oi, _ = oracle.light_cull(cam.frame, W, H, lights, depth)
    orad = oracle.shade(cam.frame, W, H, surface, lights, og, oi, None)
    for flags in paths:
        cut = int(rng.integers(0, Ty + 1))
        bands = [None] if rng.random() < 0.5 or Ty < 2 or cut in (0, Ty) else [host.band_from_tile_rows(W, H, 0, cut), host.band_from_tile_rows(W, H, cut, Ty)]
        base = 0
        for b in bands:
            fp = ForwardPlus(ctx, W, H, max(N, 1), band=b)
            bb = fp.band
            rows = slice(bb.fbRowBegin, bb.fbRowBegin + bb.fbRowCount)
            d = torch.from_numpy(np.ascontiguousarray(depth[rows])).to(ctx.device)
            s = torch.from_numpy(np.ascontiguousarray(surface[:, rows])).to(ctx.device)
            l = upload_lights(lights, ctx.device)
            fp.cull(cam.frame, l, N, d, flags)
            g, idx = fp.lists_to_host()
            t0r, t1r = bb.tileRowBegin * Tx, bb.tileRowEnd * Tx
            assert np.array_equal(g[:, 1], og[t0r:t1r, 1]), (c, W, H, N, flags, "num")
            for t in range(t1r - t0r):
                assert np.array_equal(idx[g[t, 0]: g[t, 0] + g[t, 1]], oi[og[t0r + t, 0]: og[t0r + t, 0] + og[t0r + t, 1]]), (c, W, H, N, flags, t)
            out = fp.shade(cam.frame, s, l, N, None)
            ctx.synchronize()
            got = out.cpu().numpy()
            ref = orad[rows]
            fin = np.isfinite(ref)
            if not np.array_equal(np.isfinite(got), fin):
                bad = np.argwhere(np.isfinite(got) != fin)
                print("finiteness differs at", len(bad), "values; first:", bad[:5].tolist())
                y, x, ch = bad[0]
                gy = H - 1 - (y + rows.start); t = (gy // 16) * Tx + x // 16
                print(" got", got[y, x], "ref", ref[y, x], "surface", surface[:, y + rows.start, x], "tile", t, "list", oi[og[t, 0]: og[t, 0] + og[t, 1]][:20])
                li = oi[og[t, 0]: og[t, 0] + og[t, 1]]
                print(" list light types", lights["type"][li][:20], "intensity finite", np.isfinite(lights["intensity"][li]).all(1)[:20], "pos finite", np.isfinite(lights["worldPosition"][li]).all(1)[:20], "radius", lights["bounds"][li, 0][:20])
                raise SystemExit(f"case {c}: {W}x{H}, {N} lights, flags {flags}")
            err = np.abs(got.astype(np.float64) - ref.astype(np.float64))[fin]
            tol = 1e-4 * np.abs(ref.astype(np.float64))[fin]
            # split tiles of a band differ from the one-block form by the order of four partial sums: same tolerance, checked the same way
            if only is not None:
                full_err = np.where(fin, np.abs(got.astype(np.float64) - ref.astype(np.float64)) / (np.abs(ref.astype(np.float64)) + 1e-300), 0.0)
                y, x, ch = np.unravel_index(np.argmax(full_err), full_err.shape)
                gy = H - 1 - (y + rows.start); t = (gy // 16) * Tx + x // 16
                li = oi[og[t, 0]: og[t, 0] + og[t, 1]]
                print(f"flags {flags} band {None if b is None else (bb.tileRowBegin, bb.tileRowEnd)}: worst rel {full_err.max():.3e} at pixel ({x},{y + rows.start}) ch {ch}: got {got[y, x]} ref {ref[y, x]}")
                print("  surface", surface[:, y + rows.start, x].tolist(), "list", li.tolist())
                print("  types", lights["type"][li].tolist(), "radius", lights["bounds"][li, 0].tolist(), "pos", lights["worldPosition"][li].tolist(), "intensity", lights["intensity"][li].tolist())
            assert (err <= tol * (2.0 if b is not None else 1.0)).all(), (c, W, H, N, flags, float((err / (tol + 1e-300)).max()))
            m = np.abs(ref[fin]) > 0
            if m.any():
                w = float((err[m] / np.abs(ref[fin][m])).max())
                if w > worst: worst, worst_case = w, c
    if c % 10 == 9: print(f"{c + 1} cases ok, worst relative radiance error {worst:.2e}, {time.time() - t0:.0f} s", flush=True)
print("fuzz ok:", cases, "cases, worst relative radiance error", worst, "in case", worst_case)
