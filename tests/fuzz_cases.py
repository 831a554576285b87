"""Randomised parity cases shared by tests/test_fuzz_gpu.py (a bounded, fixed-seed slice inside `-m gpu`) and the scripts/fuzz_*.py sweeps
(thousands of cases through gpurun).  Every case draws a random small problem, runs it through the C-ABI and holds it against the C oracle:
  k1k2_case : viewport, light set (radii, spot share, clusters, directional / NaN / infinite / behind-the-eye lights, negative radii), sky tiles,
              roughness-0 pixels; every cull path, random two-band splits -- lists bit for bit, radiance within 1e-4 relative, same finiteness
  k3_case   : viewport, shadow-map size and type, several directional lights, depth ranges that reach all cascades -- radiance within 1e-4 relative
  k4_case   : hierarchies (depth 1..6, ragged levels, degenerate / mirrored / huge scales, zero-size boxes) -- matrices / boxes / visibility bit for bit
A failure names the seed and the case so that `scripts/fuzz_parity.py <cases> <seed> <case>` replays it."""
import numpy as np
import torch

from oracle import oracle
from sailor_amd import _lib, host, synth
from sailor_amd.forward_plus import EcsSweep, ForwardPlus, PreparedLights, upload_lights, upload_shadow_maps

CULL_PATHS = [_lib.CULL_DEFAULT, _lib.CULL_BRUTE_FORCE, _lib.CULL_INTERVAL_MASKS]
BAND_SELECT_EVERY = 2   # every second case's bands go through the band selection as well (round 4: the band-local light selection, forced on these small sets)



def nonfinite_mismatch(got, ref):
    """the non-finite values compared BY CLASS -- a NaN where the oracle has an infinity, or an infinity of the other sign, is a mismatch
    (np.isfinite alone lets both through).  -> None, or (class name, index array of the first mismatching values)"""
    for name, fn in (("NaN", np.isnan), ("+Inf", np.isposinf), ("-Inf", np.isneginf)):
        a, b = fn(got), fn(ref)
        if not np.array_equal(a, b):
            return name, np.argwhere(a != b)
    return None


def finite_abs_diff(got, ref, fin):
    """|got - ref| in float64 where `fin` (both finite), 0 elsewhere: no Inf - Inf on the way (the RuntimeWarning the old form printed came from exactly that)"""
    full = np.zeros(ref.shape, np.float64)
    full[fin] = np.abs(got[fin].astype(np.float64) - ref[fin].astype(np.float64))
    return full


def k1k2_case(ctx, rng, c, run=True, verbose=False):
    """one K1 + K2 case; run=False only draws the case's random numbers (replaying a later case of the same seed).  Returns the worst relative
    radiance error over the case's finite, non-zero values."""
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
        if rng.random() < 0.3:   # divisors far outside the staged reciprocal's range (shaded per pixel, behind the pair queue; splits bands' long tiles too)
            j = rng.integers(0, N, 2)
            lights["type"][j[0]] = host.LIGHT_POINT; lights["bounds"][j[0], 0] = np.float32(rng.choice([1e15, 1e-15]))
            lights["type"][j[1]] = host.LIGHT_SPOT; lights["cutOff"][j[1], 0] = lights["cutOff"][j[1], 1] - np.float32(rng.choice([0.0, 1e-15]))
    if rng.random() < 0.2:
        depth = depth.copy(); depth[: H // 3] = np.inf   # sky: NaN frustum centres
    surface = synth.make_surface(cam, np.where(np.isfinite(depth), depth, 1000.0).astype(np.float32), seed)
    if rng.random() < 0.3:
        surface[1, :, ::7, 3] = 0.0   # roughness 0 pixels: 0 / 0 in NdfGGX, must see every light
    Tx, Ty = host.num_tiles(W, H)
    if not run:
        for flags in CULL_PATHS:
            rng.integers(0, Ty + 1); rng.random()
        return 0.0
    what = f"case {c}: {W}x{H}, {N} lights"
    worst = 0.0
    og, oi, _ = oracle.light_cull(cam.frame, W, H, lights, depth)
    orad = oracle.shade(cam.frame, W, H, surface, lights, og, oi, None)
    for flags in CULL_PATHS:
        cut = int(rng.integers(0, Ty + 1))
        bands = [None] if rng.random() < 0.5 or Ty < 2 or cut in (0, Ty) else [host.band_from_tile_rows(W, H, 0, cut), host.band_from_tile_rows(W, H, cut, Ty)]
        for b in bands:
            fp = ForwardPlus(ctx, W, H, max(N, 1), band=b)
            bb = fp.band
            rows = slice(bb.fbRowBegin, bb.fbRowBegin + bb.fbRowCount)
            d = torch.from_numpy(np.ascontiguousarray(depth[rows])).to(ctx.device)
            s = torch.from_numpy(np.ascontiguousarray(surface[:, rows])).to(ctx.device)
            l = upload_lights(lights, ctx.device)
            # every other cull path through the prepared-lights entry points (the path the HIP backend drives), with a capacity above the count
            prep = PreparedLights(ctx, l, N, capacity=N + 5) if (CULL_PATHS.index(flags) + c) % 2 == 0 else None
            fp.prepared = prep
            fp.cull(cam.frame, l, N, d, flags | (_lib.CULL_BAND_SELECT if (b is not None and c % BAND_SELECT_EVERY == 0) else 0))
            g, idx = fp.lists_to_host()
            t0r, t1r = bb.tileRowBegin * Tx, bb.tileRowEnd * Tx
            assert np.array_equal(g[:, 1], og[t0r:t1r, 1]), (what, flags, "num")
            for t in range(t1r - t0r):
                assert np.array_equal(idx[g[t, 0]: g[t, 0] + g[t, 1]], oi[og[t0r + t, 0]: og[t0r + t, 0] + og[t0r + t, 1]]), (what, flags, "tile", t)
            out = fp.shade(cam.frame, s, l, N, None)
            ctx.synchronize()
            got = out.cpu().numpy()
            ref = orad[rows]
            fin = np.isfinite(ref)
            mism = nonfinite_mismatch(got, ref)
            if mism is not None:
                cls, bad = mism
                y, x, ch = bad[0]
                gy = H - 1 - (y + rows.start); t = (gy // 16) * Tx + x // 16
                li = oi[og[t, 0]: og[t, 0] + og[t, 1]]
                raise AssertionError(f"{what}, flags {flags}: the {cls} masks differ at {len(bad)} values; first {bad[0].tolist()}: got {got[y, x]} ref {ref[y, x]} "
                                     f"surface {surface[:, y + rows.start, x].tolist()} tile {t} list {li[:20].tolist()} types {lights['type'][li][:20].tolist()}")
            full = finite_abs_diff(got, ref, fin)
            err = full[fin]
            tol = 1e-4 * np.abs(ref.astype(np.float64))[fin]
            if verbose:
                full_err = np.where(fin, full / (np.abs(ref.astype(np.float64)) + 1e-300), 0.0)
                y, x, ch = np.unravel_index(np.argmax(full_err), full_err.shape)
                gy = H - 1 - (y + rows.start); t = (gy // 16) * Tx + x // 16
                li = oi[og[t, 0]: og[t, 0] + og[t, 1]]
                print(f"flags {flags} band {None if b is None else (bb.tileRowBegin, bb.tileRowEnd)}: worst rel {full_err.max():.3e} at pixel ({x},{y + rows.start}) ch {ch}: got {got[y, x]} ref {ref[y, x]}")
                print("  surface", surface[:, y + rows.start, x].tolist(), "list", li.tolist())
                print("  types", lights["type"][li].tolist(), "radius", lights["bounds"][li, 0].tolist(), "pos", lights["worldPosition"][li].tolist(), "intensity", lights["intensity"][li].tolist())
            # (a band's split tiles add four partial sums in a fixed order: the same 1e-4 against the oracle as everything else -- VERDICT r03 item 3)
            assert (err <= tol).all(), (what, flags, "radiance", float((err / (tol + 1e-300)).max()))
            m = np.abs(ref[fin]) > 0
            if m.any():
                worst = max(worst, float((err[m] / np.abs(ref[fin][m])).max()))
    return worst


def k3_case(ctx, rng, c):
    """one K3 case (cull + shade with cascaded shadow maps); returns the worst relative radiance error"""
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
    if N >= 3 and rng.random() < 0.3:   # divisors far outside the staged reciprocal's range: these lights are shaded per pixel, beside the directional ones
        k = rng.integers(1, N, 2)
        lights["type"][k[0]] = host.LIGHT_POINT; lights["bounds"][k[0], 0] = np.float32(rng.choice([1e15, 1e-15]))
        lights["type"][k[1]] = host.LIGHT_SPOT; lights["cutOff"][k[1], 0] = lights["cutOff"][k[1], 1] - np.float32(rng.choice([0.0, 1e-15]))
    depth = f.depth
    if rng.random() < 0.5:  # stretch the depth so that the far cascades are selected too
        depth = (depth * np.float32(rng.choice([3.0, 8.0]))).astype(np.float32)
        f.surface = synth.make_surface(f.cam, depth, seed)
    og, oi, _ = oracle.light_cull(f.cam.frame, W, H, lights, depth)
    desc, keep = oracle.make_csm(f.shadows.lights_matrices, f.shadows.maps)
    ref = oracle.shade(f.cam.frame, W, H, f.surface, lights, og, oi, desc)
    l = upload_lights(lights, ctx.device)
    fp = ForwardPlus(ctx, W, H, N, prepared=PreparedLights(ctx, l, N) if c % 2 == 0 else None)
    fp.cull(f.cam.frame, l, N, torch.from_numpy(np.ascontiguousarray(depth)).to(ctx.device))
    gdesc, gkeep = upload_shadow_maps(f.shadows, ctx.device)
    got = fp.shade(f.cam.frame, torch.from_numpy(np.ascontiguousarray(f.surface)).to(ctx.device), l, N, gdesc).cpu().numpy()
    what = f"K3 case {c}: {W}x{H}, {N} lights, shadow size {f.shadows.size}, type {lights['shadowType'][0]}"
    fin = np.isfinite(ref)
    assert nonfinite_mismatch(got, ref) is None, (what, "non-finite classes", nonfinite_mismatch(got, ref)[0])
    err = finite_abs_diff(got, ref, fin)[fin]
    tol = 1e-4 * np.abs(ref.astype(np.float64))[fin]
    if not (err <= tol).all():
        bad = np.argwhere(np.abs(got.astype(np.float64) - ref) > 1e-4 * np.abs(ref))
        raise AssertionError(f"{what}: {len(bad)} values off, first {bad[0]}, got {got[tuple(bad[0][:2])]} ref {ref[tuple(bad[0][:2])]}")
    m = np.abs(ref[fin]) > 0
    worst = float((err[m] / np.abs(ref[fin][m])).max()) if m.any() else 0.0
    # every second case also as two bands of a split frame (round 4: a band's long tiles go to the split blocks under shadow maps too -- k2_shade_band_csm*)
    Tx, Ty = host.num_tiles(W, H)
    if c % 2 == 1 and Ty >= 2:
        cut = int(rng.integers(1, Ty))
        for b in (host.band_from_tile_rows(W, H, 0, cut), host.band_from_tile_rows(W, H, cut, Ty)):
            fpb = ForwardPlus(ctx, W, H, N, band=b, prepared=PreparedLights(ctx, l, N) if c % 4 == 1 else None)
            rows = slice(b.fbRowBegin, b.fbRowBegin + b.fbRowCount)
            fpb.cull(f.cam.frame, l, N, torch.from_numpy(np.ascontiguousarray(depth[rows])).to(ctx.device))
            gb = fpb.shade(f.cam.frame, torch.from_numpy(np.ascontiguousarray(f.surface[:, rows])).to(ctx.device), l, N, gdesc).cpu().numpy()
            rb = ref[rows]
            fb = np.isfinite(rb)
            assert nonfinite_mismatch(gb, rb) is None, (what, "band", (b.tileRowBegin, b.tileRowEnd), "non-finite classes", nonfinite_mismatch(gb, rb)[0])
            eb = finite_abs_diff(gb, rb, fb)[fb]
            assert (eb <= 1e-4 * np.abs(rb.astype(np.float64))[fb]).all(), (what, "band", (b.tileRowBegin, b.tileRowEnd), float(eb.max()))
            mb = np.abs(rb[fb]) > 0
            if mb.any():
                worst = max(worst, float((eb[mb] / np.abs(rb[fb][mb])).max()))
    return worst


_K4_PLANES = None


def k4_case(ctx, rng, c):
    """one K4 case: a random hierarchy through the ECS sweep, bit for bit"""
    global _K4_PLANES
    if _K4_PLANES is None:
        cam = synth.make_camera(1920, 1080)
        _K4_PLANES, _ = host.extract_frustum_planes(cam.world, cam.aspect, cam.fov, cam.z_near, cam.z_far)
    planes = _K4_PLANES
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
