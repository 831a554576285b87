"""Trip-count model of k2_shade on the C3 frame (CPU, numpy): for sampled tile rows, per 8x8 quadrant: list length, lights that survive
the quadrant-box test, lights that reach >= 1 pixel, lights with >= 1 queued (reach & facing) pair, queued pairs, pair passes.
Feeds the VALU budget in DESIGN.md (static instruction counts x these trips).  Usage: python scripts/analysis/shade_trips.py [rows...]"""
import sys, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from oracle import oracle
from sailor_amd import synth

cfg = synth.CONFIGS["C3"]
cam = synth.make_camera(cfg["width"], cfg["height"])
depth = synth.make_linear_depth(cam.width, cam.height)
lights = synth.make_lights(cam, depth, cfg["lights"])
W, H = cam.width, cam.height
Tx = W // 16
rows = [int(a) for a in sys.argv[1:]] or [10, 40, 67, 77, 100, 125]
stats = []
alt = []
for tr in rows:
    g, idx, _ = oracle.light_cull(cam.frame, W, H, lights, depth, tile_rows=(tr, tr + 1))
    r0, r1 = H - 16 * (tr + 1), H - 16 * tr
    surf = synth.make_surface(cam, depth, row_begin=r0, row_end=r1)     # [3, 16, W, 4]
    pos, nrm = surf[0, :, :, :3].astype(np.float64), surf[1, :, :, :3].astype(np.float64)
    for tx in range(Tx):
        off, num = int(g[tx, 0]), int(g[tx, 1])
        L = lights[idx[off:off + num]]
        lp = L["worldPosition"].astype(np.float64); lr = L["bounds"][:, 0].astype(np.float64); ltype = L["type"]
        ldir = -L["direction"].astype(np.float64)
        ldn = ldir / np.linalg.norm(ldir, axis=1, keepdims=True)
        cut_y = L["cutOff"][:, 1].astype(np.float64)
        for q in range(4):
            # tile row tr covers fb rows r0..r1; shader rows count from the bottom: gy = ty*16 + (q>>1)*8 + ly, py = H-1-gy
            ys = [(r1 - 1) - ((q >> 1) * 8 + ly) - r0 for ly in range(8)]
            xs = [tx * 16 + (q & 1) * 8 + lx for lx in range(8)]
            P = pos[np.ix_(ys, xs)].reshape(64, 3); Nn = nrm[np.ix_(ys, xs)].reshape(64, 3)
            bmin, bmax = P.min(0), P.max(0)
            ex = np.maximum(np.maximum(bmin - lp, lp - bmax), 0.0)
            box_ok = ((ex ** 2).sum(1) <= lr ** 2 * 1.0001) | (ltype != 1)
            d = lp[:, None, :] - P[None, :, :]                       # [num, 64, 3]
            d2 = (d ** 2).sum(2)
            reach_pt = d2 <= (lr ** 2)[:, None]
            theta = (-(d * ldn[:, None, :]).sum(2)) / np.sqrt(np.maximum(d2, 1e-30)) * -1.0
            theta = (( -d) * (-ldn[:, None, :])).sum(2) / np.sqrt(np.maximum(d2, 1e-30))  # dot(normalize(pos - wp) , ndir) with d = pos - wp
            reach_sp = theta >= cut_y[:, None]
            reach = np.where((ltype == 1)[:, None], reach_pt, np.where((ltype == 2)[:, None], reach_sp, True))
            facing = (Nn[None, :, :] * ldir[:, None, :]).sum(2) > 0
            surv = box_ok
            # alternative pre-filters: bounding sphere around the quadrant's centre pixel / around the box centre
            c0 = P[27]; R0 = np.sqrt(((P - c0) ** 2).sum(1).max())
            sph0 = (np.sqrt(((lp - c0) ** 2).sum(1)) <= lr * 1.0001 + R0) | (ltype != 1)
            c1 = 0.5 * (bmin + bmax); R1 = np.sqrt(((P - c1) ** 2).sum(1).max())
            sph1 = (np.sqrt(((lp - c1) ** 2).sum(1)) <= lr * 1.0001 + R1) | (ltype != 1)
            # finer pre-filters: the union of sub-spheres over the quadrant's halves (lanes 0-31 / 32-63) resp. its four 16-lane rows of two pixel rows
            def sub_spheres(groups, centres):
                ok = np.zeros(len(lp), bool)
                for lanes, c in zip(groups, centres):
                    cc = P[c]; RR = np.sqrt(((P[lanes] - cc) ** 2).sum(1).max())
                    ok |= np.sqrt(((lp - cc) ** 2).sum(1)) <= lr * 1.0001 + RR
                return ok | (ltype != 1)
            half = sub_spheres([np.arange(0, 32), np.arange(32, 64)], [11, 43])
            quart = sub_spheres([np.arange(16 * k, 16 * k + 16) for k in range(4)], [3 + 16 * k for k in range(4)])
            # spot lights against the quadrant's sphere: outside the cone iff angle(c - L, axis) > theta_c + asin(R / |c - L|)
            v = c0[None, :] - lp; dist = np.sqrt((v ** 2).sum(1)); cosA = (v * (-ldn)).sum(1) / np.maximum(dist, 1e-30)
            sin_d = np.minimum(R0 / np.maximum(dist, 1e-30), 1.0); cos_d = np.sqrt(1.0 - sin_d ** 2)
            cos_c = np.clip(cut_y, -1.0, 1.0); sin_c = np.sqrt(1.0 - cos_c ** 2)
            cone_ok = (dist <= R0) | (cosA >= cos_c * cos_d - sin_c * sin_d) | (cos_c * cos_d - sin_c * sin_d <= -1.0) | (np.arccos(cos_c) + np.arcsin(sin_d) >= np.pi)
            spot = ltype == 2
            alt.append((int(sph0.sum()), int(sph1.sum()), int((sph0 & box_ok).sum()), int(half.sum()), int(quart.sum()), int((ltype != 1).sum()),
                        int((reach & sph0[:, None]).any(1).sum()), int(spot.sum()), int((reach.any(1) & spot).sum()), int((cone_ok & spot).sum()),
                        int(((reach.any(1) & spot) & ~cone_ok).sum())))
            any_reach = (reach & surv[:, None]).any(1)
            pair = reach & facing & surv[:, None]
            any_pair = pair.any(1)
            stats.append((num, int(surv.sum()), int(any_reach.sum()), int(any_pair.sum()), int(pair.sum()), int(pair.sum(0).max()) if num else 0))
    print("row", tr, "done", flush=True)
s = np.array(stats, dtype=np.float64)
names = ["list length", "survive quadrant box", "reach >=1 pixel", ">=1 queued pair", "queued pairs", "max pairs of one pixel"]
for i, n in enumerate(names):
    print(f"{n:26s} mean {s[:, i].mean():8.2f}  p50 {np.percentile(s[:, i], 50):6.1f}  p90 {np.percentile(s[:, i], 90):6.1f}  max {s[:, i].max():6.0f}")
passes = np.ceil(s[:, 4] / 64.0)
print("pair passes (ceil(pairs/64)) mean", passes.mean(), " lane utilisation of the passes", s[:, 4].sum() / max(passes.sum() * 64, 1))
print("quadrants with list < 8:", (s[:, 0] < 8).mean(), " < 16:", (s[:, 0] < 16).mean(), " > 64:", (s[:, 0] > 64).mean())
a = np.array(alt, dtype=np.float64)
print("survivors: box", s[:, 1].mean(), " sphere@centre pixel", a[:, 0].mean(), " sphere@box centre", a[:, 1].mean(), " both", a[:, 2].mean())
print("           two half-spheres", a[:, 3].mean(), " four row-pair spheres", a[:, 4].mean(), " of which not point lights", a[:, 5].mean(), " reach >= 1 pixel", a[:, 6].mean())
print("           spot lights per list", a[:, 7].mean(), " that reach >= 1 pixel", a[:, 8].mean(), " that pass a cone-vs-sphere test", a[:, 9].mean(), " (wrongly rejected:", a[:, 10].sum(), ")")
short = s[:, 0] < 32
print("lists < 32 lights: share", short.mean(), " mean list", s[short, 0].mean(), " mean survivors (box)", s[short, 1].mean(), " sphere", a[short, 0].mean())
