"""What a wave over a 16 x 8 half tile (two pixels per lane) would test, against today's two waves over its two 8 x 8 quadrants (C3, sampled tile
rows, numpy): survivors of the bounding-sphere + cone filter per quadrant (a, b) and for the half as one region (u), lights reaching a pixel.
Loop cost today 16 (a + b), as one wave ~22 u.  Usage: python scripts/analysis/shade_two_pixels_per_lane.py [tile rows...]"""
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


def survivors(P, centre, lp, lr, ltype, ldn, cut_y):
    R0 = np.sqrt(((P - centre) ** 2).sum(1).max())
    dist = np.sqrt(((lp - centre) ** 2).sum(1))
    sph = (dist <= lr * 1.0001 + R0) | (ltype != 1)
    v = centre[None, :] - lp; cosA = (v * (-ldn)).sum(1) / np.maximum(dist, 1e-30)
    sin_d = np.minimum(R0 / np.maximum(dist, 1e-30), 1.0); cos_d = np.sqrt(1.0 - sin_d ** 2)
    cos_c = np.clip(cut_y, -1.0, 1.0); sin_c = np.sqrt(1.0 - cos_c ** 2)
    cone_ok = (dist <= R0) | (cosA >= cos_c * cos_d - sin_c * sin_d) | (np.arccos(cos_c) + np.arcsin(sin_d) >= np.pi)
    return np.where(ltype == 1, sph, np.where(ltype == 2, cone_ok, True))


rec = []
for tr in rows:
    g, idx, _ = oracle.light_cull(cam.frame, W, H, lights, depth, tile_rows=(tr, tr + 1), threads=8)
    r0, r1 = H - 16 * (tr + 1), H - 16 * tr
    surf = synth.make_surface(cam, depth, row_begin=r0, row_end=r1)
    pos = surf[0, :, :, :3].astype(np.float64)
    for tx in range(Tx):
        off, num = int(g[tx, 0]), int(g[tx, 1])
        L = lights[idx[off:off + num]]
        lp = L["worldPosition"].astype(np.float64); lr = L["bounds"][:, 0].astype(np.float64); ltype = L["type"]
        ldir = -L["direction"].astype(np.float64); ldn = ldir / np.linalg.norm(ldir, axis=1, keepdims=True)
        cut_y = L["cutOff"][:, 1].astype(np.float64)
        for half in range(2):
            q = []
            for side in range(2):
                ys = [(r1 - 1) - (half * 8 + ly) - r0 for ly in range(8)]
                xs = [tx * 16 + side * 8 + lx for lx in range(8)]
                q.append(pos[np.ix_(ys, xs)].reshape(64, 3))
            sa = survivors(q[0], q[0][27], lp, lr, ltype, ldn, cut_y)
            sb = survivors(q[1], q[1][27], lp, lr, ltype, ldn, cut_y)
            both = np.concatenate(q)
            su = survivors(both, q[0][31], lp, lr, ltype, ldn, cut_y)   # centre: the left quadrant's pixel (7, 3), next to the middle of the half
            d2 = ((lp[:, None, :] - both[None, :, :]) ** 2).sum(2)
            reach = (np.where((ltype == 1)[:, None], d2 <= (lr ** 2)[:, None], True)).any(1)
            rec.append((num, sa.sum(), sb.sum(), (sa | sb).sum(), su.sum(), (reach & su).sum()))
r = np.array(rec, np.float64)
m = r.mean(0)
print(f"half tiles {len(r)}: list {m[0]:.1f}; quadrant survivors a {m[1]:.2f} b {m[2]:.2f} (a + b {m[1] + m[2]:.2f}, union {m[3]:.2f}); the half as one region {m[4]:.2f}; "
      f"of those reaching a pixel (point lights exactly, spots counted as reaching) {m[5]:.2f}")
print(f"loop instructions per half tile: today 16 (a + b) = {16 * (m[1] + m[2]):.0f}; one wave, ~22 per light = {22 * m[4]:.0f}")
