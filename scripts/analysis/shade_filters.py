"""Round 3: what a finer pre-filter and a block-wide pair queue would buy k2_shade on the C3 frame (CPU, numpy; sampled tile rows).
Per 8x8 quadrant: survivors of today's tests (bounding sphere + cone), of a 2.5-D "depth occupancy" test (Harada's 2.5D culling at quadrant level:
the quadrant's 64 surface points lie near one ray, a light can only reach the depths within sqrt((r + E)^2 - h^2) of its foot point on that ray, and
only where a pixel actually is -- a bit mask over B depth bins), the lights that really reach a pixel, queued pairs; per tile: pair passes with one
queue per wave against one queue per block.   Usage: python scripts/analysis/shade_filters.py [rows...]"""
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
BINS = (32, 64)
rec = []      # per quadrant
tiles = []    # per tile: pairs of its four quadrants
for tr in rows:
    g, idx, _ = oracle.light_cull(cam.frame, W, H, lights, depth, tile_rows=(tr, tr + 1), threads=8)
    r0, r1 = H - 16 * (tr + 1), H - 16 * tr
    surf = synth.make_surface(cam, depth, row_begin=r0, row_end=r1)
    pos, nrm = surf[0, :, :, :3].astype(np.float64), surf[1, :, :, :3].astype(np.float64)
    for tx in range(Tx):
        off, num = int(g[tx, 0]), int(g[tx, 1])
        L = lights[idx[off:off + num]]
        lp = L["worldPosition"].astype(np.float64); lr = L["bounds"][:, 0].astype(np.float64); ltype = L["type"]
        ldir = -L["direction"].astype(np.float64)
        ldn = ldir / np.linalg.norm(ldir, axis=1, keepdims=True)
        cut_y = L["cutOff"][:, 1].astype(np.float64)
        tp = []
        for q in range(4):
            ys = [(r1 - 1) - ((q >> 1) * 8 + ly) - r0 for ly in range(8)]
            xs = [tx * 16 + (q & 1) * 8 + lx for lx in range(8)]
            P = pos[np.ix_(ys, xs)].reshape(64, 3); Nn = nrm[np.ix_(ys, xs)].reshape(64, 3)
            d = lp[:, None, :] - P[None, :, :]
            d2 = (d ** 2).sum(2)
            reach_pt = d2 <= (lr ** 2)[:, None]
            theta = ((-d) * (-ldn[:, None, :])).sum(2) / np.sqrt(np.maximum(d2, 1e-30))
            reach_sp = theta >= cut_y[:, None]
            reach = np.where((ltype == 1)[:, None], reach_pt, np.where((ltype == 2)[:, None], reach_sp, True))
            facing = (Nn[None, :, :] * ldir[:, None, :]).sum(2) > 0
            c0 = P[27]; R0 = np.sqrt(((P - c0) ** 2).sum(1).max())
            dist = np.sqrt(((lp - c0) ** 2).sum(1))
            sph = (dist <= lr * 1.0001 + R0) | (ltype != 1)
            v = c0[None, :] - lp; cosA = (v * (-ldn)).sum(1) / np.maximum(dist, 1e-30)
            sin_d = np.minimum(R0 / np.maximum(dist, 1e-30), 1.0); cos_d = np.sqrt(1.0 - sin_d ** 2)
            cos_c = np.clip(cut_y, -1.0, 1.0); sin_c = np.sqrt(1.0 - cos_c ** 2)
            cone_ok = (dist <= R0) | (cosA >= cos_c * cos_d - sin_c * sin_d) | (np.arccos(cos_c) + np.arcsin(sin_d) >= np.pi)
            today = np.where(ltype == 1, sph, np.where(ltype == 2, cone_ok, True))
            # 2.5-D: the ray from the eye through the centre pixel; t_p = the pixel's coordinate along it, E = the largest lateral distance of a pixel
            eye = np.array([cam.frame.cameraPosition[0], cam.frame.cameraPosition[1], cam.frame.cameraPosition[2]], np.float64) if hasattr(cam.frame, "cameraPosition") else None
            if eye is None:
                eye = np.asarray(cam.position, np.float64)
            u = c0 - eye; u /= np.linalg.norm(u)
            t = (P - c0) @ u
            lat = np.sqrt(np.maximum(((P - c0) ** 2).sum(1) - t ** 2, 0.0))
            E = lat.max()
            s = (lp - c0) @ u
            h2 = np.maximum(((lp - c0) ** 2).sum(1) - s ** 2, 0.0)
            w2 = (lr * 1.0001 + E) ** 2 - h2
            out = []
            for B in BINS:
                tmin, tmax = t.min(), t.max()
                scale = (B - 1e-6) / max(tmax - tmin, 1e-9)
                occ = np.zeros(B, bool); occ[np.clip(((t - tmin) * scale).astype(int), 0, B - 1)] = True
                ok = np.zeros(len(lp), bool)
                for j in range(len(lp)):
                    if ltype[j] != 1:
                        ok[j] = today[j]
                        continue
                    if w2[j] < 0:
                        continue
                    wj = np.sqrt(w2[j])
                    lo = int(np.floor((s[j] - wj - tmin) * scale)); hi = int(np.floor((s[j] + wj - tmin) * scale))
                    if hi < 0 or lo > B - 1:
                        continue
                    ok[j] = occ[max(lo, 0):min(hi, B - 1) + 1].any()
                assert not (reach.any(1) & ~ok & (ltype == 1)).any(), "the 2.5-D test dropped a light that reaches a pixel"
                out.append(int(ok.sum()))
            any_reach = reach.any(1)
            pair = reach & facing
            tp.append(int(pair.sum()))
            rec.append((num, int(today.sum()), out[0], out[1], int(any_reach.sum()), int(pair.any(1).sum()), int(pair.sum()), int((ltype == 1).sum()),
                        int((today & (ltype == 1)).sum()), int((any_reach & (ltype == 1)).sum())))
        tiles.append(tp)
    print("row", tr, "done", flush=True)
r = np.array(rec, np.float64)
names = ["list length", "today (sphere + cone)", f"2.5-D {BINS[0]} bins", f"2.5-D {BINS[1]} bins", "reach >= 1 pixel", ">= 1 pair", "pairs", "point lights in list",
         "point lights after sphere", "point lights that reach"]
for i, n in enumerate(names):
    print(f"{n:28s} mean {r[:, i].mean():8.2f}  p50 {np.percentile(r[:, i], 50):6.1f}  p90 {np.percentile(r[:, i], 90):6.1f}  max {r[:, i].max():6.0f}")
pairs = r[:, 6]
print("quadrants with 0 pairs:", (pairs == 0).mean(), " 1..16:", ((pairs > 0) & (pairs <= 16)).mean(), " 17..32:", ((pairs > 16) & (pairs <= 32)).mean(),
      " 33..64:", ((pairs > 32) & (pairs <= 64)).mean(), " > 64:", (pairs > 64).mean(), " > 128:", (pairs > 128).mean())
t = np.array(tiles, np.float64)
per_wave = np.ceil(t / 64.0).sum(1)
per_block = np.ceil(t.sum(1) / 64.0)
print("pair passes per tile: one queue per wave", per_wave.mean(), " one queue per block", per_block.mean(), " pairs per tile", t.sum(1).mean(),
      " lane use", t.sum() / (per_wave.sum() * 64), "->", t.sum() / (per_block.sum() * 64))
# with a block-wide queue the passes are dealt to four waves: the block's pass phase lasts ceil(passes / 4) rounds
print("rounds of the pass phase per tile: per-wave queues", np.ceil(t / 64.0).max(1).mean(), " block queue", np.ceil(per_block / 4.0).mean())
