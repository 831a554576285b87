"""Scratch probe run on the GPU box: generator statistics for the frozen configs + ad-hoc debugging."""
import sys, time
import numpy as np, torch
sys.path.insert(0, ".")
from sailor_amd import synth, host, _lib
from sailor_amd.forward_plus import HipContext, ForwardPlus, upload_lights
from oracle import oracle

ctx = HipContext("cuda:0")

def stats(name, s, clusters, cc, spread=3.0):
    cfg = dict(synth.CONFIGS[name])
    lc = cfg["lights"]
    lc = synth.LightSetConfig(count=lc.count, spot_fraction=lc.spot_fraction, radius_scale=s, cluster_lights=clusters, cluster_count=cc, cluster_spread=spread)
    cam = synth.make_camera(cfg["width"], cfg["height"])
    depth = synth.make_linear_depth(cam.width, cam.height)
    lights = synth.make_lights(cam, depth, lc)
    fp = ForwardPlus(ctx, cam.width, cam.height, len(lights))
    d = torch.from_numpy(depth).to(ctx.device); l = upload_lights(lights, ctx.device)
    fp.cull(cam.frame, l, len(lights), d)
    g, idx = fp.lists_to_host()
    num = g[:, 1]
    print(f"{name} s={s} clusters={clusters}x{cc} spread {spread}: mean {num.mean():.2f} max {num.max()} full(=128) {100*(num==128).mean():.3f}% total {idx[0]} distinct {len(np.unique(idx[1:]))}", flush=True)

stats("C3", 0.85, 6000, 12, 3.5)
stats("C3", 0.85, 7200, 12, 3.0)
stats("C5", 0.2, 6000, 12, 3.5)
sys.exit(0)
# sentinel debugging
f = synth.make_frame("tiny")
W, H = f.cam.width, f.cam.height
fp = ForwardPlus(ctx, W, H, len(f.lights))
d = torch.from_numpy(f.depth).to(ctx.device); l = upload_lights(f.lights, ctx.device); s = torch.from_numpy(f.surface).to(ctx.device)
fp.cull(f.cam.frame, l, len(f.lights), d)
g, idx = fp.lists_to_host()
t = int(np.argmax(g[:, 1])); cut = int(g[t, 0]) + 2
print("tile", t, "num", g[t, 1], "offset", g[t, 0], "cut", cut)
culled = fp.culled.clone(); culled[cut] = -1; fp.culled = culled
out = fp.shade(f.cam.frame, s, l, len(f.lights), None).cpu().numpy()
ref_idx = np.zeros(1 + len(g) * 128, np.uint32); ref_idx[: len(idx)] = idx; ref_idx[cut] = 0xFFFFFFFF
ref = oracle.shade(f.cam.frame, W, H, f.surface, f.lights, g, ref_idx, None)
bad = np.abs(out - ref).max(-1) > 1e-3
ys, xs = np.nonzero(bad)
print("bad pixels", bad.sum(), "x range", xs.min() if len(xs) else None, xs.max() if len(xs) else None, "y range", ys.min() if len(ys) else None, ys.max() if len(ys) else None)
tx, ty = t % 8, t // 8
print("tile px x", tx*16, "y rows", H-1-(ty*16+15), H-1-ty*16)
if len(xs):
    print("sample", out[ys[0], xs[0]], ref[ys[0], xs[0]])
