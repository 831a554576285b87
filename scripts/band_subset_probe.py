"""How much of a C5 band's cull chain is the lights that cannot reach the band?  Times the band's chain on all lights and on the subset whose spheres can touch
the band's rows (host-side estimate: the distinct lights of the band's lists plus those of the two neighbouring bands) -- the upper bound of what a
band-local pre-cull of the light set can buy.  usage: band_subset_probe.py [config] [R/G]"""
import sys
import numpy as np, torch
sys.path.insert(0, ".")
from sailor_amd import host
from sailor_amd.forward_plus import HipContext, ForwardPlus, upload_lights, PreparedLights
import bench
cfg = sys.argv[1] if len(sys.argv) > 1 else "C5"
r, g = (int(v) for v in (sys.argv[2] if len(sys.argv) > 2 else "3/8").split("/"))
frame = bench.BenchFrame(cfg)
cam, W, H = frame.cam, frame.cam.width, frame.cam.height
N = len(frame.lights)
dev = torch.device("cuda", 0)
ctx = HipContext(dev)
def chain_ms(lights_np, band, dynamic):
    n = len(lights_np)
    dl = upload_lights(lights_np, dev)
    prep = PreparedLights(ctx, dl, n)
    fp = ForwardPlus(ctx, W, H, n, band=band, prepared=prep)
    rows = slice(band.fbRowBegin, band.fbRowBegin + band.fbRowCount)
    dd = torch.from_numpy(np.ascontiguousarray(frame.depth[rows])).to(dev)
    for _ in range(3):
        fp.cull(cam.frame, dl, n, dd, prepare_lights=dynamic)
    torch.cuda.synchronize()
    names = ctx.launches_of(lambda: fp.cull(cam.frame, dl, n, dd, prepare_lights=dynamic))   # (the library's own record of the chain's kernels)
    acc = np.zeros(len(names))
    reps = 10
    for _ in range(reps):
        ctx.time_launches(0, len(names))
        fp.cull(cam.frame, dl, n, dd, prepare_lights=dynamic)
        torch.cuda.synchronize()
        acc += np.array([ctx.timed_launch_ms(i) for i in range(len(names))])
    g_, idx = fp.lists_to_host()
    return dict(zip(names, np.round(acc / reps * 1e3, 1))), g_, idx
band = host.band_for_rank(W, H, r, g)
wide = host.band_from_tile_rows(W, H, max(band.tileRowBegin - 8, 0), min(band.tileRowEnd + 8, host.num_tiles(W, H)[1]))
for dyn in (False, True):
    full, _, _ = chain_ms(frame.lights, band, dyn)
    print("dynamic" if dyn else "static ", "all %d lights:" % N, full, "sum %.1f" % sum(full.values()))
_, gw, iw = chain_ms(frame.lights, wide, False)
tot = int(iw[0])
keep = np.unique(iw[1:1 + tot])
print("lights in the lists of the band widened by 8 tile rows each side: %d (%.1f %%)" % (len(keep), 100.0 * len(keep) / N))
sub = frame.lights[np.sort(keep)]
for dyn in (False, True):
    part, _, _ = chain_ms(sub, band, dyn)
    print("dynamic" if dyn else "static ", "subset of %d lights:" % len(sub), part, "sum %.1f" % sum(part.values()))
