"""Per-block start / end times of the SHADOWED shade kernel (k2_shade_csm_pt, the per-tile grid; library built with EXTRA=-DSHADE_PROF) on a band of C4 and
on the whole frame: why does an eighth of the frame take 70 us when the whole takes 228?  usage: shade_prof_csm.py [R/G | 0/1]"""
import ctypes, sys
import numpy as np, torch
sys.path.insert(0, ".")
from sailor_amd import host, synth, _lib
from sailor_amd.forward_plus import HipContext, ForwardPlus, upload_lights, PreparedLights, upload_shadow_maps
import bench
frame = bench.BenchFrame("C4")
cam, W, H = frame.cam, frame.cam.width, frame.cam.height
N = len(frame.lights)
dev = torch.device("cuda", 0)
ctx = HipContext(dev)
r, g = (int(v) for v in (sys.argv[1] if len(sys.argv) > 1 else "2/8").split("/"))
band = host.band_for_rank(W, H, r, g) if g > 1 else host.band_whole_frame(W, H)
dl = upload_lights(frame.lights, dev)
prep = PreparedLights(ctx, dl, N)
fp = ForwardPlus(ctx, W, H, N, band=band, prepared=prep)
rows = slice(band.fbRowBegin, band.fbRowBegin + band.fbRowCount)
dd = torch.from_numpy(np.ascontiguousarray(frame.depth[rows])).to(dev)
ds = torch.from_numpy(frame.surface_rows(rows.start, rows.stop)).to(dev)
csm, keep = upload_shadow_maps(synth.make_shadow_set(cam, frame.cfg["shadow_size"]), dev)
fp.cull(cam.frame, dl, N, dd)
for _ in range(4):
    fp.shade(cam.frame, ds, dl, N, csm)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record(); fp.shade(cam.frame, ds, dl, N, csm); e1.record(); torch.cuda.synchronize()
lib = _lib.load()
buf = np.zeros((65536, 4), dtype=np.uint64)
fn = lib.sailor_hip_debug_read_shade_prof
fn.argtypes = [ctypes.c_void_p, ctypes.c_size_t]
assert fn(buf.ctypes.data, buf.nbytes) == 0
nb = fp.band_tiles
p = buf[:nb].astype(np.int64)
t0 = p[:, 0].min()
us = lambda v: (v - t0) / 100.0
dur = (p[:, 3] - p[:, 0]) / 100.0
span = us(p[:, 3].max())
print("%s: %d tiles, event-bracketed launch %.1f us; span by the blocks' clocks %.1f us" % ("band %d/%d" % (r, g) if g > 1 else "whole frame", nb, e0.elapsed_time(e1) * 1e3, span))
print("block duration us: mean %.2f median %.2f p10 %.2f p90 %.2f p99 %.2f max %.2f;  block-slot time %.0f slot us = %.1f us x 2048 slots" %
      (dur.mean(), np.median(dur), np.percentile(dur, 10), np.percentile(dur, 90), np.percentile(dur, 99), dur.max(), dur.sum(), dur.sum() / 2048))
step = 2.0 if g > 1 else 8.0
ts = np.arange(0, span + step, step)
print("resident blocks every %.0f us:" % step, [int(((us(p[:, 0]) <= t) & (us(p[:, 3]) > t)).sum()) for t in ts])
print("started by t:               ", [int((us(p[:, 0]) <= t).sum()) for t in ts])
# duration by start time: do late blocks live shorter?
order = np.argsort(p[:, 0])
q = np.array_split(order, 8)
print("mean duration of the blocks by start order (eighths):", [round(float(dur[i].mean()), 2) for i in q])
