"""Per-block start / end times of the shade on the one-block-per-tile grid (k2_shade*_pt; library built with EXTRA=-DSHADE_PROF), on a band or on the whole
frame: block durations by list length, residency over time, the last blocks to end.  usage: shade_prof_grid.py [R/G | 0/1] [C3 | C4 | C5]
(a band takes this form when it has more than SAILOR_BAND_FORM_TILES tiles: set it to 0 to force the form on any band)"""
import ctypes, sys
import numpy as np, torch
sys.path.insert(0, ".")
from sailor_amd import host, synth, _lib
from sailor_amd.forward_plus import HipContext, ForwardPlus, upload_lights, PreparedLights, upload_shadow_maps
import bench
cfg = sys.argv[2] if len(sys.argv) > 2 else "C3"
frame = bench.BenchFrame(cfg)
cam, W, H = frame.cam, frame.cam.width, frame.cam.height
N = len(frame.lights)
dev = torch.device("cuda", 0)
ctx = HipContext(dev)
r, g = (int(v) for v in (sys.argv[1] if len(sys.argv) > 1 else "0/2").split("/"))
band = host.band_for_rank(W, H, r, g) if g > 1 else host.band_whole_frame(W, H)
dl = upload_lights(frame.lights, dev)
prep = PreparedLights(ctx, dl, N)
fp = ForwardPlus(ctx, W, H, N, band=band, prepared=prep)
assert not fp.tile_order, "this band takes the band form: SAILOR_BAND_FORM_TILES=0 forces the per-tile grid"
rows = slice(band.fbRowBegin, band.fbRowBegin + band.fbRowCount)
dd = torch.from_numpy(np.ascontiguousarray(frame.depth[rows])).to(dev)
ds = torch.from_numpy(frame.surface_rows(rows.start, rows.stop)).to(dev)
csm = None
if frame.cfg.get("shadow_size"):
    csm, keep = upload_shadow_maps(synth.make_shadow_set(cam, frame.cfg["shadow_size"]), dev)
fp.cull(cam.frame, dl, N, dd)
for _ in range(4):
    fp.shade(cam.frame, ds, dl, N, csm)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record(); names = ctx.launches_of(lambda: fp.shade(cam.frame, ds, dl, N, csm)); e1.record(); torch.cuda.synchronize()
lib = _lib.load()
buf = np.zeros((65536, 4), dtype=np.uint64)
fn = lib.sailor_hip_debug_read_shade_prof
fn.argtypes = [ctypes.c_void_p, ctypes.c_size_t]
assert fn(buf.ctypes.data, buf.nbytes) == 0
# the grid (8 x tiles per piece, 10 pieces, tile rows): linear block id -> tile as k2_shade_body maps it (blocks past the row's last tile leave at once)
Tx = fp.Tx; tpp = (Tx + 79) // 80; gx = 8 * tpp
nrows = band.tileRowEnd - band.tileRowBegin
nb = min(gx * 10 * nrows, 65536)
lin = np.arange(nb); bx, by, bz = lin % gx, (lin // gx) % 10, lin // (gx * 10)
rot = 0 if csm is not None else 1
btx = (((bx - bz * rot) & 7) + 8 * by) * tpp + (bx >> 3)
real = btx < Tx
p = buf[:nb].astype(np.int64)
g_host, _ = fp.lists_to_host()
num = np.zeros(nb, np.int64); num[real] = g_host[(bz * Tx + btx)[real], 1]
p, num, bz = p[real], num[real], bz[real]
t0 = p[:, 0].min()
us = lambda v: (v - t0) / 100.0
dur = (p[:, 3] - p[:, 0]) / 100.0
span = us(p[:, 3].max())
print("%s %s %s: %d tiles (%d of them in the first 65 536 blocks), event-bracketed launch %.1f us; span by the blocks' clocks %.1f us" %
      (cfg, "band %d/%d" % (r, g) if g > 1 else "whole frame", names, fp.band_tiles, len(p), e0.elapsed_time(e1) * 1e3, span))
print("block duration us: mean %.2f median %.2f p90 %.2f p99 %.2f max %.2f;  block-slot time %.0f slot us = %.1f us x 2048 slots" %
      (dur.mean(), np.median(dur), np.percentile(dur, 90), np.percentile(dur, 99), dur.max(), dur.sum(), dur.sum() / 2048))
bins = [0, 1, 8, 16, 24, 32, 40, 64, 96, 128, 129]
for lo, hi in zip(bins[:-1], bins[1:]):
    m = (num >= lo) & (num < hi)
    if m.any():
        print("   list length %3d..%3d: %5d tiles, block duration mean %6.2f us p90 %6.2f max %6.2f" % (lo, hi - 1, m.sum(), dur[m].mean(), np.percentile(dur[m], 90), dur[m].max()))
step = max(2.0, round(span / 40))
ts = np.arange(0, span + step, step)
print("resident blocks every %.0f us:" % step, [int(((us(p[:, 0]) <= t) & (us(p[:, 3]) > t)).sum()) for t in ts])
print("started by t:              ", [int((us(p[:, 0]) <= t).sum()) for t in ts])
order = np.argsort(-p[:, 3])[:10]
print("last to end:")
for b in order:
    print("  tile row %3d (num %3d)  start %7.2f end %7.2f  (%.2f us)" % (bz[b], num[b], us(p[b, 0]), us(p[b, 3]), dur[b]))
rows_end = [round(float(us(p[bz == z, 3].max())), 1) for z in range(nrows)]
rows_start = [round(float(us(p[bz == z, 0].min())), 1) for z in range(nrows)]
print("first start / last end per tile row:", list(zip(rows_start, rows_end))[:: max(1, nrows // 34)])
