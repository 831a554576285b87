"""Per-block start / end times of the band shade kernel (library built with EXTRA=-DSHADE_PROF): where do a band's ~29 us go?  usage: shade_prof.py [R/G] [C3 | C4]
(C4: with the shadow cascades -- k2_shade_band_csm*)"""
import ctypes, sys
import numpy as np, torch
sys.path.insert(0, ".")
from sailor_amd import host, _lib
from sailor_amd.forward_plus import HipContext, ForwardPlus, upload_lights, PreparedLights
import bench
cfg = sys.argv[2] if len(sys.argv) > 2 else "C3"
frame = bench.BenchFrame(cfg)
cam, W, H = frame.cam, frame.cam.width, frame.cam.height
N = len(frame.lights)
dev = torch.device("cuda", 0)
ctx = HipContext(dev)
r, g = (int(v) for v in (sys.argv[1] if len(sys.argv) > 1 else "2/8").split("/"))
assert g > 1, "only the band kernel carries the marks"
band = host.band_for_rank(W, H, r, g)
dl = upload_lights(frame.lights, dev)
prep = PreparedLights(ctx, dl, N); prep.prepare(0, N)
fp = ForwardPlus(ctx, W, H, N, band=band, prepared=prep)
rows = slice(band.fbRowBegin, band.fbRowBegin + band.fbRowCount)
dd = torch.from_numpy(np.ascontiguousarray(frame.depth[rows])).to(dev)
ds = torch.from_numpy(frame.surface_rows(rows.start, rows.stop)).to(dev)
csm = None
if frame.cfg.get("shadow_size"):
    from sailor_amd import synth
    from sailor_amd.forward_plus import upload_shadow_maps
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
SPLIT = 2048 if g > 1 else 0
nb = SPLIT + fp.band_tiles
p = buf[:nb].astype(np.int64)
t0 = p[:, 0].min()
us = lambda v: (v - t0) / 100.0
dur = (p[:, 3] - p[:, 0]) / 100.0
xcd = (p[:, 2] >> 32) & 0xF
hw = p[:, 2] & 0xFFFFFFFF
cu = (xcd << 8) | (((hw >> 13) & 7) << 5) | (((hw >> 12) & 1) << 4) | ((hw >> 8) & 0xF)   # (XCD, SE, SH, CU)
slot = hw & 0xF
g_host, _ = fp.lists_to_host()
num = g_host[:fp.band_tiles, 1]
if g == 1:   # the whole-frame kernel's grid (8 x tiles per piece, 10 pieces, tile rows): block -> tile as k2_shade_body maps it
    Tx = fp.Tx; tpp = (Tx + 79) // 80; gx = 8 * tpp
    lin = np.arange(fp.band_tiles); bx, by, bz = lin % gx, (lin // gx) % 10, lin // (gx * 10)
    btx = (((bx - bz) & 7) + 8 * by) * tpp + (bx >> 3)
    assert (btx < Tx).all()
    num = num[bz * Tx + btx]
print("band %d/%d: %d tiles, event-bracketed launch %.2f us; kernel span by the blocks' clocks %.2f us" % (r, g, fp.band_tiles, e0.elapsed_time(e1) * 1e3, us(p[:, 3].max())))
for name, sel in (("split-role blocks", np.arange(SPLIT)), ("tile blocks", SPLIT + np.arange(fp.band_tiles))):
    d = dur[sel]
    print("%-18s n %5d  duration us mean %.2f median %.2f p90 %.2f p99 %.2f max %.2f; starts %.2f .. %.2f (median %.2f), last end %.2f" %
          (name, len(sel), d.mean(), np.median(d), np.percentile(d, 90), np.percentile(d, 99), d.max(), us(p[sel, 0].min()), us(p[sel, 0].max()), np.median(us(p[sel, 0])), us(p[sel, 3].max())))
tiles = SPLIT + np.arange(fp.band_tiles)
import os
split_min = int(os.environ.get("SAILOR_SPLIT_MIN", "0")) or (64 if fp.band_tiles <= 12000 else 96)   # shade_body.h: SPLIT_MIN_SMALL / _LARGE by the band's size
own = (num < split_min) | (g == 1)     # tiles the ordinary blocks shade themselves (the others return at once: the split blocks take them)
print("tile blocks that shade (num < %d): %d, mean duration %.2f us;  that return at once: %d, mean duration %.2f us" % (split_min, own.sum(), dur[tiles][own].mean(), (~own).sum(), dur[tiles][~own].mean() if (~own).any() else 0.0))
bins = [0, 1, 8, 16, 24, 32, 40, 64, 96, 129]
for lo, hi in zip(bins[:-1], bins[1:]):
    m = own & (num >= lo) & (num < hi)
    if m.any():
        print("   list length %2d..%2d: %5d tiles, block duration mean %.2f us p90 %.2f" % (lo, hi - 1, m.sum(), dur[tiles][m].mean(), np.percentile(dur[tiles][m], 90)))
# concurrency over time: blocks resident on the chip in 1 us steps
print("resident blocks (all roles) at t = 0, 1, ... us (whole frame: every 4 us):")
ts = np.arange(0, us(p[:, 3].max()) + 1, 1.0 if g > 1 else 4.0)
res = [int(((us(p[:, 0]) <= t) & (us(p[:, 3]) > t)).sum()) for t in ts]
print("  ", res)
done = [int((us(p[tiles, 3]) <= t).sum()) for t in ts]
print("tile blocks finished by t:"); print("  ", done)
started = [int((us(p[tiles, 0]) <= t).sum()) for t in ts]
print("tile blocks started by t:"); print("  ", started)
print("distinct CUs seen: %d; tile blocks per CU min %d max %d" % (len(np.unique(cu)), np.bincount(np.unique(cu[tiles], return_inverse=True)[1]).min(), np.bincount(np.unique(cu[tiles], return_inverse=True)[1]).max()))
perx = [int((xcd[tiles] == x).sum()) for x in range(8)]
endx = [round(float(us(p[tiles][xcd[tiles] == x, 3].max())), 1) for x in range(8)]
print("tile blocks per XCD", perx, "last end per XCD", endx)
order = np.argsort(-p[:, 3])[:8]
print("last to end:")
for b in order:
    what = "split-role" if b < SPLIT else "tile %4d (num %3d)" % (b - SPLIT, num[b - SPLIT])
    print("  block %5d %-22s start %6.2f end %6.2f  (%.2f us) xcd %d" % (b, what, us(p[b, 0]), us(p[b, 3]), dur[b], xcd[b]))
