"""Per-block phase times of k1_tile_cull (library built with EXTRA=-DCULL_PROF): which blocks are the kernel's tail?  usage: cull_prof.py [R/G] [C3 | C5]"""
import ctypes, sys
import numpy as np, torch
sys.path.insert(0, ".")
from sailor_amd import host, synth, _lib
from sailor_amd.forward_plus import HipContext, ForwardPlus, upload_lights
import bench
frame = bench.BenchFrame(sys.argv[2] if len(sys.argv) > 2 else "C3")
cam, W, H = frame.cam, frame.cam.width, frame.cam.height
dev = torch.device("cuda", 0)
ctx = HipContext(dev)
band = host.band_whole_frame(W, H)
if len(sys.argv) > 1:
    r, g = (int(v) for v in sys.argv[1].split("/")); band = host.band_for_rank(W, H, r, g)
fp = ForwardPlus(ctx, W, H, len(frame.lights), band=band)
dd = torch.from_numpy(np.ascontiguousarray(frame.depth[band.fbRowBegin:band.fbRowBegin + band.fbRowCount])).to(dev)
dl = upload_lights(frame.lights, dev)
for _ in range(3):
    fp.cull(cam.frame, dl, len(frame.lights), dd)
torch.cuda.synchronize()
lib = _lib.load()
buf = np.zeros((65536, 4), dtype=np.uint64)
fn = lib.sailor_hip_debug_read_cull_prof
fn.argtypes = [ctypes.c_void_p, ctypes.c_size_t]
assert fn(buf.ctypes.data, buf.nbytes) == 0
Tx, Ty = host.num_tiles(W, H)
groupsX = (Tx + 3) // 4
rows = band.tileRowEnd - band.tileRowBegin
head_rows = (16 * 96 + groupsX - 1) // groupsX      # light_cull.hip: 16 * HEAVY_MAX head blocks (a block per cluster tile) in front of the tile rows
if len(frame.lights) >= 262144:
    head_rows = 0                                    # ... not on the wide path (k1_group_lists_wide lists no clusters)
nh = head_rows * groupsX
nb = nh + groupsX * rows
assert nb <= 65536
p = buf[:nb].astype(np.int64)
ran = p[:, 3] > 0
t0 = p[ran, 0].min()
us = lambda v: (v - t0) / 100.0   # s_memrealtime: 100 MHz
dur = np.where(ran, (p[:, 3] - p[:, 0]) / 100.0, 0.0)
print("grid blocks", nb, "head blocks that ran (cluster tiles)", int(ran[:nh].sum()), "ordinary blocks that ran", int(ran[nh:].sum()), "kernel span us %.2f" % us(p[ran, 3].max()))
for name, sel in (("cluster tile", np.arange(nh)[ran[:nh]]), ("ordinary", nh + np.arange(nb - nh)[ran[nh:]])):
    if len(sel) == 0:
        continue
    d = dur[sel]
    print("%-12s blocks: n %5d  duration us mean %.2f median %.2f p90 %.2f p99 %.2f max %.2f;  first start %.2f last end %.2f" %
          (name, len(sel), d.mean(), np.median(d), np.percentile(d, 90), np.percentile(d, 99), d.max(), us(p[sel, 0].min()), us(p[sel, 3].max())))
    lat = (p[sel, 1] - p[sel, 0]) / 100.0
    print("             start -> tests done: mean %.2f p90 %.2f max %.2f" % (lat.mean(), np.percentile(lat, 90), lat.max()))
order = np.argsort(-np.where(ran, p[:, 3], 0))[:10]
print("last to end:")
for b in order:
    kind = "cluster tile" if b < nh else "ordinary (tile row %3d, group col %2d)" % ((b - nh) // groupsX, (b - nh) % groupsX)
    print("  block %5d %s: start %6.2f end %6.2f us" % (b, kind, us(p[b, 0]), us(p[b, 3])))
g_host, _ = fp.lists_to_host()
og, oi, cnt = None, None, None
print("longest ordinary blocks and their four tiles' list lengths (128 = a 196 -> 128 selection ran if the tile had more candidates):")
for b in (nh + np.argsort(-dur[nh:]))[:6]:
    r, c = (b - nh) // groupsX, (b - nh) % groupsX
    tiles = [r * Tx + 4 * c + w for w in range(4) if 4 * c + w < Tx]
    print("  block %5d (tile row %3d, group col %2d): %.2f us, tests done after %.2f us; list lengths %s" % (b, r, c, dur[b], (p[b, 1] - p[b, 0]) / 100.0, g_host[tiles, 1].tolist()))
starts = np.sort(us(p[ran, 0]))
print("block start times us: p10 %.1f p50 %.1f p90 %.1f max %.1f" % tuple(np.percentile(starts, [10, 50, 90, 100])))

# per XCD (HW_REG_XCC_ID as read by the block itself; s_memtime is comparable inside one XCD): the timeline of the launch
print("per XCD: blocks, span (us), mean resident blocks per CU, time by which 50 / 90 / 99 / 100 %% of the blocks have ended, kind of the last")
for x in range(8):
    sel = np.arange(nb)[(p[:, 2] == x) & ran]
    if len(sel) == 0:
        continue
    s0 = p[sel, 0].min()
    ends = np.sort(p[sel, 3] - s0) / 100.0
    span = ends[-1]
    conc = ((p[sel, 3] - p[sel, 0]).sum() / 100.0) / span / 32.0
    last = sel[np.argmax(p[sel, 3])]
    # resident blocks over time: sample at 10 points
    ts = np.linspace(0, span, 11)[1:-1]
    res = [int((((p[sel, 0] - s0) / 100.0 <= t) & ((p[sel, 3] - s0) / 100.0 > t)).sum()) for t in ts]
    print("  xcd %d: %5d blocks, span %7.1f, resident/CU %.2f, ended 50%% %6.1f 90%% %6.1f 99%% %6.1f; last = %s; resident blocks at 10..90%% of the span: %s" %
          (x, len(sel), span, conc, ends[len(ends) // 2], ends[int(len(ends) * 0.9)], ends[int(len(ends) * 0.99)], "cluster tile" if last < nh else "ordinary", res))
