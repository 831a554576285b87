"""Per-block phase times of k1_tile_cull (library built with EXTRA=-DCULL_PROF): which blocks are the kernel's tail?  usage: cull_prof.py [R/G]"""
import ctypes, sys
import numpy as np, torch
sys.path.insert(0, ".")
from sailor_amd import host, synth, _lib
from sailor_amd.forward_plus import HipContext, ForwardPlus, upload_lights
import bench
frame = bench.BenchFrame("C3")
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
nb = groupsX * (band.tileRowEnd - band.tileRowBegin)
p = buf[:nb].astype(np.int64)
t0 = p[:, 0][p[:, 0] > 0].min()
dur = (p[:, 3] - p[:, 0]) / 100.0   # s_memtime: 100 MHz -> us
print("blocks", nb, "kernel span us", (p[:, 3].max() - t0) / 100.0)
print("block duration us: mean %.2f median %.2f p99 %.2f max %.2f" % (dur.mean(), np.median(dur), np.percentile(dur, 99), dur.max()))
order = np.argsort(-(p[:, 3] - t0))[:12]
for b in order:
    print("block %5d (tile row %3d, group col %2d): start %6.2f  test-done %6.2f  synced %6.2f  end %6.2f us" % (b, b // groupsX, b % groupsX, (p[b, 0] - t0) / 100.0, (p[b, 1] - t0) / 100.0, (p[b, 2] - t0) / 100.0, (p[b, 3] - t0) / 100.0))
print("longest blocks:")
for b in np.argsort(-dur)[:6]:
    print("block %5d (tile row %3d, group col %2d): start %6.2f  test-done %6.2f  synced %6.2f  end %6.2f" % (b, b // groupsX, b % groupsX, (p[b, 0] - t0) / 100.0, (p[b, 1] - t0) / 100.0, (p[b, 2] - t0) / 100.0, (p[b, 3] - t0) / 100.0))
starts = np.sort((p[:, 0] - t0) / 100.0)
print("block start times (units of 100 ticks): p10 %.1f p50 %.1f p90 %.1f max %.1f" % tuple(np.percentile(starts, [10, 50, 90, 100])))
g, idx = fp.lists_to_host()
print("lists of the latest block's tiles:", g[order[0] * 4:order[0] * 4 + 4, 1] if nb * 4 <= len(g) + 3 else "")
