"""Per-WAVE start / end times of the shade on the one-block-per-tile grid (library built with EXTRA=-DSHADE_PROF): is the chip's wave-slot time spent shading,
or waiting -- for the block's slowest quadrant, for the dispatcher to refill a finished wave's slot?  usage: shade_wave_prof.py [R/G | 0/1] [C3 | C5] [out.npz]"""
import ctypes, sys
import numpy as np, torch
sys.path.insert(0, ".")
from sailor_amd import host, synth, _lib
from sailor_amd.forward_plus import HipContext, ForwardPlus, upload_lights, PreparedLights
import bench
cfg = sys.argv[2] if len(sys.argv) > 2 else "C3"
frame = bench.BenchFrame(cfg)
cam, W, H = frame.cam, frame.cam.width, frame.cam.height
N = len(frame.lights)
dev = torch.device("cuda", 0)
ctx = HipContext(dev)
r, g = (int(v) for v in (sys.argv[1] if len(sys.argv) > 1 else "0/1").split("/"))
band = host.band_for_rank(W, H, r, g) if g > 1 else host.band_whole_frame(W, H)
dl = upload_lights(frame.lights, dev)
prep = PreparedLights(ctx, dl, N)
fp = ForwardPlus(ctx, W, H, N, band=band, prepared=prep)
assert not fp.tile_order
rows = slice(band.fbRowBegin, band.fbRowBegin + band.fbRowCount)
dd = torch.from_numpy(np.ascontiguousarray(frame.depth[rows])).to(dev)
ds = torch.from_numpy(frame.surface_rows(rows.start, rows.stop)).to(dev)
fp.cull(cam.frame, dl, N, dd)
for _ in range(4):
    fp.shade(cam.frame, ds, dl, N)
torch.cuda.synchronize()
fp.shade(cam.frame, ds, dl, N)
torch.cuda.synchronize()
lib = _lib.load()
buf = np.zeros((65536, 12), dtype=np.uint64)
fn = lib.sailor_hip_debug_read_shade_wave_prof
fn.argtypes = [ctypes.c_void_p, ctypes.c_size_t]
assert fn(buf.ctypes.data, buf.nbytes) == 0
import os
WPB = 2 if os.environ.get("SAILOR_SHADE_HALF") == "1" else 4   # waves per block: the two-wave form (round 6) has two blocks per tile, half tiles side by side in x
Tx = fp.Tx; tpp = (Tx + 79) // 80; gx = 8 * tpp * (4 // WPB)
nrows = band.tileRowEnd - band.tileRowBegin
nb = min(gx * 10 * nrows, 65536)
lin = np.arange(nb); bx, by, bz = lin % gx, (lin // gx) % 10, lin // (gx * 10)
btx = (((bx - bz) & 7) + 8 * by) * tpp + (bx >> (3 if WPB == 4 else 4))
real = btx < Tx
p = buf[:nb][real].astype(np.int64)
t0 = p[:, 0:WPB].min()
st, en = (p[:, 0:WPB] - t0) / 100.0, (p[:, 4:4 + WPB] - t0) / 100.0     # us
hw = p[:, 8:8 + WPB]
xcd = (hw >> 32) & 0xF
h = hw & 0xFFFFFFFF
cu = (xcd << 8) | (((h >> 13) & 7) << 5) | (((h >> 12) & 1) << 4) | ((h >> 8) & 0xF)   # (XCD, SE, SH, CU)
simd = (h >> 4) & 3
slot = h & 0xF
if len(sys.argv) > 3:
    np.savez_compressed(sys.argv[3], st=st.astype(np.float32), en=en.astype(np.float32), cu=cu.astype(np.int32), simd=simd.astype(np.int8), slot=slot.astype(np.int8))
span = en.max()
life = en - st
print("%s %s: %d blocks, span %.1f us" % (cfg, "band %d/%d" % (r, g) if g > 1 else "whole frame", len(p), span))
print("wave life us: mean %.2f median %.2f p90 %.2f;  per block: slowest wave %.2f, fastest %.2f, mean %.2f  => a block's waves are busy %.0f %% of the block's life" %
      (life.mean(), np.median(life), np.percentile(life, 90), life.max(1).mean(), life.min(1).mean(), life.mean(1).mean(), 100 * life.sum() / (WPB * (en.max(1) - st.min(1))).sum()))
print("start skew inside a block (last wave's start - first wave's): mean %.2f us p90 %.2f" % ((st.max(1) - st.min(1)).mean(), np.percentile(st.max(1) - st.min(1), 90)))
# wave-slot occupancy over the steady part of the launch: per (CU, SIMD, slot) sort the waves that ran there, gaps between one's end and the next one's start
key = (cu.astype(np.int64) << 8) | (simd.astype(np.int64) << 4) | slot.astype(np.int64)
k, s, e = key.reshape(-1), st.reshape(-1), en.reshape(-1)
order = np.lexsort((s, k))
k, s, e = k[order], s[order], e[order]
same = k[1:] == k[:-1]
gap = (s[1:] - e[:-1])[same]
steady = (e[:-1][same] > 0.1 * span) & (s[1:][same] < 0.85 * span)
print("distinct wave slots seen: %d (of %d = 256 CUs x 4 SIMDs x 8); waves per slot mean %.1f" % (len(np.unique(k)), 256 * 32, len(k) / len(np.unique(k))))
gs = gap[steady]
print("gap between a wave's end and the next wave's start in the SAME slot (steady part): mean %.2f us median %.2f p10 %.2f p90 %.2f;  negative (hw id reused early) %.1f %%" %
      (gs.mean(), np.median(gs), np.percentile(gs, 10), np.percentile(gs, 90), 100 * (gs < 0).mean()))
busy = life.sum()
print("wave-slot time: busy %.0f slot us; %d slots x %.1f us span = %.0f  => occupancy %.1f %% over the whole span" % (busy, len(np.unique(k)), span, len(np.unique(k)) * span, 100 * busy / (len(np.unique(k)) * span)))
ts = np.arange(0, span, max(2.0, round(span / 40)))
print("live waves at t:", [int(((st <= t) & (en > t)).sum()) for t in ts])
