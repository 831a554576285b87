"""Per-wave (= per-block) start / end times of the single-wave form of the shade (k2_shade_q_pt; library built with EXTRA=-DSHADE_PROF, SAILOR_SHADE_QUAD=1).
usage: SAILOR_SHADE_QUAD=1 shade_quad_prof.py [C3]"""
import ctypes, sys
import numpy as np, torch
sys.path.insert(0, ".")
from sailor_amd import host, _lib
from sailor_amd.forward_plus import HipContext, ForwardPlus, upload_lights, PreparedLights
import bench
cfg = sys.argv[1] if len(sys.argv) > 1 else "C3"
frame = bench.BenchFrame(cfg)
cam, W, H = frame.cam, frame.cam.width, frame.cam.height
N = len(frame.lights)
dev = torch.device("cuda", 0)
ctx = HipContext(dev)
dl = upload_lights(frame.lights, dev)
prep = PreparedLights(ctx, dl, N)
fp = ForwardPlus(ctx, W, H, N, prepared=prep)
dd = torch.from_numpy(frame.depth).to(dev)
ds = torch.from_numpy(frame.surface_rows(0, H)).to(dev)
fp.cull(cam.frame, dl, N, dd)
for _ in range(4):
    fp.shade(cam.frame, ds, dl, N)
torch.cuda.synchronize()
names = ctx.launches_of(lambda: fp.shade(cam.frame, ds, dl, N))
torch.cuda.synchronize()
lib = _lib.load()
NB = 262144
buf = np.zeros((NB, 4), dtype=np.uint64)
fn = lib.sailor_hip_debug_read_shade_prof
fn.argtypes = [ctypes.c_void_p, ctypes.c_size_t]
assert fn(buf.ctypes.data, buf.nbytes) == 0
Tx = fp.Tx; tpp = (Tx + 79) // 80; gx = 8 * tpp
nb = gx * 40 * fp.Ty
assert nb <= NB
lin = np.arange(nb); bx, by, bz = lin % gx, (lin // gx) % 40, lin // (gx * 40)
btx = (((bx - bz) & 7) + 8 * (by >> 2)) * tpp + (bx >> 3)
real = btx < Tx
p = buf[:nb][real].astype(np.int64)
g_host, _ = fp.lists_to_host()
num = g_host[(bz * Tx + btx)[real], 1]
t0 = p[:, 0].min()
st, en = (p[:, 0] - t0) / 100.0, (p[:, 3] - t0) / 100.0
hw = p[:, 2]; xcd = (hw >> 32) & 0xF; h = hw & 0xFFFFFFFF
cu = (xcd << 8) | (((h >> 13) & 7) << 5) | (((h >> 12) & 1) << 4) | ((h >> 8) & 0xF)
key = (cu << 8) | (((h >> 4) & 3) << 4) | (h & 0xF)
life = en - st
span = en.max()
print(names, "%d waves, span %.1f us; wave life mean %.2f median %.2f p90 %.2f p99 %.2f max %.2f; busy %.0f slot us" % (len(p), span, life.mean(), np.median(life), np.percentile(life, 90), np.percentile(life, 99), life.max(), life.sum()))
for lo, hi in ((0, 8), (8, 16), (16, 24), (24, 40), (40, 96), (96, 129)):
    m = (num >= lo) & (num < hi)
    if m.any():
        print("   list length %3d..%3d: %6d waves, life mean %.2f" % (lo, hi - 1, m.sum(), life[m].mean()))
order = np.lexsort((st, key)); k, s, e = key[order], st[order], en[order]
same = k[1:] == k[:-1]
gap = (s[1:] - e[:-1])[same]
steady = (e[:-1][same] > 0.1 * span) & (s[1:][same] < 0.85 * span)
gs = gap[steady]
print("distinct wave slots %d; gap end -> next start in the same slot (steady): mean %.2f median %.2f p10 %.2f p90 %.2f" % (len(np.unique(k)), gs.mean(), np.median(gs), np.percentile(gs, 10), np.percentile(gs, 90)))
print("occupancy over the span %.1f %%" % (100 * life.sum() / (len(np.unique(k)) * span)))
ts = np.arange(0, span, max(2.0, round(span / 40)))
print("live waves at t:", [int(((st <= t) & (en > t)).sum()) for t in ts])
