"""Histogram of k1_group_lists' candidate counts per 4x4-tile group on the C3 frame (reads the cull workspace: offGroupCount = align(tiles * 64, 256))."""
import sys, os
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from sailor_amd import host
from sailor_amd.forward_plus import HipContext, ForwardPlus, upload_lights
frame = bench.BenchFrame("C3")
cam, W, H = frame.cam, frame.cam.width, frame.cam.height
ctx = HipContext("cuda:0")
fp = ForwardPlus(ctx, W, H, len(frame.lights))
dd = torch.from_numpy(np.ascontiguousarray(frame.depth)).to(ctx.device)
dl = upload_lights(frame.lights, ctx.device)
fp.cull(cam.frame, dl, len(frame.lights), dd)
ctx.synchronize()
Tx, Ty = host.num_tiles(W, H)
tiles = Tx * Ty
off = (tiles * 64 + 255) // 256 * 256
groups = ((Tx + 3) // 4) * ((Ty + 3) // 4)
c = fp.workspace[off: off + 4 * groups].view(torch.int32).cpu().numpy().view(np.uint32)
listed = (c & 0x40000000) != 0
cnt = c & 0x3FFFFFFF
print("groups", groups, "listed heavy", int(listed.sum()), "mean", cnt.mean(), "max", cnt.max())
print("histogram (edges 0,64,...,576):", np.histogram(cnt, bins=list(range(0, 640, 64)) + [4096])[0].tolist())
g, _ = fp.lists_to_host()
num = g[:, 1].reshape(Ty, Tx)
sel_rows = []
for gy in range((Ty + 3) // 4):
    for gx in range((Tx + 3) // 4):
        gi = gy * ((Tx + 3) // 4) + gx
        for r in range(4):
            ty = gy * 4 + r
            if ty >= Ty: continue
            k = int((num[ty, gx * 4: gx * 4 + 4] == 128).sum())
            if k: sel_rows.append((int(cnt[gi]), k, bool(listed[gi])))
sel = np.array(sel_rows)
print("row blocks with full (128) lists:", len(sel), "of which in listed groups:", int(sel[:, 2].sum()))
un = sel[sel[:, 2] == 0]
print("unlisted ones: group candidate counts quantiles", np.percentile(un[:, 0], [0, 10, 50, 90, 100]).tolist(), "full tiles per block histogram", np.bincount(un[:, 1].astype(int)).tolist())
