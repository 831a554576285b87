"""Scratch probe: cull diagnostics (group list statistics) and per-kernel timing of one configuration."""
import sys, os
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sailor_amd import synth, host
from sailor_amd.forward_plus import HipContext, ForwardPlus, upload_lights
cfg = sys.argv[1] if len(sys.argv) > 1 else "C3"
ctx = HipContext("cuda:0")
f = synth.make_frame(cfg, with_surface=False)
W, H, N = f.cam.width, f.cam.height, len(f.lights)
fp = ForwardPlus(ctx, W, H, N)
d = torch.from_numpy(np.ascontiguousarray(f.depth)).to(ctx.device)
l = upload_lights(f.lights, ctx.device)
for _ in range(3):
    fp.cull(f.cam.frame, l, N, d)
torch.cuda.synchronize()
a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
a.record()
for _ in range(20):
    fp.cull(f.cam.frame, l, N, d)
b.record(); torch.cuda.synchronize()
print("cull ms", a.elapsed_time(b) / 20)
diag = fp.cull_diagnostics(N)
names = ["numBands", "maskBits", "numGroups", "sumGroupLists", "overflowGroups", "longestGroupList", "wordsPerBand", "colBits"]
print(diag)
g, idx = fp.lists_to_host()
num = g[:, 1].astype(np.int64)
print("tiles", len(num), "mean list", num.mean(), "tiles with 128:", int((num >= 128).sum()), "max", num.max())
