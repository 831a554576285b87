"""Scratch probe run on the GPU box: ablation of k1_tile_cull."""
import sys
import numpy as np, torch
sys.path.insert(0, ".")
from sailor_amd import synth, host, _lib
from sailor_amd.forward_plus import HipContext, ForwardPlus, upload_lights

ctx = HipContext("cuda:0")
f = synth.make_frame("C3", with_surface=False)
fp = ForwardPlus(ctx, f.cam.width, f.cam.height, len(f.lights))
d = torch.from_numpy(f.depth).to(ctx.device); l = upload_lights(f.lights, ctx.device)
for dbg in (0, 1, 2, 3, 0):
    for _ in range(3):
        fp.cull(f.cam.frame, l, len(f.lights), d, dbg << 8)
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(20):
        fp.cull(f.cam.frame, l, len(f.lights), d, dbg << 8)
    b.record(); torch.cuda.synchronize()
    print("dbg", dbg, "cull ms", a.elapsed_time(b) / 20)
