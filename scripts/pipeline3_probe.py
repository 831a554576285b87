"""Diagnostic: captures the frame pipeline with the deferred pack on a third stream for one (unroll, sets, variant) and replays it.
usage: pipeline3_probe.py UNROLL SETS VARIANT   (variant: side3 | side2tail | inline)"""
import faulthandler
import os
import sys

faulthandler.enable()
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
import torch

import bench
from sailor_amd import synth
from sailor_amd.forward_plus import ForwardPlus, HipContext, upload_lights

unroll, sets, variant = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3]
dev = torch.device("cuda", 0)
f = synth.make_frame("C2")
cam, W, H, N = f.cam, f.cam.width, f.cam.height, len(f.lights)
side, side2, side3 = torch.cuda.Stream(device=dev), torch.cuda.Stream(device=dev), torch.cuda.Stream(device=dev)
with torch.cuda.stream(side):
    ctx, ctx2, ctx3 = HipContext(dev, stream=side), HipContext(dev, stream=side2), HipContext(dev, stream=side3)
    d_depth = torch.from_numpy(np.ascontiguousarray(f.depth)).to(dev)
    d_surface = torch.from_numpy(np.ascontiguousarray(f.surface)).to(dev)
    d_l = upload_lights(f.lights, dev)
    fps = [ForwardPlus(ctx, W, H, N) for _ in range(sets)]
    for fp in fps:
        fp.cull(cam.frame, d_l, N, d_depth); fp.shade(cam.frame, d_surface, d_l, N, None)
    torch.cuda.synchronize()
    deferred = variant != "inline"
    shade_fns = [lambda fp=fp: fp.shade(cam.frame, d_surface, d_l, N, None) for fp in fps]
    cull_fns = [lambda fp=fp: fp.cull(cam.frame, d_l, N, d_depth, ctx=ctx2, defer_pack=deferred) for fp in fps]
    if variant == "side2tail":   # the pack on the cull's own stream, behind the event the shade waits for
        g = bench.capture_frame_pipeline(side, side2, unroll, shade_fns, cull_fns, None, [lambda fp=fp: fp.pack(ctx2) for fp in fps], side2)
    else:
        g = bench.capture_frame_pipeline(side, side2, unroll, shade_fns, cull_fns, None, [lambda fp=fp: fp.pack(ctx3) for fp in fps] if deferred else None, side3)
    fps[0].cull(cam.frame, d_l, N, d_depth)
    torch.cuda.synchronize()
    for _ in range(5):
        g.replay()
    torch.cuda.synchronize()
print("ok", unroll, sets, variant)
