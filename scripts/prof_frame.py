"""Profiling target: a few cull + shade passes of one synthetic frame (default C3), nothing else on the GPU.
Usage under rocprofv3:  rocprofv3 --kernel-trace --pmc ... -- python3 scripts/prof_frame.py [C3] [passes]"""
import sys, os
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sailor_amd import synth, host  # noqa: E402
from sailor_amd.forward_plus import HipContext, ForwardPlus, PreparedLights, upload_lights, upload_shadow_maps  # noqa: E402

cfg = sys.argv[1] if len(sys.argv) > 1 else "C3"
passes = int(sys.argv[2]) if len(sys.argv) > 2 else 3
ctx = HipContext("cuda:0")
f = synth.make_frame(cfg)
W, H, N = f.cam.width, f.cam.height, len(f.lights)
d_depth = torch.from_numpy(np.ascontiguousarray(f.depth)).to(ctx.device)
d_lights = upload_lights(f.lights, ctx.device)
fp = ForwardPlus(ctx, W, H, N, prepared=None if os.environ.get("SAILOR_PLAIN_LIGHTS") else PreparedLights(ctx, d_lights, N))  # the path as the HIP backend drives it
d_surface = torch.from_numpy(np.ascontiguousarray(f.surface)).to(ctx.device)
csm = keep = None
if f.shadows is not None:
    csm, keep = upload_shadow_maps(f.shadows, ctx.device)
for _ in range(passes):
    fp.cull(f.cam.frame, d_lights, N, d_depth)
    fp.shade(f.cam.frame, d_surface, d_lights, N, csm)
torch.cuda.synchronize()
print("done", cfg, passes)
