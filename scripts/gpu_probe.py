import sys, os, subprocess
code = """
import sys
import numpy as np, torch
sys.path.insert(0, ".")
from sailor_amd import synth
from sailor_amd.forward_plus import HipContext, ForwardPlus, upload_lights
ctx = HipContext("cuda:0")
f = synth.make_frame("C3")
fp = ForwardPlus(ctx, f.cam.width, f.cam.height, len(f.lights))
d = torch.from_numpy(f.depth).to(ctx.device); l = upload_lights(f.lights, ctx.device); s = torch.from_numpy(f.surface).to(ctx.device)
fp.cull(f.cam.frame, l, len(f.lights), d)
for _ in range(3): fp.shade(f.cam.frame, s, l, len(f.lights), None)
torch.cuda.synchronize()
a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
a.record()
for _ in range(20): fp.shade(f.cam.frame, s, l, len(f.lights), None)
b.record(); torch.cuda.synchronize()
print("shade ms", a.elapsed_time(b) / 20)
"""
for dbg in ("0", "1", "2", "3", "4"):
    env = dict(os.environ, SAILOR_SHADE_DBG=dbg)
    r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True)
    print("dbg", dbg, r.stdout.strip().splitlines()[-1] if r.stdout.strip() else r.stderr[-300:])
