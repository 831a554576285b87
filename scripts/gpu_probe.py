import sys, os, subprocess
code = """
import sys
import numpy as np, torch
sys.path.insert(0, ".")
from sailor_amd import synth, host
from sailor_amd.forward_plus import HipContext, ForwardPlus, upload_lights
ctx = HipContext("cuda:0")
f = synth.make_frame("C3", with_surface=False)
for band in (host.band_whole_frame(3840, 2160), host.band_for_rank(3840, 2160, 3, 8)):
    fp = ForwardPlus(ctx, f.cam.width, f.cam.height, len(f.lights), band=band)
    d = torch.from_numpy(np.ascontiguousarray(f.depth[band.fbRowBegin:band.fbRowBegin+band.fbRowCount])).to(ctx.device); l = upload_lights(f.lights, ctx.device)
    for _ in range(3): fp.cull(f.cam.frame, l, len(f.lights), d)
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(20): fp.cull(f.cam.frame, l, len(f.lights), d)
    b.record(); torch.cuda.synchronize()
    print("rows", band.tileRowBegin, band.tileRowEnd, "cull ms", a.elapsed_time(b) / 20)
"""
for dbg in ("0", "1", "2"):
    env = dict(os.environ, SAILOR_CULL_DBG=dbg)
    r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True)
    print("dbg", dbg, r.stdout.strip().splitlines()[-2:] if r.stdout.strip() else r.stderr[-300:])
