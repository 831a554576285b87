"""Probe: the two-frames-in-flight pipeline launched EAGERLY on two streams over S list sets (no hipGraph), against bench.py's graph form.
usage: pipeline_probe.py [config] [sets] [steps] [band r/G]"""
import sys, os, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sailor_amd import synth, host  # noqa: E402
from sailor_amd.forward_plus import HipContext, ForwardPlus, PreparedLights, upload_lights, upload_shadow_maps  # noqa: E402

cfg = sys.argv[1] if len(sys.argv) > 1 else "C3"
S = int(sys.argv[2]) if len(sys.argv) > 2 else 3
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 600
band = None
f = synth.make_frame(cfg)
W, H, N = f.cam.width, f.cam.height, len(f.lights)
if len(sys.argv) > 4:
    r, G = (int(v) for v in sys.argv[4].split("/"))
    band = host.band_for_rank(W, H, r, G)
dev = torch.device("cuda:0")
side, side2 = torch.cuda.Stream(device=dev), torch.cuda.Stream(device=dev)
ctx, ctx2 = HipContext(dev, stream=side), HipContext(dev, stream=side2)
r0, r1 = (0, H) if band is None else (band.fbRowBegin, band.fbRowBegin + band.fbRowCount)
d_depth = torch.from_numpy(np.ascontiguousarray(f.depth[r0:r1])).to(dev)
d_lights = upload_lights(f.lights, dev)
prep = PreparedLights(ctx, d_lights, N)
d_surface = torch.from_numpy(np.ascontiguousarray(f.surface[:, r0:r1])).to(dev)
csm = keep = None
if f.shadows is not None:
    csm, keep = upload_shadow_maps(f.shadows, dev)
fps = [ForwardPlus(ctx, W, H, N, band=band, prepared=prep) for _ in range(S)]
with torch.cuda.stream(side):
    for fp in fps:
        fp.cull(f.cam.frame, d_lights, N, d_depth)
        fp.shade(f.cam.frame, d_surface, d_lights, N, csm)
torch.cuda.synchronize()
shade_done = [torch.cuda.Event() for _ in range(S)]
cull_done = [torch.cuda.Event() for _ in range(S)]
for e in shade_done: e.record(side)
for e in cull_done: e.record(side)
torch.cuda.synchronize()

def run(n):
    for k in range(n):
        p, q = k % S, (k + 1) % S
        side.wait_event(cull_done[p])
        with torch.cuda.stream(side):
            fps[p].shade(f.cam.frame, d_surface, d_lights, N, csm)
        shade_done[p].record(side)
        side2.wait_event(shade_done[q])
        with torch.cuda.stream(side2):
            fps[q].cull(f.cam.frame, d_lights, N, d_depth, ctx=ctx2)
        cull_done[q].record(side2)

t = time.perf_counter()
while time.perf_counter() - t < 0.4:
    run(30); torch.cuda.synchronize()
th = time.perf_counter(); run(steps); host_ms = (time.perf_counter() - th) / steps * 1e3
torch.cuda.synchronize()
t0 = time.perf_counter(); run(steps); torch.cuda.synchronize()
ms = (time.perf_counter() - t0) / steps * 1e3
print(f"{cfg} band {sys.argv[4] if band else 'whole'}: eager two-stream pipeline over {S} list sets: {ms:.4f} ms per step (host launch cost {host_ms:.4f} ms per step)")
