"""Does the frame pipeline lose time at the shade -> shade boundary?  In the captured pipeline (bench.py) consecutive shades sit on ONE stream and each has a
cross-stream edge to its frame's cull: the kernel trace shows ~10 us between the end of one shade and the start of the next.  Eager launches, C3:
  A: shades on one stream, culls on a second (the pipeline's dependencies, no graph)
  B: shades ALTERNATING between two streams (no order between consecutive shades: each waits for its own frame's cull only), culls on a third
usage: pipeline_streams_probe.py [C3] [steps]"""
import sys, time
import numpy as np, torch
sys.path.insert(0, ".")
from sailor_amd.forward_plus import HipContext, ForwardPlus, upload_lights, PreparedLights
import bench
cfg = sys.argv[1] if len(sys.argv) > 1 else "C3"
K = int(sys.argv[2]) if len(sys.argv) > 2 else 300
frame = bench.BenchFrame(cfg)
cam, W, H, N = frame.cam, frame.cam.width, frame.cam.height, len(frame.lights)
dev = torch.device("cuda", 0)
sA1, sA2, sC = (torch.cuda.Stream(device=dev) for _ in range(3))
torch.cuda.set_stream(sA1)
cA1, cA2, cC = HipContext(dev, stream=sA1), HipContext(dev, stream=sA2), HipContext(dev, stream=sC)
dl = upload_lights(frame.lights, dev)
prep = PreparedLights(cA1, dl, N)
S = 3
fps = [ForwardPlus(cA1, W, H, N, prepared=prep) for _ in range(S)]
dd = torch.from_numpy(frame.depth).to(dev)
ds = torch.from_numpy(frame.surface_rows(0, H)).to(dev)
for f in fps:
    f.cull(cam.frame, dl, N, dd); f.shade(cam.frame, ds, dl, N)
torch.cuda.synchronize()

def run(two_shade_streams, steps):
    shade_done = [None] * S
    cull_done = [None] * S
    # prologue: frame 0's lists
    fps[0].cull(cam.frame, dl, N, dd, ctx=cC, defer_pack=True); fps[0].pack(cC)
    e = torch.cuda.Event(); e.record(sC); cull_done[0] = e
    for k in range(steps):
        p, q = k % S, (k + 1) % S
        # cull(k + 1) into set q: behind the shade that last read set q
        if shade_done[q] is not None:
            sC.wait_event(shade_done[q])
        fps[q].cull(cam.frame, dl, N, dd, ctx=cC, defer_pack=True)
        e = torch.cuda.Event(); e.record(sC); cull_done[q] = e
        fps[q].pack(cC)
        # shade(k) from set p
        st, cx = ((sA1, cA1) if (k & 1) == 0 else (sA2, cA2)) if two_shade_streams else (sA1, cA1)
        st.wait_event(cull_done[p])
        fps[p].shade(cam.frame, ds, dl, N, ctx=cx)
        e = torch.cuda.Event(); e.record(st); shade_done[p] = e
    torch.cuda.synchronize()

for name, two in (("A: one shade stream", False), ("B: two shade streams", True), ("A: one shade stream", False), ("B: two shade streams", True)):
    run(two, 60)
    t0 = time.perf_counter()
    run(two, K)
    dt = time.perf_counter() - t0
    # host-only cost of issuing the same calls (nothing to wait for): is the host ahead of the GPU?
    print("%s: %.1f us per step over %d steps (eager)" % (name, dt / K * 1e6, K), flush=True)
