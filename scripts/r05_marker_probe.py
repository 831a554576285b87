"""Does a kernel's dispatch-packet reading include its predecessor's drain?  The C3 frame, eager, one stream:
  A: cull chain (no pack) -> shade          B: ... -> marker -> shade          C: ... -> k1_pack -> shade
  D: ... -> a torch reduction over tileNum (130 KB read) -> shade           E: ... -> a torch reduction over the whole tile-list slot region (16.6 MB read) -> shade
per-kernel direct readings and the event-bracketed time of the whole sequence (what really passes)."""
import sys
import numpy as np, torch
sys.path.insert(0, ".")
from sailor_amd import _lib
from sailor_amd.forward_plus import HipContext, ForwardPlus, upload_lights, PreparedLights
import bench
cfg = sys.argv[1] if len(sys.argv) > 1 else "C3"
frame = bench.BenchFrame(cfg)
cam, W, H, N = frame.cam, frame.cam.width, frame.cam.height, len(frame.lights)
dev = torch.device("cuda", 0)
side = torch.cuda.Stream(device=dev)
torch.cuda.set_stream(side)
ctx = HipContext(dev, stream=side)
dl = upload_lights(frame.lights, dev)
prep = PreparedLights(ctx, dl, N)
fp = ForwardPlus(ctx, W, H, N, prepared=prep)
dd = torch.from_numpy(frame.depth).to(dev)
ds = torch.from_numpy(frame.surface_rows(0, H)).to(dev)
lib = ctx._lib
base = fp.workspace.data_ptr()
T = fp.band_tiles
v_num = fp.workspace[fp.tile_num - base: fp.tile_num - base + 4 * T].view(torch.int32)
v_lists = fp.workspace[fp.tile_lists - base: fp.tile_lists - base + 4 * 128 * T].view(torch.int32)
sink = torch.zeros(1, dtype=torch.int64, device=dev)
def seq(kind):
    fp.cull(cam.frame, dl, N, dd, defer_pack=True)
    if kind == "B":
        _lib.check(lib.sailor_hip_marker(ctx.handle), "marker", ctx.handle)
    if kind == "C":
        fp.pack()
    if kind == "D":
        sink.copy_(v_num.sum())
    if kind == "E":
        sink.copy_(v_lists.sum())
    fp.shade(cam.frame, ds, dl, N)
for kind in "ABCDE":
    names = ctx.launches_of(lambda: seq(kind))
    for _ in range(20):
        seq(kind)
    torch.cuda.synchronize()
    reps = 60
    acc = np.zeros((reps, len(names)))
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for i in range(reps):
        ctx.time_launches(i * len(names), len(names))
        seq(kind)
    b.record()
    torch.cuda.synchronize()
    for i in range(reps):
        acc[i] = [ctx.timed_launch_ms(i * len(names) + k) * 1e3 for k in range(len(names))]
    med = np.median(acc, 0)
    print(kind, dict(zip(names, np.round(med, 1))), "sum of readings %.1f us; per sequence by events %.1f us" % (med.sum(), a.elapsed_time(b) / reps * 1e3))
