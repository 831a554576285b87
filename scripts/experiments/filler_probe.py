"""What does the SHAPE of a block cost beside the shade?  The C3 shade on one stream, a latency-bound stand-in for k1_tile_cull (scripts/experiments/filler.hip:
the same number of waves, dependent L2 loads, the LDS a cull block claims) on another, as 256-thread blocks (one wave per SIMD: needs a free slot on all four
at once, like the shade's own blocks) and as 64-thread blocks (one wave: fits the slot a finished shade wave leaves).  Events around both; the shade's own
dispatch-packet reading beside it."""
import ctypes, sys
import numpy as np, torch
sys.path.insert(0, ".")
from sailor_amd.forward_plus import HipContext, ForwardPlus, upload_lights, PreparedLights
import bench

frame = bench.BenchFrame("C3")
cam, W, H, N = frame.cam, frame.cam.width, frame.cam.height, len(frame.lights)
dev = torch.device("cuda", 0)
sa, sb = torch.cuda.Stream(device=dev), torch.cuda.Stream(device=dev)
torch.cuda.set_stream(sa)
ctx = HipContext(dev, stream=sa)
dl = upload_lights(frame.lights, dev)
prep = PreparedLights(ctx, dl, N)
fp = ForwardPlus(ctx, W, H, N, prepared=prep)
dd = torch.from_numpy(frame.depth).to(dev)
ds = torch.from_numpy(frame.surface_rows(0, H)).to(dev)
fp.cull(cam.frame, dl, N, dd)
torch.cuda.synchronize()
lib = ctypes.CDLL("sailor_amd/csrc/ab/libfiller.so")
lib.filler_launch.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_uint32, ctypes.c_void_p]
words = 1 << 20  # 4 MB: L2 / Infinity-Cache resident
table = torch.randint(0, 2**31 - 1, (words,), dtype=torch.int32, device=dev)
sink = torch.zeros(4, dtype=torch.int32, device=dev)
WAVES = 35200

def filler(stream, threads, lds, iters, alu):
    blocks = WAVES * 64 // threads
    rc = lib.filler_launch(ctypes.c_void_p(stream.cuda_stream), blocks, threads, lds, iters, alu, table.data_ptr(), words - 1, sink.data_ptr())
    assert rc == 0, rc

def shade():
    fp.shade(cam.frame, ds, dl, N)

def timed(fn, reps=40, warm=8):
    out = []
    for i in range(reps + warm):
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(sa)
        fn()
        b.record(sa)
        torch.cuda.synchronize()
        if i >= warm:
            out.append(a.elapsed_time(b) * 1e3)
    return float(np.median(out))

def both(threads, lds, iters, alu, filler_first=True):
    def fn():
        ev = torch.cuda.Event()
        ev.record(sa)
        sb.wait_event(ev)
        if filler_first:
            filler(sb, threads, lds, iters, alu)
            shade()
        else:
            shade()
            filler(sb, threads, lds, iters, alu)
        done = torch.cuda.Event()
        done.record(sb)
        sa.wait_event(done)
    return fn

def shade_reading(fn, reps=30):
    r = []
    for i in range(reps):
        torch.cuda.synchronize()
        ctx.time_launches(i, 1)
        fn()
    torch.cuda.synchronize()
    return float(np.median([ctx.timed_launch_ms(i) * 1e3 for i in range(reps)]))

t_shade = timed(shade)
print("shade alone: %.1f us (events), %.1f us (its own dispatch reading)" % (t_shade, shade_reading(shade)))
for iters, alu in ((6, 8), (12, 8), (6, 64)):
    print("filler: %d dependent loads x %d multiply-adds per thread, %d waves" % (iters, alu, WAVES))
    for threads, lds in ((256, 16544), (256, 0), (64, 4136), (64, 0)):
        t_f = timed(lambda: filler(sa, threads, lds, iters, alu))
        t_b = timed(both(threads, lds, iters, alu, True))
        t_b2 = timed(both(threads, lds, iters, alu, False))
        s_in = shade_reading(both(threads, lds, iters, alu, True))
        print("   %3d-thread blocks, %5d B LDS: alone %6.1f us; beside the shade: both done after %6.1f us (filler launched first) / %6.1f (shade first) = shade + %5.1f / %5.1f; the shade's own reading %6.1f"
              % (threads, lds, t_f, t_b, t_b2, t_b - t_shade, t_b2 - t_shade, s_in))
