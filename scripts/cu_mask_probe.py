"""Can frame k+1's cull run on a few CUs while frame k's shade has the rest?  Eager launches on two CU-masked streams
(hipExtStreamCreateWithCUMask), two sets of list buffers, events between the streams.   usage: cu_mask_probe.py [cull CUs of every 8, e.g. 1]"""
import ctypes, sys, time
import numpy as np, torch
sys.path.insert(0, ".")
from sailor_amd import host
from sailor_amd.forward_plus import HipContext, ForwardPlus, upload_lights
import bench

frame = bench.BenchFrame("C3")
cam, W, H = frame.cam, frame.cam.width, frame.cam.height
N = len(frame.lights)
dev = torch.device("cuda", 0)
torch.cuda.set_device(0)
hip = ctypes.CDLL("libamdhip64.so")


def masked_stream(bits):
    words = (ctypes.c_uint32 * 8)(*bits)
    s = ctypes.c_void_p()
    rc = hip.hipExtStreamCreateWithCUMask(ctypes.byref(s), 8, words)
    assert rc == 0, rc
    return torch.cuda.ExternalStream(s.value, device=dev)


def run(cull_of_8, steps=300):
    # CU i belongs to the cull stream if (i % 8) < cull_of_8 (the bits of a mask word are spread over the XCDs by the driver; any regular pattern will do)
    if cull_of_8 is None:
        sa, sb = torch.cuda.Stream(device=dev), torch.cuda.Stream(device=dev)
    elif cull_of_8 < 0:
        sa, sb = torch.cuda.Stream(device=dev), torch.cuda.Stream(device=dev, priority=-1)
    else:
        cull_bits = [sum(1 << b for b in range(32) if (b % 8) < cull_of_8)] * 8
        shade_bits = [(~w) & 0xFFFFFFFF for w in cull_bits]
        sa, sb = masked_stream(shade_bits), masked_stream(cull_bits)
    ca, cb = HipContext(dev, stream=sa), HipContext(dev, stream=sb)
    band = host.band_whole_frame(W, H)
    fps = [ForwardPlus(ca, W, H, N, band=band) for _ in range(2)]
    dd = torch.from_numpy(np.ascontiguousarray(frame.depth)).to(dev)
    ds = torch.from_numpy(frame.surface_rows(0, H)).to(dev)
    dl = upload_lights(frame.lights, dev)
    torch.cuda.synchronize()
    for f in fps:
        f.cull(cam.frame, dl, N, dd, ctx=cb); torch.cuda.synchronize()
        f.shade(cam.frame, ds, dl, N, None); torch.cuda.synchronize()
    # is the mask honoured?  the shade alone on its stream
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(sa)
    with torch.cuda.stream(sa):
        for _ in range(20):
            fps[0].shade(cam.frame, ds, dl, N, None)
    e1.record(sa); torch.cuda.synchronize()
    print("   shade alone on its stream: %.1f us" % (e0.elapsed_time(e1) * 50.0), end="")
    e0.record(sb)
    with torch.cuda.stream(sb):
        for _ in range(20):
            fps[0].cull(cam.frame, dl, N, dd, ctx=cb)
    e1.record(sb); torch.cuda.synchronize()
    print("   cull alone on its stream: %.1f us" % (e0.elapsed_time(e1) * 50.0))
    cull_done = [torch.cuda.Event() for _ in range(2)]
    shade_done = [torch.cuda.Event() for _ in range(2)]
    fps[0].cull(cam.frame, dl, N, dd, ctx=cb); cull_done[0].record(sb)
    shade_done[1].record(sa)
    torch.cuda.synchronize()

    def step(k):
        p = k & 1
        sa.wait_event(cull_done[p])
        with torch.cuda.stream(sa):
            fps[p].shade(cam.frame, ds, dl, N, None)
        shade_done[p].record(sa)
        sb.wait_event(shade_done[1 - p])       # frame k-1 has read the other set
        with torch.cuda.stream(sb):
            fps[1 - p].cull(cam.frame, dl, N, dd, ctx=cb)
        cull_done[1 - p].record(sb)

    for k in range(50):
        step(k)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for k in range(50, 50 + steps):
        step(k)
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / steps * 1e3
    t0 = time.perf_counter()
    for k in range(steps):  # host cost of issuing a step (nothing to wait for on the GPU side is not possible: report wall of issue loop only)
        pass
    return ms


for c in [None] + [int(a) for a in sys.argv[1:]]:
    print("cull CUs of every 8:", c, " ms per step: %.4f" % run(c), flush=True)
