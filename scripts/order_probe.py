import sys, os, ctypes as C
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np, torch
from sailor_amd import synth, host, _lib
from sailor_amd.forward_plus import HipContext, ForwardPlus, upload_lights
ctx = HipContext("cuda:0")
f = synth.make_frame("C3")
W, H, N = f.cam.width, f.cam.height, len(f.lights)
lib = _lib.load()
lib.sailor_hip_debug_set_tile_order.argtypes = [C.c_void_p]
d_lights = upload_lights(f.lights, ctx.device)
def run(band, label):
    fp = ForwardPlus(ctx, W, H, N, band=band)
    rows = slice(band.fbRowBegin, band.fbRowBegin + band.fbRowCount)
    d = torch.from_numpy(np.ascontiguousarray(f.depth[rows])).to(ctx.device)
    s = torch.from_numpy(np.ascontiguousarray(f.surface[:, rows])).to(ctx.device)
    fp.cull(f.cam.frame, d_lights, N, d)
    g, idx = fp.lists_to_host()
    num = g[:, 1].astype(np.int64)
    Tx = fp.Tx
    tiles = np.arange(len(num))
    packed = ((tiles % Tx) | ((tiles // Tx) << 16)).astype(np.uint32)
    for mode in ("natural", "heavy_first", "sorted_desc"):
        if mode == "natural": lib.sailor_hip_debug_set_tile_order(None); keep = None
        else:
            if mode == "heavy_first": o = np.concatenate([tiles[num >= 64], tiles[num < 64]])
            else: o = np.argsort(-num, kind="stable")
            keep = torch.from_numpy(packed[o]).to(ctx.device); lib.sailor_hip_debug_set_tile_order(keep.data_ptr())
        for _ in range(3): fp.shade(f.cam.frame, s, d_lights, N)
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(20): fp.shade(f.cam.frame, s, d_lights, N)
        b.record(); torch.cuda.synchronize()
        print(label, mode, "shade ms", round(a.elapsed_time(b) / 20, 4))
run(host.band_whole_frame(W, H), "whole")
run(host.band_for_rank(W, H, 2, 8), "band2/8")
run(host.band_for_rank(W, H, 3, 8), "band3/8")
