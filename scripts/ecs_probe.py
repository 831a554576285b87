"""Scratch probe: K4 sweep time for the standard 3-level hierarchy and for the same entities as roots only."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from sailor_amd import synth, host
from sailor_amd.forward_plus import HipContext, EcsSweep
ctx = HipContext("cuda:0")
cam = synth.make_camera(3840, 2160)
planes, _ = host.extract_frustum_planes(cam.world, cam.aspect, cam.fov, cam.z_near, cam.z_far)
for mode in ("standard", "roots"):
    ents = synth.make_entities(1 << 20)
    if mode == "roots":
        ents.parent[:] = 0xFFFFFFFF
        ents.level_offsets = np.array([0, 1 << 20], np.uint32)
    sw = EcsSweep(ctx, ents)
    for _ in range(5): sw.run(planes)
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(20): sw.run(planes)
    b.record(); torch.cuda.synchronize()
    print(mode, "levels", list(ents.level_offsets), "ms", a.elapsed_time(b) / 20)
