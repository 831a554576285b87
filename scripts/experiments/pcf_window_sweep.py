"""A longer run of tests/test_shade_gpu.py::test_pcf_window_on_noise_maps_of_odd_sizes: N frames, three random map sizes each (6 .. 3000), GPU against the oracle."""
import sys
import numpy as np
sys.path.insert(0, "."); sys.path.insert(0, "tests")
import torch
from sailor_amd.forward_plus import HipContext
import test_shade_gpu as t
n = int(sys.argv[1]) if len(sys.argv) > 1 else 100
ctx = HipContext(torch.device("cuda", 0))
worst = 0.0
for seed in range(n):
    rng = np.random.default_rng(9000 + seed)
    f = t._deep_frame(seed=100 + seed)
    shapes = []
    for k in (1, 2, 3):
        hi = 3000 if seed % 4 == 0 else 400
        shape = (int(rng.integers(6, hi)), int(rng.integers(6, hi)))
        if seed % 10 == 3:   # the widest maps the window takes, and one past them
            shape = (int(rng.integers(6, 48)), int(rng.choice([8192, 8191, 8193, 6000, 5000, 4097])))
        shapes.append(shape)
        f.shadows.maps[k] = t._hostile_map(("noise", "steps", "nonfinite", "signed_denormal")[(seed + k) % 4], shape, 7000 + 10 * seed + k)
    got, _ = t.gpu_frame(ctx, f)
    ref = t.oracle_frame(f)
    bad = np.abs(got - ref) > 1e-4 * np.abs(ref)
    rel = np.abs(got - ref) / np.maximum(np.abs(ref), 1e-30)
    worst = max(worst, float(rel[np.abs(ref) > 0].max()))
    if bad.any():
        print("MISMATCH seed", seed, shapes, int(bad.sum()), "values")
        sys.exit(1)
print("%d frames, three maps each: no mismatch; worst relative error %.2e" % (n, worst))
