"""Randomised parity sweep, part 2 (run through gpurun; a bounded slice runs inside `pytest -m gpu`: tests/test_fuzz_gpu.py): K3 -- random viewports,
shadow-map sizes, shadow types, several directional lights, depth ranges that reach all cascades -- radiance within 1e-4 relative; K4 -- random hierarchies
(depth 1..6, ragged level sizes, degenerate scales, zero-size boxes) -- world matrices / boxes / visibility bit for bit.   usage: fuzz_csm_ecs.py [cases] [seed]"""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from sailor_amd.forward_plus import HipContext
import fuzz_cases

cases = int(sys.argv[1]) if len(sys.argv) > 1 else 40
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
ctx = HipContext("cuda:0")
worst = 0.0
for c in range(cases):
    worst = max(worst, fuzz_cases.k3_case(ctx, rng, c))
print("K3 fuzz ok:", cases, "cases, worst relative radiance error", worst, flush=True)
for c in range(cases):
    fuzz_cases.k4_case(ctx, rng, c)
print("K4 fuzz ok:", cases, "cases", flush=True)
