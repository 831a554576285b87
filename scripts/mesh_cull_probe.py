"""Times sailor_hip_mesh_cull_compact on one GPU (bench.py's mesh_cull_compact block alone).  usage: mesh_cull_probe.py [count] [batches]"""
import json
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import bench  # noqa: E402
from sailor_amd.forward_plus import HipContext  # noqa: E402

count = int(sys.argv[1]) if len(sys.argv) > 1 else 1 << 20
batches = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
ctx = HipContext("cuda:0")
print(json.dumps(bench.mesh_cull_block(ctx, count, batches, 20)))
