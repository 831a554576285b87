"""Times the shadow-pass chain on one GPU (bench.py's shadow_passes block alone).  usage: shadow_pass_probe.py [entities] [map size] [front to back: 1 | 0]"""
import json
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import bench  # noqa: E402
from sailor_amd.forward_plus import HipContext  # noqa: E402

count = int(sys.argv[1]) if len(sys.argv) > 1 else 1 << 20
size = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
ftb = (sys.argv[3] != "0") if len(sys.argv) > 3 else True
print(json.dumps(bench.shadow_pass_block(HipContext("cuda:0"), count, size, 5, front_to_back=ftb)))
