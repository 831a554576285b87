"""The shadow passes of bench.py alone (shadow_pass_block: 1 M entities as boxes, four 4096^2 cascades) -> one JSON line; with a -DRASTER_STATS library
(SAILOR_HIP_LIB) also the rasteriser's counters of ONE draw per cascade: superblocks looked at / alive, blocks looked at / alive, texels inside / written,
blocks written whole."""
import ctypes as C
import json
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import bench  # noqa: E402
from sailor_amd.forward_plus import HipContext  # noqa: E402

ctx = HipContext("cuda:0")
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 5
cascades = tuple(int(c) for c in sys.argv[2].split(",")) if len(sys.argv) > 2 else (0, 1, 2, 3)   # (one cascade alone: for a counter pass over its draws)
out = bench.shadow_pass_block(ctx, 1 << 20, 4096, steps, cascades=cascades)
if hasattr(ctx._lib, "sailor_hip_raster_stats"):
    st = (C.c_ulonglong * 16)()
    ctx._lib.sailor_hip_raster_stats(st, 1)
    out["stats_all_draws"] = {"superblocks_seen": st[0], "superblocks_alive": st[1], "blocks_seen": st[2], "blocks_alive": st[3], "texels_inside": st[4],
                              "texels_written": st[5], "blocks_written_whole": st[6], "extra": st[7]}
print(json.dumps(out))
