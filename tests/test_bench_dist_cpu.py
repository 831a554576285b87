"""bench.py's N > 1 control flow on the CPU: the SAME bench.main() -- rank arithmetic, the calibration cull, gather_row_entries,
balanced_tile_rows, band set-up, the two-frames-in-flight pipeline over its list sets, static and dynamic lights, the exchange, alternate-frame
rendering, the JSON assembly -- on a `gloo` group of two, through a stand-in for bench.HipDevice whose "kernels" are the oracle.  What it cannot cover
is the device layer itself (streams, hipGraphs, RCCL): tests/test_bench_gpu.py and tests/test_runtime_gpu.py do that on the GPU box.  The stand-in
lives HERE, in the tests: bench.py and sailor_amd/ have no CPU path."""
import json
import os
import socket
import sys
import time
from contextlib import contextmanager
from pathlib import Path

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

ROOT = Path(__file__).resolve().parents[1]


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


class _Stream:
    cuda_stream = 0

    def wait_event(self, ev):
        pass

    def wait_stream(self, other):
        pass


class _Event:
    def __init__(self):
        self.t = None

    def record(self, stream=None):
        self.t = time.perf_counter()   # the stand-in's "kernels" run synchronously: recording an event is reading the clock

    def elapsed_time(self, other):
        return max((other.t - self.t) * 1e3, 1e-6)


class _Graph:
    def __init__(self, body):
        self.body = body

    def replay(self):
        self.body()


class _Context:
    handle = None

    def synchronize(self):
        pass

    def time_launches(self, first_slot, count):
        pass

    def timed_launch_ms(self, slot):
        return 0.01

    def launches_of(self, fn):   # (HipContext.launches_of: the kernels fn() launched, as the library's launch log names them)
        fn()
        return ["k01_prepare", "k1_group_lists", "k1_tile_cull", "k1_pack"]


class _EcsSweep:
    """EcsSweep's surface with the oracle behind it: a rank's slice fills its own words of the bitmask and nothing else"""

    def __init__(self, ents, rank, world):
        from sailor_amd.forward_plus import ecs_range_for_rank
        self.ents, self.rank, self.world = ents, rank, world
        self.n = len(ents.parent)
        self.begin, self.end, self.words_per_rank = ecs_range_for_rank(self.n, rank, world)
        self.visibility = torch.full((max((self.n + 63) // 64, world * self.words_per_rank),), -1, dtype=torch.int64)

    def run(self, planes):
        from oracle import oracle
        _, _, ov = oracle.ecs_sweep(self.ents.transforms, self.ents.parent, self.ents.local_aabb, planes)
        lo, hi = self.begin // 64, (self.end + 63) // 64
        if self.world == 1:
            self.visibility[: len(ov)] = torch.from_numpy(ov.view(np.int64).copy())
        else:
            self.visibility[self.rank * self.words_per_rank:(self.rank + 1) * self.words_per_rank] = 0
            self.visibility[lo:hi] = torch.from_numpy(ov[lo:hi].view(np.int64).copy())
        return None, None, self.visibility

    def exchange_visibility(self, comm=None, group=None):
        from sailor_amd import dist as sdist
        return sdist.allgather_visibility(self.visibility, self.rank, self.world, self.words_per_rank, group)


class _Prepared:
    def __init__(self, n):
        self.capacity, self.calls = n, 0

    def prepare(self, first, count, ctx=None):
        assert 0 <= first and first + count <= self.capacity
        self.calls += 1


class _ForwardPlus:
    """ForwardPlus's surface (sailor_amd/forward_plus.py) with the oracle behind it: band-local grid / culled in the C-ABI's layout"""

    def __init__(self, W, H, N, band, prepared):
        from sailor_amd import host
        self.W, self.H, self.N, self.band, self.prepared = W, H, N, band, prepared
        self.Tx, self.Ty = host.num_tiles(W, H)
        self.band_tiles = (band.tileRowEnd - band.tileRowBegin) * self.Tx
        self.grid = torch.zeros(max(self.band_tiles, 1) * 2, dtype=torch.int32)
        self.culled = torch.zeros(1 + max(self.band_tiles, 1) * 128, dtype=torch.int32)
        self.tile_order, self.use_tile_order = None, False
        self.lights = None

    def _full(self, rows, shape_tail=()):
        b = self.band
        full = np.zeros((self.H, self.W) + shape_tail, np.float32)
        full[b.fbRowBegin:b.fbRowBegin + b.fbRowCount] = rows
        return full

    def pack(self, ctx=None):
        assert self.deferred, "a pack follows a cull recorded with defer_pack"

    def cull(self, frame, lights, lights_num, depth, flags=0, ctx=None, prepared=None, defer_pack=False, prepare_lights=False):
        self.deferred = defer_pack
        if prepare_lights:
            self.prepared.prepare(0, lights_num)
        from oracle import oracle
        assert tuple(depth.shape) == (self.band.fbRowCount, self.W)
        g, idx, _ = oracle.light_cull(frame, self.W, self.H, lights[:lights_num], self._full(depth.numpy()), tile_rows=(self.band.tileRowBegin, self.band.tileRowEnd))
        self.grid[: self.band_tiles * 2] = torch.from_numpy(g.astype(np.int64).astype(np.int32).reshape(-1).copy())
        self.culled[: len(idx)] = torch.from_numpy(idx.view(np.int32).copy())
        self.lights = lights
        return self.grid, self.culled

    def shade(self, frame, surface, lights, lights_num, csm=None, out=None, ibl=None, prepared=None):
        from oracle import oracle
        b = self.band
        assert tuple(surface.shape) == (3, b.fbRowCount, self.W, 4)
        planes = np.zeros((3, self.H, self.W, 4), np.float32)
        planes[:, b.fbRowBegin:b.fbRowBegin + b.fbRowCount] = surface.numpy()
        grid = np.zeros((self.Tx * self.Ty, 2), np.uint32); grid[:, 0] = 1
        grid[b.tileRowBegin * self.Tx: b.tileRowEnd * self.Tx] = self.grid[: self.band_tiles * 2].numpy().view(np.uint32).reshape(-1, 2)
        rad = oracle.shade(frame, self.W, self.H, planes, lights[:lights_num], grid, self.culled.numpy().view(np.uint32), None, rows=(b.fbRowBegin, b.fbRowBegin + b.fbRowCount))
        self.radiance = torch.from_numpy(rad[b.fbRowBegin:b.fbRowBegin + b.fbRowCount])
        return self.radiance

    def lists_to_host(self):
        g = self.grid.numpy().view(np.uint32).reshape(-1, 2)[: self.band_tiles]
        c = self.culled.numpy().view(np.uint32)
        return g.copy(), c[: 1 + int(c[0])].copy()


class _Device:
    """bench.HipDevice's interface on the CPU: gloo, wall-clock "events", graphs that re-run their body, the oracle as the kernels"""
    dist_backend = "gloo"
    exchange_how = "sailor_amd.dist.exchange_lists over gloo (the CPU tests' stand-in for the C-ABI exchange)"

    def __init__(self, local_rank):
        self.device = torch.device("cpu")
        self.exchanges = 0

    def stream(self, priority=0):
        return _Stream()

    def set_stream(self, s):
        pass

    @contextmanager
    def on_stream(self, s):
        yield

    def synchronize(self):
        pass

    def event(self, timing=True):
        return _Event()

    def capture(self, stream, body):
        return _Graph(body)

    def context(self, stream):
        return _Context()

    def upload(self, a):
        return torch.from_numpy(np.ascontiguousarray(a))

    def upload_lights(self, lights):
        return lights

    def prepared_lights(self, ctx, d_lights, n):
        return _Prepared(n)

    def forward_plus(self, ctx, W, H, N, band, prepared):
        return _ForwardPlus(W, H, N, band, prepared)

    def upload_shadow_maps(self, shadows):
        return None, None   # (the stand-in's shade ignores the shadow maps: the control flow is what is under test)

    def ecs_sweep(self, ctx, entities, rank=0, world=1):
        return _EcsSweep(entities, rank, world)

    def exchange_visibility(self, sweep):
        return sweep.exchange_visibility()

    def make_comm(self, rank, world):
        self.rank, self.world = rank, world

    def exchange(self, ctx, W, H, bounds, fp):
        from sailor_amd import dist as sdist
        from sailor_amd import host
        assert len(bounds) == self.world + 1 and bounds[0] == 0 and bounds[-1] == host.num_tiles(W, H)[1]
        assert (fp.band.tileRowBegin, fp.band.tileRowEnd) == (bounds[self.rank], bounds[self.rank + 1]), "this rank's band is its slot of the split"
        self.exchanges += 1
        return sdist.exchange_lists(fp.grid[: fp.band_tiles * 2], fp.culled)

    def close(self):
        pass


def _worker(rank, world, port, extra, out_dir):
    sys.path.insert(0, str(ROOT))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world))
    import bench
    bench.BATCHES, bench.BATCH_LAUNCHES = 2, 2   # (the oracle is the kernel here: two launches per batch are as good as fifty)
    made = []

    def factory(local_rank):
        made.append(_Device(local_rank))
        return made[-1]
    line = bench.main(["--gpus", str(world), "--config", "tiny", "--steps", "3", "--warmup", "1", "--spinup-ms", "0", "--no-cpu-baseline"] + list(extra), device_factory=factory)
    assert (line is not None) == (rank == 0), "rank 0 alone assembles the JSON line"
    with open(Path(out_dir) / f"rank{rank}.json", "w") as f:
        json.dump({"line": json.loads(line) if line else None, "exchanges": made[0].exchanges}, f)


def _oracle_whole_frame():
    sys.path.insert(0, str(ROOT))
    from oracle import oracle
    from sailor_amd import synth
    f = synth.make_frame("tiny", with_surface=False)
    g, idx, _ = oracle.light_cull(f.cam.frame, f.cam.width, f.cam.height, f.lights, f.depth)
    return g, idx


@pytest.mark.parametrize("extra", [(), ("--equal-bands",), ("--frame-per-gpu",), ("--dynamic-lights", "--list-sets", "2"), ("--exchange-every-step",),
                                   ("--split-configs", "tiny_csm,tiny", "--split-config-steps", "3")],
                         ids=["balanced", "equal", "frame_per_gpu", "dynamic_two_sets", "exchange_every_step", "split_configs"])
def test_two_ranks_run_the_whole_of_bench_main(extra, tmp_path):
    world = 2
    mp.spawn(_worker, args=(world, _free_port(), extra, str(tmp_path)), nprocs=world, join=True)
    res = [json.load(open(tmp_path / f"rank{r}.json")) for r in range(world)]
    d = res[0]["line"]
    assert res[1]["line"] is None
    g, idx = _oracle_whole_frame()
    total = int(idx[0])
    weak = "--frame-per-gpu" in extra
    assert d["n_gpus"] == 2 and d["steps"] == 3 and d["scaling"] == ("weak" if weak else "strong") and d["value"] > 0 and d["unit"] == "Mpixels/s"
    assert abs(d["value"] - (2 if weak else 1) * 128 * 96 * 3 / (d["ms_per_step"] * 3e-3) / 1e6) < 1e-6 * d["value"]
    ex = d["split_frame"]["exchange"] if weak else d["exchange"]
    # the stitched global lists are the whole frame's: every band's entries at their canonical place
    assert ex["tiles"] == len(g) and ex["global_sum_num"] == total and ex["checksum"] == int(idx[1:1 + total].astype(np.int64).sum())
    bounds = d["split_frame"]["tile_row_bounds"] if weak else ex["tile_row_bounds"]
    assert bounds[0] == 0 and bounds[-1] == 6 and len(bounds) == 3 and bounds[0] < bounds[1] < bounds[2]
    if "--equal-bands" in extra:
        assert bounds == [0, 3, 6] and d["config"]["partition"] == "equal tile rows"
    if weak:
        assert d["split_frame"]["scaling"] == "strong" and d["split_frame"]["value"] > 0
    else:
        afr = d["alternate_frame_rendering"]
        assert afr["scaling"] == "weak" and afr["value"] > 0 and d["speedup_vs_one_gpu_whole_frame"] > 0
        assert d["value_serial"] > 0 and d["serial_step_ms"]["max_over_ranks"] >= d["serial_step_ms"]["median"] * (1 - 1e-9)
    # both light modes are in the line, and the mode says which one `value` is
    dynamic = "--dynamic-lights" in extra
    assert d["lights"]["mode"].startswith("dynamic" if dynamic else "static") and d["lights"]["prepare_lights_ms"] > 0
    other = "static" if dynamic else "dynamic"
    assert d[f"value_{other}_lights"] > 0 and d[f"ms_per_step_{other}"] > 0
    if "--list-sets" in extra:
        assert "2 list sets" in d["launch"]
    import bench
    timed = 1 + bench.EXCHANGE_TIMED   # (one exchange on the worst-case slots, then the event-timed ones: bench.exchange_stats)
    if "--exchange-every-step" in extra:
        assert d["launch"] == "eager" and res[0]["exchanges"] == res[1]["exchanges"] > 3 + timed   # every rank took part in every exchange
    elif weak:
        assert res[0]["exchanges"] == res[1]["exchanges"] == timed
    else:
        per_config = timed * (len(d["split_configs"]) if d.get("split_configs") else 0)
        assert res[0]["exchanges"] == res[1]["exchanges"] == timed + max(d["warmup"], 2) + d["steps"] + per_config
        # round 6: the exchange's cost is in the line -- its own event-timed duration, the bytes a rank gathers, and the step with it in it
        assert ex["ms_median"] > 0 and ex["ms_p90"] >= ex["ms_median"] and ex["timed_exchanges"] == bench.EXCHANGE_TIMED and ex["bytes_gathered"] > 0
        assert d["value_exchange_every_step"] > 0 and d["ms_per_step_exchange_every_step"] > 0 and d["ms_per_step_same_form_without_exchange"] > 0
    if weak:
        sx = d["split_frame"]["exchange"]
        assert sx["ms_median"] > 0 and sx["timed_exchanges"] == bench.EXCHANGE_TIMED
    r = d["roofline"]
    assert r["bound"] == "hbm" and r["avg_launch_ms"] > 0 and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-12
    assert list(r["cull"]["kernels_ms"]) == ["k01_prepare", "k1_group_lists", "k1_tile_cull", "k1_pack"], "the chain's kernels as the library's launch log names them"
    # K4 across the ranks (entity ranges + one all-gather of the visibility words) beside K4 replicated, and the gathered bitmask is the replicated one
    e = d["ecs_sweep"]["split"]
    assert e["default"] == "replicated" and e["replicated_ms"] > 0 and e["split"]["ranks"] == 2 and e["split"]["bitmask_equals_replicated_on_every_rank"] is True
    assert e["split"]["slice_plus_allgather_ms"] >= e["split"]["slice_ms"] * 0.5
    if "--split-configs" in extra:
        sc = d["split_configs"]
        assert list(sc) == ["tiny_csm", "tiny"]
        for name, v in sc.items():
            assert v["config"] == name and v["split_ms_per_step"] > 0 and v["whole_frame_per_gpu_ms_per_step"] > 0 and v["value"] > 0, v
            assert v["tile_row_bounds"][0] == 0 and v["tile_row_bounds"][-1] == 6 and len(v["tile_row_bounds"]) == 3
            assert abs(v["speedup_vs_one_gpu_whole_frame"] - v["whole_frame_per_gpu_ms_per_step"] / v["split_ms_per_step"]) < 1e-9
    else:
        assert "split_configs" not in d


def test_a_bare_gpus_n_starts_its_ranks_as_child_processes(monkeypatch):
    """`python bench.py --gpus N` with WORLD_SIZE unset must not exit with a usage message: it launches torch.distributed.run
    as a child BEFORE touching any device and leaves with the child's code."""
    sys.path.insert(0, str(ROOT))
    import subprocess

    import bench
    calls = []
    monkeypatch.delenv("WORLD_SIZE", raising=False)
    monkeypatch.setattr(subprocess, "call", lambda cmd, **kw: calls.append(cmd) or 7)
    monkeypatch.setattr(bench, "HipDevice", lambda *_: pytest.fail("the parent must not create a device"))
    with pytest.raises(SystemExit) as e:
        bench.main(["--gpus", "4", "--steps", "5"])
    assert e.value.code == 7 and len(calls) == 1
    cmd = calls[0]
    assert cmd[:3] == [sys.executable, "-m", "torch.distributed.run"] and "--nproc-per-node=4" in cmd and "--nnodes=1" in cmd
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1" and int(cmd[cmd.index("--master-port") + 1]) > 0
    assert cmd[-5:] == [str(ROOT / "bench.py"), "--gpus", "4", "--steps", "5"]
