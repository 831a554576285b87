#!/usr/bin/env python3
"""Headline benchmark of the MI355X-native Forward+ lighting path (BASELINE.json: "lit Mpixels/s + Mlights culled/s at
4K / 65 536 lights; 1/2/4/8-GPU scaling").

    python bench.py --gpus N --steps K --warmup W        (N > 1 without a launcher: this process starts the N ranks itself, as child processes)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...

One step = one pass of the hot path over one synthetic frame: K0 light view transform + K1 tile light cull + K2 PBR shade
over the per-tile lists (BASELINE.json configs[2]: 4K, 65 536 point+spot lights, synthetic G-buffer/surface tiles), with
every input already resident in HBM.  With N > 1 the step is ONE frame cut into N cost-balanced tile-row bands, one band per
rank (BASELINE.json's multi-GPU metric, SURVEY.md 8e: strong scaling; cull + shade need no collective), and
`value` = W*H*K / max-over-ranks(time); after the timed steps the band lists are exchanged once through the SHIPPED exchange --
sailor_hip_exchange_light_lists_rows (exchange.hip: three ncclAllGather on an ncclComm_t of the job's ranks + one stitch kernel), the
call HipGraphicsDriver::ExchangeLightLists makes -- and the stitched global buffers are summarised by a checksum, so the distributed
path is exercised end to end (`--exchange-every-step` puts the exchange inside the timed region).  The same invocation also measures
the other way of using N GPUs -- every rank renders whole frames of its own -- and reports it as `alternate_frame_rendering`, together
with `speedup_vs_one_gpu_whole_frame` = that whole-frame step time / the split step time.  `--frame-per-gpu` swaps the roles.

Lights: static (the prepared views derived once, outside the timed region) or dynamic (every light dirty every frame: the
preparation of all N lights inside every step).  C5 ("1 M dynamic lights") is quoted dynamic, the others static; the line always
carries both step times (`value_dynamic_lights` / `ms_per_step_dynamic`, resp. `..._static`) and `lights.mode` says which is `value`.

Rank 0 prints ONE JSON line.  `roofline` is for the dominant kernel (k2_shade): algorithmic bytes per launch
/ its launch duration read DIRECTLY: one HIP event pair around every launch, each launch between its real neighbours
(kernel_in_frame_ms).  `cpu_baseline` = the CPU oracle (a port: the reference cannot be built here) timed on a bounded
sample of the same workload on this box's host cores.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

import numpy as np
import torch
from ctypes import c_float as C_float
import ctypes as _C
C_float_p = _C.POINTER(_C.c_float)
L_R16F, L_RGBA32F = 0, 1

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

from sailor_amd import _lib, host, synth  # noqa: E402
from sailor_amd.forward_plus import PreparedLights, EcsSweep, ForwardPlus, HipContext, upload_lights  # noqa: E402

HBM_PEAK_GBS = 8000.0       # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
HBM_COPY_GBS = 6290.0       # same table: 6.29 TB/s measured float4 copy
FP32_VALU_TFLOPS = 157.3


def parse(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--spinup-ms", type=float, default=300.0, help="untimed clock spin-up before the warm-up steps (0 = none)")
    ap.add_argument("--config", default="C3", choices=["C2", "C3", "C4", "C5", "tiny"], help="`tiny` (128 x 96, 512 lights) is the CPU control-flow test's frame, not a bench line")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-sample-tile-rows", type=int, default=96)
    ap.add_argument("--exchange-every-step", action="store_true", help="include the RCCL list exchange in the timed region (N > 1)")
    ap.add_argument("--list-sets", type=int, default=3, choices=[2, 3],
                    help="sets of list buffers (grid / culledLights / workspace) the two-frames-in-flight pipeline rotates through: see capture_frame_pipeline")
    ap.add_argument("--frames-in-flight", type=int, default=2, choices=[1, 2],
                    help="2 (the reference's MaxFramesInQueue, RHI/Renderer.h:34): frame k+1's cull is recorded on a second stream beside frame k's shade")
    ap.add_argument("--plain-lights", action="store_true", help="no prepared lights: cull and shade read the 112-byte light records (rounds 1-2)")
    ap.add_argument("--dynamic-lights", action="store_true", default=None,
                    help="every light is dirty every frame (LightingECS::Tick re-uploads it, ECS/LightingECS.cpp:152-191): sailor_hip_prepare_lights over all N lights inside "
                         "every step, in front of the cull.  The default for C5 (\"1 M dynamic lights\"); the other configurations report it beside `value` as value_dynamic_lights")
    ap.add_argument("--static-lights", dest="dynamic_lights", action="store_false", help="the lights' prepared views are derived once, outside the timed region (the default except for C5)")
    ap.add_argument("--separate-prepare", action="store_true", help="dynamic lights: sailor_hip_prepare_lights as a launch of its own in front of every cull (round 3) instead of "
                                                                    "folded into the cull's per-light pass (SAILOR_CULL_PREPARE_LIGHTS)")
    ap.add_argument("--single-mode", action="store_true", help="profiling runs (rocprofv3 kernel stats, PMC passes): only the headline's light mode and launch form -- no second "
                                                               "pass in the other light mode, no one-frame-in-flight reading with the pack beside the shade -- so that every kernel of the "
                                                               "trace was launched the same way")
    ap.add_argument("--pack", default="deferred", choices=["never", "deferred", "inline"],
                    help="when k1_pack -- the compaction of the per-tile lists into the reference's lightsGrid / culledLights -- runs.  deferred (default): every frame, "
                         "behind the event the shade waits for.  never: no frame runs it (the shade reads the per-tile lists; the reference's only reader of the two buffers "
                         "IS the shade, Standard.shader:422-436) -- they stay available bit for bit on demand (sailor_hip_light_cull_pack: the N > 1 exchange and the list "
                         "read-back call it) and the kernel is reported as `pack_ms`; measured in round 5: no gain -- the frame pipeline hides the launch, and a shade that "
                         "follows k1_tile_cull directly takes ~9 us LONGER than one that follows the pack (profiles/r05/README.md).  inline: inside every cull (rounds 1-3)")
    ap.add_argument("--pack-inline", action="store_true", help="= --pack inline")
    ap.add_argument("--no-graph", action="store_true", help="launch every kernel eagerly instead of replaying one captured hipGraph per step")
    ap.add_argument("--equal-bands", action="store_true", help="N > 1: equal tile-row bands instead of cost-balanced ones")
    ap.add_argument("--split-frame", action="store_true", help="N > 1: ONE frame split into tile-row bands is `value` (the default; kept for old command lines)")
    ap.add_argument("--frame-per-gpu", action="store_true",
                    help="N > 1: make a whole frame per GPU (weak scaling, no data-path collective) the primary measurement; by default ONE frame is "
                         "split into tile-row bands (strong scaling, BASELINE.json's metric) and the frame-per-GPU reading is reported next to it")
    ap.add_argument("--no-afr", action="store_true", help="N > 1: skip the supplementary measurement (the other of the two multi-GPU modes)")
    ap.add_argument("--simulate-split", type=int, default=0, help="G: time each band of a cost-balanced G-way split one after the other on this GPU and print the predicted speed-up; diagnostic")
    ap.add_argument("--simulate-band", default=None, help="R/G: time only band R of a G-way split in this single process (no collectives); diagnostic")
    ap.add_argument("--force-dist", action="store_true", help="initialise the RCCL process group and run the list exchange even with one rank")
    ap.add_argument("--split-configs", default=None,
                    help="N > 1: comma-separated configurations whose split frame gets a bounded reading (~20 steps each) appended to the line as `split_configs`; "
                         "default: C4,C5 -- the two BASELINE.json names for eight GPUs -- when the headline is C3, none otherwise; '' = none")
    ap.add_argument("--split-config-steps", type=int, default=24)
    ap.add_argument("--rebalance", type=int, default=2,
                    help="N > 1 (and --simulate-split): rounds of re-cutting the cost-balanced bands on MEASURED band times (a renderer has the previous frame's); 0 = the "
                         "list-volume model alone")
    return ap.parse_args(argv)


class HipDevice:
    """The device layer of this benchmark: torch-ROCm streams, events and hipGraphs around the C-ABI library.  Everything main() does to a GPU goes
    through one of these, so that tests/test_bench_dist_cpu.py can drive the SAME main() -- rank arithmetic, calibration, band set-up, exchange, JSON
    assembly -- through a stand-in with this interface on a `gloo` group of two (the oracle as the kernels): a rank-count bug must not first appear
    on an 8-GPU box.  There is no such stand-in in the product or in this file: without a HIP device this class refuses to exist."""
    dist_backend = "nccl"   # = RCCL on ROCm

    def __init__(self, local_rank: int):
        if not torch.cuda.is_available():
            raise SystemExit("bench.py needs a HIP device: the path has no CPU fallback")
        # SAILOR_BENCH_SHARE_GPU=1 (diagnostic, 1-GPU boxes): every rank of an N > 1 run takes device 0 and the process group is `gloo` -- RCCL refuses two
        # ranks on one device, so the communicator of the shipped exchange is not created either and the ranks agree on the torch.distributed
        # exchange.  The ranks time-share the GPU: the throughput means nothing, but the N > 1 path runs for real on a GPU (band kernels,
        # calibration, re-split, exchange, every collective of main()): tests/test_bench_gpu.py::test_two_ranks_sharing_the_gpu.
        if os.environ.get("SAILOR_BENCH_SHARE_GPU") == "1":
            local_rank = 0
            self.dist_backend = "gloo"
        torch.cuda.set_device(local_rank)
        self.device = torch.device("cuda", local_rank)
        self._comm = None
        self._exchanges, self._last_exchange = {}, None

    # -- streams, events, graphs
    def stream(self, priority: int = 0):
        return torch.cuda.Stream(device=self.device, priority=priority)

    def set_stream(self, stream):
        torch.cuda.set_stream(stream)

    def on_stream(self, stream):
        return torch.cuda.stream(stream)

    def synchronize(self):
        torch.cuda.synchronize()

    def event(self, timing: bool = True):
        return torch.cuda.Event(enable_timing=timing)

    def capture(self, stream, body):
        """body() recorded into one hipGraph on `stream`; .replay() launches it"""
        g = torch.cuda.CUDAGraph()
        # In an NCCL process group the watchdog thread polls its work objects' events with hipEventQuery until it has seen them complete (every 100 ms), and a
        # query that lands inside a stream capture of this thread fails -- the watchdog then takes the process down with a C++ back trace.  Seen about once in
        # fifteen one-rank runs when a capture follows a collective closely (round 6).  Two defences: the capture is THREAD-LOCAL (other threads' calls are then
        # legal by the API's own rules), and -- because one failure was seen with that alone -- no collective is left for the watchdog to poll: everything this
        # rank enqueued is complete (synchronize) and the watchdog gets two of its periods to retire it.  Outside every timed region.
        if self.dist_backend == "nccl" and torch.distributed.is_available() and torch.distributed.is_initialized():
            torch.cuda.synchronize(self.device)
            time.sleep(0.25)
        with torch.cuda.graph(g, stream=stream, capture_error_mode="thread_local"):
            body()
        return g

    # -- the path (C-ABI) and its buffers
    def context(self, stream):
        return HipContext(self.device, stream=stream)

    def upload(self, array):
        return torch.from_numpy(np.ascontiguousarray(array)).to(self.device)

    def upload_lights(self, lights):
        return upload_lights(lights, self.device)

    def prepared_lights(self, ctx, d_lights, n):
        return PreparedLights(ctx, d_lights, n)

    def forward_plus(self, ctx, W, H, N, band, prepared):
        return ForwardPlus(ctx, W, H, N, band=band, prepared=prepared)

    def upload_shadow_maps(self, shadows):
        from sailor_amd.forward_plus import upload_shadow_maps
        return upload_shadow_maps(shadows, self.device)

    def ecs_sweep(self, ctx, entities, rank=0, world=1):
        """K4 over this rank's slice of an equal split of the entities (the whole set when world == 1)"""
        return EcsSweep(ctx, entities, rank, world)

    def exchange_visibility(self, sweep):
        """the slices' visibility words -> the whole bitmask on every rank: sailor_hip_exchange_visibility on the job's ncclComm_t, or the same all-gather
        over torch.distributed where there is none"""
        return sweep.exchange_visibility(self._comm)

    def copy_probe_ms(self, ctx, nbytes=512 << 20, launches=20):
        """THIS box's yardstick: a float4 streaming copy of `nbytes` (sailor_hip_copy_probe), each launch's own dispatch-packet timestamps; median ms"""
        src = torch.empty(nbytes, dtype=torch.uint8, device=self.device).fill_(1)
        dst = torch.empty_like(src)
        lib = ctx._lib
        for _ in range(3):
            _lib.check(lib.sailor_hip_copy_probe(ctx.handle, src.data_ptr(), dst.data_ptr(), nbytes), "sailor_hip_copy_probe", ctx.handle)
        ctx.synchronize()
        for i in range(launches):
            ctx.time_launches(i, 1)
            _lib.check(lib.sailor_hip_copy_probe(ctx.handle, src.data_ptr(), dst.data_ptr(), nbytes), "sailor_hip_copy_probe", ctx.handle)
        ctx.synchronize()
        return float(np.median([ctx.timed_launch_ms(i) for i in range(launches)]))

    # -- the split frame's exchange: the SHIPPED one (exchange.hip over an ncclComm_t), not torch.distributed's collectives
    def make_comm(self, rank: int, world: int):
        """the job's ncclComm_t for the C-ABI exchange.  If it cannot be had on EVERY rank (agreed by an all-reduce: a rank that went on alone would hang
        the others in the first collective), the exchange falls back to torch.distributed's collectives and the line says so."""
        import torch.distributed as dist
        from sailor_amd import dist as sdist
        ok = 1
        try:
            if self.dist_backend != "nccl":
                raise RuntimeError("the ranks share one device (SAILOR_BENCH_SHARE_GPU): RCCL takes one rank per device")
            self._comm = sdist.RcclComm(rank, world)
        except Exception as e:
            ok = 0
            print(f"[bench] rank {rank}: no ncclComm_t for the C-ABI exchange ({type(e).__name__}: {e})", file=sys.stderr)
        flag = torch.tensor([ok], dtype=torch.int32, device=self.device)
        dist.all_reduce(flag, op=dist.ReduceOp.MIN)
        if int(flag.item()) != 1:
            if self._comm is not None:
                self._comm.close()
            self._comm = None
            self.exchange_how = f"sailor_amd.dist.exchange_lists over torch.distributed ({self.dist_backend}): the C-ABI exchange's own communicator could not be created on every rank"

    def exchange(self, ctx, W, H, bounds, fp):
        """this rank's band lists -> the frame's canonical (lightsGrid, culledLights) on every rank: sailor_hip_exchange_light_lists_rows, RECORDED on the
        context's stream (round 6: the call reads nothing back and does not wait; workspace and global buffers are kept from call to call)"""
        from sailor_amd import dist as sdist
        if self._comm is None:
            return sdist.exchange_lists(fp.grid[: fp.band_tiles * 2], fp.culled)
        key = (id(ctx), W, H, tuple(int(b) for b in bounds))
        ex = self._exchanges.get(key)
        if ex is None:
            ex = self._exchanges[key] = sdist.ListExchange(ctx, self._comm, W, H, bounds, fp.culled.device)
        self._last_exchange = ex
        return ex.record(fp.grid[: fp.band_tiles * 2], fp.culled)

    def exchange_adapt(self):
        """sailor_hip_exchange_adapt on the context of the last exchange (every rank, same point): {largest_band_total, clipped, slot_words, bytes_gathered}"""
        ex = self._last_exchange
        if ex is None:
            return None
        largest, clipped, slot = ex.adapt()
        return {"largest_band_total": largest, "clipped": clipped, "slot_words": slot, "worst_case_slot_words": ex.worst_case_slot_words,
                "bytes_gathered_per_rank": ex.bytes_gathered()}

    exchange_how = "sailor_hip_exchange_light_lists_rows (C-ABI: 3 x ncclAllGather on an ncclComm_t of the job's ranks + one stitch kernel)"

    def close(self):
        self._exchanges.clear()
        self._last_exchange = None
        if self._comm is not None:
            self._comm.close()
            self._comm = None


_DEV = None   # the device layer the timing helpers below use (set by main(); a HipDevice of the current device otherwise)


def _dev():
    global _DEV
    if _DEV is None:
        _DEV = HipDevice(torch.cuda.current_device() if torch.cuda.is_available() else 0)
    return _DEV


EXCHANGE_TIMED = 10   # event-timed exchanges per reading (after one on the worst-case slots and the adapt call)


def exchange_stats(dev, do_exchange, stream, dist=None, device=None, n=EXCHANGE_TIMED):
    """The split frame's exchange, timed: ONE exchange on the worst-case slots, sailor_hip_exchange_adapt on every rank (the slots of the
    second gather sized from that exchange's gathered totals), then `n` exchanges, each between a HIP event pair on the launch stream.  -> (global grid,
    global culledLights, {ms_median, ms_p90, ms_min, ms_first_worst_case_slots, bytes_gathered, ...}); the medians are the MAX over the ranks."""
    a0, b0 = dev.event(), dev.event()
    a0.record(stream)
    gg, gi = do_exchange()
    b0.record(stream)
    dev.synchronize()
    first_ms = a0.elapsed_time(b0)
    adapt = dev.exchange_adapt() if hasattr(dev, "exchange_adapt") else None
    pairs = []
    for _ in range(n):
        a, b = dev.event(), dev.event()
        a.record(stream)
        gg, gi = do_exchange()
        b.record(stream)
        pairs.append((a, b))
    dev.synchronize()
    ms = np.array([a.elapsed_time(b) for a, b in pairs])
    after = dev.exchange_adapt() if hasattr(dev, "exchange_adapt") else None
    med, p90, mn = float(np.median(ms)), float(np.percentile(ms, 90)), float(ms.min())
    if dist is not None:
        t = torch.tensor([med, p90, first_ms], dtype=torch.float64, device=device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        med, p90, first_ms = (float(v) for v in t.tolist())
    out = {"ms_median": med, "ms_p90": p90, "ms_min": mn, "timed_exchanges": n, "ms_first_worst_case_slots": first_ms,
           "timing": "one exchange on the worst-case slots, sailor_hip_exchange_adapt on every rank, then %d exchanges each between a HIP event pair on the launch "
                  "stream (record-only calls: nothing read back, no stream synchronisation inside); median / p90 = the max over the ranks" % n}
    if adapt is not None:
        out.update({"bytes_gathered": adapt["bytes_gathered_per_rank"], "slot_words": adapt["slot_words"], "worst_case_slot_words": adapt["worst_case_slot_words"],
                    "largest_band_total": adapt["largest_band_total"], "clipped": bool(after["clipped"]) if after else None})
    else:
        tot = int(gi[0].item())
        out.update({"bytes_gathered": 4 * (1 + tot) + 8 * int(gg.numel() // 2), "slot_words": None,
                    "note": "torch.distributed stand-in (no ncclComm_t on this job): bytes_gathered = the global lists' own size"})
    return gg, gi, out


def event_ms(fn, steps):
    """average duration of fn() in ms, HIP events on the current (= launch) stream, one pair per call."""
    dev = _dev()
    pairs = []
    for _ in range(steps):
        a, b = dev.event(), dev.event()
        a.record(); fn(); b.record()
        pairs.append((a, b))
    dev.synchronize()
    t = np.array([a.elapsed_time(b) for a, b in pairs])
    return float(t.mean()), float(np.median(t)), float(np.percentile(t, 10)), float(np.percentile(t, 90))


def kernel_in_frame_ms(ctx, before, kernel, launches, kernels_per_call=1, which=0):
    """The duration of ONE kernel, read DIRECTLY and in place: every launch of it carries a HIP event pair on its own dispatch packet
    (sailor_hip_context_time_launches -> hipExtLaunchKernel's start / stop events), i.e. the command processor's timestamps of that kernel -- the
    figure rocprofv3's kernel trace reports -- while the launch sits between its real neighbours: before(); kernel(); before(); ... on the launch
    stream, eagerly, nothing drained around it.  Nothing is subtracted and no two medians are combined.  (Events recorded in front of and behind a
    launch drain the stream on both sides -- isolated_* -- and fifty launches of the kernel alone back to back run into each other's write-back --
    back_to_back_launch_ms.)  kernels_per_call / which: kernel() launches several kernels and the `which`-th is the one wanted."""
    dev = _dev()
    for i in range(launches):
        before()
        ctx.time_launches(i * kernels_per_call, kernels_per_call)
        kernel()
    dev.synchronize()
    t = np.array([ctx.timed_launch_ms(i * kernels_per_call + which) for i in range(launches)])
    return {"mean": float(t.mean()), "median": float(np.median(t)), "p10": float(np.percentile(t, 10)), "p90": float(np.percentile(t, 90)),
            "min": float(t.min()), "max": float(t.max()), "launches": int(len(t))}


def chain_kernels_ms(ctx, chain, behind, launches, count):
    """the durations of the `count` kernels one chain() call launches (the cull chain), each by the event pair on its own dispatch packet, between
    their real neighbours (behind() = the rest of the frame): median over `launches` frames per kernel, in launch order"""
    dev = _dev()
    for i in range(launches):
        ctx.time_launches(i * count, count)
        chain()
        behind()
    dev.synchronize()
    t = np.array([[ctx.timed_launch_ms(i * count + k) for k in range(count)] for i in range(launches)])
    return [float(v) for v in np.median(t, axis=0)]


BATCHES = 5          # per-kernel figures: the median of this many batches ...
BATCH_LAUNCHES = 50  # ... of at least this many back-to-back launches each, whatever --steps is (the driver runs --steps 20)


def event_batch_stats(fn, steps, graph_stream=None, batches=BATCHES):
    """Duration of fn() in ms from HIP event pairs around batches of back-to-back calls on the launch stream: what a launch costs inside a running
    pipeline (the next launch ramps up while the previous one drains).  `batches` batches of max(steps, BATCH_LAUNCHES) calls each; the figure is
    the MEDIAN batch (min / max kept alongside), so a short --steps or one disturbed batch does not move it.  An event pair around every single launch
    (event_ms) drains the GPU on both sides of the kernel and reads 10-15 % longer for a 0.2 ms kernel.  With graph_stream the calls of a batch are
    captured into one hipGraph first and the pair brackets its replay: eager launches of one kernel back to back leave the command processor's
    dispatch gap (5-15 us) between them, which is no part of the kernel; inside a graph the gap is what it is in the timed region.  What the pair
    includes besides the kernels' own durations (rocprofv3's kernel trace reports those): the gap between consecutive launches and the end-of-kernel
    write-back -- see `launch_gap_ms` in the JSON line."""
    dev = _dev()
    n = max(int(steps), BATCH_LAUNCHES)
    times = []
    if graph_stream is not None:
        try:
            def body():
                for _ in range(n):
                    fn()
            g = dev.capture(graph_stream, body)
            g.replay()
            dev.synchronize()
            pairs = []
            for _ in range(batches):
                a, b = dev.event(), dev.event()
                a.record(graph_stream)
                g.replay()
                b.record(graph_stream)
                pairs.append((a, b))
            dev.synchronize()
            times = [a.elapsed_time(b) / n for a, b in pairs]
        except Exception as e:  # (a failed capture leaves the eager measurement)
            print(f"[bench] per-kernel hipGraph capture failed ({type(e).__name__}: {e}); timing eager launches", file=sys.stderr)
            times = []
    if not times:
        pairs = []
        for _ in range(batches):
            a, b = dev.event(), dev.event()
            a.record()
            for _ in range(n):
                fn()
            b.record()
            pairs.append((a, b))
        dev.synchronize()
        times = [a.elapsed_time(b) / n for a, b in pairs]
    t = np.array(times)
    return {"median": float(np.median(t)), "min": float(t.min()), "max": float(t.max()), "batches": int(len(t)), "launches_per_batch": n}


def event_batch_ms(fn, steps, graph_stream=None):
    """the median batch of event_batch_stats"""
    return event_batch_stats(fn, steps, graph_stream)["median"]


def side_ms(fn, steps):
    """the side blocks' timing: event_batch_ms in event_ms's (mean, median, p10, p90) shape -- one figure, the in-pipeline cost of a launch"""
    b = event_batch_ms(fn, steps)
    return b, b, b, b


def trace_kernel_ms(kernel: str, config: str, world: int):
    """the kernel's average duration in the newest committed `rocprofv3 --kernel-trace --stats` summary of this configuration
    (profiles/rNN/kernel_stats[_<config>].csv), for comparison with the event figure; None when there is none"""
    if world != 1:
        return None
    prof = os.path.join(ROOT, "profiles")
    rounds = sorted((d for d in os.listdir(prof) if d.startswith("r") and d[1:].isdigit()), reverse=True) if os.path.isdir(prof) else []
    name = "kernel_stats.csv" if config == "C3" else f"kernel_stats_{config}.csv"
    for r in rounds:
        try:
            import csv
            with open(os.path.join(prof, r, name), newline="") as f:
                for row in csv.DictReader(f):
                    if row["Name"].startswith(kernel + "(") or row["Name"].startswith("void " + kernel + "("):
                        return float(row["AverageNs"]) * 1e-6
        except (OSError, ValueError, KeyError):
            pass
    return None


def measured_traffic(kernel: str, config: str, world: int):
    """HBM bytes per launch of the dominant kernel from the committed PMC passes (profiles/<latest round>/traffic[_<config>].json: separate
    rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE runs of this same command, corrected as MI355X_MICROARCH.md prescribes).
    bench.py cannot sample PMC counters itself; None when no matching profile is committed."""
    prof = os.path.join(ROOT, "profiles")
    rounds = sorted((d for d in os.listdir(prof) if d.startswith("r") and d[1:].isdigit()), reverse=True) if os.path.isdir(prof) else []
    for r in rounds:
        for name in ("traffic.json", f"traffic_{config}.json"):
            try:
                with open(os.path.join(prof, r, name)) as f:
                    t = json.load(f)
                if t.get("config") == config and world == 1:
                    return t["kernels"][kernel]["hbm_bytes_per_launch"]
            except (OSError, KeyError, ValueError):
                pass
    return None


def csm_distinct_texels(frame, shadows, surface_pos: np.ndarray, normals: np.ndarray):
    """D_c of SURVEY.md 8(d): the number of DISTINCT texels that the K3 lookups of the frame's directional lights fetch in cascade c (EVSM: one
    bilinear 2 x 2 footprint in cascade 0; PCF: sixteen of them) -- the algorithmic bytes of K3 are 256 + sum_c b_c D_c with b_0 = 16, b_1..3 = 2.
    Vectorised restatement of the lookups' addressing only (Standard.shader:266-283, Lighting.glsl:168-216, 242-284), float32 like the kernel."""
    fb = np.frombuffer(bytes(frame.cam.frame), np.uint8)
    view = fb[:64].view(np.float32).reshape(4, 4).astype(np.float32)            # view[c][r]
    z_far = float(fb[216:224].view(np.float32)[1])
    levels = np.float32([0.05, 0.1, 0.333333, 0.5]) * np.float32(z_far)
    poisson = np.float32([(-0.94201624, -0.39906216), (0.94558609, -0.76890725), (-0.094184101, -0.92938870), (0.34495938, 0.29387760),
                          (-0.91588581, 0.45771432), (-0.81544232, -0.87912464), (-0.38277543, 0.27676845), (0.97484398, 0.75648379),
                          (0.44323325, -0.97511554), (0.53742981, -0.47373420), (-0.26496911, -0.41893023), (0.79197514, 0.19090188),
                          (-0.24188840, 0.99706507), (-0.81409955, 0.91437590), (0.19984126, 0.78641367), (0.14383161, -0.14100790)])
    directional = [l for l in frame.lights if int(l["type"]) == 0]
    sizes = [(m.shape[1], m.shape[0]) for m in shadows.maps]
    seen = [np.zeros(h * w, bool) for (w, h) in sizes]
    P = surface_pos.reshape(-1, 3).astype(np.float32)
    for light in directional:
        pz_view = (P @ view[:3, 2] + view[3, 2]) / (P @ view[:3, 3] + view[3, 3])
        cascade = np.minimum(np.searchsorted(levels, np.abs(pz_view), side="right"), 3)
        for c in range(4):
            sel = np.nonzero(cascade == c)[0]
            if sel.size == 0:
                continue
            M = shadows.lights_matrices[c].reshape(4, 4).astype(np.float32)        # M[col][row]
            q = P[sel] @ M[:3, :] + M[3, :]
            px = (q[:, 0] / q[:, 3]) * np.float32(0.5) + np.float32(0.5)
            py = np.float32(1.0) - ((q[:, 1] / q[:, 3]) * np.float32(0.5) + np.float32(0.5))
            pz = q[:, 2] / q[:, 3]
            evsm = int(light["shadowType"]) == 2 and c == 0
            inside = (px <= 1) & (py <= 1) & (px >= 0) & (py >= 0) & ((pz >= 0) if evsm else (pz * np.float32(0.5) + np.float32(0.5) >= 0.5))
            px, py = px[inside], py[inside]
            w, h = sizes[c]
            taps = [(np.float32(0), np.float32(0))] if evsm else [(poisson[i, 0] * 2 / np.float32(w), poisson[i, 1] * 2 / np.float32(h)) for i in range(16)]
            for ox, oy in taps:
                x = (px + ox) * np.float32(w) - np.float32(0.5); y = (py + oy) * np.float32(h) - np.float32(0.5)
                x0 = np.floor(x).astype(np.int64); y0 = np.floor(y).astype(np.int64)
                for dx in (0, 1):
                    for dy in (0, 1):
                        seen[c][np.clip(y0 + dy, 0, h - 1) * w + np.clip(x0 + dx, 0, w - 1)] = True
    return [int(m.sum()) for m in seen]


def cpu_baseline(frame, sample_rows: int):
    """The oracle (CPU restatement of the reference algorithm) on a bounded band of the same frame, 1 thread."""
    from oracle import oracle
    cam, W, H = frame.cam, frame.cam.width, frame.cam.height
    Tx, Ty = host.num_tiles(W, H)
    r0 = (Ty - sample_rows) // 2
    r1 = r0 + sample_rows
    fb0, fb1 = H - 16 * r1, H - 16 * r0
    surface = frame.surface_rows(fb0, fb1)
    t0 = time.perf_counter()
    g, idx, _ = oracle.light_cull(cam.frame, W, H, frame.lights, frame.depth, tile_rows=(r0, r1))
    t_cull = time.perf_counter() - t0
    grid = np.zeros((Tx * Ty, 2), np.uint32); grid[:, 0] = 1
    grid[r0 * Tx:r1 * Tx] = g
    planes = np.zeros((3, H, W, 4), np.float32)  # the oracle addresses rows of full-frame planes; untouched pages stay virtual
    planes[:, fb0:fb1] = surface
    t0 = time.perf_counter()
    oracle.shade(cam.frame, W, H, planes, frame.lights, grid, idx, None, rows=(fb0, fb1))
    t_shade = time.perf_counter() - t0
    pixels = (fb1 - fb0) * W
    return {"value": pixels / (t_cull + t_shade) / 1e6, "unit": "Mpixels/s", "cores": 1, "kind": "port",
            "sample": f"tile rows [{r0},{r1}) of {Ty} ({pixels} pixels): oracle cull {t_cull:.2f} s + shade {t_shade:.2f} s, scalar C, gcc -O2 -ffp-contract=off",
            "mlights_culled_per_s": len(frame.lights) * (sample_rows / Ty) / t_cull / 1e6,
            "host_cpu": _cpu_model(), "host_cores": os.cpu_count()}


def _cpu_model():
    try:
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.startswith("model name"):
                    return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def ecs_baseline(ctx, count: int, steps: int):
    """K4 next to the reference's CPU ECS/frustum-cull loop: oracle port, 1 thread and all host cores."""
    from oracle import oracle
    ents = synth.make_entities(count)
    cam = synth.make_camera(3840, 2160)
    planes, _ = host.extract_frustum_planes(cam.world, cam.aspect, cam.fov, cam.z_near, cam.z_far)
    sweep = EcsSweep(ctx, ents)
    for _ in range(3):
        sweep.run(planes)
    mean, med, _, _ = side_ms(lambda: sweep.run(planes), steps)
    world = np.zeros((count, 16), np.float32); aabb = np.zeros((count, 6), np.float32); vis = np.zeros((count + 63) // 64, np.uint64)
    t0 = time.perf_counter()
    oracle.ecs_sweep(ents.transforms, ents.parent, ents.local_aabb, planes, world=world, world_aabb=aabb, visibility=vis)
    t1 = time.perf_counter() - t0
    cores = os.cpu_count() or 1
    # all host cores: the reference's own parallel form -- 1 024-entity chunks on worker threads (ECS/StaticMeshRendererECS.cpp:19), one hierarchy
    # level after the other -- inside the C oracle (pthreads; the seconds are taken between the start barrier and the last level's barrier)
    tn = min(oracle.ecs_sweep_threads(ents.transforms, ents.parent, ents.local_aabb, planes, ents.level_offsets, cores, world, aabb, vis)[3] for _ in range(3))
    # the cull loop alone, scalar (Math/Bounds.cpp:245-260) and the reference's SSE batch form (:264-325, restated literally --
    # including its layout bug -- purely as a cost proxy): 16-byte aligned input, n % 4 == 0
    import ctypes as C
    L = oracle.lib()
    n4 = count & ~3
    boxes = np.zeros(n4 * 6 + 4, np.float32)
    off = (-boxes.ctypes.data % 16) // 4
    al = boxes[off:off + n4 * 6]
    al[:] = aabb[:n4].reshape(-1)
    res = np.zeros(n4, np.int32)
    pl = np.ascontiguousarray(planes, np.float32).reshape(24)
    t0 = time.perf_counter()
    L.oracle_overlaps_aabb_sse(pl.ctypes.data_as(C.c_void_p), al.ctypes.data_as(C.c_void_p), C.c_uint32(n4), res.ctypes.data_as(C.c_void_p))
    t_sse = time.perf_counter() - t0
    # shadow-pass planning on the sweep's world boxes: four cascade frusta (LightingECS.cpp:287-296) x all entities -> four bitmasks
    from sailor_amd.forward_plus import csm_caster_masks
    sh = synth.make_shadow_set(cam, 16)
    cplanes = np.stack([host.extract_frustum_planes_matrix(sh.lights_matrices[k])[0] for k in range(4)])
    _, casc_ms, _, _ = side_ms(lambda: csm_caster_masks(ctx, sweep.world_aabb, cplanes), steps)
    bytes_per_entity = 164.125
    return {"entities": count, "gpu_ms": med, "csm_caster_masks_ms": casc_ms, "csm_caster_masks_gbs": count * 24.5 / casc_ms / 1e6, "gpu_mentities_per_s": count / med / 1e3, "gpu_hbm_gbs": count * bytes_per_entity / med / 1e6,
            "gpu_hbm_frac": count * bytes_per_entity / med / 1e6 / HBM_PEAK_GBS,
            "cpu_1thread_mentities_per_s": count / t1 / 1e6, "cpu_ns_per_entity_1thread": t1 / count * 1e9,
            "cpu_allcores_mentities_per_s": count / tn / 1e6, "cpu_allcores_speedup_vs_1thread": t1 / tn,
            "cpu_allcores_gbs": count * bytes_per_entity / tn / 1e9, "cpu_sse_cull_only_mboxes_per_s_1thread": n4 / t_sse / 1e6, "cpu_cores": cores, "cpu_model": _cpu_model(), "kind": "port"}


def mesh_cull_block(ctx, count: int, num_batches: int, steps: int):
    """SURVEY.md 8f rank 4 (the half with defined semantics): ComputeMeshCulling.shader main() without OCCLUSION_CULLING = frustum
    flags + per-draw stable compaction of the 96-byte instance records.  The buffers are compacted in place, so every timed call
    starts from a fresh device copy (the copy is outside the events).  Algorithmic bytes: 84 read + 4 written per instance for
    the flags; 96 read per instance + 96 written per moved record + 28 per batch for the compaction."""
    from oracle import oracle
    from sailor_amd.forward_plus import MeshCull
    cam = synth.make_camera(3840, 2160)
    s = synth.make_instance_set(count, num_batches)
    mc = MeshCull(ctx, s.instances, s.batches)
    inst0, batch0 = mc.instances.clone(), mc.batches.clone()
    t = []
    for _ in range(steps + 3):
        mc.instances.copy_(inst0); mc.batches.copy_(batch0)
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); mc.run(cam.frame); b.record()
        torch.cuda.synchronize()
        t.append(a.elapsed_time(b))
    med = float(np.median(t[3:]))
    t0 = time.perf_counter()
    ref_i, ref_b = oracle.mesh_cull_compact(cam.frame, s.instances, count, 0, s.batches)
    t1 = time.perf_counter() - t0
    kept = int(ref_b[:, 1].sum())
    moved = int((ref_i["materialInstance"] != np.arange(count, dtype=np.uint32)).sum())
    algo = count * (84 + 4) + count * 96 + moved * 96 + num_batches * 28
    # the shader as shipped (OCCLUSION_CULLING): Hi-Z pyramid of the half-resolution raw depth (DefaultRenderer.renderer: ViewportWidth/2 squared,
    # 8 B per output texel and level) + frustum || occlusion + compaction
    from sailor_amd.forward_plus import hiz_build
    hw, hh, levels = cam.width // 2, cam.height // 2, 11
    raw = torch.from_numpy(synth.make_raw_depth(synth.make_linear_depth(hw, hh, 9, d_min=200.0, d_max=2500.0), cam.frame.cameraZNearZFar[0])).to(ctx.device)
    pyr = hiz_build(ctx, raw, hw, hw, levels)
    _, hiz_ms, _, _ = side_ms(lambda: hiz_build(ctx, raw, hw, hw, levels), steps)
    t = []
    for _ in range(steps + 3):
        mc.instances.copy_(inst0); mc.batches.copy_(batch0)
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); mc.run(cam.frame, hiz=(pyr, hw, hw, levels)); b.record()
        torch.cuda.synchronize()
        t.append(a.elapsed_time(b))
    occ_ms = float(np.median(t[3:]))
    _, occ_b = mc.download()
    lv = [max(hw >> l, 1) ** 2 for l in range(levels)]
    hiz_bytes = 4 * (hw * hh + sum(lv[:-1]) + sum(lv))  # every level read once and written once, the depth read once
    return {"instances": count, "batches": num_batches, "kept": kept, "moved_records": moved, "gpu_ms": med,
            "with_occlusion": {"hiz_pyramid": f"{hw}x{hw}, {levels} mips from {hw}x{hh} depth", "hiz_build_ms": hiz_ms, "hiz_gbs": hiz_bytes / hiz_ms / 1e6,
                               "cull_compact_ms": occ_ms, "kept": int(occ_b[:, 1].sum())},
            "gpu_minstances_per_s": count / med / 1e3, "algorithmic_bytes": algo, "gpu_hbm_gbs": algo / med / 1e6,
            "gpu_hbm_frac": algo / med / 1e6 / HBM_PEAK_GBS, "cpu_1thread_minstances_per_s": count / t1 / 1e6, "kind": "port"}


def ibl_prefilter_block(ctx, steps: int):
    """SURVEY.md 8f rank 2, the one-off half: EnvironmentNode's cubemap pre-filters at the reference's sizes (EnvironmentNode.h:16-20:
    512 x 512 x 6 environment cube with 10 mips, 32 x 32 x 6 irradiance cube).  Sample counts are the shaders' constants: 1 024 GGX samples per
    env texel (8 texel gathers each), 65 536 hemisphere samples per irradiance texel (4 gathers each).  CPU: the oracle on a bounded
    sample (one mip level / a 2 x 2 x 6 irradiance cube), scaled per sample."""
    from oracle import oracle
    from sailor_amd.forward_plus import compute_irradiance_map, prefilter_env_map
    ibl = synth.make_ibl_set(16, 16, np.zeros((2, 2, 2), np.float32), env_size=512, with_ao=False)
    raw = torch.from_numpy(ibl.env_chain).to(ctx.device)
    env = prefilter_env_map(ctx, raw, 512, ibl.env_levels)
    _, pre_ms, _, _ = side_ms(lambda: prefilter_env_map(ctx, raw, 512, ibl.env_levels), steps)
    _, irr_ms, _, _ = side_ms(lambda: compute_irradiance_map(ctx, env, 512, ibl.env_levels, 32), steps)
    env_samples = sum(6 * max(512 >> l, 1) ** 2 for l in range(1, ibl.env_levels)) * 1024
    irr_samples = 6 * 32 * 32 * 65536
    small = synth.make_ibl_set(16, 16, np.zeros((2, 2, 2), np.float32), env_size=64, with_ao=False)
    t0 = time.perf_counter()
    out = np.zeros_like(small.env_chain)
    oracle.lib().oracle_prefilter_env_level(oracle._p(small.env_chain), 64, small.env_levels, oracle._p(out), 1, C_float(1.0 / (small.env_levels - 1)))
    t_env = time.perf_counter() - t0
    t0 = time.perf_counter()
    oracle.compute_irradiance_map(small.env_chain, 64, small.env_levels, 2)
    t_irr = time.perf_counter() - t0
    # the raw cube in front of them (EnvironmentNode.cpp:116-138): a 2048 x 1024 RGBA32F panorama -> 512 x 512 x 6, then the 2:1 blit chain.
    # Algorithmic bytes: the panorama read once + level 0 written (conversion); level 0 read + 1/3 of it written (mips).
    from sailor_amd.forward_plus import raw_env_cubemap
    gen = torch.Generator(device="cpu").manual_seed(5)
    pano = (torch.rand((1024, 2048, 4), generator=gen, dtype=torch.float32) * 4.0).to(ctx.device)
    raw_env_cubemap(ctx, pano, 512, 10)
    _, raw_ms, _, _ = side_ms(lambda: raw_env_cubemap(ctx, pano, 512, 10), steps)
    chain_floats = sum(6 * max(512 >> l, 1) ** 2 * 4 for l in range(10))
    zero_ms = side_ms(lambda: torch.zeros(chain_floats, dtype=torch.float32, device=ctx.device), steps)[1]  # the wrapper's allocation + clear, not the path's
    raw_bytes = 2048 * 1024 * 16 + 2 * 6 * 512 * 512 * 16 + (6 * 512 * 512 * 16) // 3
    t0 = time.perf_counter()
    oracle.equirect_to_cube(pano.cpu().numpy(), 512)
    t_raw = time.perf_counter() - t0
    raw = {"raw_cube_ms": raw_ms - zero_ms, "raw_cube_gbs": raw_bytes / ((raw_ms - zero_ms) * 1e-3) / 1e9, "raw_cube_bytes": raw_bytes,
           "cpu_1thread_equirect_mtexels_per_s": 6 * 512 * 512 / t_raw / 1e6, "gpu_equirect_plus_mips_mtexels_per_s": 6 * 512 * 512 / ((raw_ms - zero_ms) * 1e-3) / 1e6}
    return {**raw, "env_prefilter_ms": pre_ms, "env_gsamples_per_s": env_samples / pre_ms / 1e6, "irradiance_ms": irr_ms,
            "irradiance_gsamples_per_s": irr_samples / irr_ms / 1e6, "env_samples": env_samples, "irradiance_samples": irr_samples,
            "cpu_1thread_env_msamples_per_s": 6 * 32 * 32 * 1024 / t_env / 1e6, "cpu_1thread_irradiance_msamples_per_s": 6 * 2 * 2 * 65536 / t_irr / 1e6,
            "kind": "port"}


def shadow_pass_block(ctx, count: int, size: int, steps: int, use_coarse: bool = True, front_to_back: bool = True, cascades=(0, 1, 2, 3)):
    """SURVEY.md 8f rank 3, the producer: the shadow passes of one directional light over `count` entities drawn as their bounding boxes (12
    triangles each) -- cascade sets from the sweep's world boxes, the caster draws of the four cascades into size x size depth buffers (compute
    rasteriser), ShadowCaster's fragment stage (cascade 0: EVSM moments, 1-3: R16F) and the EVSM blur of cascade 0.  CPU: the oracle rasteriser
    on a bounded sample of the instances of cascade 3."""
    from oracle import oracle
    from sailor_amd.forward_plus import csm_caster_masks, evsm_blur, raster_depth, shadow_resolve
    cam = synth.make_camera(3840, 2160)
    ents = synth.make_entities(count)
    sweep = EcsSweep(ctx, ents)
    planes, _ = host.extract_frustum_planes(cam.world, cam.aspect, cam.fov, cam.z_near, cam.z_far)
    world, aabb, _ = sweep.run(planes)
    sh = synth.make_shadow_set(cam, 16)
    cplanes = np.stack([host.extract_frustum_planes_matrix(sh.lights_matrices[k])[0] for k in range(4)])
    masks = csm_caster_masks(ctx, aabb, cplanes).cpu().numpy().view(np.uint64)
    pos_h, tris_h = synth.unit_cube_mesh()
    models_h = synth.caster_models(world.cpu().numpy(), ents.local_aabb)
    pos = torch.from_numpy(pos_h).to(ctx.device); tris = torch.from_numpy(tris_h.view(np.int32)).to(ctx.device)
    models = torch.from_numpy(models_h).to(ctx.device)
    ids_h = [np.nonzero(np.unpackbits(masks[k].view(np.uint8), bitorder="little")[:count])[0].astype(np.uint32) for k in range(4)]
    if front_to_back:  # nearest to the light first (reversed Z: largest clip z): the coarse depth is tight after the first few boxes
        for k in range(4):
            lm = np.asarray(sh.lights_matrices[k], np.float64).reshape(4, 4).T
            centre = models_h[ids_h[k]].reshape(-1, 4, 4)[:, 3, :3].astype(np.float64)
            z = centre @ lm[2, :3] + lm[2, 3]
            ids_h[k] = np.ascontiguousarray(ids_h[k][np.argsort(-z, kind="stable")])
    ids = [torch.from_numpy(i.view(np.int32)).to(ctx.device) for i in ids_h]
    depth = [torch.empty((size, size), dtype=torch.float32, device=ctx.device) for _ in range(4)]
    coarse = torch.empty(int(ctx._lib.sailor_hip_raster_coarse_words(size, size)), dtype=torch.int32, device=ctx.device) if use_coarse else None
    out = {"entities": count, "map_size": size, "coarse_depth": use_coarse, "front_to_back": front_to_back, "cascades": []}
    total = 0.0
    for k in cascades:   # (all four in the line; one alone for a profile of its draw: scripts/r06_raster_probe.py)
        def draw(k=k):
            ctx._lib.sailor_hip_raster_depth(ctx.handle, np.ascontiguousarray(sh.lights_matrices[k], np.float32).ctypes.data_as(C_float_p), pos.data_ptr(), tris.data_ptr(), 12,
                                             models.data_ptr(), ids[k].data_ptr(), len(ids_h[k]), size, size, depth[k].data_ptr(), 3, coarse.data_ptr() if coarse is not None else None)  # flags: CLEAR | CULL_BACK (the shadow material, ShadowPrepassNode.cpp:39)
        _, ms, _, _ = side_ms(draw, steps)
        _, rs, _, _ = side_ms(lambda k=k: shadow_resolve(ctx, depth[k], L_RGBA32F if k == 0 else L_R16F), steps)
        cover = float((depth[k] > 0).float().mean().item())
        out["cascades"].append({"instances": int(len(ids_h[k])), "triangles": int(len(ids_h[k])) * 12, "raster_ms": ms, "resolve_ms": rs, "covered": cover,
                                "mtriangles_per_s": len(ids_h[k]) * 12 / ms / 1e3})
        if hasattr(ctx._lib, "sailor_hip_raster_stats"):   # (a -DRASTER_STATS build through SAILOR_HIP_LIB: the counters of ONE draw of this cascade)
            st = (_C.c_ulonglong * 16)()
            ctx.synchronize(); ctx._lib.sailor_hip_raster_stats(st, 1)
            draw(); ctx.synchronize(); ctx._lib.sailor_hip_raster_stats(st, 1)
            out["cascades"][-1]["stats"] = dict(zip(("superblocks_seen", "superblocks_alive", "blocks_seen", "blocks_alive", "texels_inside", "texels_written",
                                                      "blocks_written_whole", "extra", "wave_ticks_sum", "wave_ticks_max", "waves", "waves_over_100us", "waves_over_1ms"),
                                                     (int(v) for v in st)))
        total += ms + rs
    if 0 not in cascades:
        out["all_passes_ms"] = total
        return out
    moments = shadow_resolve(ctx, depth[0], L_RGBA32F)
    tmp = torch.empty_like(moments)
    _, bs, _, _ = side_ms(lambda: evsm_blur(ctx, moments, 2, 5, tmp), steps)
    out["blur_cascade0_ms"] = bs
    out["all_passes_ms"] = total + bs
    # The same four passes with nothing between them but their own dependencies (round 6): the cascades' draws write different targets, so they may run side by
    # side -- one stream and one coarse-depth workspace per cascade (a context per stream), draw -> resolve (-> blur for cascade 0) in order on each, one event
    # pair on the launch stream around the fork and the join.  What that buys is the launches' tails: a draw ends in a few long waves while the chip idles.
    # `all_passes_ms` above stays the sum of the passes timed one by one.
    try:
        from sailor_amd.forward_plus import HipContext
        main = torch.cuda.current_stream(ctx.device)
        streams = [torch.cuda.Stream(device=ctx.device) for _ in cascades]
        ctxs = [HipContext(ctx.device, stream=st) for st in streams]
        coarses = [torch.empty_like(coarse) for _ in cascades] if coarse is not None else [None] * len(cascades)
        outs = [torch.empty((size, size, 4), dtype=torch.float32, device=ctx.device) if k == 0 else torch.empty((size, size), dtype=torch.float16, device=ctx.device) for k in cascades]
        lms = [np.ascontiguousarray(sh.lights_matrices[k], np.float32) for k in cascades]

        def all_at_once():
            fork = torch.cuda.Event(); fork.record(main)
            for i, k in enumerate(cascades):
                c, st = ctxs[i], streams[i]
                st.wait_event(fork)
                c._lib.sailor_hip_raster_depth(c.handle, lms[i].ctypes.data_as(C_float_p), pos.data_ptr(), tris.data_ptr(), 12, models.data_ptr(), ids[k].data_ptr(), len(ids_h[k]),
                                               size, size, depth[k].data_ptr(), 3, coarses[i].data_ptr() if coarses[i] is not None else None)
                c._lib.sailor_hip_shadow_resolve(c.handle, depth[k].data_ptr(), size, size, L_RGBA32F if k == 0 else L_R16F, outs[i].data_ptr())
                if k == 0:
                    c._lib.sailor_hip_evsm_blur(c.handle, outs[i].data_ptr(), tmp.data_ptr(), size, size, 2, 5)
                done = torch.cuda.Event(); done.record(st)
                main.wait_event(done)
        one_by_one = [depth[k].clone() for k in cascades]   # (the buffers the passes above left, drawn one after the other)
        for _ in range(2):
            all_at_once()
        torch.cuda.synchronize(ctx.device)
        out["side_by_side_equals_one_by_one"] = all(bool(torch.equal(depth[k].view(torch.int32), one_by_one[i].view(torch.int32))) for i, k in enumerate(cascades))
        del one_by_one
        evs = []
        for _ in range(steps):
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record(main); all_at_once(); b.record(main)
            evs.append((a, b))
        torch.cuda.synchronize(ctx.device)
        out["all_passes_side_by_side_ms"] = float(np.median([a.elapsed_time(b) for a, b in evs]))
        out["all_passes_side_by_side_how"] = ("the four cascades' draw -> resolve (-> blur) chains on four streams between one fork and one join, median of %d; "
                                              "all_passes_ms = the same passes timed one by one and added up" % steps)
        for c in ctxs:
            c.close()
    except Exception as e:   # (a probe beside the figure of record: it must not take the line down)
        out["all_passes_side_by_side_ms"] = None
        out["all_passes_side_by_side_how"] = f"not measured ({type(e).__name__}: {e})"
    sample = ids_h[3][:: max(1, len(ids_h[3]) // 20000)]
    t0 = time.perf_counter()
    oracle.raster_depth(sh.lights_matrices[3], pos_h, tris_h, models_h, size, size, instance_ids=sample)
    t1 = time.perf_counter() - t0
    out["cpu_1thread_mtriangles_per_s"] = len(sample) * 12 / t1 / 1e6
    out["cpu_sample"] = f"{len(sample)} instances of cascade 3"
    out["kind"] = "port"
    return out


def linearize_block(ctx, frame, fp, d_lights, steps: int):
    """the LinearizeDepth pass in front of K1, standalone (8 algorithmic bytes per pixel) and folded into
    the cull's depth pass (SAILOR_CULL_RAW_DEPTH: no extra pass, no extra bytes), next to the oracle on one host core."""
    from oracle import oracle
    from sailor_amd import _lib as L
    from sailor_amd.forward_plus import linearize_depth
    cam, W, H, N = frame.cam, frame.cam.width, frame.cam.height, len(frame.lights)
    zn = cam.frame.cameraZNearZFar[0]
    raw = synth.make_raw_depth(frame.depth, zn)
    d_raw = torch.from_numpy(raw).to(ctx.device)
    d_lin = torch.empty_like(d_raw)
    for _ in range(3):
        linearize_depth(ctx, cam.frame, d_raw, d_lin)
    lin_ms = side_ms(lambda: linearize_depth(ctx, cam.frame, d_raw, d_lin), steps)
    two = side_ms(lambda: (linearize_depth(ctx, cam.frame, d_raw, d_lin), fp.cull(cam.frame, d_lights, N, d_lin)), steps)
    fused = side_ms(lambda: fp.cull(cam.frame, d_lights, N, d_raw, L.CULL_RAW_DEPTH), steps)
    t0 = time.perf_counter()
    oracle.linearize_depth(zn, raw)
    t_cpu = time.perf_counter() - t0
    b = 8 * W * H
    return {"pixels": W * H, "gpu_ms": lin_ms[1], "gpu_hbm_gbs": b / lin_ms[1] / 1e6, "gpu_hbm_frac": b / lin_ms[1] / 1e6 / HBM_PEAK_GBS,
            "linearize_then_cull_ms": two[1], "cull_on_raw_depth_ms": fused[1],
            "cpu_1thread_mpixels_per_s": W * H / t_cpu / 1e6, "kind": "port"}


def ambient_block(ctx, frame, fp, d_lights, d_surface, steps: int):
    """the shade pass with Standard.shader's ambient / IBL term (irradiance cube 16^2, pre-filtered environment
    cube 64^2 with 7 levels, 64x64 BRDF table computed on the GPU, AO target): extra algorithmic bytes = 4 per pixel (AO) + the
    textures once (0.7 MB)."""
    from sailor_amd.forward_plus import compute_brdf_lut, upload_ibl
    cam, W, H, N = frame.cam, frame.cam.width, frame.cam.height, len(frame.lights)
    lut_ms = side_ms(lambda: compute_brdf_lut(ctx, 256, 256), 5)
    lut = compute_brdf_lut(ctx, 64, 64).cpu().numpy()
    ibl = synth.make_ibl_set(W, H, lut)
    desc, keep = upload_ibl(ibl, ctx.device)
    for _ in range(3):
        fp.shade(cam.frame, d_surface, d_lights, N, None, ibl=desc)
    with_ibl = side_ms(lambda: fp.shade(cam.frame, d_surface, d_lights, N, None, ibl=desc), steps)
    without = side_ms(lambda: fp.shade(cam.frame, d_surface, d_lights, N, None), steps)
    return {"shade_with_ambient_ms": with_ibl[1], "shade_without_ms": without[1], "extra_bytes_per_pixel": 4,
            "brdf_lut_256x256_ms": lut_ms[1], "kernel": "k2_shade_ibl"}


def blur_block(ctx, steps: int):
    """ShadowPrepassNode's EVSM blur of the cascade-0 moments map (4096^2 RGBA32F, radii (2, 5) =
    ShadowCascadeBlur[0]): two passes, 32 algorithmic bytes per texel and pass."""
    from oracle import oracle
    from sailor_amd.forward_plus import evsm_blur
    S = 4096
    m = torch.rand((S, S, 4), dtype=torch.float32, device=ctx.device)
    tmp = torch.empty_like(m)
    for _ in range(2):
        evsm_blur(ctx, m, 2, 5, tmp)
    ms = side_ms(lambda: evsm_blur(ctx, m, 2, 5, tmp), steps)
    b = 2 * 32 * S * S
    sample = np.random.default_rng(0).random((512, 512, 4)).astype(np.float32)
    t0 = time.perf_counter()
    oracle.evsm_blur(sample, 2, 5)
    t_cpu = time.perf_counter() - t0
    return {"texels": S * S, "gpu_ms": ms[1], "gpu_hbm_gbs": b / ms[1] / 1e6, "gpu_hbm_frac": b / ms[1] / 1e6 / HBM_PEAK_GBS,
            "cpu_1thread_mtexels_per_s": 512 * 512 / t_cpu / 1e6, "kind": "port"}


def capture_frame_pipeline(side, side2, unroll, shade_fns, cull_fns, dev=None, pack_fns=None, side3=None):
    """One hipGraph holding `unroll` steps (a multiple of the number S of list sets) of the two-frames-in-flight pipeline: shade_fns[p]() records
    frame k's shade from set p = k % S on `side`, cull_fns[q]() records frame k + 1's cull into set q = (k + 1) % S on `side2`.  The only
    dependencies are the frames' own: shade(k) waits for cull(k), cull(k + 1) for the shade that last read its set, shade(k + 1 - S); one join at
    the end of the graph.  The lists of frame 0 (set 0) must exist before the first replay.
    S = 3 (the default): with two sets cull(k + 1) and shade(k) both start the moment shade(k - 1) ends, and the cross-queue wait sits on the
    critical path -- the kernel trace shows ~10 us between the end of one shade and the start of the next (scripts/analysis/pipeline_timeline.py);
    with a third set the cull waits for a shade that ended a frame ago, and the shades follow each other like launches on one stream.
    pack_fns / side3 (round 4): the culls were recorded with SAILOR_CULL_DEFER_PACK -- the shade reads the per-tile lists, so nothing on the frame's
    path needs lightsGrid / culledLights -- and pack_fns[q]() writes set q's canonical buffers behind the event shade(k + 1) waits for: on the cull's
    own stream (side3 is side2: the form bench.py uses -- in order with everything else that touches the set, no further dependency) or on a third
    stream (cull(k + 2) then does not wait for it either, only the next cull into the SAME set does).  The third-stream form makes hipGraph capture on
    ROCm 7.0 / torch 2.10 fall over (a segmentation fault in capture_end as soon as a forked stream waits for an event of another forked stream:
    scripts/pipeline3_probe.py); eager launches take it."""
    dev = dev or _dev()
    S = len(shade_fns)
    assert len(cull_fns) == S and (pack_fns is None or (len(pack_fns) == S and side3 is not None))

    def body():
        shade_done = [None] * S
        pack_done = [None] * S
        cull_done = None
        for k in range(unroll):
            p, q = k % S, (k + 1) % S
            if cull_done is not None:
                side.wait_event(cull_done)
            shade_fns[p]()
            shade_done[p] = dev.event(False); shade_done[p].record(side)
            if shade_done[q] is not None:
                side2.wait_event(shade_done[q])
            elif k == 0:
                side2.wait_stream(side)  # fork
            if pack_done[q] is not None:
                side2.wait_event(pack_done[q])
            with dev.on_stream(side2):
                cull_fns[q]()
                cull_done = dev.event(False); cull_done.record(side2)
            if pack_fns is not None:
                if side3 is side2:   # behind the event shade(k + 1) waits for, on the cull's own stream: in order with everything that touches the set
                    with dev.on_stream(side2):
                        pack_fns[q]()
                else:
                    side3.wait_event(cull_done)
                    with dev.on_stream(side3):
                        pack_fns[q]()
                        pack_done[q] = dev.event(False); pack_done[q].record(side3)
        side.wait_stream(side2)  # join
        if pack_fns is not None and side3 is not side2:
            side.wait_stream(side3)
    return dev.capture(side, body)


def pipeline_unroll(steps: int, sets: int = 2):
    """(U, r): the K timed steps are K // U replays of a U-step pipeline graph (U a multiple of the number of list sets: a replay ends where it
    began) plus one replay of an r-step graph for the rest, so that the timed region is exactly K steps whatever K is.  U = 0: only the tail graph."""
    cands = [u for u in range(18, 0, -1) if u % sets == 0 and u >= 2]
    u = next((u for u in cands if steps % u == 0), 0)
    if u:
        return u, 0
    u = next((u for u in cands if u <= max(steps - 1, 0) and u <= 12), 0)
    return u, steps - (steps // u) * u if u else steps


def simulate_split(args, dev, ctx, side, frame, d_lights, fp_full, d_depth_full, prep=None, csm=None):
    """Single-GPU estimate of the G-way split: per-band step time (hipGraph replay), equal vs cost-balanced bands."""
    from sailor_amd import dist as sdist
    cam, W, H, N, G = frame.cam, frame.cam.width, frame.cam.height, len(frame.lights), args.simulate_split
    Tx, Ty = host.num_tiles(W, H)
    fp_full.cull(cam.frame, d_lights, N, d_depth_full)
    g, _ = fp_full.lists_to_host()
    row_entries = sdist.row_cost_entries(g[:, 1].astype(np.int64), Tx)
    out = {"config": args.config, "split": G}

    unroll = 0 if args.frames_in_flight == 1 else pipeline_unroll(args.steps, args.list_sets)[0]
    side2 = dev.stream(priority=int(os.environ.get('SAILOR_CULL_PRIORITY', '0')))
    ctx2 = dev.context(side2)

    def time_band(b):
        """ms per step of band b alone on this GPU, launched the way a rank of the split frame launches it (the main path's pipeline graph)"""
        fs = [dev.forward_plus(ctx, W, H, N, b, prep) for _ in range(args.list_sets if unroll else 1)]
        dd = dev.upload(frame.depth[b.fbRowBegin:b.fbRowBegin + b.fbRowCount])
        ds = dev.upload(frame.surface_rows(b.fbRowBegin, b.fbRowBegin + b.fbRowCount))
        for f in fs:
            f.cull(cam.frame, d_lights, N, dd)
            f.shade(cam.frame, ds, d_lights, N, csm)
        dev.synchronize()
        if unroll:
            mode = "inline" if args.pack_inline else args.pack   # as the main path records it
            defer = mode != "inline"
            graph = capture_frame_pipeline(side, side2, unroll, [lambda f=f: f.shade(cam.frame, ds, d_lights, N, csm) for f in fs],
                                           [lambda f=f: f.cull(cam.frame, d_lights, N, dd, ctx=ctx2, defer_pack=defer) for f in fs], dev,
                                           [lambda f=f: f.pack(ctx2) for f in fs] if mode == "deferred" else None, side2)
            fs[0].cull(cam.frame, d_lights, N, dd)
            per = unroll
        else:
            graph = dev.capture(side, lambda: (fs[0].cull(cam.frame, d_lights, N, dd, defer_pack=(args.pack == "never" and not args.pack_inline)), fs[0].shade(cam.frame, ds, d_lights, N, csm)))
            per = 1
        t_spin = time.perf_counter()   # the same clock spin-up as the main path (a band's K steps are over in 2-4 ms)
        while (time.perf_counter() - t_spin) * 1e3 < args.spinup_ms:
            for _ in range(8):
                graph.replay()
            dev.synchronize()
        for _ in range((args.warmup + per - 1) // per):
            graph.replay()
        dev.synchronize()
        reps = []
        for _ in range(5):   # (the median of five timed runs of K steps: a band's K steps are over in 1-2 ms and single runs scatter by +-10 %)
            t0 = time.perf_counter()
            for _ in range(args.steps // per):
                graph.replay()
            dev.synchronize()
            reps.append((time.perf_counter() - t0) / (args.steps // per * per) * 1e3)
        return float(np.median(reps))

    whole = time_band(host.band_whole_frame(W, H))
    for name, bounds in (("equal", [host.band_for_rank(W, H, r, G).tileRowBegin for r in range(G)] + [Ty]),
                         ("balanced", sdist.balanced_tile_rows(row_entries, Tx, G))):
        ms = [time_band(host.band_from_tile_rows(W, H, bounds[r], bounds[r + 1])) for r in range(G)]
        out[name] = {"bounds": [int(b) for b in bounds], "band_ms": ms, "max_ms": max(ms), "predicted_speedup": whole / max(ms)}
    # ... and re-cut on the measured band times, as a renderer re-cuts on the previous frame's (sdist.rebalance_on_measured_times; what main() does for N > 1):
    # `--rebalance` rounds, each from the round before
    prev = out["balanced"]
    for it in range(args.rebalance):
        bounds = sdist.rebalance_on_measured_times(prev["bounds"], prev["band_ms"], row_entries, Tx)
        if bounds == prev["bounds"]:
            break
        ms = [time_band(host.band_from_tile_rows(W, H, bounds[r], bounds[r + 1])) for r in range(G)]
        cur = {"bounds": [int(b) for b in bounds], "band_ms": ms, "max_ms": max(ms), "predicted_speedup": whole / max(ms), "round": it + 1}
        if cur["max_ms"] < out.get("rebalanced", prev)["max_ms"]:
            out["rebalanced"] = cur
        prev = cur
    out["whole_frame_ms"] = whole
    out["launch"] = f"hipGraph replay ({unroll} steps of the frame pipeline per graph), 2 frames in flight over {args.list_sets} list sets" if unroll else "hipGraph replay, 1 frame in flight"
    emit(json.dumps(out))


def ecs_split_block(dev, ctx, dist, rank: int, world: int, count: int, steps: int, simulate=()):
    """K4 across the ranks of a node (contiguous entity ranges, one all-gather of the visibility words) beside K4 replicated (every rank
    sweeps the whole set: no collective, and every rank holds every world matrix).  world > 1: both forms timed on the job's ranks (max over ranks);
    world == 1: the slices of a G-way split timed one after the other on this GPU for G in `simulate` (the all-gather cannot be timed on one GPU).
    The default of the path is the REPLICATED sweep: DESIGN.md 6 has the arithmetic."""
    ents = synth.make_entities(count)
    cam = synth.make_camera(3840, 2160)
    planes, _ = host.extract_frustum_planes(cam.world, cam.aspect, cam.fov, cam.z_near, cam.z_far)
    whole = dev.ecs_sweep(ctx, ents)
    for _ in range(3):
        whole.run(planes)
    rep_ms = event_batch_ms(lambda: whole.run(planes), steps)
    out = {"entities": count, "replicated_ms": rep_ms, "default": "replicated",
           "why": "a split saves at most the sweep's 1-GPU time minus a slice's (~30 us at 1 M entities) and pays an all-gather of the visibility words (128 KB: "
                  "a latency-bound RCCL collective) -- and leaves world matrices / boxes on the rank that computed them, which the draws of every rank need "
                  "(64 + 24 B per entity over xGMI cost more than the whole sweep); built, tested, off by default (DESIGN.md 6)"}
    if world > 1:
        mine = dev.ecs_sweep(ctx, ents, rank, world)
        for _ in range(3):
            mine.run(planes)
            dev.exchange_visibility(mine)
        dev.synchronize()
        slice_ms = event_batch_ms(lambda: mine.run(planes), steps)
        both_ms = event_batch_ms(lambda: (mine.run(planes), dev.exchange_visibility(mine)), steps)
        t = torch.tensor([rep_ms, slice_ms, both_ms], dtype=torch.float64, device=dev.device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        rep_ms, slice_ms, both_ms = (float(v) for v in t.tolist())
        # the gathered bitmask against the replicated sweep's, on every rank
        whole.run(planes); mine.run(planes); dev.exchange_visibility(mine)
        dev.synchronize()
        words = (count + 63) // 64
        same = bool(torch.equal(whole.visibility[:words].cpu(), mine.visibility[:words].cpu()))
        ok = torch.tensor([1 if same else 0], dtype=torch.int32, device=dev.device)
        dist.all_reduce(ok, op=dist.ReduceOp.MIN)
        out.update({"replicated_ms": rep_ms, "split": {"ranks": world, "slice_ms": slice_ms, "slice_plus_allgather_ms": both_ms, "allgather_ms": both_ms - slice_ms,
                                                       "visibility_words_per_rank": mine.words_per_rank, "bitmask_equals_replicated_on_every_rank": bool(int(ok.item()))},
                    "split_vs_replicated": rep_ms / both_ms if both_ms > 0 else None})
    for G in simulate:
        ms = []
        for r in range(G):
            sw = dev.ecs_sweep(ctx, ents, r, G)
            sw.run(planes)
            ms.append(event_batch_ms(lambda sw=sw: sw.run(planes), steps))
            del sw
        out.setdefault("split_simulated_on_one_gpu", {})[str(G)] = {"slice_ms_max": max(ms), "slice_ms_min": min(ms),
                                                                    "allgather_must_cost_less_than_ms": rep_ms - max(ms)}
    return out


def split_config_reading(name: str, args, dev, dist, ctx, side, side2, ctx2, rank: int, world: int, steps: int):
    """A BOUNDED reading of another configuration's split frame inside an N > 1 run (BASELINE.json names C4 and C5 for eight GPUs, the headline is C3):
    cost-balanced tile-row bands, the band kernels in the two-frames-in-flight pipeline graph exactly as the headline launches them, `steps` timed steps
    between barriers, max over ranks -- and the whole frame on every rank's own GPU (one frame in flight) for the speed-up.  C5 with every light
    dirty every frame (its headline mode).  Every rank takes part in every collective whether or not its own set-up succeeded."""
    from sailor_amd import dist as sdist
    from sailor_amd import _lib as slib

    def all_ok(flag_value):
        fl = torch.tensor([flag_value], dtype=torch.int32, device=dev.device)
        dist.all_reduce(fl, op=dist.ReduceOp.MIN)
        return int(fl.item()) == 1

    def max_over_ranks(v):
        t = torch.tensor([v], dtype=torch.float64, device=dev.device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item())

    t_setup = time.perf_counter()
    ok, st = 1, {}
    try:
        frame = BenchFrame(name)
        cam, W, H, N = frame.cam, frame.cam.width, frame.cam.height, len(frame.lights)
        Tx, Ty = host.num_tiles(W, H)
        d_lights = dev.upload_lights(frame.lights)
        prep = dev.prepared_lights(ctx, d_lights, N)
        dyn = name == "C5"
        csm = keep = None
        if frame.cfg.get("shadow_size"):
            csm, keep = dev.upload_shadow_maps(synth.make_shadow_set(cam, frame.cfg["shadow_size"]))
        b0 = host.band_for_rank(W, H, rank, world)
        f0 = dev.forward_plus(ctx, W, H, N, b0, prep)
        f0.cull(cam.frame, d_lights, N, dev.upload(frame.depth[b0.fbRowBegin:b0.fbRowBegin + b0.fbRowCount]))
        dev.synchronize()
        st = dict(frame=frame, cam=cam, W=W, H=H, N=N, Tx=Tx, Ty=Ty, d_lights=d_lights, prep=prep, dyn=dyn, csm=csm, keep=keep, b0=b0, f0=f0)
    except Exception as e:
        ok = 0
        print(f"[bench] rank {rank}: split reading of {name}: calibration failed ({type(e).__name__}: {e})", file=sys.stderr)
    if not all_ok(ok):
        return {"config": name, "error": "calibration failed on some rank"}
    frame, cam, W, H, N, Tx, Ty, d_lights, prep, dyn, csm, b0, f0 = (st[k] for k in ("frame", "cam", "W", "H", "N", "Tx", "Ty", "d_lights", "prep", "dyn", "csm", "b0", "f0"))
    rows_entries = sdist.gather_row_entries(f0.grid[: f0.band_tiles * 2], Tx, b0.tileRowEnd - b0.tileRowBegin, Ty, b0.tileRowBegin)
    bounds = [int(b) for b in sdist.balanced_tile_rows(rows_entries, Tx, world)]
    del f0, st
    ok, run, per, wrun = 1, None, 1, None
    try:
        band = host.band_from_tile_rows(W, H, bounds[rank], bounds[rank + 1])
        dd = dev.upload(frame.depth[band.fbRowBegin:band.fbRowBegin + band.fbRowCount])
        ds = dev.upload(frame.surface_rows(band.fbRowBegin, band.fbRowBegin + band.fbRowCount))
        fs = [dev.forward_plus(ctx, W, H, N, band, prep if i == 0 else dev.prepared_lights(ctx, d_lights, N)) for i in range(args.list_sets)]

        def cull_of(f, c, defer):
            fl = slib.CULL_PREPARE_SELECTED if dyn else slib.CULL_DEFAULT
            f.cull(cam.frame, d_lights, N, dd, fl, ctx=c, defer_pack=defer, prepare_lights=dyn)
        for f in fs:
            cull_of(f, None, False)
            f.shade(cam.frame, ds, d_lights, N, csm)
        dev.synchronize()
        per = pipeline_unroll(steps, args.list_sets)[0] or args.list_sets
        mode = "inline" if (args.pack_inline or not hasattr(fs[0], "pack")) else args.pack
        defer = mode != "inline"
        graph = capture_frame_pipeline(side, side2, per, [lambda f=f: f.shade(cam.frame, ds, d_lights, N, csm) for f in fs],
                                       [lambda f=f: cull_of(f, ctx2, defer) for f in fs], dev, [lambda f=f: f.pack(ctx2) for f in fs] if mode == "deferred" else None, side2)
        cull_of(fs[0], None, False)
        dev.synchronize()
        run = graph.replay
        # the whole frame on this rank's own GPU (one frame in flight), for the speed-up
        wf = dev.forward_plus(ctx, W, H, N, host.band_whole_frame(W, H), prep)
        wd, wsf = dev.upload(frame.depth), dev.upload(frame.surface_rows(0, H))

        def wstep():
            wf.cull(cam.frame, d_lights, N, wd, prepare_lights=dyn, defer_pack=(mode == "never"))
            wf.shade(cam.frame, wsf, d_lights, N, csm)
        wstep(); dev.synchronize()
        try:
            wrun = dev.capture(side, wstep).replay
        except Exception:
            wrun = wstep
            dev.synchronize()
    except Exception as e:
        ok = 0
        print(f"[bench] rank {rank}: split reading of {name}: set-up failed ({type(e).__name__}: {e})", file=sys.stderr)
    if not all_ok(ok):
        return {"config": name, "error": "set-up failed on some rank", "tile_row_bounds": bounds}
    setup_s = time.perf_counter() - t_setup
    reps = max(steps // per, 1)
    out = {"config": name, "tile_row_bounds": bounds, "steps": reps * per, "lights": "dynamic (all dirty every frame, SAILOR_CULL_PREPARE_SELECTED on the bands)" if dyn else "static"}
    for key, fn, n_steps in (("split", run, reps * per), ("whole_frame_per_gpu", wrun, reps * per)):
        calls = reps if key == "split" else n_steps
        for _ in range(max(2, calls // 4)):
            fn()
        if dist is not None:
            dist.barrier()
        dev.synchronize()
        t0 = time.perf_counter()
        for _ in range(calls):
            fn()
        if dist is not None:
            dist.barrier()
        dev.synchronize()
        out[key + "_ms_per_step"] = max_over_ranks(time.perf_counter() - t0) / n_steps * 1e3
    out["value"] = W * H / (out["split_ms_per_step"] * 1e-3) / 1e6
    out["unit"] = "Mpixels/s"
    out["speedup_vs_one_gpu_whole_frame"] = out["whole_frame_per_gpu_ms_per_step"] / out["split_ms_per_step"]
    # the exchange of THIS configuration's band lists, timed like the headline's (exchange_stats), and what a step with it in it would cost
    try:
        fx = fs[0]
        if mode != "inline":
            fx.pack()
        dev.synchronize()
        _, gi, xs = exchange_stats(dev, lambda: dev.exchange(ctx, W, H, bounds, fx), side, dist, dev.device)
        xs["global_sum_num"] = int(gi[0].item())
        out["exchange"] = xs
        out["value_exchange_every_step"] = W * H / ((out["split_ms_per_step"] + xs["ms_median"]) * 1e-3) / 1e6
        out["value_exchange_every_step_how"] = "split_ms_per_step + exchange.ms_median (the exchange behind every step on the same stream: an upper bound on its cost)"
    except Exception as e:
        out["exchange"] = {"error": f"{type(e).__name__}: {e}"}
    out["setup_s"] = setup_s
    return out


class BenchFrame:
    """Synthetic frame with lazily generated surface rows (the full 4K surface is 531 MB; ranks only make their band)."""

    def __init__(self, name):
        cfg = synth.CONFIGS[name]
        self.cfg = cfg
        self.cam = synth.make_camera(cfg["width"], cfg["height"])
        self.depth = synth.make_linear_depth(self.cam.width, self.cam.height)
        self.lights = synth.make_lights(self.cam, self.depth, cfg["lights"])

    def surface_rows(self, r0, r1):
        return synth.make_surface(self.cam, self.depth, row_begin=r0, row_end=r1)


def launch_ranks(n: int, argv):
    """`python bench.py --gpus N` with no launcher around it (WORLD_SIZE unset): this process starts the N ranks itself, as CHILD processes of
    `python -m torch.distributed.run`, and leaves with their exit code.  It has not touched the GPU at this point and never does (importing torch
    initialises nothing; an exec from a process that has is what takes a box down) -- rank 0 of the children prints the JSON line on the shared stdout."""
    import socket
    import subprocess
    with socket.socket() as so:
        so.bind(("127.0.0.1", 0))
        port = so.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1", "--master-port", str(port),
           os.path.abspath(__file__)] + list(argv)
    return subprocess.call(cmd)


def main(argv=None, device_factory=None):
    """device_factory: the device layer (HipDevice; the CPU control-flow test passes its stand-in)"""
    global _DEV
    argv = list(sys.argv[1:] if argv is None else argv)
    args = parse(argv)
    if "WORLD_SIZE" not in os.environ and args.gpus > 1 and device_factory is None:
        raise SystemExit(launch_ranks(args.gpus, argv))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    dev = _DEV = (device_factory or HipDevice)(local_rank)
    device = dev.device
    dist = None
    if world > 1 or args.force_dist:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_PORT", "29533")
        os.environ.setdefault("RANK", "0"); os.environ.setdefault("WORLD_SIZE", "1")
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if dev.dist_backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=device)
        else:
            dist.init_process_group(dev.dist_backend, rank=rank, world_size=world)
        dev.make_comm(rank, world)   # the ncclComm_t of the shipped exchange (exchange.hip), beside torch's own

    frame = BenchFrame(args.config)
    cam, W, H = frame.cam, frame.cam.width, frame.cam.height
    N = len(frame.lights)
    # everything runs on one side stream: the C-ABI records on it, torch events time it, and a hipGraph can capture it
    side = dev.stream(priority=int(os.environ.get('SAILOR_SHADE_PRIORITY', '0')))   # (A / B knob: -1 = the launch stream above the cull's)
    dev.set_stream(side)
    ctx = dev.context(side)
    d_lights = dev.upload_lights(frame.lights)
    # The lights' prepared views (sailor_hip_prepare_lights: the cull's 20-byte records, the shade's staged records) are derived where the HIP backend
    # derives them: behind the copy that writes the `light` SSBO (LightingECS::Tick uploads dirty runs only, ECS/LightingECS.cpp:152-191).  STATIC lights
    # (the default for C2-C4: the synthetic set does not move): once, here, outside the timed region.  DYNAMIC lights (the default for C5, "1 M dynamic
    # lights"; --dynamic-lights elsewhere): every light is dirty every frame, so the preparation of all N runs inside every step, in front of the
    # cull, on the cull's stream.  The line always carries both step times.  --plain-lights: the kernels read the 112-byte records as rounds 1-2 did.
    prep = None if args.plain_lights else dev.prepared_lights(ctx, d_lights, N)
    dynamic = (args.dynamic_lights if args.dynamic_lights is not None else args.config == "C5") and prep is not None
    Tx, Ty = host.num_tiles(W, H)

    def resident(b, own_views=False):
        """ForwardPlus for band b with the band's depth rows resident.  own_views: a further list set of the frame pipeline -- with its own copy of the
        lights' prepared views, so that frame k + 1's re-preparation (dynamic lights, on the cull's stream) never writes what frame k's shade reads"""
        f = dev.forward_plus(ctx, W, H, N, b, dev.prepared_lights(ctx, d_lights, N) if (own_views and prep is not None) else prep)
        dd = dev.upload(frame.depth[b.fbRowBegin:b.fbRowBegin + b.fbRowCount])
        return f, dd

    csm = keep = None
    if frame.cfg.get("shadow_size"):
        shadows = synth.make_shadow_set(cam, frame.cfg["shadow_size"])
        csm, keep = dev.upload_shadow_maps(shadows)

    band = host.band_for_rank(W, H, rank, world)
    bounds = [host.band_for_rank(W, H, r, world).tileRowBegin for r in range(world)] + [Ty]   # tile-row boundaries of the split (equal bands)
    partition = "whole frame"
    rebalance_info = None
    # N > 1, default: ONE frame cut into tile-row bands, one per GPU (BASELINE.json's metric: strong scaling).  --frame-per-gpu: every GPU
    # takes a whole frame per step (per-GPU work fixed: weak scaling, no collective on the data path).
    weak = args.frame_per_gpu
    if world > 1 and weak:
        band = host.band_whole_frame(W, H)
        partition = "one whole frame per GPU and step (frames are independent; no data-path collective)"
    elif world > 1 and not args.equal_bands:
        # Cost-balanced bands (sailor_amd/dist.py:balanced_tile_rows): one calibration cull on equal bands, all ranks learn
        # every tile row's list volume (a renderer would use the previous frame's), and re-split.  Not in the timed region.
        from sailor_amd import dist as sdist
        f0, d0 = resident(band)
        f0.cull(cam.frame, d_lights, N, d0)
        dev.synchronize()
        rows_entries = sdist.gather_row_entries(f0.grid[: f0.band_tiles * 2], Tx, band.tileRowEnd - band.tileRowBegin, Ty, band.tileRowBegin)
        bounds = [int(b) for b in sdist.balanced_tile_rows(rows_entries, Tx, world)]
        band = host.band_from_tile_rows(W, H, bounds[rank], bounds[rank + 1])
        del f0, d0
        # ... then re-cut on MEASURED band times (sdist.rebalance_on_measured_times: a renderer re-cuts on the previous frame's): every rank times its
        # band's cull + shade (eager, one stream), all ranks learn all times, the boundaries move; `--rebalance` rounds.  Not in the timed region.
        # A re-cut is KEPT only if the slowest band it gives measures faster than the best cut so far (the readings are ten eager steps on one
        # stream, not the two-frames-in-flight graph that is timed below -- a noisy reading must not move the headline onto worse bounds; simulate_split
        # applies the same rule).  `rebalance_log` = the slowest band's ms of every cut measured, in order; the first entry is the model's balanced cut.
        rebalanced = 0
        best_bounds, best_max, rebalance_log = list(bounds), None, []
        for it in range(args.rebalance + 1):
            fb, db = resident(band)
            sb = dev.upload(frame.surface_rows(band.fbRowBegin, band.fbRowBegin + band.fbRowCount))

            def band_step():
                fb.cull(cam.frame, d_lights, N, db)
                fb.shade(cam.frame, sb, d_lights, N, csm)
            for _w in range(3):
                band_step()
            dev.synchronize()
            t = torch.zeros(world, dtype=torch.float64, device=device)
            t[rank] = event_batch_ms(band_step, 10)
            dist.all_reduce(t, op=dist.ReduceOp.SUM)
            del fb, db, sb
            worst = float(t.max())
            rebalance_log.append(round(worst, 4))
            if best_max is None or worst < best_max:
                best_bounds, best_max = list(bounds), worst
            if it == args.rebalance:
                break
            new = [int(b) for b in sdist.rebalance_on_measured_times(bounds, t.tolist(), rows_entries, Tx)]
            if new == bounds:
                break
            bounds, rebalanced = new, rebalanced + 1
            band = host.band_from_tile_rows(W, H, bounds[rank], bounds[rank + 1])
        if args.rebalance > 0:
            kept = best_bounds != bounds
            bounds = best_bounds
            band = host.band_from_tile_rows(W, H, bounds[rank], bounds[rank + 1])
            rebalance_info = {"slowest_band_ms_per_cut": rebalance_log, "slowest_band_ms_balanced": rebalance_log[0], "slowest_band_ms_kept": round(best_max, 4),
                              "cuts_measured": len(rebalance_log), "kept": "an earlier cut (the last re-cut measured slower)" if kept else "the last cut measured"}
        partition = f"cost-balanced tile rows {bounds}" + (f" (re-cut {rebalanced}x on measured band times)" if rebalanced else "")
    elif world > 1:
        partition = "equal tile rows"
    if args.simulate_band:
        r_, g_ = (int(v) for v in args.simulate_band.split("/"))
        band = host.band_for_rank(W, H, r_, g_)
        partition = f"diagnostic: band {r_} of an equal {g_}-way tile-row split, alone on this GPU"
    rows = slice(band.fbRowBegin, band.fbRowBegin + band.fbRowCount)
    fp, d_depth = resident(band)
    d_surface = dev.upload(frame.surface_rows(rows.start, rows.stop))

    if args.simulate_split:
        simulate_split(args, dev, ctx, side, frame, d_lights, fp, d_depth, prep, csm)
        return

    def cull_of(f, c, dyn, defer_pack=False):
        """one frame's cull chain into f's list set, recorded on context c's stream (None: the launch stream); with dynamic lights the preparation
        of every light goes in front of it, on the same stream.  defer_pack: stop after the per-tile lists (the shade reads those); f.pack() writes
        lightsGrid / culledLights wherever the caller records it"""
        if dyn and args.separate_prepare:
            f.prepared.prepare(0, N, ctx=c)
        # (a band with every light dirty every frame: the staged shade records only of the lights the band's selection keeps -- SAILOR_CULL_PREPARE_SELECTED; no
        # effect where no selection runs: the whole frame, bands under fewer than 131 072 lights)
        from sailor_amd import _lib as slib
        fl = slib.CULL_PREPARE_SELECTED if (dyn and not args.separate_prepare and (world > 1 or args.simulate_band)) else slib.CULL_DEFAULT
        f.cull(cam.frame, d_lights, N, d_depth, fl, ctx=c, defer_pack=defer_pack, prepare_lights=dyn and not args.separate_prepare)

    # when k1_pack runs (--pack): never (the default) / deferred behind the event the shade waits for / inline in every cull
    pack_mode = "inline" if (args.pack_inline or not hasattr(fp, "pack")) else args.pack
    defer_pack = pack_mode != "inline"

    def cull():
        """one frame's cull as a frame of the timed region runs it, on the launch stream (with --pack deferred the frame's pack follows it at once)"""
        cull_of(fp, None, dynamic, defer_pack)
        if pack_mode == "deferred":
            fp.pack()

    def shade():
        fp.shade(cam.frame, d_surface, d_lights, N, csm)

    def exchange():
        if pack_mode == "never":   # the exchange IS a consumer of the canonical buffers: it asks for them
            fp.pack()
        return dev.exchange(ctx, W, H, bounds, fp)

    def barrier():
        if dist is not None:
            dist.barrier()
        dev.synchronize()

    # One frame = 8 short kernels: at N = 8 a band's kernels last ~50 us in total, less than eight eager launches cost
    # on the host.  Capture the step once and replay it (launch-bound inner loop -> hipGraph).
    # Frames in flight = 2 (the reference keeps MaxFramesInQueue = 2, RHI/Renderer.h:34): a step then shades frame k from
    # one set of list buffers while frame k+1's cull -- a chain of short, latency-bound kernels -- fills the other set on
    # a second stream.  Every step still performs exactly one cull and one shade of a frame.
    # One replay = `unroll` steps of the software pipeline, with the dependencies of the frames themselves and nothing else: shade(k) waits for
    # cull(k), cull(k + 1) for shade(k - 1) (it overwrites the list set that frame read).  One fork / join per STEP -- round 1's form, two graphs of
    # one step each -- makes every step last as long as the longer of its two branches plus the join (0.2335 ms); without it the chain of cull
    # kernels slides under the neighbouring shades (0.219 ms, measured first with eager launches on two streams: scripts/cu_mask_probe.py).
    # The main graph's length is a multiple of the number of list sets (a replay ends where it began); a K that no such length divides gets a
    # second, shorter graph for the rest, so the timed region is exactly K steps (pipeline_unroll).
    want_pipeline = args.frames_in_flight == 2 and not args.no_graph and not args.exchange_every_step
    unroll, tail = pipeline_unroll(args.steps, args.list_sets)
    # (round 4) the shade reads the cull's per-tile lists, so the compaction into the reference's lightsGrid / culledLights (k1_pack) is off the
    # frame's path: recorded on a third stream behind its cull -- every frame still produces both canonical buffers.  --pack-inline: rounds 1-3's form.
    # (a band's shade takes its long tiles from the order hint: written by k1_tile_cull, not by k1_pack, so the band's pack is deferred like the frame's)
    side2 = dev.stream(priority=int(os.environ.get('SAILOR_CULL_PRIORITY', '0')))
    ctx2 = dev.context(side2)
    side3, ctx3 = side2, ctx2   # (the pack's stream: the cull's own -- see capture_frame_pipeline)
    fps = (fp,)
    if want_pipeline:
        fps = (fp,) + tuple(resident(band, True)[0] for _ in range(args.list_sets - 1))   # further sets of grid / culledLights / workspace / prepared views
        for f in fps:               # eager warm-up of every set (also sizes every internal buffer before capture)
            f.cull(cam.frame, d_lights, N, d_depth)
            f.shade(cam.frame, d_surface, d_lights, N, csm)
        dev.synchronize()

    def build_runner(dyn):
        """-> (run, finish, per_run, launch): run() = `per_run` steps (the main pipeline graph, or one step of the other forms), finish() = the
        pipeline's shorter graph for the rest of K, launch = how the steps are launched (for the JSON line)"""
        def step():
            cull_of(fp, None, dyn, defer_pack)
            if pack_mode == "deferred":
                fp.pack()
            shade()
            if world > 1 and args.exchange_every_step:
                exchange()
        if want_pipeline:
            try:
                graphs = [capture_frame_pipeline(side, side2, length, [lambda f=f: f.shade(cam.frame, d_surface, d_lights, N, csm) for f in fps],
                                                 [lambda f=f: cull_of(f, ctx2, dyn, defer_pack) for f in fps], dev,
                                                 [lambda f=f: f.pack(ctx3) for f in fps] if pack_mode == "deferred" else None, side3) if length else None for length in (unroll, tail)]
                cull_of(fps[0], None, dyn, defer_pack)         # prologue: frame 0's lists
                dev.synchronize()
                main_graph = graphs[0] if graphs[0] is not None else graphs[1]
                rest = graphs[1] if graphs[0] is not None else None
                per = unroll if unroll else tail
                how = (f"hipGraph replay ({per} steps of the frame pipeline per graph" + (f", {tail} in the last" if unroll and tail else "") +
                       f"), 2 frames in flight over {args.list_sets} list sets" + (", k1_pack behind the event the shade waits for" if pack_mode == "deferred" else (", no k1_pack (no consumer of the canonical buffers)" if pack_mode == "never" else "")))
                return main_graph.replay, (rest.replay if rest is not None else (lambda: None)), per, how
            except Exception as e:
                print(f"[bench] two-frames-in-flight capture failed ({type(e).__name__}: {e}); falling back to one frame in flight", file=sys.stderr)
                dev.synchronize()
        if not args.no_graph and not args.exchange_every_step:
            try:
                step(); dev.synchronize()
                return dev.capture(side, step).replay, (lambda: None), 1, "hipGraph replay"
            except Exception as e:  # capture unsupported: stay eager, say so in the JSON line
                print(f"[bench] hipGraph capture failed ({type(e).__name__}: {e}); launching eagerly", file=sys.stderr)
                dev.synchronize()
        return step, (lambda: None), 1, "eager"

    def time_steps(run, finish, per):
        """W warm-up steps, then EXACTLY K timed steps between barrier + synchronize on both sides; the MAX over ranks, in seconds"""
        # Clock spin-up (untimed, before the W warm-up steps): K steps of a 0.17 ms frame are over in a few milliseconds, less than the GPU
        # needs to leave its idle power state -- a renderer runs continuously, so the steady state is what the K timed steps should see.
        if args.spinup_ms > 0:
            t_spin = time.perf_counter()
            while (time.perf_counter() - t_spin) * 1e3 < args.spinup_ms:
                for _ in range(max(1, 32 // per)):
                    run()
                dev.synchronize()
        for _ in range((args.warmup + per - 1) // per):   # at least W warm-up steps
            run()
        barrier()
        t0 = time.perf_counter()
        for _ in range(args.steps // per):
            run()
        finish()
        barrier()
        el = time.perf_counter() - t0
        if dist is not None:
            t = torch.tensor([el], dtype=torch.float64, device=device)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            el = float(t.item())
        return el

    run, finish, per_run, launch = build_runner(dynamic)
    elapsed = time_steps(run, finish, per_run)
    ms_per_step = elapsed / args.steps * 1e3
    frames_per_step = world if (weak and world > 1) else 1
    # per-step distribution (median, p10 / p90): a second pass of K steps with one HIP event per step on the launch stream --
    # outside the timed region above, which stays free of event records
    n_runs = max(args.steps // per_run, 1)
    evs = [dev.event() for _ in range(n_runs + 1)]
    evs[0].record(side)
    for i in range(n_runs):
        run()
        evs[i + 1].record(side)
    dev.synchronize()
    per_step = np.array([evs[i].elapsed_time(evs[i + 1]) for i in range(n_runs)]) / per_run
    step_stats = {"median": float(np.median(per_step)), "p10": float(np.percentile(per_step, 10)), "p90": float(np.percentile(per_step, 90)),
                  "how": ("HIP events between consecutive steps on the launch stream" if per_run == 1 else
                          f"HIP events between consecutive replays of the {per_run}-step pipeline graph on the launch stream, divided by {per_run}") +
                         ", a separate pass of K steps (this rank)"}
    value = frames_per_step * W * H * args.steps / elapsed / 1e6

    # ---- the other light mode, same launch form, same K steps: static lights beside a dynamic headline and the other way round ----
    other_mode = None
    if prep is not None and not args.single_mode:
        run2, finish2, per2, _ = build_runner(not dynamic)
        el2 = time_steps(run2, finish2, per2)
        other_mode = {"ms_per_step": el2 / args.steps * 1e3, "value": frames_per_step * W * H * args.steps / el2 / 1e6}
        prep.prepare(0, N)   # (the views as the per-kernel measurements below expect them)
        dev.synchronize()

    # ---- per-kernel timing on the launch stream (HIP events) + algorithmic bytes ----
    cull_ms = event_ms(cull, args.steps)
    shade_ms = event_ms(shade, args.steps)
    batch_stream = None if args.no_graph else side
    cull_batch = event_batch_stats(cull, args.steps, batch_stream)
    shade_batch = event_batch_stats(shade, args.steps, batch_stream)
    cull_batch_ms, shade_batch_ms = cull_batch["median"], shade_batch["median"]
    prepare_ms = event_batch_ms(lambda: prep.prepare(0, N), args.steps, batch_stream) if prep is not None else None
    # SURVEY.md 8d's own definition of the step: t(K1 + K2) by events on ONE stream, one frame in flight -- cull, then shade, then the next frame's cull
    serial = event_batch_stats(lambda: (cull(), shade()), args.steps, batch_stream)
    serial_max = serial["median"]
    if dist is not None:   # (N > 1: the frame is done when its slowest band is)
        t = torch.tensor([serial_max], dtype=torch.float64, device=device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        serial_max = float(t.item())
    # ... and the same frame with k1_pack beside the shade (the form the frame pipeline above launches; still ONE frame in flight: the next
    # frame's cull waits for this frame's shade AND pack)
    serial_deferred = None
    if pack_mode == "deferred" and not args.single_mode:
        def frame_deferred():
            cull_of(fp, None, dynamic, True)
            side2.wait_stream(side)
            with dev.on_stream(side2):
                fp.pack(ctx2)
            shade()
            side.wait_stream(side2)
        serial_deferred = event_batch_stats(frame_deferred, args.steps, batch_stream)
    pipeline_ms = event_batch_ms(lambda: (cull(), shade()), args.steps)   # eager cull + shade chains back to back: the step without graphs or overlap
    cull_eager_ms = event_batch_ms(cull, args.steps)
    # THE ROOFLINE'S DURATION: the dominant kernel's own dispatch-packet timestamps (an event pair riding on every launch of it), each launch between
    # its real neighbours (the cull chain in front, the next frame's behind) -- a direct reading of the kernel, the figure rocprofv3's kernel trace
    # reports, no difference of two measurements (VERDICT r03 / ADVICE r03)
    direct = kernel_in_frame_ms(ctx, cull, shade, max(args.steps, BATCH_LAUNCHES))
    # which kernels one cull consists of (a selection in front of a band's chain or not, the wide list builder, brute force ...) is the LIBRARY's
    # decision: read from its launch log (sailor_hip_context_launch_log), not re-derived here -- one timing slot per kernel it names
    chain_names = ctx.launches_of(cull)
    chain_ms = chain_kernels_ms(ctx, cull, shade, max(args.steps, BATCH_LAUNCHES), len(chain_names))
    # k1_pack on its own (--pack never: no frame runs it; this is what a consumer of lightsGrid / culledLights pays when it asks for them)
    pack_ms = None
    if pack_mode == "never":
        pack_ms = kernel_in_frame_ms(ctx, lambda: (cull(), shade()), fp.pack, max(args.steps, BATCH_LAUNCHES))["median"]
    # THIS box's yardstick, in this process: a float4 streaming copy of 512 MB timed like the kernels above (boxes of the pool differ by +-5 %: a
    # slower line on a slower box shows as the same frac_of_box_copy)
    box = None
    if hasattr(dev, "copy_probe_ms"):
        nbytes = 512 << 20
        copy_ms = dev.copy_probe_ms(ctx, nbytes)
        box = {"copy_gbs": 2 * nbytes / (copy_ms * 1e-3) / 1e9, "copy_ms": copy_ms, "copy_bytes": nbytes,
               "how": "sailor_hip_copy_probe: float4 per lane, 512 MB read + 512 MB written, median of 20 launches by their own dispatch-packet timestamps, this process, this rank",
               "guide_copy_gbs": HBM_COPY_GBS}
    if pack_mode == "never":
        fp.pack()   # (the read-back below wants the canonical buffers)
    g, idx = fp.lists_to_host()
    sum_nt = int(idx[0])
    distinct = int(len(np.unique(idx[1:]))) if sum_nt else 0
    band_pixels = band.fbRowCount * W
    b_shade = 64 * band_pixels + 8 * fp.band_tiles + 4 * sum_nt + 112 * distinct
    csm_info = None
    if csm is not None:   # K3 add-on: 256 B of matrices + b_c bytes per DISTINCT texel fetched in cascade c (b_0 = 16 RGBA32F, b_1..3 = 2 R16F)
        surf_host = frame.surface_rows(rows.start, rows.stop)
        d_c = csm_distinct_texels(frame, shadows, surf_host[0, :, :, :3], surf_host[1, :, :, :3])
        b_csm = 256 + 16 * d_c[0] + 2 * (d_c[1] + d_c[2] + d_c[3])
        b_shade += b_csm
        csm_info = {"distinct_texels_per_cascade": d_c, "bytes": b_csm}
        del surf_host
    b_cull = 20 * N + 4 * band_pixels + 8 * fp.band_tiles + 4 * sum_nt + 4 + (112 * N if dynamic else 0)   # (dynamic: + the records the preparation reads)
    evals = int((g[:, 1].astype(np.int64) * 256).sum())
    shade_launch_ms = direct["mean"]
    shade_gbs = b_shade / (shade_launch_ms * 1e-3) / 1e9
    # the same kernel as the difference (one-frame-in-flight step) - (cull chain alone), medians of event-bracketed graph batches: round 3's figure,
    # kept beside the direct one (it reads a few per cent lower: the next cull's head overlaps the shade's drain)
    shade_diff_ms = serial["median"] - cull_batch_ms
    # the shade's kernel as the library's launch log names it (k2_shade[_band][_csm][_p|_t|_pt]: the band form or not, shadow maps, prepared lights, the
    # lists from the cull's per-tile slots -- the library's decisions); the stand-in of the CPU control-flow test has no log of its own
    logged = [n for n in ctx.launches_of(shade) if n.startswith("k2_shade")]
    if logged:
        shade_kernel = logged[-1]
    else:
        shade_kernel = ("k2_shade_band" if fp.tile_order and fp.use_tile_order else "k2_shade") + ("_csm" if csm is not None else "")
        tl = bool(getattr(fp, "shade_from_tile_lists", False))
        if prep is not None or tl:
            shade_kernel += "_" + ("p" if prep is not None else "") + ("t" if tl else "")
    trace_ms = trace_kernel_ms(shade_kernel, args.config, world)
    roofline = {"bound": "hbm", "kernel": shade_kernel, "achieved": shade_gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": shade_gbs / HBM_PEAK_GBS,
                "traffic": measured_traffic(shade_kernel, args.config, world), "bytes_per_launch": b_shade, "avg_launch_ms": shade_launch_ms,
                "timing": "the kernel's own launches, measured directly: a HIP event pair on the dispatch packet of EVERY launch (hipExtLaunchKernel start / stop events = the "
                          "command processor's timestamps of that kernel, what rocprofv3 --kernel-trace reports), each launch between its real neighbours on the launch stream "
                          "(this frame's cull chain in front of it, the next frame's behind it; eager launches, one frame in flight), mean of %d launches -- compare "
                          "rocprof_kernel_avg_ms, the kernel's duration in rocprofv3 --kernel-trace --stats of the same frame (profiles/<round>/kernel_stats.csv).  "
                          "in_frame_by_difference_ms = median(serial step) - median(cull chain alone), event-bracketed hipGraph batches (round 3's figure); "
                          "back_to_back_launch_ms = the same kernel fifty times in a row (each launch runs into its predecessor's write-back of the radiance); "
                          "isolated_* = an event pair around every single launch with the GPU drained on both sides" % direct["launches"],
                "avg_launch_ms_stats": direct,
                "in_frame_by_difference_ms": shade_diff_ms, "frac_in_frame_by_difference": b_shade / (shade_diff_ms * 1e-3) / 1e9 / HBM_PEAK_GBS if shade_diff_ms > 0 else None,
                "back_to_back_launch_ms": shade_batch_ms, "back_to_back_launch_ms_min_max": [shade_batch["min"], shade_batch["max"]],
                "frac_back_to_back": b_shade / (shade_batch_ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                "rocprof_kernel_avg_ms": trace_ms,
                "launch_gap_ms": (shade_batch_ms - trace_ms) if trace_ms is not None else None,  # back-to-back launches against the committed trace's kernel duration
                "isolated_avg_launch_ms": shade_ms[0], "isolated_median_launch_ms": shade_ms[1],
                "in_pipeline_launch_ms": pipeline_ms - cull_eager_ms,  # eager (cull + shade) x K minus eager (cull) x K
                "eager_step_ms": pipeline_ms,
                "frac_of_measured_copy_peak": shade_gbs / HBM_COPY_GBS,   # (the guide's 6.29 TB/s constant; frac_of_box_copy divides by THIS box's own copy)
                "frac_of_box_copy": (shade_gbs / box["copy_gbs"]) if box else None,
                "valu_sidebar": {"pixel_light_evals": evals, "gevals_per_s": evals / (shade_launch_ms * 1e-3) / 1e9,
                                 "note": "~110 fp32 ops per (pixel,light): VALU-bound once mean list length exceeds ~10 (SURVEY.md 7, hard part 2)"},
                "csm": csm_info,
                "cull": {"kernels": (("k_prepare_lights+" if args.separate_prepare else "(lights prepared inside) ") if dynamic else "") + "k01_prepare+k1_*", "kernels_ms": dict(zip(chain_names, chain_ms)), "kernels_sum_ms": float(sum(chain_ms)),
                         "kernels_how": "median over the frames of the direct reading above, per kernel of the chain (each kernel's own dispatch-packet timestamps)",
                         "avg_ms": cull_batch_ms, "isolated_avg_ms": cull_ms[0], "isolated_median_ms": cull_ms[1],
                         "bytes": b_cull, "achieved_gbs": b_cull / (cull_batch_ms * 1e-3) / 1e9, "frac": b_cull / (cull_batch_ms * 1e-3) / 1e9 / HBM_PEAK_GBS},
                "whole_path": {"what": "cull chain + shade, SURVEY.md 8d's bytes of both over the serial step (one frame in flight) and over `ms_per_step`",
                               "bytes": b_cull + b_shade, "frac_serial": (b_cull + b_shade) / (serial["median"] * 1e-3) / 1e9 / HBM_PEAK_GBS,
                               "frac_pipelined": (b_cull + b_shade) / (ms_per_step * 1e-3) / 1e9 / HBM_PEAK_GBS if world == 1 else None}}

    # ---- supplementary, N > 1 only: alternate-frame rendering.  The split frame above is latency-bound (a 0.17 ms frame leaves ~20 us
    # per GPU); the reference keeps two frames in flight (RHI/Renderer.h:34), and whole frames are independent, so a node can also give
    # every GPU its own frame.  Same step (one cull + one shade of the whole frame) on every rank, K steps, max over ranks; reported next
    # to `value`, never instead of it.  Every rank takes part in the collectives below whether or not its own set-up succeeded.
    afr = None
    if dist is not None and not args.no_afr and not weak:
        ok = 1
        wrun = None
        try:
            wf = dev.forward_plus(ctx, W, H, N, host.band_whole_frame(W, H), prep)
            wd = dev.upload(frame.depth)
            ws = dev.upload(frame.surface_rows(0, H))

            def wstep():
                wf.cull(cam.frame, d_lights, N, wd, defer_pack=(pack_mode == "never"))
                wf.shade(cam.frame, ws, d_lights, N, csm)
            wstep(); dev.synchronize()
            wrun = wstep
            if not args.no_graph:
                try:
                    wrun = dev.capture(side, wstep).replay
                except Exception:
                    wrun = wstep
                    dev.synchronize()
        except Exception as e:
            ok = 0
            print(f"[bench] rank {rank}: alternate-frame set-up failed ({type(e).__name__}: {e})", file=sys.stderr)
        flag = torch.tensor([ok], dtype=torch.int32, device=device)
        dist.all_reduce(flag, op=dist.ReduceOp.MIN)
        if int(flag.item()) == 1:
            for _ in range(args.warmup):
                wrun()
            barrier()
            t0 = time.perf_counter()
            for _ in range(args.steps):
                wrun()
            barrier()
            t = torch.tensor([time.perf_counter() - t0], dtype=torch.float64, device=device)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            afr_elapsed = float(t.item())
            afr = {"what": "every GPU renders whole frames of its own (frames are independent; one frame in flight per GPU)", "scaling": "weak",
                   "value": world * W * H * args.steps / afr_elapsed / 1e6, "unit": "Mpixels/s", "ms_per_frame_per_gpu": afr_elapsed / args.steps * 1e3}
        else:
            afr = {"error": "set-up failed on some rank"}

    # ---- supplementary in the --frame-per-gpu (weak) mode: ONE frame split into cost-balanced tile-row bands, one band per GPU -- the strong-scaling
    # reading, with the exchange that rebuilds the reference's global lists.  Set-up failures are agreed on by all ranks before any further collective.
    split = None
    if dist is not None and not args.no_afr and weak:
        from sailor_amd import dist as sdist

        def all_ok(flag_value):
            fl = torch.tensor([flag_value], dtype=torch.int32, device=device)
            dist.all_reduce(fl, op=dist.ReduceOp.MIN)
            return int(fl.item()) == 1

        ok, f0 = 1, None
        try:
            b0 = host.band_for_rank(W, H, rank, world)
            f0, d0 = resident(b0)
            f0.cull(cam.frame, d_lights, N, d0)
            dev.synchronize()
        except Exception as e:
            ok = 0
            print(f"[bench] rank {rank}: split-frame calibration failed ({type(e).__name__}: {e})", file=sys.stderr)
        if not all_ok(ok):
            split = {"error": "calibration failed on some rank"}
        else:
            rows_entries = sdist.gather_row_entries(f0.grid[: f0.band_tiles * 2], Tx, b0.tileRowEnd - b0.tileRowBegin, Ty, b0.tileRowBegin)
            sbounds = [int(b) for b in sdist.balanced_tile_rows(rows_entries, Tx, world)]
            del f0, d0
            ok, brun = 1, None
            try:
                bb = host.band_from_tile_rows(W, H, sbounds[rank], sbounds[rank + 1])
                bf, bd = resident(bb)
                bs = dev.upload(frame.surface_rows(bb.fbRowBegin, bb.fbRowBegin + bb.fbRowCount))

                def bstep():
                    bf.cull(cam.frame, d_lights, N, bd, defer_pack=(pack_mode == "never"))
                    bf.shade(cam.frame, bs, d_lights, N, csm)
                bstep(); dev.synchronize()
                brun = bstep
                if not args.no_graph:
                    try:
                        brun = dev.capture(side, bstep).replay
                    except Exception:
                        brun = bstep
                        dev.synchronize()
            except Exception as e:
                ok = 0
                print(f"[bench] rank {rank}: split-frame set-up failed ({type(e).__name__}: {e})", file=sys.stderr)
            if not all_ok(ok):
                split = {"error": "set-up failed on some rank"}
            else:
                for _ in range(max(args.warmup, 20)):
                    brun()
                barrier()
                t0 = time.perf_counter()
                for _ in range(args.steps):
                    brun()
                barrier()
                t = torch.tensor([time.perf_counter() - t0], dtype=torch.float64, device=device)
                dist.all_reduce(t, op=dist.ReduceOp.MAX)
                split_elapsed = float(t.item())
                if pack_mode == "never":
                    bf.pack()
                gg, gi, sx_stats = exchange_stats(dev, lambda: dev.exchange(ctx, W, H, sbounds, bf), side, dist, device)
                tot = int(gi[0].item())
                split = {"what": "ONE frame split into cost-balanced tile-row bands, one band per GPU; cull and shade need no collective, an RCCL all-gather "
                                 "rebuilds the reference's global lists for consumers that want them (outside the timed steps)",
                         "scaling": "strong", "value": W * H * args.steps / split_elapsed / 1e6, "unit": "Mpixels/s",
                         "ms_per_step": split_elapsed / args.steps * 1e3, "speedup_vs_one_gpu_whole_frame": ms_per_step / (split_elapsed / args.steps * 1e3),
                         "tile_row_bounds": sbounds,
                         "exchange": {"global_sum_num": tot, "checksum": int(gi[1:1 + tot].to(torch.int64).sum().item()), "tiles": int(gg.numel() // 2), "how": dev.exchange_how}}
                split["exchange"].update(sx_stats)

    exchange_info = None
    exchange_step = None
    if dist is not None and not weak:
        gg, gi, ex_stats = exchange_stats(dev, exchange, side, dist, device)
        tot = int(gi[0].item())
        exchange_info = {"global_sum_num": tot, "checksum": int(gi[1:1 + tot].to(torch.int64).sum().item()), "tiles": int(gg.numel() // 2), "how": dev.exchange_how,
                         "tile_row_bounds": bounds}
        exchange_info.update(ex_stats)
        # ... and the step WITH the exchange in it (a consumer that wants the global lists of every frame): one frame in flight -- cull chain with k1_pack,
        # shade, exchange -- against the same form without the exchange; the kernels of the step replayed from a hipGraph, the exchange's calls recorded
        # behind every replay (eagerly: whether three RCCL collectives may sit in a hipGraph on a node is not something this box can try at N > 1)
        if world > 1 and not args.exchange_every_step and not args.single_mode:
            def step_kernels():
                cull_of(fp, None, dynamic, False)
                shade()
            run_k = step_kernels
            if not args.no_graph:
                try:
                    step_kernels(); dev.synchronize()
                    run_k = dev.capture(side, step_kernels).replay
                except Exception:
                    dev.synchronize()
            def timed_steps(fn):   # (a fixed number of calls on every rank -- the exchange is collective; no time-based spin-up here)
                for _ in range(max(args.warmup, 2)):
                    fn()
                barrier()
                t0 = time.perf_counter()
                for _ in range(args.steps):
                    fn()
                barrier()
                t = torch.tensor([time.perf_counter() - t0], dtype=torch.float64, device=device)
                dist.all_reduce(t, op=dist.ReduceOp.MAX)
                return float(t.item())
            el_x = timed_steps(lambda: (run_k(), exchange()))
            el_n = timed_steps(run_k)
            exchange_step = {"value_exchange_every_step": W * H * args.steps / el_x / 1e6, "ms_per_step_exchange_every_step": el_x / args.steps * 1e3,
                             "ms_per_step_same_form_without_exchange": el_n / args.steps * 1e3,
                             "form": "one frame in flight: cull chain (k1_pack inline) + shade" + (" replayed from a hipGraph" if run_k is not step_kernels else " launched eagerly") +
                                     ", the exchange's calls recorded behind every step; K steps between barriers, max over ranks"}

    # ---- N > 1: the other configurations BASELINE.json names for a node (C4, C5), bounded, and K4 across the ranks beside K4 replicated
    split_configs = None
    ecs_split = None
    if dist is not None and world > 1:
        names = args.split_configs if args.split_configs is not None else ("C4,C5" if args.config == "C3" else "")
        for nm in [n for n in names.split(",") if n]:
            split_configs = split_configs or {}
            split_configs[nm] = split_config_reading(nm, args, dev, dist, ctx, side, side2, ctx2, rank, world, args.split_config_steps)
        ecs_split = ecs_split_block(dev, ctx, dist, rank, world, (1 << 20) if args.config != "tiny" else 5000, 10)

    line = None
    if rank == 0:
        mode = (("dynamic: every light dirty every frame -- the prepared views of all %d lights re-derived inside every step, " % N) +
                ("by sailor_hip_prepare_lights in front of the cull" if args.separate_prepare else "folded into the cull's per-light pass (SAILOR_CULL_PREPARE_LIGHTS)")) if dynamic else \
               ("static: the lights' prepared views derived ONCE, outside the timed region (LightingECS::Tick uploads dirty runs only; the synthetic set does not move)" if prep is not None
                else "plain: no prepared views, cull and shade read the 112-byte records")
        out = {
            "metric": "lit Mpixels/s (K0+K1 tile light cull + K2 PBR shade over per-tile lists)", "value": value, "unit": "Mpixels/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "spinup_ms": args.spinup_ms, "ms_per_step": ms_per_step, "higher_is_better": True,
            "scaling": "weak" if weak else "strong", "launch": launch, "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"{args.config}: {W}x{H}, {N} point+spot lights" + (" (all dirty every frame)" if dynamic else "") + f", 16x16 tiles ({fp.Tx}x{fp.Ty}), cull + PBR shade"
                                   + (" + 4-cascade CSM" if csm is not None else ""),
                       "width": W, "height": H, "lights": N, "parallelism": (f"dp{world}: a whole frame per GPU" if weak else f"tile-row bands x{world}"), "partition": partition, "rebalance": rebalance_info,
                       "mean_list_length": sum_nt / max(fp.band_tiles, 1), "sum_num_rank0_band": sum_nt, "distinct_lights_rank0_band": distinct,
                       "generator": {"seed": synth.SEED, "radius_scale": frame.cfg["lights"].radius_scale}},
            "lights": {"mode": mode, "prepare_lights_ms": prepare_ms,
                       "prepare_lights_how": "sailor_hip_prepare_lights(0, N) back to back, median of event-bracketed batches (what one frame's re-upload of every light adds)"},
            "step_ms": step_stats,
            "value_serial": frames_per_step * W * H / (serial_max * 1e-3) / 1e6 if not (weak and world > 1) else None,
            "serial_step_ms": {"median": serial["median"], "min": serial["min"], "max": serial["max"], "max_over_ranks": serial_max,
                               "how": "SURVEY.md 8d: t(K0+K1+K2) of ONE frame in flight -- cull then shade on one stream, %d batches of %d frames, one HIP event pair per batch "
                                      "(median / min / max: this rank's band when N > 1; value_serial divides by the slowest rank's median); `value` above keeps two frames in "
                                      "flight as the reference does (RHI/Renderer.h:34)" % (serial["batches"], serial["launches_per_batch"])},
            "value_serial_pack_beside_shade": (frames_per_step * W * H / (serial_deferred["median"] * 1e-3) / 1e6) if (serial_deferred and not (weak and world > 1)) else None,
            "serial_step_pack_beside_shade_ms": serial_deferred,
            "mlights_culled_per_s": N / (cull_batch_ms * 1e-3) / 1e6,
            "cull_ms": cull_batch_ms, "shade_ms": shade_launch_ms, "shade_back_to_back_ms": shade_batch_ms,
            "pack": {"mode": pack_mode, "pack_ms": pack_ms,
                     "what": "k1_pack = the per-tile lists compacted into the reference's lightsGrid / culledLights (bit for bit); never: only when a consumer asks "
                             "(sailor_hip_light_cull_pack), its kernel time reported here; deferred / inline: inside every step"},
            "roofline": roofline,
            "box": box,
        }
        if other_mode is not None:
            tag = "static_lights" if dynamic else "dynamic_lights"
            out["value_" + tag] = other_mode["value"]
            out["ms_per_step_" + ("static" if dynamic else "dynamic")] = other_mode["ms_per_step"]
        if exchange_info:
            out["exchange"] = exchange_info
        if exchange_step:
            out.update(exchange_step)
        if afr:
            out["alternate_frame_rendering"] = afr
            if "ms_per_frame_per_gpu" in afr:
                out["speedup_vs_one_gpu_whole_frame"] = afr["ms_per_frame_per_gpu"] / ms_per_step
        if split:
            out["split_frame"] = split
        if split_configs:
            out["split_configs"] = split_configs
        if ecs_split:
            out["ecs_sweep"] = {"split": ecs_split, "note": "K4 replicated on every rank (the default); `split` = the entity ranges + visibility all-gather form, timed beside it"}
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(frame, args.cpu_sample_tile_rows)
            out["ecs_sweep"] = ecs_baseline(ctx, 1 << 20, 20)
            out["ecs_sweep"]["split"] = ecs_split_block(dev, ctx, None, 0, 1, 1 << 20, 10, simulate=(2, 4, 8))
            out["mesh_cull_compact"] = mesh_cull_block(ctx, 1 << 20, 4096, 20)
            out["linearize_depth"] = linearize_block(ctx, frame, fp, d_lights, 30)
            if csm is None:
                out["ambient_ibl"] = ambient_block(ctx, frame, fp, d_lights, d_surface, 30)
            out["evsm_blur"] = blur_block(ctx, 10)
            out["ibl_prefilter"] = ibl_prefilter_block(ctx, 3)
            out["shadow_passes"] = shadow_pass_block(ctx, 1 << 20, 4096, 5)
        line = json.dumps(out)
    if dist is not None:
        dist.barrier()
        dev.close()
        dist.destroy_process_group()
    if rank == 0:
        emit(line)
    return line


_JSON_FD = None   # set when this process is run as a rank: the descriptor stdout had at start-up


def emit(line: str):
    """The JSON line, alone on stdout.  Run as a program, a rank moves descriptor 1 onto stderr before anything else runs (`claim_stdout`), so what libraries
    print through C stdio -- RCCL's version banner, Gloo's connection notes, on every rank -- lands on stderr, and rank 0 writes its line to the real one."""
    import ctypes
    sys.stdout.flush()
    ctypes.CDLL(None).fflush(None)
    if _JSON_FD is None:
        print(line, flush=True)
    else:
        os.write(_JSON_FD, (line + "\n").encode())


def claim_stdout(argv):
    """(not in the parent of `--gpus N` without a launcher: its children are the ranks and inherit the real stdout)"""
    global _JSON_FD
    gpus = 1
    for i, a in enumerate(argv):
        if a == "--gpus" and i + 1 < len(argv):
            gpus = int(argv[i + 1])
        elif a.startswith("--gpus="):
            gpus = int(a.split("=", 1)[1])
    if "WORLD_SIZE" in os.environ or gpus <= 1:
        sys.stdout.flush()
        _JSON_FD = os.dup(1)
        os.dup2(2, 1)


if __name__ == "__main__":
    claim_stdout(sys.argv[1:])
    main()
