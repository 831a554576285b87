"""Split-frame exchange: tile-row bands, one process per GPU, torch.distributed (backend "nccl" = RCCL
over xGMI on the GPU box, "gloo" in the CPU tests).

Cull and shade need NO communication on a band partition: every rank culls and shades its own tile rows.  Only a consumer
that wants the reference's *global* compact buffers (`lightsGrid` + `culledLights` in canonical tile order) triggers the
two collectives below:
  1. all-gather of one uint32 per rank (the band total sum(num)) -> every rank prefix-sums the global base offsets
     (Appendix A step 6: offset(tile) = 1 + sum of num over earlier tiles);
  2. all-gather of the band index segments, padded to the largest band total, placed at their canonical offsets.
Message sizes at 4K / 65 536 lights: 4 B and <= 2.1 MB per rank -- latency-bound on xGMI, so they are issued once per
frame, never per tile.
"""
from __future__ import annotations

import torch
import torch.distributed as dist

from . import host


def bands(width: int, height: int, world_size: int):
    return [host.band_for_rank(width, height, r, world_size) for r in range(world_size)]


# Relative cost of one tile (streaming its 256 pixels + fixed per-tile work) and of one light-list entry.  Round 4 re-fit on the bands of an 8-way split
# of the 4K / 65 536-light frame (bench.py --simulate-split 8): with equal bands the step of a band is ~25 us of launch floors plus (a) 21.7 us for an
# edge band (4 080 tiles, ~10 entries a tile) and (b) 32 us for a middle band (~35 entries a tile) -- a ratio of 1.47, which fixes tile : entry at 43 : 1
# (round 1's 4.1 : 0.28 = 15 : 1 came from the whole frame's kernels and gave the edge bands 24 rows of 135: they became the slowest).
TILE_COST, ENTRY_COST = 12.0, 0.28
# A tile in a light cluster (>= LONG_TILE lights) costs its band more than its entries: its cull goes through a block of its own and its shade through the
# split blocks.  Charged as LONG_TILE_ENTRIES extra entries per such tile (round 2: 1000, when a cluster row block took 24 us of a band's cull; since round
# 4's block-per-tile cull and block-wide selection the cluster band's cull is 9 against 6 us).
LONG_TILE, LONG_TILE_ENTRIES = 96, 300


def row_cost_entries(num_per_tile, tiles_per_row: int):
    """Per tile row: the sum of the list lengths plus the long-tile charge (what balanced_tile_rows takes as row_entries).  num_per_tile: a
    torch or numpy integer array of rows * tiles_per_row list lengths."""
    n = num_per_tile.reshape(-1, tiles_per_row)
    return n.sum(1) + LONG_TILE_ENTRIES * (n >= LONG_TILE).sum(1)


def balanced_tile_rows(row_entries, tiles_per_row: int, world_size: int):
    """Cost-balanced contiguous partition of the tile rows.  row_entries[r] = sum of list lengths of tile row r (from the
    previous frame -- light lists are temporally coherent -- or from a calibration cull).  Returns world_size + 1 boundaries.
    Equal-row bands leave the middle of a perspective frame 1.7x heavier than its top (measured: 166 us vs 95 us per band
    of 8), which caps 8-GPU scaling at 2.4x; balancing by last frame's cost is the standard split-frame remedy."""
    import numpy as np
    cost = tiles_per_row * TILE_COST + np.asarray(row_entries, dtype=np.float64) * ENTRY_COST
    n = len(cost)
    cum = np.concatenate([[0.0], np.cumsum(cost)])
    bounds = [0]
    for g in range(1, world_size):
        target = cum[-1] * g / world_size
        r = int(np.searchsorted(cum, target))
        # nearest boundary, but keep every band non-empty where possible
        if r > 0 and abs(cum[r - 1] - target) <= abs(cum[min(r, n)] - target):
            r -= 1
        r = max(r, bounds[-1] + (1 if n - bounds[-1] > world_size - g else 0))
        r = min(r, n - (world_size - g) if n >= world_size else n)
        bounds.append(max(r, bounds[-1]))
    bounds.append(n)
    return bounds


def rebalance_on_measured_times(bounds, band_ms, row_entries, tiles_per_row: int):
    """Tile-row boundaries re-cut on MEASURED band times (a renderer has the previous frame's band times; a model that knows list lengths
    does not know cascades, clusters or launch floors).  bounds: the world + 1 boundaries the times were taken with; band_ms[r]: band r's step time;
    row_entries: the model's per-row list volumes (row_cost_entries).  Every row's model cost is scaled so that each band's modelled cost equals its
    measured time -- the model keeps the variation INSIDE a band, the measurement fixes the level of each band -- and the rows are cut again into
    bands of equal cost.  Returns the new boundaries (same conventions as balanced_tile_rows: non-decreasing, every band non-empty where possible)."""
    import numpy as np
    world = len(bounds) - 1
    model = tiles_per_row * TILE_COST + np.asarray(row_entries, dtype=np.float64) * ENTRY_COST
    n = len(model)
    cost = np.zeros(n, np.float64)
    for r in range(world):
        lo, hi = int(bounds[r]), int(bounds[r + 1])
        if hi <= lo:
            continue
        m = model[lo:hi]
        cost[lo:hi] = m * (float(band_ms[r]) / max(float(m.sum()), 1e-30))
    cum = np.concatenate([[0.0], np.cumsum(cost)])
    out = [0]
    for g in range(1, world):
        target = cum[-1] * g / world
        r = int(np.searchsorted(cum, target))
        if r > 0 and abs(cum[r - 1] - target) <= abs(cum[min(r, n)] - target):
            r -= 1
        r = max(r, out[-1] + (1 if n - out[-1] > world - g else 0))
        r = min(r, n - (world - g) if n >= world else n)
        out.append(max(r, out[-1]))
    out.append(n)
    return out


def gather_row_entries(band_grid: torch.Tensor, tiles_per_row: int, band_rows: int, total_rows: int, row_begin: int, group=None):
    """All-gather of the per-tile-row costs (row_cost_entries) of every rank's band -> float64 numpy [total_rows] on every rank."""
    num = row_cost_entries(band_grid.reshape(-1, 2)[:, 1].to(torch.int64), tiles_per_row) if band_rows else torch.zeros(0, dtype=torch.int64, device=band_grid.device)
    full = torch.zeros(total_rows, dtype=torch.int64, device=band_grid.device)
    full[row_begin:row_begin + band_rows] = num
    dist.all_reduce(full, op=dist.ReduceOp.SUM, group=group)
    return full.cpu().numpy().astype("float64")


def exchange_lists(band_grid: torch.Tensor, band_culled: torch.Tensor, group=None):
    """band_grid: int32[bandTiles*2] ({offset, num} pairs, band-local offsets); band_culled: int32[1 + ...] with [0] = band total.
    Returns (global_grid int32[T*2], global_culled int32[1 + sum]) in the reference's canonical layout, on every rank."""
    world = dist.get_world_size(group)
    dev = band_culled.device
    total = band_culled[:1].clone()
    totals = [torch.zeros_like(total) for _ in range(world)]
    dist.all_gather(totals, total, group=group)                      # collective 1
    totals_host = torch.stack(totals).reshape(-1).cpu().tolist()
    max_total = max(max(totals_host), 1)
    seg = torch.zeros(max_total, dtype=band_culled.dtype, device=dev)
    my_total = int(totals_host[dist.get_rank(group)])
    seg[:my_total] = band_culled[1:1 + my_total]
    segs = [torch.empty_like(seg) for _ in range(world)]
    dist.all_gather(segs, seg, group=group)                          # collective 2
    # grids differ in size per band: gather them padded as well (8 B per tile)
    tiles = torch.tensor([band_grid.numel()], dtype=torch.int64, device=dev)
    all_tiles = [torch.zeros_like(tiles) for _ in range(world)]
    dist.all_gather(all_tiles, tiles, group=group)
    tiles_host = [int(t.item()) for t in all_tiles]
    gpad = torch.zeros(max(max(tiles_host), 1), dtype=band_grid.dtype, device=dev)
    gpad[: band_grid.numel()] = band_grid
    grids = [torch.empty_like(gpad) for _ in range(world)]
    dist.all_gather(grids, gpad, group=group)
    out_grid, out_idx, base = [], [torch.zeros(1, dtype=band_culled.dtype, device=dev)], 0
    for r in range(world):
        g = grids[r][: tiles_host[r]].clone().reshape(-1, 2)
        g[:, 0] += base                                              # rebase to the canonical global offsets
        out_grid.append(g.reshape(-1))
        out_idx.append(segs[r][: int(totals_host[r])])
        base += int(totals_host[r])
    culled = torch.cat(out_idx)
    culled[0] = base
    return torch.cat(out_grid), culled


def gather_rows(band_rows: torch.Tensor, group=None) -> torch.Tensor:
    """All-gather per-band framebuffer rows (e.g. radiance) into the full frame.  Band 0 is the BOTTOM of the framebuffer
    (tile row 0 = last framebuffer rows), so bands are concatenated in reverse rank order."""
    world = dist.get_world_size(group)
    dev = band_rows.device
    n = torch.tensor([band_rows.shape[0]], dtype=torch.int64, device=dev)
    counts = [torch.zeros_like(n) for _ in range(world)]
    dist.all_gather(counts, n, group=group)
    counts = [int(c.item()) for c in counts]
    pad = torch.zeros((max(counts),) + tuple(band_rows.shape[1:]), dtype=band_rows.dtype, device=dev)
    pad[: band_rows.shape[0]] = band_rows
    parts = [torch.empty_like(pad) for _ in range(world)]
    dist.all_gather(parts, pad, group=group)
    return torch.cat([parts[r][: counts[r]] for r in reversed(range(world))], 0)


def allgather_visibility(visibility: torch.Tensor, rank: int, world: int, words_per_rank: int, group=None) -> torch.Tensor:
    """K4 split across ranks: rank r has swept the entities of words [r * words_per_rank, (r + 1) * words_per_rank) into its part of
    `visibility` (int64 [>= world * words_per_rank]); one all-gather completes the bitmask on every rank, in place.  The torch.distributed form of
    sailor_hip_exchange_visibility (gloo in the CPU tests, RCCL through torch on a node)."""
    if world == 1 or words_per_rank == 0:
        return visibility
    mine = visibility[rank * words_per_rank:(rank + 1) * words_per_rank].clone()
    parts = [torch.empty_like(mine) for _ in range(world)]
    dist.all_gather(parts, mine, group=group)
    visibility[: world * words_per_rank] = torch.cat(parts)
    return visibility


# ---- the shipped exchange: sailor_hip_exchange_light_lists_rows over an ncclComm_t (exchange.hip) -------------------------------
def _load_rccl():
    """The librccl this process already has mapped (torch brings its own copy under torch/lib; exchange.hip looks for the same one), else the loader's."""
    import ctypes as C
    try:
        with open("/proc/self/maps") as f:
            for line in f:
                path = line.rsplit(" ", 1)[-1].strip()
                if "librccl.so" in path:
                    return C.CDLL(path)
    except OSError:
        pass
    try:
        return C.CDLL("librccl.so.1")
    except OSError:
        return C.CDLL("librccl.so")


class RcclComm:
    """An ncclComm_t over the ranks of a torch.distributed group, created the way a host engine creates one: rank 0's ncclGetUniqueId travels over the
    existing group, every rank calls ncclCommInitRank.  `handle` is what the C-ABI's `comm` arguments take.  One rank per GPU (RCCL refuses two ranks
    on one device)."""

    def __init__(self, rank: int, world_size: int, group=None):
        import ctypes as C
        try:
            self.lib = _load_rccl()
        except OSError:
            self.lib = None   # (raised below, behind the broadcast every rank takes part in)

        class UniqueId(C.Structure):
            _fields_ = [("internal", C.c_char * 128)]
        uid = UniqueId()
        payload = None
        if rank == 0:
            try:
                if self.lib is not None and self.lib.ncclGetUniqueId(C.byref(uid)) == 0:
                    payload = bytes(bytearray(uid))
            except Exception:
                payload = None
        everyone_has_lib = self.lib is not None
        if world_size > 1:
            # Every rank takes part in both collectives whatever it found: rank 0's id (None on failure) travels to all, and every rank tells all whether
            # it has the library -- ncclCommInitRank is itself collective, so a rank that raised here alone would leave the others blocked in it forever
            # (ADVICE r04).  Only when ALL ranks can go on does any of them go on.
            box = [payload]
            dist.broadcast_object_list(box, src=0, group=group)
            payload = box[0]
            have = [None] * world_size
            dist.all_gather_object(have, self.lib is not None, group=group)
            everyone_has_lib = all(have)
        if payload is None or not everyone_has_lib:
            raise RuntimeError("no librccl on this rank" if self.lib is None else
                               ("ncclGetUniqueId failed on rank 0" if payload is None else "another rank has no librccl"))
        C.memmove(C.byref(uid), payload, 128)
        comm = C.c_void_p()
        self.lib.ncclCommInitRank.argtypes = [C.POINTER(C.c_void_p), C.c_int, UniqueId, C.c_int]
        rc = self.lib.ncclCommInitRank(C.byref(comm), world_size, uid, rank)
        ok = rc == 0
        if world_size > 1:   # (a failure on some rank only: every rank learns of it and none keeps a communicator the others do not have)
            oks = [None] * world_size
            dist.all_gather_object(oks, ok, group=group)
            ok_all = all(oks)
        else:
            ok_all = ok
        if not ok_all:
            if ok:
                self.lib.ncclCommDestroy.argtypes = [C.c_void_p]
                self.lib.ncclCommDestroy(comm)
            raise RuntimeError(f"ncclCommInitRank failed: {rc}" if not ok else "ncclCommInitRank failed on another rank")
        self.handle, self.rank, self.world_size = comm, rank, world_size

    def close(self):
        if getattr(self, "handle", None):
            import ctypes as C
            self.lib.ncclCommDestroy.argtypes = [C.c_void_p]
            self.lib.ncclCommDestroy(self.handle)
            self.handle = None


class ListExchange:
    """The split frame's exchange with its buffers kept: workspace and the two global buffers are allocated once, record() only RECORDS on the
    context's stream (sailor_hip_exchange_light_lists_rows: three ncclAllGather + one stitch kernel; nothing read back, nothing waits), adapt() is the
    one synchronising call (sailor_hip_exchange_adapt: waits for the last recorded exchange, sizes the next one's slots from its gathered totals).
    Every rank must call record / adapt in the same order."""

    def __init__(self, ctx, comm: "RcclComm", width: int, height: int, tile_row_bounds, device):
        import numpy as np

        from . import _lib
        self.ctx, self.comm, self.W, self.H = ctx, comm, width, height
        self.world = comm.world_size
        self.bounds = np.ascontiguousarray(tile_row_bounds, np.int32)
        assert len(self.bounds) == self.world + 1
        self.Tx, self.Ty = host.num_tiles(width, height)
        need = ctx._lib.sailor_hip_exchange_workspace_size_rows(width, height, self.world, self.bounds.ctypes.data)
        if need == 0:
            raise ValueError(f"invalid tile-row bounds {self.bounds.tolist()} for {width}x{height} over {self.world} ranks")
        self.ws = torch.empty(need, dtype=torch.uint8, device=device)
        self.out_grid = torch.zeros(self.Tx * self.Ty * 2, dtype=torch.int32, device=device)
        self.out_culled = torch.zeros(1 + self.Tx * self.Ty * _lib.LIGHTS_PER_TILE, dtype=torch.int32, device=device)
        self.max_tiles = int(max(self.bounds[1:] - self.bounds[:-1])) * self.Tx
        self.worst_case_slot_words = self.max_tiles * _lib.LIGHTS_PER_TILE
        self.slot_words = 0   # what the context will use for the next record(): 0 = the worst case

    def record(self, band_grid: torch.Tensor, band_culled: torch.Tensor):
        from . import _lib
        lib = self.ctx._lib
        _lib.check(lib.sailor_hip_exchange_light_lists_rows(self.ctx.handle, self.comm.handle, self.comm.rank, self.world, self.W, self.H, self.bounds.ctypes.data,
                                                             band_grid.data_ptr(), band_culled.data_ptr(), self.out_grid.data_ptr(), self.Tx * self.Ty,
                                                             self.out_culled.data_ptr(), self.out_culled.numel(), self.ws.data_ptr(), self.ws.numel()),
                   "sailor_hip_exchange_light_lists_rows", self.ctx.handle)
        return self.out_grid, self.out_culled

    def adapt(self):
        """-> (largest band total of the last exchange, clipped?, slot words of the next one)"""
        import ctypes as C

        from . import _lib
        largest, clipped, slot = C.c_uint32(0), C.c_int32(0), C.c_size_t(0)
        _lib.check(self.ctx._lib.sailor_hip_exchange_adapt(self.ctx.handle, C.byref(largest), C.byref(clipped), C.byref(slot)), "sailor_hip_exchange_adapt", self.ctx.handle)
        self.slot_words = int(slot.value)
        return int(largest.value), bool(clipped.value), int(slot.value)

    def bytes_gathered(self) -> int:
        """what ONE rank receives per exchange at the current slot size: world x (total word + index slot + grid slot)"""
        slot = self.slot_words or self.worst_case_slot_words
        return self.world * 4 * (1 + slot + 2 * self.max_tiles)


def exchange_lists_rccl(ctx, comm: "RcclComm", width: int, height: int, tile_row_bounds, band_grid: torch.Tensor, band_culled: torch.Tensor):
    """The split frame's exchange as the C++ host runs it (HipGraphicsDriver::ExchangeLightLists -> sailor_hip_exchange_light_lists_rows: three
    ncclAllGather on `comm` and the context's stream + one stitch kernel).  tile_row_bounds: world + 1 tile-row boundaries (equal or cost-balanced
    bands).  Returns (global_grid int32[T*2], global_culled int32[1 + T*128]) on this rank, canonical layout; exchange_lists above is the same
    exchange over torch.distributed (the gloo tests' stand-in)."""
    import ctypes as C

    import numpy as np

    from . import _lib
    lib = ctx._lib
    world = comm.world_size
    bounds = np.ascontiguousarray(tile_row_bounds, np.int32)
    assert len(bounds) == world + 1
    Tx, Ty = host.num_tiles(width, height)
    need = lib.sailor_hip_exchange_workspace_size_rows(width, height, world, bounds.ctypes.data)
    if need == 0:
        raise ValueError(f"invalid tile-row bounds {bounds.tolist()} for {width}x{height} over {world} ranks")
    dev = band_culled.device
    ws = torch.empty(need, dtype=torch.uint8, device=dev)
    out_grid = torch.zeros(Tx * Ty * 2, dtype=torch.int32, device=dev)
    out_culled = torch.zeros(1 + Tx * Ty * _lib.LIGHTS_PER_TILE, dtype=torch.int32, device=dev)
    _lib.check(lib.sailor_hip_exchange_light_lists_rows(ctx.handle, comm.handle, comm.rank, world, width, height, bounds.ctypes.data, band_grid.data_ptr(),
                                                         band_culled.data_ptr(), out_grid.data_ptr(), Tx * Ty, out_culled.data_ptr(), out_culled.numel(),
                                                         ws.data_ptr(), ws.numel()),
               "sailor_hip_exchange_light_lists_rows", ctx.handle)
    ctx.synchronize()   # (the workspace is released on return)
    return out_grid, out_culled
