"""Seed sharding across the GPUs of one node: one process per GPU, rank g owns a contiguous slab of the candidate
seeds of cppflow/planners.py:231-251, the robot / target path / obstacles are replicated, and all-gathers give every rank what
cppflow/search.py:146-151 consumes.  The reference has no collective anywhere; this module is the product entry point for the
sharded path -- `bench.py` only times it.

Three layers:
  * shard arithmetic and the two payloads (`seed_shard`, `padded_shard_size`, `allgather_seed_summaries`: 8 floats per SEED,
    latency-bound, every step; `allgather_seed_outputs`: the packed 15 B / row buffer, once per planning call);
  * transports: one object with `all_gather(out, inp)` enqueued on torch's CURRENT stream -- RCCL through the library's own C ABI
    (`CAbiAllGather`: ncclAllGather on the launch stream itself), RCCL / gloo through torch.distributed (`C10dAllGather`), a
    host-staged one for ranks sharing one GPU (`HostStagedAllGather`, rehearsal only); `pick_transport` agrees on one across ranks;
  * `ShardedRefiner`: the per-rank engine -- a ring of output-buffer sets, launches of B consecutive steps each
    (`Robot.lm_batch_plan`: B independent problems in one full-width grid), buckets of G steps whose [S,8] summaries travel in
    ONE in-stream all-gather followed by x_is_valid's seed selection over every rank's seeds
    (`Robot.select_valid_seed` on the gathered buffer, cppflow/optimization_utils.py:856-909), and, once per planning call,
    `gather_and_search`: the all-gather of the packed per-row outputs + candidate paths and `dp_search` over all of them
    (cppflow/search.py:128-191).  `sharded_candidate_evaluation` is the one-call form `Planner._run_pipeline` uses when a
    process group is up.

Throughput, not latency: a rank's launch carries B steps = B INDEPENDENT requests (at N = 8 the 32 768-row shard of one request is
an eighth of the chip; eight consecutive requests' shards make one full-width launch).  One planning call's own shard is
latency-bound: a 32 768-row launch takes ~26 us against 49 us for the unsharded 262 144 rows.
"""

import ctypes
import sys
import time
from dataclasses import dataclass
from typing import List, Optional, Tuple

import numpy as np
import torch
import torch.distributed as dist

PACKED_BYTES_PER_ROW = 15
SEED_SUMMARY_FLOATS = 8  # Robot.SEED_SUMMARY_FIELDS
FULL_WIDTH_ROWS = 262144  # four wavefronts per SIMD of the row shape on 256 CUs: the launch width of the unsharded C4 workload
LM_POSE = dict(lm_lambda=1e-6, alpha_position=3.5, alpha_rotation=0.35)  # ALT_LOSS_V2_1_POSE (cppflow/lm_hyper_parameters.py:119-151)


# ---- payloads -------------------------------------------------------------------------------------------------------------------
def _all_gather_flat(out: torch.Tensor, inp: torch.Tensor, group=None, async_op: bool = False):
    """`dist.all_gather_into_tensor(out, inp)` -- `out` the concatenation of every rank's `inp` along dim 0 -- staged through the
    host when the group's backend cannot take device tensors (gloo with HIP tensors: several ranks SHARING one GPU, which RCCL
    refuses; a rehearsal of the N > 1 path on the hardware there is, exact for semantics and meaningless for timing)."""
    if inp.is_cuda and dist.get_backend(group) == "gloo":
        h_out = torch.empty(out.shape, dtype=out.dtype)
        dist.all_gather_into_tensor(h_out, inp.cpu(), group=group)
        out.copy_(h_out)
        return None
    return dist.all_gather_into_tensor(out, inp, group=group, async_op=async_op)


def allgather_seed_summaries(summary: torch.Tensor, group: Optional[dist.ProcessGroup] = None,
                             out: Optional[torch.Tensor] = None, async_op: bool = False):
    """All-gather the [S_local, 8] per-seed summaries (`Robot.seed_summary`) -> [world * S_local, 8]: 32 bytes per seed,
    so the collective is latency-bound and can be issued every step; with `async_op=True` it returns (out, work) and runs
    on the communicator's own stream beside the next step's kernel.  Every rank must pass the same S_local (uneven shards:
    allocate `padded_shard_size` seeds and strip the filler with `drop_padding` once the collective has completed)."""
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    if world == 1:
        return (summary, None) if async_op else summary
    if out is None:
        out = torch.empty((world * summary.shape[0], summary.shape[1]), dtype=summary.dtype, device=summary.device)
    work = _all_gather_flat(out, summary.contiguous(), group, async_op)
    return (out, work) if async_op else out


def drop_padding(gathered: torch.Tensor, seeds_per_rank: int, counts) -> torch.Tensor:
    """[world * seeds_per_rank, ...] gathered from padded shards -> [sum(counts), ...], filler seeds removed."""
    if all(c == seeds_per_rank for c in counts):
        return gathered
    return torch.cat([gathered[r * seeds_per_rank : r * seeds_per_rank + c] for r, c in enumerate(counts)], dim=0)


def seed_shard(n_seeds_total: int, rank: int, world: int) -> Tuple[int, int]:
    """[begin, end) of the seeds rank owns: contiguous, sizes differ by at most one."""
    base, rem = divmod(n_seeds_total, world)
    begin = rank * base + min(rank, rem)
    return begin, begin + base + (1 if rank < rem else 0)


def padded_shard_size(n_seeds_total: int, W: int, world: int) -> int:
    """Seeds per rank of the BUFFERS every rank allocates: `all_gather_into_tensor` needs identical sizes on every rank and
    the packed per-row buffer needs S_pad * W % 4 == 0 (its fp32 part starts each rank's slice 4-byte aligned), so the
    largest shard is rounded up accordingly; a rank fills the seeds it does not own with copies of its last seed (any
    finite rows do) and the gathers below drop them again."""
    s = -(-n_seeds_total // world)
    while (s * W) % 4 != 0:
        s += 1
    return s


def shard_counts(n_seeds_total: int, world: int):
    """Number of seeds each rank really owns (the rest of its padded buffer is filler)."""
    return [seed_shard(n_seeds_total, r, world)[1] - seed_shard(n_seeds_total, r, world)[0] for r in range(world)]


@dataclass
class GatheredSeedOutputs:
    ext_cost: torch.Tensor  # float32 [S_total, W]
    pos_err_m: torch.Tensor  # float32 [S_total, W]
    rot_err_rad: torch.Tensor  # float32 [S_total, W]
    self_mask: torch.Tensor  # bool [S_total, W]
    env_mask: torch.Tensor  # bool [S_total, W]
    jlim_mask: torch.Tensor  # bool [S_total, W]


def unpack_rows(packed: torch.Tensor, n: int):
    """Views into one rank's packed buffer: (ext_cost, pos_err_m, rot_err_rad) float32 [n], 3 x uint8 [n]."""
    assert packed.dtype == torch.uint8 and packed.numel() == PACKED_BYTES_PER_ROW * n
    f = packed[: 12 * n].view(torch.float32)
    m = packed[12 * n :]
    return f[:n], f[n : 2 * n], f[2 * n :], m[:n], m[n : 2 * n], m[2 * n :]


def allgather_seed_outputs(
    packed: torch.Tensor, seeds_per_rank: int, W: int, group: Optional[dist.ProcessGroup] = None,
    out: Optional[torch.Tensor] = None, counts=None,
) -> GatheredSeedOutputs:  # fmt: skip
    """One all-gather of every rank's packed buffer -> the per-seed matrices over ALL seeds.  Every rank passes a buffer of
    the same `seeds_per_rank` (`padded_shard_size` when the seeds do not divide evenly); `counts` (seeds really owned per
    rank, `shard_counts`) strips the filler seeds from the result."""
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    n = seeds_per_rank * W
    assert packed.numel() == PACKED_BYTES_PER_ROW * n
    assert world == 1 or n % 4 == 0, (
        f"seeds_per_rank * W = {n} must be a multiple of 4 (each rank's fp32 slice of the gathered buffer must start "
        "4-byte aligned): allocate distributed.padded_shard_size(...) seeds per rank")
    assert counts is None or (len(counts) == world and all(0 <= c <= seeds_per_rank for c in counts)), counts
    if world == 1:
        gathered = packed.view(1, -1)
    else:
        if out is None:
            out = torch.empty(world * packed.numel(), dtype=torch.uint8, device=packed.device)
        _all_gather_flat(out, packed, group)
        gathered = out.view(world, -1)
    parts = [unpack_rows(gathered[r], n) for r in range(world)]

    def cat(i, as_bool=False):
        keep = counts if counts is not None else [seeds_per_rank] * world
        t = torch.cat([p[i].reshape(seeds_per_rank, W)[:c] for p, c in zip(parts, keep)], dim=0)
        return t.view(torch.bool) if as_bool else t

    return GatheredSeedOutputs(cat(0), cat(1), cat(2), cat(3, True), cat(4, True), cat(5, True))


# ---- transports: all_gather(out [world, ...], inp [...]) on torch's CURRENT stream ------------------------------------------------
class C10dAllGather:
    """torch.distributed's all_gather_into_tensor ("nccl" = RCCL on ROCm; gloo on CPU tensors in the unit tests).  On a GPU the
    communicator runs on its own stream; `work.wait()` makes the current stream wait for it (stream-side, no host block)."""

    name = "torch.distributed all_gather_into_tensor"

    def __init__(self, group=None):
        self.group = group
        self.world = dist.get_world_size(group)

    def all_gather(self, out, inp):
        flat = out.view((out.shape[0] * inp.shape[0],) + tuple(inp.shape[1:]))  # the concatenated form of the same memory
        work = dist.all_gather_into_tensor(flat, inp, group=self.group, async_op=True)
        work.wait()

    def close(self):
        pass


class HostStagedAllGather:
    """Rehearsal transport for several ranks sharing ONE GPU (which RCCL refuses): gloo with the payload staged on the host.
    Same call sites, same dependency structure, meaningless timing."""

    name = "gloo, host-staged (one-GPU rehearsal: NOT a multi-GPU result)"

    def __init__(self, group=None):
        self.group = group
        self.world = dist.get_world_size(group)

    def all_gather(self, out, inp):
        host_in = inp.cpu()  # synchronises on the current stream
        host_out = torch.empty((out.shape[0] * inp.shape[0],) + tuple(inp.shape[1:]), dtype=out.dtype)
        dist.all_gather_into_tensor(host_out, host_in, group=self.group)
        out.copy_(host_out.view(out.shape))

    def close(self):
        pass


class LocalAllGather:
    """world = 1 without a process group, or a diagnostic with the collective taken out: everything of an exchange step but the wire."""

    name = "none (one rank / diagnostic)"
    world = 1

    def all_gather(self, out, inp):
        out.view((out.shape[0] * inp.shape[0],) + tuple(inp.shape[1:]))[: inp.shape[0]].copy_(inp)

    def close(self):
        pass


class CAbiAllGather:
    """RCCL through the library's own C ABI (cppf_comm_init_rank / cppf_allgather_bytes): ONE ncclAllGather enqueued on the
    launch stream itself -- no second stream, no c10d bookkeeping (3 us of host time and 6 us on the stream against 28 / 33 us for
    torch.distributed's call on one rank, scripts/gather_latency.py).  Built by `pick_transport`."""

    name = "RCCL through the C ABI (cppf_allgather_bytes on the launch stream)"

    def __init__(self, comm, world):
        from cppflow_amd import _hip

        self._hip, self.comm, self.world = _hip, comm, world

    def all_gather(self, out, inp):
        nbytes = inp.numel() * inp.element_size()
        assert out.numel() * out.element_size() == nbytes * self.world
        self._hip.check(self._hip.lib().cppf_allgather_bytes(self.comm, inp.data_ptr(), out.data_ptr(), nbytes,
                                                              torch.cuda.current_stream(inp.device).cuda_stream))

    def close(self):
        comm, self.comm = self.comm, None
        if comm is not None:
            self._hip.lib().cppf_comm_destroy(comm)


def _all_ok(ok: bool, device, group=None) -> bool:
    flag = torch.tensor([1 if ok else 0], dtype=torch.int32, device=device)
    dist.all_reduce(flag, op=dist.ReduceOp.MIN, group=group)
    return int(flag.item()) == 1


def pick_transport(device: torch.device, group=None, prefer: str = "cabi"):
    """(transport, record) for the ranks of an initialised "nccl" group: the C-ABI communicator if EVERY rank can bring it up and
    its all-gather equals torch.distributed's own on a probe, else the c10d call on every rank.

    Every rank executes the SAME sequence of collectives on the c10d group whatever happens to it locally: local failures become
    flags, and after each stage all ranks MIN-reduce their flag and leave together.  Stage 0 (no collective): can this rank load
    RCCL through the library (cppf_comm_available)?  1: rank 0 draws the unique id and ALWAYS broadcasts (status, id).  2: all
    ranks agree to go on, then call cppf_comm_init_rank.  3: agree again, then the probe (both all-gathers on every rank).  4:
    agree on the comparison.  A successful probe is the validation of this transport on the hardware the run is on."""
    from cppflow_amd import _hip

    rank, world = dist.get_rank(group), dist.get_world_size(group)
    if prefer != "cabi":
        t = {"c10d": lambda: C10dAllGather(group), "none": LocalAllGather}[prefer]()
        return t, {"requested": prefer, "transport": t.name, "world_seen": world, "unique_id_via": "torch.distributed (c10d store)"}
    rec = {"requested": "cabi", "stages": []}
    why, lib = "", None
    try:
        lib = _hip.lib()
        ok = lib.cppf_comm_available() == 0
        if not ok:
            why = lib.cppf_last_error().decode("utf-8", "replace")
    except Exception as e:  # noqa: BLE001 -- any local failure becomes a flag
        ok, why = False, repr(e)
    box = [None]
    if rank == 0:
        uid = (ctypes.c_char * 128)()
        st = False
        if ok:
            try:
                st = lib.cppf_comm_unique_id(uid) == 0
            except Exception as e:  # noqa: BLE001
                why = repr(e)
        box = [(bool(st), bytes(uid))]
    dist.broadcast_object_list(box, src=dist.get_global_rank(group, 0) if group is not None else 0, group=group)
    ok = ok and bool(box[0][0])
    go = _all_ok(ok, device, group)
    rec["stages"].append({"stage": "load RCCL through the C ABI + unique id from rank 0 (torch.distributed broadcast)", "ok": go})
    comm = ctypes.c_void_p()
    if go:
        try:
            uid = (ctypes.c_char * 128).from_buffer_copy(box[0][1])
            ok = lib.cppf_comm_init_rank(uid, rank, world, device.index, ctypes.byref(comm)) == 0 and lib.cppf_comm_world(comm) == world
            if not ok:
                why = lib.cppf_last_error().decode("utf-8", "replace")
        except Exception as e:  # noqa: BLE001
            ok, why = False, repr(e)
        go = _all_ok(ok, device, group)
        rec["stages"].append({"stage": "cppf_comm_init_rank on every rank", "ok": go})
    if go:
        cabi = CAbiAllGather(comm, world)
        probe = torch.full((1, 4, 8), float(rank + 1), dtype=torch.float32, device=device)
        got = torch.zeros((world, 1, 4, 8), dtype=torch.float32, device=device)
        want = torch.zeros_like(got)
        try:
            cabi.all_gather(got, probe)
        except Exception as e:  # noqa: BLE001
            ok, why = False, repr(e)
        dist.all_gather_into_tensor(want.view(world, 4, 8), probe, group=group)  # (every rank, whatever the C-ABI call did)
        torch.cuda.synchronize()
        ok = ok and bool(torch.equal(got, want))
        if not ok and not why:
            why = "probe mismatch"
        go = _all_ok(ok, device, group)
        rec["stages"].append({"stage": "probe: cppf_allgather_bytes == torch.distributed all_gather_into_tensor", "ok": go})
        if go:
            rec.update(transport=cabi.name, world_seen=int(lib.cppf_comm_world(comm)),
                       unique_id_via="cppf_comm_unique_id on rank 0 -> torch.distributed broadcast_object_list")
            return cabi, rec
    if comm.value:
        try:
            lib.cppf_comm_destroy(comm)
        except Exception:  # noqa: BLE001
            pass
    print(f"cppflow_amd.distributed: rank {rank}: C-ABI RCCL transport not used ({why or 'failed on another rank'}); every rank uses "
          "torch.distributed's all-gather", file=sys.stderr)
    t = C10dAllGather(group)
    rec.update(transport=t.name, world_seen=world, unique_id_via="torch.distributed (c10d store)", fallback_reason=why or "failed on another rank")
    return t, rec


def device_census(rank: int, dev_index: int, gather: bool = True):
    """[{rank, device ordinal, PCI bus id, uuid, name}] of every rank (gathered through the default process group): lets a reader of a
    multi-GPU record check that N ranks sat on N different GPUs."""
    p = torch.cuda.get_device_properties(dev_index)
    bus = None
    if all(hasattr(p, a) for a in ("pci_domain_id", "pci_bus_id", "pci_device_id")):
        bus = f"{p.pci_domain_id:04x}:{p.pci_bus_id:02x}:{p.pci_device_id:02x}.0"
    mine = {"rank": rank, "device": dev_index, "pci_bus_id": bus, "uuid": str(getattr(p, "uuid", "")) or None, "name": p.name}
    if not (gather and dist.is_initialized()):
        return [mine]
    everyone = [None] * dist.get_world_size()
    dist.all_gather_object(everyone, mine)
    return everyone


# ---- how a rank issues its steps -----------------------------------------------------------------------------------------------
def launch_plan(rows_step: int, steps_hint: int, batch: int = 0, gather_every: int = 0, streams: int = 0, quad: bool = False,
                max_batch: int = 16):
    """How a rank issues a run of `steps_hint` steps of `rows_step` rows each -> (steps per launch B, steps per collective G, the
    bucket size requested before clamping, launch streams).  Pure host logic (tests/test_sharded_refiner_order.py holds it to the
    measured choices for 1024 / 512 / 256 / 128 seeds per rank).
      B: as many of this rank's steps as make one full-width launch (1 for the unsharded C4 step; 2 / 4 / 8 for the shards of 2 / 4 /
         8 GPUs), so that every GPU issues launches of the same width at every N.
      G: the all-gather's latency (tens of microseconds across a node) is paid once per bucket: 8 steps, 32 for shards of <= 65 536
         rows, 64 for <= 32 768; a multiple of B; never more than half of a short run (a run of K steps holds at least two buckets
         when it can), and in a run of fewer than four buckets every launch is followed by its own exchange so that consecutive
         launches alternate between the streams (profiles/r4_short_region_buckets.txt).
      streams: launches of <= 2 wavefronts per SIMD: two in flight cannot fill the chip, four can (profiles/r2_hwq_sweep.txt)."""
    B = batch if batch > 0 else max(1, min(max_batch, FULL_WIDTH_ROWS // max(rows_step, 1)))
    if quad:
        B = 1
    G_req = gather_every if gather_every > 0 else (64 if rows_step <= 32768 else (32 if rows_step <= 65536 else 8))
    G = max(B, (min(G_req, max(steps_hint // 2, 1)) // B) * B)
    if gather_every <= 0 and steps_hint < 4 * G:
        G = B if (rows_step > 65536 or 2 * B > 8) else 2 * B
    n_streams = streams if streams > 0 else (4 if rows_step * B <= 131072 else 2)
    return B, G, G_req, n_streams


def _hip_keys():
    from cppflow_amd import _hip

    return _hip.TUNE_KEYS


class ShardedRefiner:
    """One rank's engine for a stream of refinement steps over its shard of the seeds: S_local seeds x W waypoints per step, K
    fused LM iterations + pose metrics + collision masks + search cost + per-seed summary per row (cppf_lm_batch_launch).

    A LAUNCH GROUP is B consecutive ring slots; a launch starts at a group's first slot and carries 1 .. B of its steps (the K mod
    B steps left over at the end of a run go out as one shorter launch, after which the ring moves on to the next group).  Without
    a transport consecutive groups alternate between the streams.  With one, the ring is `n_streams` BUCKETS of G steps (G a
    multiple of B); a bucket's launches all go to ONE stream and its exchange step -- the all-gather of the G [S,8] summaries and
    the seed selection over every rank's seeds -- is enqueued on that same stream right behind them, so producer -> collective ->
    consumer -> reuse of the bucket's buffers are ordered by the stream itself.  No cross-stream event anywhere (measured on a
    32 768-row shard: making an auxiliary stream wait on events of four launch streams cost 22.5 us per step against 7.2).

    `transport`: an object of this module (`pick_transport`), or None for a single rank that exchanges nothing.  `graphs=True`
    replays each bucket's launches as one captured hipGraph; the decision is made COLLECTIVELY when a group is given (a capture that
    fails on one rank turns graphs off on all: the ranks must issue identical sequences of collectives).
    `pace`: fair-share pacing of the fused launches (include/cppflow_hip_debug.h: CPPF_TUNE_LM_PACE; csrc/kernels_fused.h: lm_pace) --
    for launches that run ALONE on the chip, one after the other: None (default) = on exactly when there is one stream (a dependency
    chain: -3 .. -6 % per step), off with several (overlapping launches lose up to 5 % with it).  No effect on results."""

    def __init__(self, robot, x0: torch.Tensor, target: torch.Tensor, n_lm_steps: int, *, transport=None, group=None,
                 collide: bool = True, batch: int = 1, bucket: int = 1, n_streams: int = 2, shape: int = 0, solver: int = 0,
                 graphs: bool = False, constraints=None, lm=LM_POSE, pace: Optional[bool] = None):
        from cppflow_amd import _hip
        from cppflow_amd.data_types import DEFAULT_CONSTRAINTS

        self.robot, self.x0, self.target, self.K, self.collide = robot, x0, target, int(n_lm_steps), bool(collide)
        self.device, self.group, self.lm = x0.device, group, dict(lm)
        n, W = x0.shape[0], target.shape[0]
        assert n % W == 0
        self.n, self.S, self.W = n, n // W, W
        self.transport = transport if collide else None
        self.world = self.transport.world if self.transport is not None else 1
        self.n_streams = max(1, int(n_streams))
        self.pace = (self.n_streams == 1) if pace is None else bool(pace)
        self.shape, self.solver = shape, solver
        self.B = max(1, min(int(batch), _hip.MAX_BATCH))
        self.use_batch_api = shape != _hip.SHAPE_QUAD  # an explicit quad shape keeps the plain per-step launches (B = 1)
        if not self.use_batch_api:
            self.B = 1
        self.buckets = self.transport is not None or bool(graphs)
        self.G = max(self.B, (max(1, int(bucket)) // self.B) * self.B) if self.buckets else self.B
        self.NBUF = self.n_streams * self.G if self.buckets else max(4, self.n_streams) * self.B
        self.constraints = constraints if constraints is not None else DEFAULT_CONSTRAINTS
        self.step_no = 0  # always a multiple of B: the ring position of the next launch
        self.start_bucket: Optional[int] = None  # where a run (`run_region`) restarts the ring; None = wherever it stands
        self.stream_calibration = None
        self.graphs = None
        self._allocate()
        self._make_launches()
        first = self.launches[0][0].outputs
        self.outputs = first[0] if isinstance(first, list) else first  # ring slot 0's output views
        self.streams = self._make_streams(self.n_streams)
        if graphs:
            self._capture_graphs()

    # ---- device-facing pieces (tests/test_distributed_gloo.py substitutes CPU stand-ins for these four) ----------------------------
    def _allocate(self):
        dev, n, NBUF = self.device, self.n, self.NBUF
        self.x_outs = [torch.empty_like(self.x0) for _ in range(NBUF)]
        self.packeds = [torch.empty(PACKED_BYTES_PER_ROW * n, dtype=torch.uint8, device=dev) if self.collide else None for _ in range(NBUF)]
        self.summ_all = torch.empty((NBUF, self.S, 8), dtype=torch.float32, device=dev) if self.collide else None
        self.errs = None if self.collide else [(torch.empty(n, device=dev), torch.empty(n, device=dev)) for _ in range(NBUF)]
        if self.transport is not None:
            self.gathered = [torch.empty((self.world, self.G, self.S, 8), dtype=torch.float32, device=dev) for _ in range(self.n_streams)]
            self.selected = [torch.empty((self.G, 4), dtype=torch.int32, device=dev) for _ in range(self.n_streams)]
        else:
            self.gathered = self.selected = None

    def _item(self, b):
        it = dict(x=self.x0, target=self.target, x_out=self.x_outs[b])
        if self.collide:
            it.update(packed_out=self.packeds[b], summary_out=self.summ_all[b])
        else:  # FK + Jacobian + LM only (BASELINE configs[1]): the result and its pose errors, no collision stage
            it.update(errors_out=self.errs[b])
        return it

    def _make_launches(self):
        """launches[g][c - 1]: the launch of the first c steps of group g (c = B: the group; c < B: what is left at the end of a run)"""
        rb, B, K = self.robot, self.B, self.K
        self.launches = []
        # (a launch plan takes the handle's scheduling switch as it is when the plan is made: set for the making only, then restored)
        pace_was = rb._tuning.get(_hip_keys()["lm_pace"]) if hasattr(rb, "_tuning") else None
        if self.pace and hasattr(rb, "debug_set"):
            rb.debug_set("lm_pace", -1)
        try:
            self._make_launch_plans(rb, B, K)
        finally:
            if self.pace and hasattr(rb, "debug_set"):
                rb.debug_set("lm_pace", pace_was)

    def _make_launch_plans(self, rb, B, K):
        for g in range(self.NBUF // B):
            if self.use_batch_api:
                self.launches.append([rb.lm_batch_plan([self._item(g * B + j) for j in range(c)], n_steps=K, solver=self.solver, **self.lm)
                                      for c in range(1, B + 1)])
            elif self.collide:
                self.launches.append([rb.lm_launch_plan(self.x0, self.target, n_steps=K, x_out=self.x_outs[g], packed_out=self.packeds[g],
                                                        summary_out=self.summ_all[g], shape=self.shape, solver=self.solver, **self.lm)])
            else:
                self.launches.append([rb.lm_launch_plan(self.x0, self.target, n_steps=K, x_out=self.x_outs[g], errors_out=self.errs[g],
                                                        shape=self.shape, solver=self.solver, **self.lm)])

    def _make_streams(self, count):
        cur = torch.cuda.current_stream(self.device)
        streams = [torch.cuda.Stream(device=self.device) for _ in range(count)]
        for st in streams:
            st.wait_stream(cur)  # the inputs were produced on the current stream
        return streams

    def _on(self, stream):
        return torch.cuda.stream(stream)

    def _select(self, gathered, out):
        """the consumer: cppflow/optimization_utils.py:856-909 over ALL ranks' seeds, one row of `out` per step of the bucket"""
        self.robot.select_valid_seed(gathered, self.constraints, out=out)

    def synchronize(self):
        torch.cuda.synchronize(self.device)

    # ---- graphs -----------------------------------------------------------------------------------------------------------------
    def _capture_graphs(self):
        """One hipGraph per bucket = its G / B launches in stream order, captured on the bucket's own stream (every launch once
        eagerly first: nothing lazy may happen inside a capture).  A replay costs the host one call per G steps."""
        B, G = self.B, self.G
        for g in range(self.NBUF // B):
            self.launches[g][B - 1].launch_on(self.streams[(g * B) // G])
        self.synchronize()
        graphs, ok = [], True
        try:
            for bucket in range(self.n_streams):
                gr = torch.cuda.CUDAGraph()
                with torch.cuda.graph(gr, stream=self.streams[bucket], capture_error_mode="thread_local"):
                    for g in range(bucket * G // B, (bucket + 1) * G // B):
                        self.launches[g][B - 1].launch_on(self.streams[bucket])
                graphs.append(gr)
        except RuntimeError as e:  # capture refused on this box: the eager path does the same work, one host call per launch
            print(f"cppflow_amd.distributed: hipGraph capture failed ({e}); continuing with eager launches", file=sys.stderr)
            ok = False
        self.synchronize()
        if dist.is_initialized() and self.transport is not None and self.world > 1:
            ok = _all_ok(ok, self.device, self.group)  # every rank or none (the ranks' collective sequences must not diverge)
        self.graphs = graphs if ok else None

    # ---- issuing steps ------------------------------------------------------------------------------------------------------------
    def stream_of(self, b):
        return self.streams[b // self.G] if self.buckets else self.streams[(b // self.B) % self.n_streams]

    def launch(self):
        """one full launch (B steps) on torch's current stream (group 0)"""
        self.launches[0][self.B - 1].launch()

    def run_steps(self, n):
        """`n` steps = floor(n / B) launches of B steps and one of n mod B; whole buckets go out as one graph replay each when
        graphs are on.  A bucket's exchange step follows its last launch ON ITS STREAM; the host issues it one launch late -- after
        the next bucket's first launch has gone to ITS stream -- so that the second stream's kernels are not held back by the
        ~10 us of host time the collective and the selection launch take."""
        B, G = self.B, self.G
        pending = None  # a complete bucket whose exchange step has not been issued yet

        def flush():
            nonlocal pending
            if pending is not None:
                self.exchange(pending)
                pending = None

        while n > 0:
            b = self.step_no % self.NBUF
            if self.graphs is not None and b % G == 0 and n >= G:
                bucket = b // G
                if pending == bucket:
                    flush()
                with self._on(self.streams[bucket]):
                    self.graphs[bucket].replay()
                self.step_no += G
                n -= G
                flush()
                if self.transport is not None:
                    pending = bucket
                continue
            c = min(B, n)
            if pending is not None and self.buckets and pending == b // G:
                flush()  # (a one-bucket ring: the launch below would overwrite the summaries still to be gathered)
            self.launches[b // B][c - 1].launch_on(self.stream_of(b))
            self.step_no += B  # (a shorter launch leaves the rest of its group unused: the ring moves on to the next group)
            n -= c
            flush()
            if self.buckets and self.transport is not None and (b + B) % G == 0:
                pending = b // G  # the bucket is complete: gather its G summaries from every rank and consume them
        flush()

    def exchange(self, bucket):
        G = self.G
        with self._on(self.streams[bucket]):
            self.transport.all_gather(self.gathered[bucket], self.summ_all[bucket * G : (bucket + 1) * G])
            self._select(self.gathered[bucket], self.selected[bucket])

    def drain(self):
        """the exchange of a partly filled bucket; the next step then starts a fresh bucket"""
        if self.buckets and self.step_no % self.G != 0:
            if self.transport is not None:
                self.exchange((self.step_no % self.NBUF) // self.G)
            self.step_no += self.G - self.step_no % self.G

    def run_region(self, steps):
        """`steps` steps + the exchange of a partly filled bucket, from the ring position `start_bucket` when one is set (everything
        issued before must have completed).  Asynchronous: the caller synchronises."""
        if self.start_bucket is not None:
            self.step_no = self.start_bucket * self.G
        self.run_steps(steps)
        self.drain()

    def prewarm(self, ms: float, count: Optional[int] = None):
        """untimed bare launches (sustained clocks, full pipeline) round-robin over the groups and streams for `ms` milliseconds, or
        exactly `count` launches: no collectives inside (a time-based loop would make the ranks issue different numbers of them)"""
        ngroups = self.NBUF // self.B
        if count is not None:
            for i in range(count):
                g = i % ngroups
                self.launches[g][self.B - 1].launch_on(self.stream_of(g * self.B))
            self.synchronize()
            return
        t0 = time.perf_counter()
        while (time.perf_counter() - t0) * 1e3 < ms:
            for i in range(max(1, 48 // self.B)):
                g = i % ngroups
                self.launches[g][self.B - 1].launch_on(self.stream_of(g * self.B))
            self.synchronize()

    # ---- optional: which streams a short run uses -----------------------------------------------------------------------------------
    def calibrate_streams(self, steps, barrier=None, pool_extra: int = 4, regions_per_candidate: int = 3):
        """Which hardware queue a stream lands on, and what else shares it, is decided when the runtime creates it: the same 8 / 8 /
        4-step run of a 32 768-row shard took 5.8 ... 7.0 us per step with different stream pairs of ONE process
        (profiles/r4_start_bucket.txt).  This times `regions_per_candidate` runs of `steps` steps for every ordered pair out of the
        two streams + `pool_extra` fresh ones (two buckets, eager launches), or for every start bucket otherwise, and keeps the
        fastest for `run_region`.  NOT the default: a fresh ShardedRefiner runs on the streams it created, and bench.py reports
        that as the headline and this as a secondary figure.  Every rank runs the same number of regions (they contain
        collectives) and decides for itself.  Returns the record (also kept as `stream_calibration`)."""
        if not (self.buckets and self.transport is not None and self.n_streams >= 2):
            return None

        def region():
            if barrier is not None:
                barrier()
            t0 = time.perf_counter()
            self.run_region(steps)
            self.synchronize()
            return time.perf_counter() - t0

        if self.n_streams == 2 and self.graphs is None:
            pool = list(self.streams) + self._make_streams(pool_extra)
            cands = [((i, j), 0) for i in range(len(pool)) for j in range(len(pool)) if i != j]
        else:
            pool = list(self.streams)
            cands = [(tuple(range(self.n_streams)), sb) for sb in range(self.n_streams)]
        med = []
        for idx, sb in cands:
            self.streams = [pool[i] for i in idx]
            self.start_bucket = sb
            med.append(float(np.median([region() for _ in range(regions_per_candidate)])))
        best = int(np.argmin(med))
        self.streams = [pool[i] for i in cands[best][0]]
        self.start_bucket = cands[best][1]
        self.stream_calibration = {"candidates": len(cands), "chosen_streams": list(cands[best][0]), "chosen_start_bucket": cands[best][1],
                                   "us_per_step": {"best": 1e6 * med[best] / steps, "median": 1e6 * float(np.median(med)) / steps,
                                                   "worst": 1e6 * max(med) / steps, "first": 1e6 * med[0] / steps}}
        return self.stream_calibration

    # ---- diagnostics ---------------------------------------------------------------------------------------------------------------
    def allgather_latency_us(self, reps: int = 100):
        """Untimed diagnostic: `reps` bare all-gathers of ONE step's [S,8] summaries on an otherwise idle stream, host clock around
        the lot (synchronised): what one collective of the exchange step costs on this transport and topology."""
        if self.transport is None:
            return None
        out = torch.empty((self.world, 1, self.S, 8), dtype=torch.float32, device=self.device)
        inp = self.summ_all[:1]
        with self._on(self.streams[0]):
            for _ in range(5):
                self.transport.all_gather(out, inp)
            self.synchronize()
            t0 = time.perf_counter()
            for _ in range(reps):
                self.transport.all_gather(out, inp)
            self.synchronize()
        return 1e6 * (time.perf_counter() - t0) / reps

    # ---- once per planning call ------------------------------------------------------------------------------------------------------
    def gather_candidates(self, slot: int = 0):
        """Every rank gets ALL ranks' per-row costs / masks and candidate paths of ring slot `slot` with one all-gather each
        (cppflow/search.py:146-151 consumes every candidate's cost row): (q_all [k, W, d], GatheredSeedOutputs)."""
        assert self.collide
        d = self.x0.shape[1]
        if self.world > 1 and dist.is_initialized():
            g = allgather_seed_outputs(self.packeds[slot], self.S, self.W, group=self.group)
            q_all = torch.empty((self.world * self.n, d), dtype=torch.float32, device=self.device)  # (the concatenated form: gloo insists on it)
            _all_gather_flat(q_all, self.x_outs[slot], self.group)
            return q_all.view(self.world * self.S, self.W, d), g
        cost, pe, re, sm, em, jm = (t.view(self.S, self.W) for t in unpack_rows(self.packeds[slot], self.n))
        g = GatheredSeedOutputs(cost, pe, re, sm.view(torch.bool), em.view(torch.bool), jm.view(torch.bool))
        return self.x_outs[slot].view(self.S, self.W, d), g

    def gather_and_search(self, slot: int = 0):
        """`gather_candidates` + `dp_search` over every rank's candidates (cppflow/search.py:128-191): (best_path [W, d], best_idx [W])."""
        q_all, g = self.gather_candidates(slot)
        best_path, best_idx, _ = self._dp_search(q_all.contiguous(), g.ext_cost.contiguous())
        return best_path, best_idx

    def _dp_search(self, q_all, cost_all):
        return self.robot.dp_search(q_all, cost_all)

    def close(self):
        self.graphs = None
        self.launches = []


def sharded_candidate_evaluation(problem, qs_local: torch.Tensor, lm_steps: int = 0, group=None, counts: Optional[List[int]] = None):
    """One planning call's sharded stage: this rank's candidates qs_local [k_local, T, d] (every rank the same k_local; `counts` =
    candidates really owned per rank when the last shards are padded) -> (qs_all [k, T, d], self_mask [k, T] bool, env_mask [k, T]
    bool) over ALL ranks' candidates, on every rank: optionally `lm_steps` fused pose-only LM iterations on every (candidate,
    waypoint) row (cppflow/optimization.py:61-92), the collision masks of cppflow/collision_detection.py:27-69 in the same
    launch, then ONE all-gather of the packed per-row buffer and one of the paths.  What `Planner._run_pipeline` calls between the
    seed provider and dp_search (cppflow/planners.py:231-274) when torch.distributed is initialised."""
    from cppflow_amd.search import DEFAULT_JLIM_SAFETY_PADDING_PRISMATIC, DEFAULT_JLIM_SAFETY_PADDING_REVOLUTE

    rb = problem.robot
    k_local, T, d = qs_local.shape
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    problem.bind_obstacles()
    rb.set_joint_limit_padding(DEFAULT_JLIM_SAFETY_PADDING_REVOLUTE, DEFAULT_JLIM_SAFETY_PADDING_PRISMATIC)  # search.py:20-21
    qs_local = qs_local.contiguous()
    n = k_local * T
    assert world == 1 or n % 4 == 0, "k_local * T must be a multiple of 4: allocate distributed.padded_shard_size(...) candidates per rank"
    packed = torch.empty(PACKED_BYTES_PER_ROW * n, dtype=torch.uint8, device=qs_local.device)
    if lm_steps > 0:
        x = rb.lm_pose_steps(qs_local.view(n, d), problem.target_path, n_steps=int(lm_steps), packed_out=packed, **LM_POSE)["x"]
    else:
        x = qs_local.view(n, d)
        m = rb.collision_masks(qs_local)
        cost, pe, re, sm, em, jm = unpack_rows(packed, n)
        pe_, re_ = rb.pose_error_metrics(x, problem.target_path)
        cost.copy_(m["ext_cost"].view(-1)), pe.copy_(pe_), re.copy_(re_)
        sm.copy_(m["self_mask"].view(-1).to(torch.uint8)), em.copy_(m["env_mask"].view(-1).to(torch.uint8))
        jm.copy_(m["jlim_mask"].view(-1).to(torch.uint8))
    g = allgather_seed_outputs(packed, k_local, T, group=group, counts=counts)
    if world > 1:
        q_all = torch.empty((world * n, d), dtype=torch.float32, device=x.device)
        _all_gather_flat(q_all, x.contiguous(), group)
        q_all = drop_padding(q_all.view(world * k_local, T, d), k_local, counts if counts is not None else [k_local] * world)
    else:
        q_all = x.view(k_local, T, d)
    return q_all.contiguous(), g.self_mask, g.env_mask
