"""Seed sharding across the GPUs of one node: one process per GPU, rank g owns a contiguous slab of
seeds, the robot / target path / obstacles are replicated, and ONE all-gather of the packed per-row outputs
(`Robot.PACKED_BYTES_PER_ROW` bytes per (seed, waypoint): search cost, pose errors, the three masks) gives every rank
the `[S_total, W]` cost and mask matrices `dp_search` consumes (cppflow/search.py:146-151).

The collective is `torch.distributed.all_gather_into_tensor` -- RCCL over xGMI with the "nccl" backend on MI355X, gloo
on CPU in the unit tests.  Two payloads:
  * `allgather_seed_summaries`: 8 floats per SEED (`Robot.seed_summary`: the four x_is_valid maxima, collision counts,
    summed cost) -- 32 KB per rank at 1024 seeds, latency-bound, cheap enough for every step and asynchronous;
  * `allgather_seed_outputs`: the full per-row buffer (15 B per row: 3.9 MB per rank at 1024 x 256) when a rank needs the
    whole `[S_total, W]` cost / mask matrices (dp_search over every rank's candidates): once per planning call.
"""

from dataclasses import dataclass
from typing import Optional, Tuple

import torch
import torch.distributed as dist

PACKED_BYTES_PER_ROW = 15


SEED_SUMMARY_FLOATS = 8  # Robot.SEED_SUMMARY_FIELDS


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
    work = dist.all_gather_into_tensor(out, summary.contiguous(), group=group, async_op=async_op)
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
        dist.all_gather_into_tensor(out, packed, group=group)
        gathered = out.view(world, -1)
    parts = [unpack_rows(gathered[r], n) for r in range(world)]

    def cat(i, as_bool=False):
        keep = counts if counts is not None else [seeds_per_rank] * world
        t = torch.cat([p[i].reshape(seeds_per_rank, W)[:c] for p, c in zip(parts, keep)], dim=0)
        return t.view(torch.bool) if as_bool else t

    return GatheredSeedOutputs(cat(0), cat(1), cat(2), cat(3, True), cat(4, True), cat(5, True))
