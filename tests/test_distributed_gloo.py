"""The N > 1 path on CPU: world_size-2 gloo processes shard the seeds, each fills its packed per-row buffer (with the
oracle standing in for the kernel, there being no GPU here) and ONE all-gather gives every rank all seeds' costs / masks /
errors -- identical to the single-process result (sharding invariance: rows are independent)."""

import os
import socket

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from cppflow_amd import distributed as D
from cppflow_amd.distributed import (PACKED_BYTES_PER_ROW, allgather_seed_outputs, allgather_seed_summaries, drop_padding,
                                     padded_shard_size, seed_shard, shard_counts, unpack_rows)  # fmt: skip
from tests import helpers as H

S_TOTAL, W, K = 6, 16, 3


def _fill_packed(name, x0, target, S):
    """What cppf_lm_pose_steps writes into the packed buffer, computed by the oracle."""
    o64, o32 = H.oracle64(name), H.oracle32(name)
    tgt = H.stacked(target, S)
    x = H.f32(o64.lm_steps(x0, tgt, K))
    pe, re = o64.pose_metrics_exact(x, tgt)
    lo, hi = H.box_corners([c for c, _ in H.PANDA_2CUBES], [T for _, T in H.PANDA_2CUBES])
    ch = H.chain(name)
    m = o32.masks(x, lo, hi, ch.lo, ch.hi)
    n = S * W
    packed = torch.zeros(PACKED_BYTES_PER_ROW * n, dtype=torch.uint8)
    cost, p, r, sm, em, jm = unpack_rows(packed, n)
    cost.copy_(torch.tensor(m["ext_cost"], dtype=torch.float32))
    p.copy_(torch.tensor(pe, dtype=torch.float32))
    r.copy_(torch.tensor(re, dtype=torch.float32))
    sm.copy_(torch.tensor(m["self_mask"]))
    em.copy_(torch.tensor(m["env_mask"]))
    jm.copy_(torch.tensor(m["jlim_mask"]))
    return packed


def _worker(rank, world, port, out_dir):
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        x0, target = H.lm_problem("panda", S_TOTAL, W, seed=21)
        b, e = seed_shard(S_TOTAL, rank, world)
        packed = _fill_packed("panda", x0[b * W : e * W], target, e - b)
        g = allgather_seed_outputs(packed, e - b, W)
        # the per-step payload: 8 floats per seed, gathered asynchronously
        cost, pe, re, sm, em, jm = unpack_rows(packed, (e - b) * W)
        summary = torch.stack([100 * pe.view(e - b, W).amax(1), torch.rad2deg(re.view(e - b, W).amax(1)),
                               torch.zeros(e - b), torch.zeros(e - b), sm.view(e - b, W).sum(1).float(),
                               em.view(e - b, W).sum(1).float(), jm.view(e - b, W).sum(1).float(),
                               cost.view(e - b, W).sum(1)], dim=1).contiguous()  # fmt: skip
        gathered, work = allgather_seed_summaries(summary, async_op=True)
        work.wait()
        np.save(os.path.join(out_dir, f"summary_rank{rank}.npy"), gathered.numpy())
        np.savez(os.path.join(out_dir, f"rank{rank}.npz"), cost=g.ext_cost.numpy(), pe=g.pos_err_m.numpy(),
                 re=g.rot_err_rad.numpy(), sm=g.self_mask.numpy(), em=g.env_mask.numpy(), jm=g.jlim_mask.numpy())  # fmt: skip
    finally:
        dist.destroy_process_group()


def test_two_rank_allgather_equals_single_process(tmp_path):
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    mp.spawn(_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    x0, target = H.lm_problem("panda", S_TOTAL, W, seed=21)
    full = _fill_packed("panda", x0, target, S_TOTAL)
    cost, pe, re, sm, em, jm = (t.numpy().reshape(S_TOTAL, W) for t in unpack_rows(full, S_TOTAL * W))
    for rank in range(2):
        z = np.load(os.path.join(str(tmp_path), f"rank{rank}.npz"))
        assert np.array_equal(z["cost"], cost) and np.array_equal(z["pe"], pe) and np.array_equal(z["re"], re)
        assert np.array_equal(z["sm"], sm.astype(bool)) and np.array_equal(z["em"], em.astype(bool))
        assert np.array_equal(z["jm"], jm.astype(bool))
    assert cost.max() >= 1000.0  # the case does contain collisions
    s0, s1 = (np.load(os.path.join(str(tmp_path), f"summary_rank{r}.npy")) for r in range(2))
    assert s0.shape == (S_TOTAL, 8) and np.array_equal(s0, s1)
    np.testing.assert_allclose(s0[:, 7], cost.sum(1), rtol=1e-6)
    assert np.array_equal(s0[:, 4], sm.sum(1)) and np.array_equal(s0[:, 5], em.sum(1))


# ---- uneven shards: 7 seeds over 2 ranks (4 + 3), W = 6 -> buffers padded to 4 seeds each ------------------------------------
S_ODD, W_ODD = 7, 6


def _fake_packed(S, W, first_seed):
    """a packed buffer whose every field encodes (global seed, waypoint)"""
    n = S * W
    packed = torch.zeros(PACKED_BYTES_PER_ROW * n, dtype=torch.uint8)
    cost, p, r, sm, em, jm = unpack_rows(packed, n)
    seed = (first_seed + torch.arange(S)).repeat_interleave(W)
    w = torch.arange(W).repeat(S)
    cost.copy_((1000 * seed + w).float())
    p.copy_((seed + 0.25).float())
    r.copy_((w + 0.5).float())
    sm.copy_((seed % 2).to(torch.uint8))
    em.copy_((w % 2).to(torch.uint8))
    jm.copy_(((seed + w) % 2).to(torch.uint8))
    return packed


def _worker_uneven(rank, world, port, out_dir):
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        b, e = seed_shard(S_ODD, rank, world)
        S_pad = padded_shard_size(S_ODD, W_ODD, world)
        counts = shard_counts(S_ODD, world)
        assert counts[rank] == e - b and (S_pad * W_ODD) % 4 == 0
        packed = torch.zeros(PACKED_BYTES_PER_ROW * S_pad * W_ODD, dtype=torch.uint8)
        own = _fake_packed(e - b, W_ODD, b)
        for dst, src in zip(unpack_rows(packed, S_pad * W_ODD), unpack_rows(own, (e - b) * W_ODD)):
            dst[: (e - b) * W_ODD].copy_(src)  # the filler seeds stay zero
        g = allgather_seed_outputs(packed, S_pad, W_ODD, counts=counts)
        summ = torch.full((S_pad, 8), -1.0)
        summ[: e - b, 0] = torch.arange(b, e).float()
        allsumm = drop_padding(allgather_seed_summaries(summ), S_pad, counts)
        np.savez(os.path.join(out_dir, f"odd{rank}.npz"), cost=g.ext_cost.numpy(), pe=g.pos_err_m.numpy(), sm=g.self_mask.numpy(),
                 jm=g.jlim_mask.numpy(), summ=allsumm.numpy())  # fmt: skip
    finally:
        dist.destroy_process_group()


def test_uneven_shards_are_padded_and_the_filler_dropped(tmp_path):
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    mp.spawn(_worker_uneven, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    cost, pe, _, sm, _, jm = (t.numpy().reshape(S_ODD, W_ODD) for t in unpack_rows(_fake_packed(S_ODD, W_ODD, 0), S_ODD * W_ODD))
    for rank in range(2):
        z = np.load(os.path.join(str(tmp_path), f"odd{rank}.npz"))
        assert z["cost"].shape == (S_ODD, W_ODD)
        assert np.array_equal(z["cost"], cost) and np.array_equal(z["pe"], pe)
        assert np.array_equal(z["sm"], sm.astype(bool)) and np.array_equal(z["jm"], jm.astype(bool))
        assert np.array_equal(z["summ"][:, 0], np.arange(S_ODD))


def test_unaligned_equal_shards_are_refused_with_a_clear_message():
    # 3 seeds x 5 waypoints per rank = 15 rows: rank 1's fp32 slice would start at byte 225
    import pytest

    packed = torch.zeros(PACKED_BYTES_PER_ROW * 15, dtype=torch.uint8)
    assert padded_shard_size(6, 5, 2) == 4
    # single process: world == 1 never trips the check
    allgather_seed_outputs(packed, 3, 5)
    with pytest.raises(AssertionError):
        allgather_seed_outputs(packed, 3, 5, counts=[3, 3])


# ---- the package's own engine (cppflow_amd.distributed.ShardedRefiner) on two gloo ranks ---------------------------------------------
# No GPU here, so the four device-facing pieces of the class are replaced by CPU stand-ins -- the ORACLE computes what a fused launch
# writes (x_out, the packed per-row buffer, the [S,8] summary), the seed selection is evaluated on the host with the package's own
# threshold rule, streams are absent -- while everything the N > 1 path adds runs as shipped: the launch ring and its buckets,
# batched launches, the in-"stream" exchange through `C10dAllGather` (gloo), the gathered selection's indexing, `gather_and_search`.
import contextlib  # noqa: E402


class _OracleLaunch:
    def __init__(self, refiner, slots):
        self.r, self.slots = refiner, slots
        self.outputs = [dict(x=refiner.x_outs[b]) for b in slots]

    def launch_on(self, stream):
        r = self.r
        for b in self.slots:
            packed, x = _fill_packed_and_x("panda", r.x0.numpy().astype(np.float64), r.target.numpy().astype(np.float64), r.S, r.K)
            r.packeds[b].copy_(packed)
            r.x_outs[b].copy_(torch.tensor(x, dtype=torch.float32))
            cost, pe, re, sm, em, jm = unpack_rows(r.packeds[b], r.n)
            S, Wl = r.S, r.W
            val = H.oracle64("panda").seed_validity(x, H.stacked(r.target.numpy().astype(np.float64), S), S, Wl)
            r.summ_all[b].copy_(torch.cat([torch.tensor(val, dtype=torch.float32), sm.view(S, Wl).sum(1, keepdim=True).float(),
                                           em.view(S, Wl).sum(1, keepdim=True).float(), jm.view(S, Wl).sum(1, keepdim=True).float(),
                                           cost.view(S, Wl).sum(1, keepdim=True)], dim=1))
            r.log.append(("launch", b))

    launch = launch_on


def _fill_packed_and_x(name, x0, target, S, K_steps):
    o64, o32 = H.oracle64(name), H.oracle32(name)
    tgt = H.stacked(target, S)
    x = H.f32(o64.lm_steps(x0, tgt, K_steps))
    pe, re = o64.pose_metrics_exact(x, tgt)
    lo, hi = H.box_corners([c for c, _ in H.PANDA_2CUBES], [T for _, T in H.PANDA_2CUBES])
    ch = H.chain(name)
    m = o32.masks(x, lo, hi, ch.lo, ch.hi)
    n = x.shape[0]
    packed = torch.zeros(PACKED_BYTES_PER_ROW * n, dtype=torch.uint8)
    for dst, src in zip(unpack_rows(packed, n), (m["ext_cost"], pe, re, m["self_mask"], m["env_mask"], m["jlim_mask"])):
        dst.copy_(torch.tensor(src, dtype=dst.dtype))
    return packed, x


def _host_selection(constraints, gathered):
    """cppf_select_valid_seed_gathered on the host: gathered [world, G, S, 8] -> [G, 4]; seed index = rank * S + s"""
    from cppflow_amd.evaluation_utils import seed_metrics_are_below_threshold

    world, G, S, _ = gathered.shape
    out = torch.zeros((G, 4), dtype=torch.int32)
    for g in range(G):
        rows = gathered[:, g].reshape(world * S, 8)
        valid = [i for i in range(world * S) if seed_metrics_are_below_threshold(constraints, rows[i, :4])[0] and rows[i, 4] == 0 and rows[i, 5] == 0]
        out[g] = torch.tensor([valid[0] if valid else -1, len(valid), int(torch.argmin(rows[:, 7])), 0])
    return out


class _CpuRefiner(D.ShardedRefiner):
    def _make_launches(self):
        self.log = []
        B = self.B
        self.launches = [[_OracleLaunch(self, list(range(g * B, g * B + c))) for c in range(1, B + 1)] for g in range(self.NBUF // B)]

    def _make_streams(self, count):
        return [f"s{i}" for i in range(count)]

    def _on(self, stream):
        return contextlib.nullcontext()

    def _select(self, gathered, out):
        out.copy_(_host_selection(self.constraints, gathered))
        self.log.append(("exchange",))

    def synchronize(self):
        pass

    def _dp_search(self, q_all, cost_all):
        idx, _ = H.oracle64("panda").dp_search(q_all.numpy().astype(np.float64), cost_all.numpy().astype(np.float64))
        path = q_all[torch.tensor(idx, dtype=torch.long), torch.arange(q_all.shape[1])]
        return path, torch.tensor(idx), None


S_ENG, W_ENG, K_ENG = 8, 8, 2


def _loose_constraints():
    from cppflow_amd.data_types import Constraints

    return Constraints(max_allowed_position_error_cm=5.0, max_allowed_rotation_error_deg=10.0, max_allowed_mjac_deg=400.0, max_allowed_mjac_cm=100.0)


def _worker_engine(rank, world, port, out_dir):
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        x0, target = H.lm_problem("panda", S_ENG, W_ENG, seed=35)
        b, e = seed_shard(S_ENG, rank, world)
        x_local = torch.tensor(x0[b * W_ENG : e * W_ENG], dtype=torch.float32)
        B, G, _, n_streams = D.launch_plan(x_local.shape[0], 5, batch=2, gather_every=2, streams=2)
        r = _CpuRefiner(None, x_local, torch.tensor(target, dtype=torch.float32), K_ENG, transport=D.C10dAllGather(), batch=B, bucket=G,
                        n_streams=n_streams, constraints=_loose_constraints())
        assert (r.B, r.G, r.NBUF, r.world) == (2, 2, 4, world)
        r.run_region(5)  # launches of 2 / 2 / 1 steps; buckets 0, 1 complete, bucket 0 again partly filled and drained
        assert sum(1 for ev in r.log if ev[0] == "exchange") == 3 and r.step_no % r.G == 0
        lat = r.allgather_latency_us(3)
        assert lat is not None and lat > 0
        path, idx = r.gather_and_search(0)
        q_all, g = r.gather_candidates(0)
        np.savez(os.path.join(out_dir, f"engine{rank}.npz"), sel0=r.selected[0].numpy(), sel1=r.selected[1].numpy(), path=path.numpy(), idx=idx.numpy(),
                 q_all=q_all.numpy(), cost=g.ext_cost.numpy(), gathered=r.gathered[1].numpy())
    finally:
        dist.destroy_process_group()


def test_sharded_refiner_on_two_gloo_ranks_equals_the_single_process_result(tmp_path):
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    mp.spawn(_worker_engine, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    # the single-process answer: all seeds in one refiner without a transport (world = 1), the same stand-ins
    x0, target = H.lm_problem("panda", S_ENG, W_ENG, seed=35)
    one = _CpuRefiner(None, torch.tensor(x0, dtype=torch.float32), torch.tensor(target, dtype=torch.float32), K_ENG, transport=D.LocalAllGather(),
                      batch=1, bucket=1, n_streams=1, constraints=_loose_constraints())
    one.run_region(1)
    want_sel = one.selected[0][0].numpy()
    want_path, want_idx = one.gather_and_search(0)
    z = [np.load(os.path.join(str(tmp_path), f"engine{r}.npz")) for r in range(2)]
    for zr in z:
        # every step of every bucket selected the same seed as the single process over the same 8 seeds (seed index = rank * S_local + s)
        assert (zr["sel0"] == want_sel[None]).all() and (zr["sel1"] == want_sel[None]).all(), (zr["sel0"], want_sel)
        assert zr["gathered"].shape == (2, 2, S_ENG // 2, 8)
        assert np.array_equal(zr["q_all"].reshape(S_ENG * W_ENG, -1), one.x_outs[0].numpy())
        assert np.array_equal(zr["cost"], unpack_rows(one.packeds[0], S_ENG * W_ENG)[0].numpy().reshape(S_ENG, W_ENG))
        assert np.array_equal(zr["idx"], want_idx.numpy()) and np.array_equal(zr["path"], want_path.numpy())
    assert 1 <= want_sel[1] < S_ENG  # the case does contain valid seeds (the selection is not vacuous)
