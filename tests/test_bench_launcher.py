"""`python bench.py --gpus N` must start its own ranks.  Without a GPU the launcher is driven in its
dry-run mode: two fresh gloo ranks, seed sharding (uneven on purpose), one all-gather, the seed selection, ONE JSON line."""

import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(extra_env, *argv):
    env = dict(os.environ, CPPF_BENCH_DRYRUN="1", **extra_env)
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), *argv], env=env, capture_output=True, text=True,
                          timeout=300)  # fmt: skip


def test_launcher_starts_two_ranks_and_relays_one_json_line():
    r = _run({}, "--gpus", "2", "--steps", "3", "--warmup", "1", "--seeds", "7", "--waypoints", "6")
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, r.stdout
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["config"]["world_size"] == 2 and d["scaling"] == "strong"
    # 7 seeds over 2 ranks: shards of 4 and 3, padded to 4 each (4 * 6 rows % 4 == 0); the filler seed is dropped again
    assert d["config"]["seeds_per_gpu_padded"] == 4
    assert d["config"]["gathered_seed_ids"] == list(range(7)) and d["config"]["max_rank"] == 1
    assert d["config"]["n_valid"] == 7


def test_launcher_at_the_driver_s_largest_world_sizes_with_uneven_shards():
    """World 8 (what the driver's scaling run ends with) and world 7 with 1024 seeds (1024 % 7 != 0: shards of 147 and 146,
    padded to 148 so that the packed per-row buffer of every rank starts 4-byte aligned): every seed arrives exactly once and
    in order, the filler seeds are dropped, every rank computes the same selection, and the JSON carries the records a reader
    needs to check an N > 1 run from the line alone (`rccl.world_seen`, one entry per rank, `selection_check`)."""
    for world, seeds, W in ((8, 1024, 6), (7, 1024, 6)):
        r = _run({}, "--gpus", str(world), "--steps", "2", "--warmup", "1", "--seeds", str(seeds), "--waypoints", str(W))
        assert r.returncode == 0, r.stderr[-2000:]
        lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
        assert len(lines) == 1, r.stdout
        d = json.loads(lines[0])
        assert d["n_gpus"] == world and d["config"]["world_size"] == world
        assert d["config"]["gathered_seed_ids"] == list(range(seeds)) and d["config"]["n_valid"] == seeds
        per = -(-seeds // world)
        while (per * W) % 4:
            per += 1
        assert d["config"]["seeds_per_gpu_padded"] == per and d["config"]["max_rank"] == world - 1
        assert d["rccl"]["world_seen"] == world and sorted(e["rank"] for e in d["rccl"]["ranks"]) == list(range(world))
        sc = d["selection_check"]
        assert len(sc["selected_by_rank"]) == world and sc["identical_on_every_rank"] and sc["equals_single_process"]


def test_launcher_propagates_a_rank_failure():
    r = _run({"CPPF_BENCH_DRYRUN_FAIL_RANK": "1"}, "--gpus", "2", "--seeds", "4", "--waypoints", "4")
    assert r.returncode == 7, (r.returncode, r.stderr[-2000:])


def test_world_size_mismatch_is_refused_before_the_gpu_is_touched():
    env = dict(os.environ, WORLD_SIZE="2", RANK="0", LOCAL_RANK="0")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "4"], env=env, capture_output=True, text=True,
                       timeout=120)  # fmt: skip
    assert r.returncode != 0 and "--gpus 4 but WORLD_SIZE=2" in r.stderr
