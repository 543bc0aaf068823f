"""bench.py itself on the GPU (pytest -m gpu): the JSON line of the driver's command, and one rank's N = 8 shard with the exchange step of
a one-rank RCCL group on -- the only piece of the N > 1 path a one-GPU box can run on the real transport.  Each run is a child process
(bench.py initialises the GPU itself; nothing of it may already live in the test process)."""

import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench(args, env=None, timeout=300):
    e = dict(os.environ)
    e.update(env or {})
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, capture_output=True, text=True, timeout=timeout, env=e, cwd=ROOT)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, lines  # exactly ONE JSON line on stdout
    return json.loads(lines[0])


def test_the_driver_s_command_prints_the_contract_s_line():
    """`python bench.py --gpus 1 --steps 20 --warmup 5` (the round driver's command; CPU baselines off here: they take 10 - 30 s)."""
    d = _bench(["--gpus", "1", "--steps", "20", "--warmup", "5", "--no-cpu-baseline", "--no-siblings"])
    assert d["metric"].startswith("LM-IK iterations") and d["unit"] == "LM-IK iterations/s" and d["higher_is_better"] is True
    assert (d["n_gpus"], d["steps"], d["warmup"]) == (1, 20, 5) and d["dtype"] == "f32" and d["data"] == "synthetic" and d["vs_baseline"] is None
    assert "workload" in d["config"] and "model" not in d["config"]
    # value = rows x K x steps / time of exactly 20 steps
    rows = 1024 * 256
    assert abs(d["value"] - rows * 10 / (d["ms_per_step"] * 1e-3)) / d["value"] < 1e-6
    assert 20e-3 < d["ms_per_step"] < 80e-3, d["ms_per_step"]  # 36 us on an MI355X; a silent fall-back would be off by orders of magnitude
    r = d["roofline"]
    assert r["bound"] == "valu" and r["peak"] == 157.3 and r["unit"] == "TFLOP/s" and r["kernel"] == "lm_fused_kernel"
    assert r["library_build_id"] and 0.02 < r["kernel_ms"] < 0.2
    # frac only ever on the executed basis: a number when the committed counters are of this build, else null with the reason
    assert (r["frac"] is None) == (not r["basis"].startswith("executed")), r["basis"]
    if r["frac"] is not None:
        assert 0.2 < r["frac"] < 0.6 and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-9 and r["at_step_rate"]["frac"] < 1.0
    t = d["config"]["timed_region"]
    assert t["repeats"] >= 5 and len(t["ms_per_step_all"]) == t["repeats"] and t["reported"] == "median"
    assert d["config"]["converged_frac_pos_err_lt_1e-4"] > 0.9


def test_one_rank_s_shard_of_eight_with_the_rccl_exchange_on():
    """128 seeds x 256 waypoints = what each of 8 GPUs runs, `--steps 20 --warmup 5`, the exchange step (all-gather of the per-seed
    summaries through RCCL + the seed selection over all ranks' seeds) on the launch streams of a ONE-rank group: 8 steps per launch,
    one launch per bucket, the default pair of streams in the headline and the calibrated pair beside it, the selection equal to a single process's, the region repeated."""
    d = _bench(["--gpus", "1", "--seeds", "128", "--steps", "20", "--warmup", "5", "--no-cpu-baseline", "--no-siblings"], env={"CPPF_BENCH_FORCE_DIST": "1"})
    c = d["config"]
    assert (c["steps_per_launch"], c["steps_per_allgather"], c["streams"]) == (8, 8, 2)
    assert d["selection_check"]["identical_on_every_rank"] and d["selection_check"]["equals_single_process"]
    assert d["rccl"]["world_seen"] == 1
    assert c["engine"].startswith("cppflow_amd.distributed.ShardedRefiner") and "INDEPENDENT" in c["throughput_not_latency"]
    t = c["timed_region"]
    assert t["repeats"] >= 21 and t["closing_barrier_us"] is not None  # a 0.12 ms region: repeated, the closing barrier outside the clock
    # the headline is the DEFAULT stream pair; the calibrated pair (a best-of-30 pick) is reported beside it, never instead of it
    cal = d["calibrated_streams"]["calibration"]
    u = cal["us_per_step"]
    assert cal["candidates"] == 30 and u["best"] <= u["median"] <= u["worst"] and len(set(cal["chosen_streams"])) == 2
    assert d["ms_per_step_calibrated_streams"] == d["calibrated_streams"]["ms_per_step"] and 3e-3 < d["ms_per_step_calibrated_streams"] < 15e-3
    assert d["rccl"]["allgather_latency_us"] > 0.5  # 100 bare [S,8] all-gathers through the transport, per call
    assert 3e-3 < d["ms_per_step"] < 15e-3, d["ms_per_step"]  # 6 - 8 us on an MI355X (5.8 with the calibrated pair)
    assert d["value"] == pytest.approx(128 * 256 * 10 / (d["ms_per_step"] * 1e-3), rel=1e-6)
