"""cppflow_amd.distributed.ShardedRefiner.run_steps / drain without a GPU: the ORDER in which a rank issues its launches and
exchange steps -- what the N > 1 path rests on and what no one-GPU box can show on RCCL.  The refiner is built around stand-ins (no
robot, no streams): launches and exchanges only record themselves."""

import contextlib

import pytest

from cppflow_amd import distributed as D


class _Launch:
    def __init__(self, log, group, steps):
        self.log, self.group, self.steps = log, group, steps

    def launch_on(self, stream):
        self.log.append(("launch", self.group, self.steps, stream))


def _runner(B, G, n_streams, transport=True):
    r = object.__new__(D.ShardedRefiner)
    r.B, r.G, r.n_streams = B, G, n_streams
    r.buckets = transport
    r.transport = object() if transport else None
    r.NBUF = n_streams * G if transport else max(4, n_streams) * B
    r.graphs = None
    r.step_no = 0
    r.start_bucket = None
    r._on = lambda stream: contextlib.nullcontext()
    r.streams = [f"s{i}" for i in range(n_streams)]
    r.log = []
    r.launches = [[_Launch(r.log, g, c + 1) for c in range(B)] for g in range(r.NBUF // B)]
    r.exchange = lambda bucket: r.log.append(("exchange", bucket))
    return r


def _steps(log):
    return sum(e[2] for e in log if e[0] == "launch")


@pytest.mark.parametrize("B,G,steps", [(8, 8, 20), (4, 8, 20), (2, 2, 20), (1, 1, 20), (8, 64, 2000), (1, 8, 37), (4, 4, 3)])
def test_exactly_the_requested_steps_and_every_bucket_exchanged_once_behind_its_last_launch(B, G, steps):
    r = _runner(B, G, 2)
    r.run_steps(steps)
    r.drain()
    assert _steps(r.log) == steps
    # on each stream the order is: the bucket's launches, then its exchange, then the next use of that bucket
    last_launch_of = {}
    open_bucket = None
    for i, e in enumerate(r.log):
        if e[0] == "launch":
            bucket = (e[1] * B) // G
            assert e[3] == f"s{bucket}"  # a bucket's launches all go to its own stream
            assert last_launch_of.get(bucket, (None, True))[1], (i, e, "launch into a bucket whose exchange has not been issued")
            last_launch_of[bucket] = (i, (e[1] * B + B) % G != 0)  # second item: bucket still filling
            open_bucket = bucket
        else:
            assert e[1] in last_launch_of, (i, e)
            last_launch_of[e[1]] = (last_launch_of[e[1]][0], True)
    # nothing is left unexchanged when the region ends (the clock stops after the partly filled bucket's exchange too)
    n_launched_buckets = len({(e[1] * B) // G for e in r.log if e[0] == "launch"})
    assert len([e for e in r.log if e[0] == "exchange"]) >= n_launched_buckets
    assert r.log[-1][0] == "exchange"
    assert r.step_no % G == 0  # the next region starts a fresh bucket


def test_an_exchange_is_issued_one_launch_late_but_never_behind_a_launch_into_its_own_bucket():
    r = _runner(8, 8, 2)
    r.run_steps(20)  # launches of 8 / 8 / 4 steps: buckets 0, 1, 0
    kinds = [(e[0], e[1] if e[0] == "exchange" else (e[1] * 8) // 8 % 2) for e in r.log]
    # bucket 0's exchange comes AFTER bucket 1's launch (the second stream's kernels are not held back by its host calls) and BEFORE
    # the third launch, which goes into bucket 0 again (it would overwrite the summaries still to be gathered); bucket 1's exchange
    # in turn waits for that third launch to be out
    assert kinds == [("launch", 0), ("launch", 1), ("exchange", 0), ("launch", 0), ("exchange", 1), ("exchange", 0)], kinds
    one = _runner(4, 4, 1)  # a one-bucket ring: every exchange right behind its launch
    one.run_steps(12)
    assert [e[0] for e in one.log] == ["launch", "exchange"] * 3


def test_without_a_transport_consecutive_launches_alternate_between_the_streams():
    r = _runner(1, 1, 2, transport=False)
    r.run_steps(6)
    r.drain()
    assert [e[3] for e in r.log] == ["s0", "s1", "s0", "s1", "s0", "s1"] and all(e[0] == "launch" for e in r.log)


def test_launch_plan_for_the_driver_s_flags_and_for_long_runs():
    """D.launch_plan: steps per launch / per collective / streams.  With the driver's flags (--steps 20) the choices measured on the
    one-GPU boxes (profiles/r4_short_region_buckets.txt, r4_driverflags_by_shard.txt): every rank issues full-width launches (1 / 2 /
    4 / 8 steps of its 1024 / 512 / 256 / 128 seeds), one launch per bucket -- two for the 256-seed shard -- on two streams; over
    2 000 steps a bucket is 8 ... 64 steps.  Explicit arguments win."""
    W = 256
    assert [D.launch_plan(S * W, 20)[:2] + D.launch_plan(S * W, 20)[3:] for S in (1024, 512, 256, 128)] == [(1, 1, 2), (2, 2, 2), (4, 8, 2), (8, 8, 2)]
    assert [D.launch_plan(S * W, 2000)[:2] for S in (1024, 512, 256, 128)] == [(1, 8), (2, 8), (4, 32), (8, 64)]
    assert D.launch_plan(128 * 64, 2000)[:2] == (16, 64) and D.launch_plan(128 * 64, 2000)[3] == 4  # C2: 16 steps per launch, half width
    assert D.launch_plan(128 * W, 20, batch=4, gather_every=4, streams=3) == (4, 4, 4, 3)
    assert D.launch_plan(128 * 64, 2000, quad=True)[0] == 1  # the quad shape keeps one step per launch
    for S in (1024, 512, 256, 128, 64, 1):  # a bucket is a whole number of launches, never longer than the region allows
        for steps in (1, 5, 20, 64, 2000):
            b, G, _, _ = D.launch_plan(S * W, steps)
            assert G % b == 0 and G >= b
