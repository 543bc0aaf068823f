"""The build's own robot models against the reference's problem set (CPU).

jrl is absent, so no capsule of the reference can be compared with.  What CAN be checked: the reference ships 18 planning problems
(cppflow/problems/*.yaml -> tests/golden/reference_problems.npz, data only), 13 of which its README names as problems it solves, and
`cppflow/planners.py:237, 247` can only make progress where collision-free IK solutions exist.  scripts/problem_plausibility.py solves
IK for every waypoint with the CPU oracle and records which solutions the capsule model flags; the table is committed
(tests/golden/problem_plausibility.json) and this test re-derives it from other random starts and holds both to the
bars: every waypoint reachable inside the joint limits, a collision-free solution for >= 95 % of the waypoints of every problem."""

import json
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(ROOT, "tests", "golden")
sys.path.insert(0, os.path.join(ROOT, "scripts"))

# /root/reference/README.md:26-38 (the problems its evaluate.py accepts) + the two `_mini` problems its own tests plan on
# (tests/planners_test.py); panda__square and the *_rot_yz2 variants are in the directory but in neither list
SOLVED_BY_THE_REFERENCE = ["fetch__circle", "fetch__hello", "fetch__rot_yz", "fetch__s", "fetch__square", "fetch_arm__circle",
                           "fetch_arm__hello", "fetch_arm__rot_yz", "fetch_arm__s", "fetch_arm__square", "panda__flappy_bird",
                           "panda__2cubes", "panda__1cube", "fetch_arm__hello_mini", "panda__1cube_mini"]  # fmt: skip


def test_committed_table_meets_the_bars_and_covers_every_problem():
    table = json.load(open(os.path.join(GOLDEN, "problem_plausibility.json")))["problems"]
    z = np.load(os.path.join(GOLDEN, "reference_problems.npz"))
    assert sorted(table) == sorted(str(n) for n in z["names"]) and len(table) == 18
    for name in SOLVED_BY_THE_REFERENCE:
        rec = table[name]
        assert rec["reachable_frac"] == 1.0, (name, rec["reachable_frac"])
        assert rec["free_frac"] >= 0.95, (name, rec["free_frac"])
    # the symptom VERDICT r3 named: 42 % of the IK solutions on fetch__hello were flagged self-colliding
    for name, rec in table.items():
        if rec["robot"] != "panda":
            assert rec["solutions_self_colliding_frac"] <= 0.10, (name, rec["solutions_self_colliding_frac"])


@pytest.mark.parametrize("name", SOLVED_BY_THE_REFERENCE)
def test_reference_problems_have_collision_free_ik_solutions(name):
    import problem_plausibility as pp

    z = np.load(os.path.join(GOLDEN, "reference_problems.npz"))
    rec = pp.solve_problem(name, str(z[name + "__robot"]), z[name + "__target_path"], z[name + "__obstacles"], restarts=48, seed=3)
    assert rec["reachable_frac"] >= 0.995, rec  # (other random starts than the committed table's: a waypoint in 300 may be missed)
    assert rec["free_frac"] >= 0.95, rec
    committed = json.load(open(os.path.join(GOLDEN, "problem_plausibility.json")))["problems"][name]
    # the committed table was made with other random starts: the shares must agree to sampling noise
    for k in ("solutions_self_colliding_frac", "solutions_env_colliding_frac"):
        assert abs(rec[k] - committed[k]) <= 0.08, (name, k, rec[k], committed[k])


def test_random_configurations_are_mostly_self_collision_free():
    """uniformly random configurations: what dp_search's candidates look like before refinement (each flagged cell is priced 1000,
    cppflow/search.py:14-15).  Round 3: 44 % (Fetch) / 50 % (FetchArm) / 3 % (Panda)."""
    from tests import helpers as H

    for name, bar in (("panda", 0.06), ("fetch", 0.16), ("fetch_arm", 0.25)):
        o = H.oracle64(name)
        x = H.random_configs(name, 20000, seed=1)
        frac = float((o.self_dists(x) < 0).any(axis=1).mean())
        assert frac <= bar, (name, frac)
