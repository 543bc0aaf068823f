"""Shape / value fuzz of the fused launch and its two neighbour stages against the oracle (pytest -m gpu): the sweeps live in scripts/fuzz_dp.py,
scripts/fuzz_coupled.py and scripts/fuzz_lm.py so that they can be run and extended by hand; here they must report no disagreement."""
import os
import runpy

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("script", ["fuzz_dp.py", "fuzz_coupled.py", "fuzz_lm.py", "fuzz_masks.py"])
def test_fuzz_sweep_reports_no_disagreement(script, capsys):
    """fuzz_dp: cppf_dp_search in its table / resident / per-waypoint forms over k = 1 .. 300, T = 1 .. 257 with ties, +inf columns,
    1e33 costs and identical candidates -- cost table, argmins and path bit-identical to the oracle (920 comparisons).
    fuzz_coupled: cppf_lm_full_step in its three elimination orders over T = 1 .. 65, S = 1 .. 3, 0 .. 4 virtual configurations,
    with and without obstacles, with the differencing "satisfied" options (filter / scale-down / scale-down + shift), three robots (2349
    comparisons; where an option's threshold falls within rounding of a joint change the reference's dense formulation on the mirror's
    matrices arbitrates).
    fuzz_lm: the fused launch over four robots x eleven (S, W) shapes (W = 1 .. 300, ragged and not) x K in {1, 3, 10} x kernel
    shape x solver: x against the oracle, per-row outputs at the launch's own x (masks bit-exact), the per-seed summary against the
    separate reduction, the two kernel shapes against each other (792 launches).
    fuzz_masks: the collision stage, standalone and fused, over obstacle sets from none to the maximum (thin plates, rods, points,
    a cuboid around the base, one far away) and nine robots (shipped, random generic, random run-time-specialised, with spheres among
    their capsules): masks, cost and signed minimum distances bit for bit (216 cases x 12 arrays)."""
    with pytest.raises(SystemExit) as e:
        runpy.run_path(os.path.join(ROOT, "scripts", script), run_name="__main__")
    out = capsys.readouterr().out
    assert e.value.code == 0 and "disagreements: 0" in out, out[-2000:]
    assert int(out.split("comparisons:")[1].split()[0]) > 200
