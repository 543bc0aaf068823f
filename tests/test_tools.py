"""The measurement tooling behind the committed records (CPU): the opcode census finds the LM iteration of the headline kernel, and the
dispatch-timeline summary finds the timed regions of a kernel trace."""

import os
import shutil
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "scripts"))


@pytest.mark.skipif(shutil.which("hipcc") is None, reason="needs hipcc (cross-compiles gfx950 without a GPU)")
def test_isa_mix_finds_the_lm_iteration_of_the_headline_kernel():
    import isa_mix

    path = isa_mix.compile_dev()  # scripts/fused_dev.hip: lm_fused_kernel<StaRobot<Panda>, 1> with fused_static.hip's flags, ~5 s
    kernels = isa_mix.parse(path, "")
    (name, blocks), = [(n, b) for n, b in kernels.items() if "lm_fused_kernel" in n]
    regs = isa_mix.regions(blocks)
    loops = [r for r in regs if len(r) == 4]
    assert len(loops) >= 2  # the lean loop and the general (last / early-out) iteration's loop at least
    lean = [r for r in loops if isa_mix.census(r[1])["vmem"] == 0 and isa_mix.census(r[1])["lds"] == 0]
    hot = max(lean, key=lambda r: isa_mix.census(r[1])["valu"])
    c = isa_mix.census(hot[1])
    # the likely path of one lean LM iteration (update and clamp included): ~590 VALU, >= 90 % of it multiply-adds, multiplies and adds,
    # at most five transcendentals (three block pivots, one shared reciprocal of the angle functions, none for asin on the fast path),
    # no fp64, no LDS
    assert 500 <= c["valu"] <= 700, c["valu"]
    assert (c["fma"] + c["mul"] + c["addsub"]) / c["valu"] >= 0.9 and c["fp64"] == 0 and c["mfma"] == 0 and c["trans"] <= 5, dict(c)
    whole = isa_mix.census([x for b in blocks for x in b["ins"]])
    assert whole["valu"] > 5000 and whole["fp64"] > 0  # the gate's double-precision rounds live behind the rare branches


def test_overlap_summary_separates_overlapped_regions_from_isolated_launches():
    import overlap_summary as o

    rows, t = [], 0
    for _ in range(3):  # three timed regions of 20 dispatches, two in flight, a synchronise (idle) between them
        for i in range(20):
            rows.append((t, t + (50_000 if i == 0 else 70_000), str(i % 2), ""))
            t += 36_000
        t += 80_000
    for _ in range(50):  # isolated launches back to back on one queue: ~1 us apart, never two resident
        rows.append((t, t + 48_000, "0", ""))
        t += 49_000
    cl = o.clusters(sorted(rows))
    timed = [o.region_stats(c) for c in cl if len(c) == 20]
    assert len(timed) == 3 and all(r["max_resident"] == 2 and r["queues"] == ["0", "1"] for r in timed)
    assert all(abs(r["union_busy_us"] / 20 - 37.7) < 1.0 and r["two_or_more_resident_frac"] > 0.8 for r in timed)
    alone = [c for c in cl if o.region_stats(c)["max_resident"] == 1]
    assert sum(len(c) for c in alone) == 50
