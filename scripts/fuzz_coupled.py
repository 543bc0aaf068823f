"""Developer fuzz: cppf_lm_full_step against the oracle's dense restatement over tiny and awkward shapes (T = 1 .. 9, S = 1 .. 3,
with and without virtual configurations / obstacles), every elimination order.  Differencing preset (no pose block: the blocks are
well conditioned, so joint-space agreement is tight).  Prints every disagreement; exits 1 if any."""
import sys, os
import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tests import helpers as H  # noqa: E402
from cppflow_amd.lm_hyper_parameters import ALT_LOSS_V2_1_DIFF, OptimizationParameters  # noqa: E402
from cppflow_amd.robots import get_robot  # noqa: E402

DEV = "cuda:0"


def dev(a):
    return torch.tensor(np.asarray(a), dtype=torch.float32, device=DEV)


bad = checked = boundary = 0
from cppflow_amd.data_types import Constraints  # noqa: E402

# thresholds in the middle of the joint changes of these trajectories (0.03 rad per waypoint ~ 1.7 deg), so that rows fall on both sides
CONS = Constraints(max_allowed_position_error_cm=1.5, max_allowed_rotation_error_deg=0.02, max_allowed_mjac_deg=2.0, max_allowed_mjac_cm=1.5)
for name in ("panda", "fetch", "chain12"):
    rb, orc, ch = get_robot(name), H.oracle64(name), H.chain(name)
    d = rb.ndof
    rng = np.random.RandomState(3)
    obs = H.PANDA_2CUBES
    lo, hi = H.box_corners([c for c, _ in obs], [T_ for _, T_ in obs])
    for T in (1, 2, 3, 4, 5, 9, 17, 65):
        for S in (1, 2, 3):
            for nvc in (0, 1, 2, 4):
                if nvc and 2 * nvc >= T:
                    continue
                for with_obs, option in ((False, None), (True, None), (True, "diff_filter"), (False, "diff_scale"), (True, "diff_scale_shift")):
                    if option and (T < 3 or nvc == 0 and option != "diff_filter"):
                        continue  # (the scale-down options want virtual configurations: lm_hyper_parameters warns otherwise)
                    rb.set_obstacles([c for c, _ in obs] if with_obs else [], [T_ for _, T_ in obs] if with_obs else [])
                    base = np.clip(H.random_configs(name, 1, seed=T * 7 + S) + np.cumsum(0.03 * rng.randn(T, d), axis=0), ch.lo, ch.hi)
                    x = H.f32(np.clip(base[None] + 0.01 * rng.randn(S, T, d), ch.lo, ch.hi).reshape(S * T, d))
                    target = H.f32(orc.fk(H.f32(base)))
                    kw = dict(ALT_LOSS_V2_1_DIFF.__dict__)
                    kw.update(use_virtual_configs=bool(nvc), n_virtual_configs=nvc if nvc else None,
                              use_env_collisions=with_obs, use_differencing=T > 1 or not nvc)
                    if T == 1:
                        kw.update(use_differencing=True)
                    if option == "diff_filter":
                        kw.update(differencing_do_ignore_satisfied=True, differencing_ignore_satisfied_margin_deg=0.5, differencing_ignore_satisfied_margin_cm=0.5)
                    elif option:
                        kw.update(differencing_do_scale_satisfied=True, differencing_scale_down_satisfied_scale=0.4,
                                  differencing_scale_down_satisfied_shift_invalid_to_threshold=option == "diff_scale_shift",
                                  differencing_ignore_satisfied_margin_deg=0.5, differencing_ignore_satisfied_margin_cm=0.5)
                    try:
                        pm = OptimizationParameters(**kw)
                    except AssertionError:
                        continue
                    xv = H.f32(x + 0.01 * rng.randn(*x.shape)) if nvc else None
                    pm.virtual_configs = dev(xv) if nvc else torch.tensor([])
                    try:
                        want = orc.lm_full_step(x, target, pm, S, T, virtual_configs=xv, boxes_lo=lo if with_obs else None, boxes_hi=hi if with_obs else None,
                                                constraints=CONS if option else None)
                    except Exception as e:  # noqa: BLE001
                        print("ORACLE EXC", name, T, S, nvc, with_obs, repr(e)[:150]); continue
                    step = np.abs(want - x).max()
                    for order, sets in (("default", {}), ("sequential", {"pcr_max_rows": 0}), ("per_wave", {"pcr_max_rows": 0, "full_rows": 0})):
                        for k_, v_ in sets.items():
                            rb.debug_set(k_, v_)
                        try:
                            got = rb.lm_full_step(dev(x), dev(target), pm, virtual_configs=pm.virtual_configs,
                                                  constraints=CONS if option else None).cpu().numpy().astype(np.float64)
                            torch.cuda.synchronize()
                        except Exception as e:  # noqa: BLE001
                            print("EXC", name, "T", T, "S", S, "nvc", nvc, "obs", with_obs, option, order, repr(e)[:200]); bad += 1; continue
                        finally:
                            for k_ in sets:
                                rb.debug_set(k_, None)
                        checked += 1
                        err = np.abs(got - want).max()
                        if not np.isfinite(got).all() or err > 2e-4 + 2e-3 * step:
                            if option:
                                # An option turns rows on and off at thresholds, and the device evaluates them on fp32 kinematics, the
                                # oracle on fp64: a joint change within rounding of its threshold is a row on one side only.  The
                                # arbiter is then the reference's own dense formulation on the MIRROR's matrices (get_r_and_J over
                                # the device's per-row quantities): J^T J + lambda I solved in fp64, trajectory by trajectory.
                                from cppflow_amd.optimization_utils import LmResidualFns

                                worst = 0.0
                                for s_ in range(S):
                                    pm0 = OptimizationParameters(**kw)
                                    pm0.virtual_configs = dev(xv[s_ * T : (s_ + 1) * T]) if nvc else torch.tensor([])
                                    Tc = [torch.tensor(T_) for _, T_ in obs] if with_obs else None
                                    cub = [torch.tensor(c) for c, _ in obs] if with_obs else None
                                    Jm, rm = LmResidualFns.get_r_and_J(pm0, rb, dev(x[s_ * T : (s_ + 1) * T]), dev(target), Tcuboids=Tc, cuboids=cub, constraints=CONS)
                                    Jd, rd = Jm.get_J().double().cpu().numpy(), rm.get_r().double().cpu().numpy()[:, 0]
                                    dense = np.linalg.solve(Jd.T @ Jd + pm.lm_lambda * np.eye(Jd.shape[1]), Jd.T @ rd)
                                    worst = max(worst, np.abs(dense - (got[s_ * T : (s_ + 1) * T] - x[s_ * T : (s_ + 1) * T]).reshape(-1)).max())
                                if worst <= 2e-4 + 2e-3 * step:
                                    boundary += 1
                                    continue
                                print("   device vs the dense formulation on the mirror's matrices: %.3e" % worst)
                            bad += 1
                            print("DIFF", name, "T", T, "S", S, "nvc", nvc, "obs", with_obs, option, order, "err %.3e" % err, "step %.3e" % step)
    rb.set_obstacles([], [])
print("comparisons:", checked, " disagreements:", bad, " (threshold-boundary cases settled by the dense formulation:", boundary, ")")
sys.exit(1 if bad else 0)
