"""Developer fuzz: cppf_dp_search (table / resident / per-waypoint forms) against the oracle over small and awkward shapes and
adversarial costs (ties, +inf, huge, identical candidates).  Prints every disagreement; exits 1 if any."""
import sys, os
import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tests import helpers as H  # noqa: E402
from cppflow_amd.robots import get_robot  # noqa: E402

DEV = "cuda:0"
bad = 0
checked = 0
for name in ("panda", "fetch"):
    rb, orc, ch = get_robot(name), H.oracle32(name), H.chain(name)
    d = rb.ndof
    rng = np.random.RandomState(1)
    shapes = [(1, 1), (1, 2), (2, 1), (2, 2), (3, 2), (2, 3), (5, 7), (63, 3), (64, 3), (65, 3), (64, 64), (65, 65), (127, 9), (128, 9),
              (129, 9), (191, 5), (192, 5), (193, 5), (255, 4), (256, 4), (257, 4), (300, 3), (7, 129), (7, 257)]
    for (k, T) in shapes:
        for kind in ("random", "ties", "inf", "huge", "same_q"):
            q = H.f32(rng.uniform(ch.lo, ch.hi, size=(k, T, d)))
            ext = H.f32(rng.choice([0.0, 100.0, 1000.0, 1100.0], size=(k, T)))
            if kind == "ties":
                q = H.f32(np.round(q, 1))
                ext[:] = 0.0
            elif kind == "inf":
                ext[rng.rand(k, T) < 0.3] = np.inf
                if T > 1:
                    ext[:, T // 2] = np.inf  # a waypoint where every candidate is infeasible
            elif kind == "huge":
                ext = H.f32(ext * 1e30)
            elif kind == "same_q":
                q[:] = q[0:1]
            want_idx, want_costs = orc.dp_search(q, ext)
            for method in ("table", "resident", "auto"):
                if method == "table" and (k > 256 or T < 2):
                    continue
                for persistent in ((1, 0) if method == "resident" else (1,)):
                    rb.debug_set("dp_persistent", persistent)
                    try:
                        path, idx, costsT = rb.dp_search(torch.tensor(q, dtype=torch.float32, device=DEV), torch.tensor(ext, dtype=torch.float32, device=DEV), method=method)
                        torch.cuda.synchronize()
                    except Exception as e:  # noqa: BLE001
                        print("EXC", name, k, T, kind, method, persistent, repr(e)[:200]); bad += 1; continue
                    finally:
                        rb.debug_set("dp_persistent", None)
                    checked += 1
                    gi, gc = idx.cpu().numpy(), costsT.cpu().numpy().T.astype(np.float64)
                    same_c = np.array_equal(gc, want_costs) or (np.isnan(gc) == np.isnan(want_costs)).all() and np.array_equal(np.nan_to_num(gc, nan=-1), np.nan_to_num(want_costs, nan=-1))
                    same_i = np.array_equal(gi, want_idx)
                    want_path = q[want_idx, np.arange(T)]
                    same_p = np.array_equal(path.cpu().numpy().astype(np.float64), want_path)
                    if not (same_c and same_i and same_p):
                        bad += 1
                        print("DIFF", name, "k", k, "T", T, kind, method, "persistent", persistent, "costs", same_c, "idx", same_i, "path", same_p,
                              "got idx", gi[:8], "want", want_idx[:8])
print("comparisons:", checked, " disagreements:", bad)
sys.exit(1 if bad else 0)
