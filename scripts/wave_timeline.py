"""Developer measurement: when do the wavefronts of ONE isolated headline launch start and end?  Needs a diagnostic build of the
library (build_var/lib_stamp.so: the fused kernel writes s_memrealtime at wave start / loop end / wave end into the n_iters output;
built from a patched copy of csrc, never shipped).  Run with CPPFLOW_HIP_LIB=build_var/lib_stamp.so."""
import os, sys
import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from cppflow_amd import _hip  # noqa: E402
from cppflow_amd.problems_synthetic import PANDA_2CUBES_OBSTACLES, obstacle_arrays  # noqa: E402
from cppflow_amd.robots import get_robot  # noqa: E402

DEV = torch.device("cuda:0")
rb = get_robot("panda")
obs = obstacle_arrays(PANDA_2CUBES_OBSTACLES)
rb.set_obstacles([c for c, _ in obs], [T for _, T in obs])
rb.set_joint_limit_padding(float(np.deg2rad(1.5)), 0.03)
S, W, K = 1024, 256, 10
x0, target, _ = bench.make_inputs_problem(rb, S, W, DEV, seed=0)
n = S * W
if os.environ.get("CPPF_STAMP_FLIP"):  # the same seeds in reverse order: does the slow band follow the data or the position?
    x0 = x0.view(S, W, -1).flip(0).reshape(n, -1).contiguous()
pk = torch.empty(rb.PACKED_BYTES_PER_ROW * n, dtype=torch.uint8, device=DEV)
sm = torch.empty((S, 8), dtype=torch.float32, device=DEV)
LM = dict(lm_lambda=1e-6, alpha_position=3.5, alpha_rotation=0.35)
for solver, name in ((_hip.SOLVER_AUTO, "auto"), (_hip.SOLVER_F32, "f32")):
    for _ in range(300):  # warm clocks
        rb.lm_pose_steps(x0, target, n_steps=K, packed_out=pk, summary_out=sm, solver=solver, **LM)
    torch.cuda.synchronize()
    res = []
    for rep in range(5):
        torch.cuda.synchronize()
        r = rb.lm_pose_steps(x0, target, n_steps=K, packed_out=pk, summary_out=sm, want_iters=True, solver=solver, **LM)
        torch.cuda.synchronize()
        v = r["n_iters"].cpu().numpy().astype(np.int64).reshape(-1, 64)  # one row per wavefront
        start = v[:, 0] & 0xFFFF
        end = (v[:, 0] >> 16) & 0xFFFF
        mid = (v[:, 1] >> 16) & 0xFFFF
        t0 = start.min() if (start.max() - start.min()) < 30000 else np.sort(start)[0]
        st = ((start - t0) & 0xFFFF) * 0.01  # us
        en = ((end - t0) & 0xFFFF) * 0.01
        md = ((mid - t0) & 0xFFFF) * 0.01
        res.append((st, md, en))
    st, md, en = res[-1]
    q = lambda a: np.percentile(a, [0, 10, 50, 90, 99, 100]).round(2)
    print(f"== solver {name}: 4096 wavefronts of one isolated launch (us from the first wavefront's start; percentiles 0 10 50 90 99 100)")
    print("   start      ", q(st))
    print("   loop end   ", q(md))
    print("   end        ", q(en))
    print("   lifetime   ", q(en - st), " loop ", q(md - st), " finish stage ", q(en - md))
    if v.shape[1] > 3 and os.environ.get("CPPF_STAMP_HWID"):
        hw, xcc = v[:, 2] & 0xFFFFFFFF, v[:, 3] & 0xF
        simd, cu, sh, se = (hw >> 4) & 3, (hw >> 8) & 15, (hw >> 12) & 1, (hw >> 13) & 7
        key = ((((xcc * 8 + se) * 2 + sh) * 16 + cu) * 4 + simd)
        import collections
        groups = collections.defaultdict(list)
        for i, k in enumerate(key):
            groups[int(k)].append(i)
        sizes = collections.Counter(len(g) for g in groups.values())
        print("   SIMDs in use:", len(groups), " wavefronts per SIMD histogram:", dict(sorted(sizes.items())))
        cus = collections.Counter(int(k) // 4 for k in key)
        print("   compute units in use:", len(cus), " wavefronts per CU histogram:", dict(sorted(collections.Counter(cus.values()).items())))
        # for SIMDs with 4 waves: order of finishing by workgroup band
        wgb = (np.arange(len(st)) // 4) // 256
        lastband = collections.Counter()
        for g in groups.values():
            if len(g) == 4:
                lastband[tuple(sorted(wgb[g]))] += 1
        print("   band composition of the 4-wave SIMDs (top 5):", lastband.most_common(5))
        ends_by_count = collections.defaultdict(list)
        for g in groups.values():
            ends_by_count[len(g)].append(en[g].max())
        for c_, e_ in sorted(ends_by_count.items()):
            print(f"   SIMDs holding {c_} wavefronts: last end median {np.median(e_):.2f} us, max {np.max(e_):.2f}")
    # by wave slot within the workgroup (dispatch order within a CU is not visible; use workgroup index bands instead)
    wg = np.arange(len(st)) // 4
    for lo, hi in ((0, 256), (256, 512), (512, 768), (768, 1024)):
        m = (wg >= lo) & (wg < hi)
        print(f"   workgroups {lo:4d}-{hi:4d}: start median {np.median(st[m]):6.2f}  end median {np.median(en[m]):6.2f}  end max {en[m].max():6.2f}")
