#!/usr/bin/env bash
# round 5, GPU call 16: the negated-W sign convention of the lean block solve (A/B, results must be bit-identical), broad-phase census
set -o pipefail
ROOT="$(pwd)"; OUT="$ROOT/gpurun_out/r5"; mkdir -p "$OUT"; export TMPDIR=/tmp
F="$OUT/ab_lean_block_neg.txt"
timeout -k 10 300 python scripts/cull_stats.py panda 2>&1 | grep -v amdgpu.ids | tee "$OUT/cull_stats.txt"
echo "== bit-identical results? (x, packed outputs of a C4 launch, both libraries)" | tee "$F"
for lib in cppflow_amd/csrc/libcppflow_hip.so build_var/lib_blkpos.so; do
CPPFLOW_HIP_LIB=$lib timeout -k 10 200 python - <<'PY' 2>&1 | grep -v amdgpu.ids | tee -a "$F"
import os, hashlib, numpy as np, torch, bench
from cppflow_amd.robots import get_robot
from cppflow_amd.problems_synthetic import PANDA_2CUBES_OBSTACLES, obstacle_arrays
rb = get_robot("panda"); obs = obstacle_arrays(PANDA_2CUBES_OBSTACLES)
rb.set_obstacles([c for c, _ in obs], [T for _, T in obs]); rb.set_joint_limit_padding(float(np.deg2rad(1.5)), 0.03)
dev = torch.device("cuda:0")
for label, (x0, t) in (("planner", bench.make_inputs_problem(rb, 1024, 256, dev, seed=0)[:2]), ("random", bench.make_inputs(rb, 1024, 256, dev, 1))):
    pk = torch.empty(rb.PACKED_BYTES_PER_ROW * x0.shape[0], dtype=torch.uint8, device=dev)
    r = rb.lm_pose_steps(x0, t, 1e-6, 3.5, 0.35, n_steps=10, packed_out=pk)
    torch.cuda.synchronize()
    print(os.environ["CPPFLOW_HIP_LIB"], label, "sha256(x) %s  sha256(packed) %s" % (hashlib.sha256(r["x"].cpu().numpy().tobytes()).hexdigest()[:16], hashlib.sha256(pk.cpu().numpy().tobytes()).hexdigest()[:16]))
PY
done
echo "== A/B, default flags (2 000 steps): in-tree = negated convention, lib_blkpos = CPPF_LEAN_BLOCK_NEG=0" | tee -a "$F"
timeout -k 10 600 bash scripts/lib_ab.sh build_var/lib_blkpos.so 2>&1 | tee -a "$F"
echo "== A/B, driver flags" | tee -a "$F"
for rep in 1 2 3; do for lib in cppflow_amd/csrc/libcppflow_hip.so build_var/lib_blkpos.so; do
  echo -n "$lib  " | tee -a "$F"
  CPPFLOW_HIP_LIB=$lib timeout -k 10 200 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-siblings 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('driver flags us/step %.2f   isolated kernel %.2f us' % (d['ms_per_step']*1e3, d['roofline']['kernel_ms']*1e3))" | tee -a "$F"
done; done
