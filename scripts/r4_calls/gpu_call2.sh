#!/usr/bin/env bash
# second GPU call of round 4: the suite on the register-lean PCR / quad / blocks kernels, the coupled step under them, shard scheduling at the driver's flags
set -o pipefail
mkdir -p gpurun_out/c2
ok() { [ "$1" -ne 124 ] && [ "$1" -ne 137 ]; }
timeout -k 10 1000 python -m pytest tests -m gpu -q -x --timeout 600 > gpurun_out/c2/pytest.txt 2>&1; rc=$?; tail -5 gpurun_out/c2/pytest.txt | cut -c1-300; ok $rc || exit 1
[ $rc -eq 0 ] || { grep -n "Error\|assert\|FAILED" gpurun_out/c2/pytest.txt | head -30; exit 1; }
echo "== pcr_ab"; timeout -k 10 300 python scripts/pcr_ab.py > gpurun_out/c2/pcr_ab.txt 2>&1; rc=$?; grep -v amdgpu.ids gpurun_out/c2/pcr_ab.txt; ok $rc || exit 1
echo "== coupled_bench"; timeout -k 10 400 python scripts/coupled_bench.py --seeds 1,64,512,1024 > gpurun_out/c2/coupled_bench.txt 2>&1; rc=$?; grep -v amdgpu.ids gpurun_out/c2/coupled_bench.txt; ok $rc || exit 1
for args in "--batch 8 --streams 2" "--batch 8 --streams 3" "--batch 8 --streams 4" "--batch 4 --streams 4" "--batch 5 --streams 4" "--batch 10 --streams 2" "--batch 7 --streams 3"; do
  echo "== shard 128 seeds, one-rank RCCL, driver flags: $args"
  CPPF_BENCH_FORCE_DIST=1 timeout -k 10 300 python bench.py --gpus 1 --seeds 128 --steps 20 --warmup 5 $args --no-cpu-baseline --no-siblings > gpurun_out/c2/shard.json 2> gpurun_out/c2/shard.err; rc=$?; ok $rc || exit 1
  python -c "import json; d=json.load(open('gpurun_out/c2/shard.json')); c=d['config']; print('us/step %.2f' % (d['ms_per_step']*1e3), 'steps/launch', c['steps_per_launch'], 'steps/allgather', c['steps_per_allgather'], 'streams', c['streams'], c['hip_graphs'][:3], 'all', ['%.2f' % (1e3*v) for v in c['timed_region']['ms_per_step_all']])" || tail -5 gpurun_out/c2/shard.err
done
echo "== done"
