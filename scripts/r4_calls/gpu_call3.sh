#!/usr/bin/env bash
# third GPU call of round 4: PCR forms (in-tree: lean at d = 8 with fences, row form at d <= 7; variant: fences only at 512 lanes), hw sin/cos A/B,
# the dp_search exchange microbenchmark, the shard at the driver's flags with the exchange issued one launch late
set -o pipefail
mkdir -p gpurun_out/c3
ok() { [ "$1" -ne 124 ] && [ "$1" -ne 137 ]; }
echo "== pcr_ab in-tree"; timeout -k 10 300 python scripts/pcr_ab.py > gpurun_out/c3/pcr_ab.txt 2>&1; rc=$?; grep -v amdgpu.ids gpurun_out/c3/pcr_ab.txt; ok $rc || exit 1
echo "== pcr_ab fence512 variant"; CPPFLOW_HIP_LIB=build_var/lib_pcr_fence512.so timeout -k 10 300 python scripts/pcr_ab.py > gpurun_out/c3/pcr_ab_fence512.txt 2>&1; rc=$?; grep -v amdgpu.ids gpurun_out/c3/pcr_ab_fence512.txt; ok $rc || exit 1
echo "== dp_exchange"; timeout -k 10 300 ./build_var/dp_exchange > gpurun_out/c3/dp_exchange.txt 2>&1; rc=$?; cat gpurun_out/c3/dp_exchange.txt; ok $rc || exit 1
echo "== A/B hw sincos (in-tree) vs canonical"; bash scripts/lib_ab.sh build_var/lib_canon.so 2>&1 | tee gpurun_out/c3/ab_sincos.txt
for args in "" "--streams 3"; do
  echo "== shard 128 seeds, one-rank RCCL, driver flags $args"
  CPPF_BENCH_FORCE_DIST=1 timeout -k 10 300 python bench.py --gpus 1 --seeds 128 --steps 20 --warmup 5 $args --no-cpu-baseline --no-siblings > gpurun_out/c3/shard.json 2> gpurun_out/c3/shard.err; rc=$?; ok $rc || exit 1
  python -c "import json; d=json.load(open('gpurun_out/c3/shard.json')); c=d['config']; print('us/step %.2f' % (d['ms_per_step']*1e3), 'steps/launch', c['steps_per_launch'], 'steps/allgather', c['steps_per_allgather'], 'streams', c['streams'], 'all', ['%.2f' % (1e3*v) for v in c['timed_region']['ms_per_step_all']])" || tail -5 gpurun_out/c3/shard.err
done
echo "== N=1 driver flags"; timeout -k 10 300 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-siblings > gpurun_out/c3/n1.json 2> gpurun_out/c3/n1.err; rc=$?; ok $rc || exit 1
python -c "import json; d=json.load(open('gpurun_out/c3/n1.json')); print('us/step %.2f' % (d['ms_per_step']*1e3))"
echo "== done"
