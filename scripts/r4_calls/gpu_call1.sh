#!/usr/bin/env bash
# first GPU call of round 4: the test suite on the new kernels, then the numbers that decide what to do next
set -o pipefail
mkdir -p gpurun_out/c1
ok() { [ "$1" -ne 124 ] && [ "$1" -ne 137 ]; }
python -c "import __graft_entry__ as g; g.build()" > gpurun_out/c1/build.txt 2>&1 || { tail -5 gpurun_out/c1/build.txt; exit 1; }
timeout -k 10 1000 python -m pytest tests -m gpu -q -x --timeout 600 > gpurun_out/c1/pytest.txt 2>&1; rc=$?; tail -30 gpurun_out/c1/pytest.txt; ok $rc || exit 1
echo "== bench driver flags"; timeout -k 10 400 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/c1/bench_driver.json 2> gpurun_out/c1/bench_driver.err; rc=$?; ok $rc || exit 1
python -c "import json; d=json.load(open('gpurun_out/c1/bench_driver.json')); print('driver: us/step %.2f' % (d['ms_per_step']*1e3), 'kernel %.1f us' % (d['roofline']['kernel_ms']*1e3), d['config']['timed_region']['ms_per_step_all'], 'one_stream', d.get('one_stream',{}).get('ms_per_step'), 'random', d.get('random_inputs',{}).get('ms_per_step'))" || tail -5 gpurun_out/c1/bench_driver.err
echo "== bench 2000 steps"; timeout -k 10 400 python bench.py --no-cpu-baseline --no-siblings > gpurun_out/c1/bench_2000.json 2> gpurun_out/c1/bench_2000.err; rc=$?; ok $rc || exit 1
python -c "import json; d=json.load(open('gpurun_out/c1/bench_2000.json')); print('2000 steps: us/step %.2f' % (d['ms_per_step']*1e3), 'kernel %.1f us' % (d['roofline']['kernel_ms']*1e3))" || tail -5 gpurun_out/c1/bench_2000.err
for args in "--steps 20 --warmup 5" "--steps 2000 --warmup 100" "--steps 2000 --warmup 100 --batch 1" "--steps 2000 --warmup 100 --batch 4"; do
  echo "== shard 128 seeds, one-rank RCCL: $args"
  CPPF_BENCH_FORCE_DIST=1 timeout -k 10 300 python bench.py --gpus 1 --seeds 128 $args --no-cpu-baseline --no-siblings > gpurun_out/c1/shard.json 2> gpurun_out/c1/shard.err; rc=$?; ok $rc || exit 1
  python -c "import json; d=json.load(open('gpurun_out/c1/shard.json')); c=d['config']; print('us/step %.2f' % (d['ms_per_step']*1e3), 'steps/launch', c['steps_per_launch'], 'steps/allgather', c['steps_per_allgather'], 'streams', c['streams'], c['hip_graphs'][:3], 'launch %.1f us' % (d['roofline']['kernel_ms']*1e3), 'all', ['%.2f' % (1e3*v) for v in c['timed_region']['ms_per_step_all']], 'closing', c['timed_region']['closing_barrier_us'])" || tail -5 gpurun_out/c1/shard.err
done
for s in 256 512; do
  echo "== shard $s seeds (no dist)"
  timeout -k 10 300 python bench.py --gpus 1 --seeds $s --steps 2000 --warmup 100 --no-cpu-baseline --no-siblings > gpurun_out/c1/shard.json 2> gpurun_out/c1/shard.err; rc=$?; ok $rc || exit 1
  python -c "import json; d=json.load(open('gpurun_out/c1/shard.json')); c=d['config']; print('us/step %.2f' % (d['ms_per_step']*1e3), 'steps/launch', c['steps_per_launch'], 'streams', c['streams'], 'launch %.1f us' % (d['roofline']['kernel_ms']*1e3))" || tail -5 gpurun_out/c1/shard.err
done
echo "== C2 / C3"; for c in C2 C3; do timeout -k 10 300 python bench.py --config $c --no-cpu-baseline --no-siblings > gpurun_out/c1/bench_$c.json 2> gpurun_out/c1/bench_$c.err; rc=$?; ok $rc || exit 1
  python -c "import json; d=json.load(open('gpurun_out/c1/bench_$c.json')); c=d['config']; print('$c us/step %.2f' % (d['ms_per_step']*1e3), 'steps/launch', c['steps_per_launch'], 'launch %.1f us' % (d['roofline']['kernel_ms']*1e3))" || tail -5 gpurun_out/c1/bench_$c.err; done
echo "== dp_bench"; timeout -k 10 500 python scripts/dp_bench.py > gpurun_out/c1/dp_bench.txt 2>&1; rc=$?; cat gpurun_out/c1/dp_bench.txt | tail -32; ok $rc || exit 1
echo "== done"
