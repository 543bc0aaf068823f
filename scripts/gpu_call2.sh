#!/usr/bin/env bash
set -o pipefail
mkdir -p gpurun_out/c2
ok() { [ "$1" -ne 124 ] && [ "$1" -ne 137 ]; }
timeout -k 10 1000 python -m pytest tests -m gpu -q --timeout 600 > gpurun_out/c2/pytest.txt 2>&1; rc=$?; tail -40 gpurun_out/c2/pytest.txt | cut -c1-300; ok $rc || exit 1
echo "== valu_rate (transcendental modes)"; timeout -k 5 200 ./build_var/valu_rate > gpurun_out/c2/valu_rate.txt 2>&1; rc=$?; grep "waves/SIMD 4\|waves/SIMD 1 " gpurun_out/c2/valu_rate.txt; ok $rc || exit 1
echo "== A/B: in-tree (hardware sin/cos in the leading iterations) vs lib_canon (canonical everywhere)"
bash scripts/lib_ab.sh build_var/lib_canon.so 2>&1 | tee gpurun_out/c2/ab_sincos.txt
echo "== driver flags, N=1"; timeout -k 10 400 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-siblings > gpurun_out/c2/bench_driver.json 2> gpurun_out/c2/bench_driver.err; rc=$?; ok $rc || exit 1
python -c "import json; d=json.load(open('gpurun_out/c2/bench_driver.json')); print('driver: us/step %.2f' % (d['ms_per_step']*1e3), 'kernel %.1f us' % (d['roofline']['kernel_ms']*1e3))" || tail -5 gpurun_out/c2/bench_driver.err
for args in "--steps 20 --warmup 5" "--steps 2000 --warmup 100" "--steps 20 --warmup 5 --streams 4"; do
  echo "== shard 128 seeds, one-rank RCCL: $args"
  CPPF_BENCH_FORCE_DIST=1 timeout -k 10 300 python bench.py --gpus 1 --seeds 128 $args --no-cpu-baseline --no-siblings > gpurun_out/c2/shard.json 2> gpurun_out/c2/shard.err; rc=$?; ok $rc || exit 1
  python -c "import json; d=json.load(open('gpurun_out/c2/shard.json')); c=d['config']; print('us/step %.2f' % (d['ms_per_step']*1e3), 'steps/launch', c['steps_per_launch'], 'steps/allgather', c['steps_per_allgather'], 'streams', c['streams'], 'launch %.1f us' % (d['roofline']['kernel_ms']*1e3), 'all', ['%.2f' % (1e3*v) for v in c['timed_region']['ms_per_step_all']])" || tail -5 gpurun_out/c2/shard.err
done
echo "== 2 ranks sharing the GPU (gloo rehearsal), driver flags"
CPPF_BENCH_SHARE_GPU=1 timeout -k 10 300 python bench.py --gpus 2 --steps 20 --warmup 5 --no-siblings > gpurun_out/c2/bench_2ranks.json 2> gpurun_out/c2/bench_2ranks.err; rc=$?; ok $rc || exit 1
python -c "import json; d=json.load(open('gpurun_out/c2/bench_2ranks.json')); c=d['config']; print('2 ranks: us/step %.2f' % (d['ms_per_step']*1e3), c['steps_per_launch'], c['steps_per_allgather'], d['selection_check']['equals_single_process'])" || tail -5 gpurun_out/c2/bench_2ranks.err
echo "== C3 (new Fetch model and inputs)"; timeout -k 10 300 python bench.py --config C3 --no-cpu-baseline --no-siblings > gpurun_out/c2/bench_C3.json 2> gpurun_out/c2/bench_C3.err; rc=$?; ok $rc || exit 1
python -c "import json; d=json.load(open('gpurun_out/c2/bench_C3.json')); c=d['config']; print('C3 us/step %.2f' % (d['ms_per_step']*1e3), 'steps/launch', c['steps_per_launch'], 'conv', c['converged_frac_pos_err_lt_1e-4'])" || tail -5 gpurun_out/c2/bench_C3.err
echo "== dp_bench"; timeout -k 10 500 python scripts/dp_bench.py > gpurun_out/c2/dp_bench.txt 2>&1; rc=$?; grep "k=  175 T= 256\|k=  300\|k=  512\|k= 1024" gpurun_out/c2/dp_bench.txt; ok $rc || exit 1
echo "== coupled_bench"; timeout -k 10 300 python scripts/coupled_bench.py > gpurun_out/c2/coupled_bench.txt 2>&1; rc=$?; tail -12 gpurun_out/c2/coupled_bench.txt; ok $rc || exit 1
echo "== done"
