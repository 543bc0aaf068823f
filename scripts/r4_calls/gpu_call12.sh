#!/usr/bin/env bash
# calibrated region start with 2 / 3 / 4 streams, and the N = 1 line, driver flags
set -o pipefail
mkdir -p gpurun_out/c12
for st in 2 3 4; do
  echo "== shard 128 seeds, one-rank RCCL, driver flags, --streams $st"
  CPPF_BENCH_FORCE_DIST=1 timeout -k 10 300 python bench.py --gpus 1 --seeds 128 --steps 20 --warmup 5 --streams $st --no-cpu-baseline --no-siblings > gpurun_out/c12/shard_$st.json 2> gpurun_out/c12/shard.err || exit 1
  python -c "import json; d=json.load(open('gpurun_out/c12/shard_$st.json')); c=d['config']; t=c['timed_region']; print('us/step %.2f' % (d['ms_per_step']*1e3), 'start', t['region_start_bucket'], [round(v,2) for v in t['region_start_bucket_calibration_us_per_step']], 'min %.2f max %.2f' % (1e3*min(t['ms_per_step_all']), 1e3*max(t['ms_per_step_all'])))"
done
for s in 256 512; do
  echo "== shard $s seeds, one-rank RCCL, driver flags"
  CPPF_BENCH_FORCE_DIST=1 timeout -k 10 300 python bench.py --gpus 1 --seeds $s --steps 20 --warmup 5 --no-cpu-baseline --no-siblings > gpurun_out/c12/shard_s$s.json 2> gpurun_out/c12/shard.err || exit 1
  python -c "import json; d=json.load(open('gpurun_out/c12/shard_s$s.json')); c=d['config']; t=c['timed_region']; print('us/step %.2f' % (d['ms_per_step']*1e3), 'steps/launch', c['steps_per_launch'], 'start', t['region_start_bucket'], t['region_start_bucket_calibration_us_per_step'], 'min %.2f max %.2f' % (1e3*min(t['ms_per_step_all']), 1e3*max(t['ms_per_step_all'])))"
done
echo "== N = 1, one-rank RCCL, driver flags"; CPPF_BENCH_FORCE_DIST=1 timeout -k 10 300 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-siblings > gpurun_out/c12/n1_dist.json 2> gpurun_out/c12/n1.err || exit 1
python -c "import json; d=json.load(open('gpurun_out/c12/n1_dist.json')); c=d['config']; t=c['timed_region']; print('us/step %.2f' % (d['ms_per_step']*1e3), 'start', t['region_start_bucket'], t['region_start_bucket_calibration_us_per_step'])"
echo "== done"
