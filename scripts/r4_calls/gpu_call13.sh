#!/usr/bin/env bash
# does the opening barrier's idle gap cost the region its clocks?  bare launches between barrier and clock: 0 / 4 / 16
set -o pipefail
mkdir -p gpurun_out/c13
for pw in 0 4 16; do
  echo "== N = 1, one-rank RCCL, driver flags, CPPF_BENCH_REGION_PREWARM=$pw"
  CPPF_BENCH_REGION_PREWARM=$pw CPPF_BENCH_FORCE_DIST=1 timeout -k 10 300 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-siblings > gpurun_out/c13/n1_$pw.json 2> gpurun_out/c13/err.txt || exit 1
  python -c "import json; d=json.load(open('gpurun_out/c13/n1_$pw.json')); c=d['config']; t=c['timed_region']; print('us/step %.2f' % (d['ms_per_step']*1e3), 'start', t['region_start_bucket'], t['region_start_bucket_calibration_us_per_step'], 'min %.2f max %.2f' % (1e3*min(t['ms_per_step_all']), 1e3*max(t['ms_per_step_all'])), 'closing', [round(v) for v in t['closing_barrier_us'][:5]])"
  echo "== shard 128, CPPF_BENCH_REGION_PREWARM=$pw"
  CPPF_BENCH_REGION_PREWARM=$pw CPPF_BENCH_FORCE_DIST=1 timeout -k 10 300 python bench.py --gpus 1 --seeds 128 --steps 20 --warmup 5 --no-cpu-baseline --no-siblings > gpurun_out/c13/s128_$pw.json 2> gpurun_out/c13/err.txt || exit 1
  python -c "import json; d=json.load(open('gpurun_out/c13/s128_$pw.json')); c=d['config']; t=c['timed_region']; print('us/step %.2f' % (d['ms_per_step']*1e3), 'start', t['region_start_bucket'], t['region_start_bucket_calibration_us_per_step'], 'min %.2f max %.2f' % (1e3*min(t['ms_per_step_all']), 1e3*max(t['ms_per_step_all'])))"
done
echo "== N = 1 without a process group, driver flags"; timeout -k 10 300 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-siblings > gpurun_out/c13/n1_plain.json 2> gpurun_out/c13/err.txt || exit 1
python -c "import json; d=json.load(open('gpurun_out/c13/n1_plain.json')); print('us/step %.2f' % (d['ms_per_step']*1e3))"
echo "== done"
