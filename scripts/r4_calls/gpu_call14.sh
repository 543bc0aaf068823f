#!/usr/bin/env bash
# steps per collective = steps per launch for short regions? driver flags, one-rank RCCL
set -o pipefail
mkdir -p gpurun_out/c14
run() { CPPF_BENCH_FORCE_DIST=1 timeout -k 10 300 python bench.py --gpus 1 "$@" --steps 20 --warmup 5 --no-cpu-baseline --no-siblings > gpurun_out/c14/o.json 2> gpurun_out/c14/err.txt || exit 1
  python -c "import json; d=json.load(open('gpurun_out/c14/o.json')); c=d['config']; t=c['timed_region']; print('us/step %.2f' % (d['ms_per_step']*1e3), 'steps/launch', c['steps_per_launch'], 'steps/allgather', c['steps_per_allgather'], 'streams', c['streams'], 'start', t['region_start_bucket'], [round(v,2) for v in (t['region_start_bucket_calibration_us_per_step'] or [])], 'min %.2f max %.2f' % (1e3*min(t['ms_per_step_all']), 1e3*max(t['ms_per_step_all'])))"; }
echo "== 512 seeds"; run --seeds 512; run --seeds 512 --gather-every 2; run --seeds 512 --gather-every 4; run --seeds 512 --gather-every 2 --streams 4
echo "== 256 seeds"; run --seeds 256; run --seeds 256 --gather-every 4; run --seeds 256 --gather-every 4 --streams 4
echo "== 1024 seeds"; run; run --gather-every 1; run --gather-every 2
echo "== 128 seeds"; run --seeds 128; run --seeds 128 --gather-every 8 --streams 4
echo "== done"
