#!/usr/bin/env bash
# defaults after the short-region rules: every shard size with the driver's flags (one-rank RCCL), the two-rank rehearsal, N = 1
set -o pipefail
mkdir -p gpurun_out/c15
run() { CPPF_BENCH_FORCE_DIST=1 timeout -k 10 300 python bench.py --gpus 1 "$@" --steps 20 --warmup 5 --no-cpu-baseline --no-siblings > gpurun_out/c15/o.json 2> gpurun_out/c15/err.txt || { tail -5 gpurun_out/c15/err.txt; exit 1; }
  python -c "import json; d=json.load(open('gpurun_out/c15/o.json')); c=d['config']; t=c['timed_region']; print('us/step %.2f' % (d['ms_per_step']*1e3), 'steps/launch', c['steps_per_launch'], 'steps/allgather', c['steps_per_allgather'], 'streams', c['streams'], 'start', t['region_start_bucket_calibration_us_per_step'], 'repeats', t['repeats'], 'min %.2f max %.2f' % (1e3*min(t['ms_per_step_all']), 1e3*max(t['ms_per_step_all'])), 'selection', d.get('selection_check',{}).get('equals_single_process'))"; }
for s in 1024 512 256 128; do echo "== $s seeds per rank"; run --seeds $s; done
echo "== two ranks sharing the GPU (gloo rehearsal), driver flags"; CPPF_BENCH_SHARE_GPU=1 timeout -k 10 300 python bench.py --gpus 2 --steps 20 --warmup 5 --no-siblings --no-cpu-baseline > gpurun_out/c15/two.json 2> gpurun_out/c15/two.err || { tail -5 gpurun_out/c15/two.err; exit 1; }
python -c "import json; d=json.load(open('gpurun_out/c15/two.json')); c=d['config']; print('us/step %.2f' % (d['ms_per_step']*1e3), 'steps/launch', c['steps_per_launch'], 'steps/allgather', c['steps_per_allgather'], d['selection_check']['identical_on_every_rank'], d['selection_check']['equals_single_process'])"
echo "== N = 1 plain"; timeout -k 10 300 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-siblings > gpurun_out/c15/n1.json 2> gpurun_out/c15/err.txt || exit 1
python -c "import json; d=json.load(open('gpurun_out/c15/n1.json')); print('us/step %.2f' % (d['ms_per_step']*1e3))"
echo "== done"
