#!/usr/bin/env bash
# which stream a 20-step shard region starts on: forced start bucket 0 / 1 (two buckets of 8 steps on two streams)
set -o pipefail
mkdir -p gpurun_out/c11
for sb in 0 1 ""; do
  echo "== shard 128 seeds, one-rank RCCL, driver flags, CPPF_BENCH_START_BUCKET='$sb'"
  CPPF_BENCH_START_BUCKET=$sb CPPF_BENCH_FORCE_DIST=1 timeout -k 10 300 python bench.py --gpus 1 --seeds 128 --steps 20 --warmup 5 --no-cpu-baseline --no-siblings > gpurun_out/c11/shard_$sb.json 2> gpurun_out/c11/shard.err || exit 1
  python -c "import json; d=json.load(open('gpurun_out/c11/shard_$sb.json')); c=d['config']; print('us/step %.2f' % (d['ms_per_step']*1e3), [round(1e3*v,2) for v in c['timed_region']['ms_per_step_all']])"
done
echo "== done"
