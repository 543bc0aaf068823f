#!/usr/bin/env bash
# Step rate vs HIP hardware-queue count and launch streams (developer tool behind bench.py's defaults): independent launches on
# different HIP streams only overlap when the streams map to different hardware queues (GPU_MAX_HW_QUEUES, default 4).
for s in 1024 512 256 128; do
  for q in default 16; do
    for st in 2 4 8; do
      if [ "$q" = default ]; then unset GPU_MAX_HW_QUEUES; export CPPF_BENCH_KEEP_HWQ=1; else export GPU_MAX_HW_QUEUES=$q; fi
      timeout -k 10 100 python bench.py --seeds $s --steps 2000 --warmup 200 --streams $st --no-cpu-baseline --no-siblings 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('seeds $s hwq $q streams', d['config']['streams'], 'us/step %.2f'%(d['ms_per_step']*1e3), 'host %.1f'%d['config']['host_enqueue_us_per_step'])"
    done
  done
done
