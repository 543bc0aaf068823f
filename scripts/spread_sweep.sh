# developer sweep: the residency claim (CPPF_TUNE_SPREAD_KB) of shard-sized launches, 4 launches in flight
set -e
for seeds in 128 64 32 256; do
for kb in 0 42 0 42; do
  CPPF_BENCH_SPREAD_KB=$kb timeout -k 10 200 python bench.py --seeds $seeds --steps 1000 --warmup 100 --streams 4 --no-cpu-baseline --no-siblings 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('seeds/GPU', d['config']['seeds_per_gpu'], ' spread_kb $kb', ' us/step %.2f' % (d['ms_per_step']*1e3), ' isolated kernel %.2f us' % (d['roofline']['kernel_ms']*1e3))"
done; done
for kb in 0 42 0 42; do
  CPPF_BENCH_SPREAD_KB=$kb timeout -k 10 200 python bench.py --config C2 --no-cpu-baseline --no-siblings 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('C2  spread_kb $kb', ' us/step %.2f' % (d['ms_per_step']*1e3), ' isolated kernel %.2f us' % (d['roofline']['kernel_ms']*1e3), d['config']['kernel_shape'])"
done
