set -e
mkdir -p gpurun_out/r3h
for rep in 1 2; do
for st in 3 4 5 6; do
  timeout -k 10 200 python bench.py --seeds 128 --steps 1000 --warmup 100 --streams $st --no-cpu-baseline --no-siblings 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('seeds/GPU', d['config']['seeds_per_gpu'], ' streams', d['config']['streams'], ' graphs', d['config']['hip_graphs'][:12], ' us/step %.2f' % (d['ms_per_step']*1e3), ' host %.1f us/step' % d['config']['host_enqueue_us_per_step'])"
done; done
