# developer sweep: a 32 768-row shard (what each of 8 GPUs runs under --scaling strong), streams x graphs
set -e
for g in on off; do for st in 3 4 5 6 7; do
  python bench.py --seeds 128 --steps 2048 --warmup 256 --streams $st --graphs $g --no-cpu-baseline --no-siblings 2>>gpurun_out/graph_try.err | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('seeds 128 graphs $g streams $st  us/step %.2f  host %.2f' % (d['ms_per_step']*1e3, d['config']['host_enqueue_us_per_step']))"
done; done
