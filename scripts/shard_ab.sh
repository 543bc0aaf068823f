set -e
mkdir -p gpurun_out/r3h
for sv in auto f32; do for st in 4; do
python bench.py --seeds 128 --steps 2000 --warmup 200 --streams $st --solver $sv --no-cpu-baseline --no-siblings 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$sv seeds/GPU', d['config']['seeds_per_gpu'], ' streams', d['config']['streams'], ' us/step %.2f' % (d['ms_per_step']*1e3), ' isolated kernel %.2f us' % (d['roofline']['kernel_ms']*1e3), d['roofline']['kernel_ms_stats'])"
done; done
python scripts/shard_bench.py --shapes row --sizes 128,1024 2>&1 | tail -6
