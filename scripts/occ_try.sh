# developer run: headline bench line + one-stream + shard + C3 after an occupancy change
set -e
python bench.py --no-cpu-baseline --no-siblings 2>>gpurun_out/occ_try.err | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('C4 2 streams  us/step %.2f  kernel %.2f us' % (d['ms_per_step']*1e3, d['roofline']['kernel_ms']*1e3))"
python bench.py --no-cpu-baseline --no-siblings --streams 1 2>>gpurun_out/occ_try.err | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('C4 1 stream   us/step %.2f  kernel %.2f us' % (d['ms_per_step']*1e3, d['roofline']['kernel_ms']*1e3))"
python bench.py --no-cpu-baseline --no-siblings --inputs random 2>>gpurun_out/occ_try.err | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('C4 random     us/step %.2f  kernel %.2f us' % (d['ms_per_step']*1e3, d['roofline']['kernel_ms']*1e3))"
python bench.py --no-cpu-baseline --no-siblings --config C3 2>>gpurun_out/occ_try.err | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('C3            us/step %.2f  kernel %.2f us' % (d['ms_per_step']*1e3, d['roofline']['kernel_ms']*1e3))"
for s in 512 256 128; do
python bench.py --seeds $s --steps 1000 --warmup 100 --no-cpu-baseline --no-siblings 2>>gpurun_out/occ_try.err | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('seeds $s  us/step %.2f  kernel %.2f us' % (d['ms_per_step']*1e3, d['roofline']['kernel_ms']*1e3))"
done
