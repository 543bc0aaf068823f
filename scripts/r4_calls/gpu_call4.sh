#!/usr/bin/env bash
# fourth GPU call of round 4: suite on the build with canonical leading iterations, split PCR at d = 8, 1 024-lane dp_search forms
set -o pipefail
mkdir -p gpurun_out/c4
ok() { [ "$1" -ne 124 ] && [ "$1" -ne 137 ]; }
timeout -k 10 1000 python -m pytest tests -m gpu -q -x --timeout 600 > gpurun_out/c4/pytest.txt 2>&1; rc=$?; tail -5 gpurun_out/c4/pytest.txt | cut -c1-300; ok $rc || exit 1
[ $rc -eq 0 ] || { grep -n "Error\|assert\|FAILED" gpurun_out/c4/pytest.txt | head -30; exit 1; }
echo "== dp_bench"; timeout -k 10 500 python scripts/dp_bench.py > gpurun_out/c4/dp_bench.txt 2>&1; rc=$?; grep -v amdgpu.ids gpurun_out/c4/dp_bench.txt; ok $rc || exit 1
echo "== coupled_bench"; timeout -k 10 400 python scripts/coupled_bench.py --seeds 1,64,512,1024 > gpurun_out/c4/coupled_bench.txt 2>&1; rc=$?; grep -v amdgpu.ids gpurun_out/c4/coupled_bench.txt; ok $rc || exit 1
echo "== N=1 driver flags"; timeout -k 10 300 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/c4/n1.json 2> gpurun_out/c4/n1.err; rc=$?; ok $rc || exit 1
python -c "import json; d=json.load(open('gpurun_out/c4/n1.json')); print('us/step %.2f' % (d['ms_per_step']*1e3), 'one_stream', d.get('one_stream',{}).get('ms_per_step'), 'random', d.get('random_inputs',{}).get('ms_per_step'), 'plan_search', d.get('plan_search'))"
echo "== N=1 2000 steps"; timeout -k 10 300 python bench.py --no-cpu-baseline --no-siblings > gpurun_out/c4/n1_2000.json 2> gpurun_out/c4/n1.err; rc=$?; ok $rc || exit 1
python -c "import json; d=json.load(open('gpurun_out/c4/n1_2000.json')); print('us/step %.2f' % (d['ms_per_step']*1e3))"
echo "== done"
