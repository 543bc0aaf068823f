#!/usr/bin/env bash
# sixth GPU call of round 4: full-range polynomial sine / cosine in the leading iterations -- suite, then A/B against the all-canonical build
set -o pipefail
mkdir -p gpurun_out/c6
ok() { [ "$1" -ne 124 ] && [ "$1" -ne 137 ]; }
timeout -k 10 1000 python -m pytest tests -m gpu -q -x --timeout 600 > gpurun_out/c6/pytest.txt 2>&1; rc=$?; tail -5 gpurun_out/c6/pytest.txt | cut -c1-300; ok $rc || exit 1
[ $rc -eq 0 ] || { grep -n "Error\|assert\|FAILED" gpurun_out/c6/pytest.txt | head -30; exit 1; }
echo "== A/B polynomial sincos in the leading iterations (in-tree) vs canonical"; bash scripts/lib_ab.sh build_var/lib_canon.so 2>&1 | tee gpurun_out/c6/ab_sincos_poly.txt
echo "== N=1 driver flags"; timeout -k 10 300 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/c6/n1.json 2> gpurun_out/c6/n1.err; rc=$?; ok $rc || exit 1
python -c "import json; d=json.load(open('gpurun_out/c6/n1.json')); print('us/step %.2f' % (d['ms_per_step']*1e3), 'one_stream', d.get('one_stream',{}).get('ms_per_step'), 'random', d.get('random_inputs',{}).get('ms_per_step'), 'converged', d['config'].get('converged_frac'))"
for c in C3 C5; do timeout -k 10 300 python bench.py --config $c --no-cpu-baseline --no-siblings > gpurun_out/c6/bench_$c.json 2> gpurun_out/c6/bench_$c.err; rc=$?; ok $rc || exit 1
  python -c "import json; d=json.load(open('gpurun_out/c6/bench_$c.json')); print('$c us/step %.2f' % (d['ms_per_step']*1e3))"; done
echo "== done"
