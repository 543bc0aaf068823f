#!/usr/bin/env bash
# ninth GPU call of round 4: relative gate criterion in the lean iterations -- suite, then the bench siblings at 0 / 1000 / 10000 ppm
set -o pipefail
mkdir -p gpurun_out/c9
ok() { [ "$1" -ne 124 ] && [ "$1" -ne 137 ]; }
timeout -k 10 1000 python -m pytest tests -m gpu -q --timeout 600 > gpurun_out/c9/pytest.txt 2>&1; rc=$?; tail -5 gpurun_out/c9/pytest.txt | cut -c1-300; ok $rc || exit 1
[ $rc -eq 0 ] || { grep -n "Error\|assert\|FAILED" gpurun_out/c9/pytest.txt | head -30; }
for ppm in 0 1000 10000; do
  echo "== bench (2000 steps, siblings), CPPF_BENCH_GATE_REL_PPM=$ppm"; CPPF_BENCH_GATE_REL_PPM=$ppm timeout -k 10 400 python bench.py --no-cpu-baseline > gpurun_out/c9/bench_$ppm.json 2> gpurun_out/c9/bench.err; rc=$?; ok $rc || exit 1
  python -c "import json; d=json.load(open('gpurun_out/c9/bench_$ppm.json')); print('us/step %.2f' % (d['ms_per_step']*1e3), 'one_stream %.2f' % (1e3*d['one_stream']['ms_per_step']), 'random %.2f' % (1e3*d['random_inputs']['ms_per_step']), 'conv', d['config']['converged_frac_pos_err_lt_1e-4'], 'random conv', d['random_inputs'].get('converged_frac_pos_err_lt_1e-4'))"
done
echo "== done"
