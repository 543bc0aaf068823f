#!/usr/bin/env bash
# eighth GPU call of round 4: first iteration lean (any-angle polynomials) vs general -- suite, then A/B
set -o pipefail
mkdir -p gpurun_out/c8
ok() { [ "$1" -ne 124 ] && [ "$1" -ne 137 ]; }
timeout -k 10 1000 python -m pytest tests -m gpu -q --timeout 600 > gpurun_out/c8/pytest.txt 2>&1; rc=$?; tail -5 gpurun_out/c8/pytest.txt | cut -c1-300; ok $rc || exit 1
[ $rc -eq 0 ] || { grep -n "Error\|assert\|FAILED" gpurun_out/c8/pytest.txt | head -30; }
echo "== A/B first iteration lean (in-tree) vs general"; bash scripts/lib_ab.sh build_var/lib_first_general.so 2>&1 | tee gpurun_out/c8/ab_first_lean.txt
echo "== done"
