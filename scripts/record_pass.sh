#!/usr/bin/env bash
# Round-end measurement pass on the MI355X box (run from the repo root through gpurun); everything lands in gpurun_out/rec/.
# Counter passes are separate rocprofv3 runs with --pmc only (no trace domains), as MI355X_MICROARCH.md prescribes.
set -eo pipefail
ROOT="$(pwd)"
OUT="$ROOT/gpurun_out/rec"
mkdir -p "$OUT"
export TMPDIR=/tmp
echo "== pytest -m gpu"
timeout -k 10 600 python -m pytest tests -m gpu -q 2>&1 | tail -3 | tee "$OUT/pytest_gpu.txt"
echo "== bench (default)"
timeout -k 10 600 python bench.py > "$OUT/bench.json" 2> "$OUT/bench.err"
cat "$OUT/bench.json"
echo "== bench, one stream"
timeout -k 10 300 python bench.py --no-cpu-baseline --streams 1 > "$OUT/bench_streams1.json" 2>> "$OUT/bench.err"
echo "== bench, one-rank RCCL group"
CPPF_BENCH_FORCE_DIST=1 timeout -k 10 300 python bench.py --no-cpu-baseline > "$OUT/bench_dist1.json" 2>> "$OUT/bench.err"
for c in C2 C3 C5; do
    echo "== bench --config $c"
    timeout -k 10 300 python bench.py --no-cpu-baseline --config $c > "$OUT/bench_$c.json" 2>> "$OUT/bench.err"
done
echo "== kbench"
timeout -k 10 600 python scripts/kbench.py > "$OUT/kbench.txt" 2>&1
echo "== rocprofv3 kernel trace"
cd /tmp
timeout -k 10 600 rocprofv3 --kernel-trace --stats -d "$OUT/kt" -o kt -- python3 "$ROOT/bench.py" --steps 100 --warmup 5 --no-cpu-baseline --streams 1 > "$OUT/kt_stdout.txt" 2> "$OUT/kt_stderr.txt"
echo "== rocprofv3 pmc FETCH_SIZE"
timeout -k 10 600 rocprofv3 --pmc FETCH_SIZE -d "$OUT/pmc_fetch" -o pmc -- python3 "$ROOT/bench.py" --steps 5 --warmup 2 --no-cpu-baseline --streams 1 > /dev/null 2> "$OUT/pmc_fetch_stderr.txt"
echo "== rocprofv3 pmc WRITE_SIZE"
timeout -k 10 600 rocprofv3 --pmc WRITE_SIZE -d "$OUT/pmc_write" -o pmc -- python3 "$ROOT/bench.py" --steps 5 --warmup 2 --no-cpu-baseline --streams 1 > /dev/null 2> "$OUT/pmc_write_stderr.txt"
echo "== rocprofv3 pmc SQ_INSTS_VALU SQ_WAVES"
timeout -k 10 600 rocprofv3 --pmc SQ_INSTS_VALU SQ_WAVES -d "$OUT/pmc_valu" -o pmc -- python3 "$ROOT/scripts/pmc_probe.py" panda > /dev/null 2> "$OUT/pmc_valu_stderr.txt"
cd "$ROOT"
find "$OUT" -name "*.csv" | head -30
# keep the merge small: drop anything big that is not a csv / json / txt summary
find "$OUT" -type f -size +8M -delete
echo "== done"
