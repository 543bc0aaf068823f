#!/usr/bin/env bash
# round 5, GPU call 5: the overlap trace again (region detection fixed), then the bench records of the record pass on the committed counters
set -eo pipefail
ROOT="$(pwd)"; OUT="$ROOT/gpurun_out/rec"; mkdir -p "$OUT/profiles"; export TMPDIR=/tmp
cd /tmp
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d "$OUT/overlap" -o kt -- python3 "$ROOT/bench.py" --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-siblings > "$OUT/overlap_stdout.txt" 2> "$OUT/overlap_stderr.txt"
cd "$ROOT"
python3 scripts/overlap_summary.py "$OUT/overlap" "$OUT/overlap_stdout.txt" "$OUT/profiles/r5_overlap.json" 20 | tail -24
rm -rf "$OUT/overlap"
cp "$OUT/profiles/r5_overlap.json" "$ROOT/profiles/"
timeout -k 10 1000 bash scripts/record_pass.sh bench
