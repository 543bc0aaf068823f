#!/usr/bin/env bash
# round 5, GPU call 34: the record pass (counters, kernel trace, overlap, bench records) on the final library (with the opt-in pacing)
set -eo pipefail
timeout -k 10 1150 bash scripts/record_pass.sh quick
