#!/usr/bin/env bash
# round 5, GPU call 13: the bench records of the record pass on the final library's committed counters
set -eo pipefail
timeout -k 10 1100 bash scripts/record_pass.sh bench
