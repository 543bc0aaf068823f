#!/usr/bin/env bash
# round 5, GPU call 19: the queue, second cut (constants in the items, packed cuboid masks): parity, A/B, counters
set -o pipefail
bash scripts/r5_calls/gpu_call17.sh && bash scripts/r5_calls/gpu_call18.sh
