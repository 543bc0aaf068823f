#!/usr/bin/env bash
# round 5, GPU call 38: the bench records again (one_stream = the one-stream engine's default, paced; the unpaced figure beside it)
set -eo pipefail
timeout -k 10 1150 bash scripts/record_pass.sh bench
