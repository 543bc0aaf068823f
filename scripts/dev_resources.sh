#!/usr/bin/env bash
# scripts/dev_resources.sh scripts/pcr_dev.hip [filter]: compile a one-kernel translation unit with the library's flags and print its kernels' resources
set -e
src=$1; out=build_var/$(basename ${src%.hip}).so
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fhip-fp32-correctly-rounded-divide-sqrt -fno-slp-vectorize -Wno-comment -Icppflow_amd/csrc -Iinclude -shared -o $out $src
python scripts/kernel_resources.py $out ${2:-} | cut -c1-160
