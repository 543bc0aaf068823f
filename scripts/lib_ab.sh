# developer A/B of alternative builds of the library on ONE box: bash scripts/lib_ab.sh build_var/lib_x.so build_var/lib_y.so ...
# (each run prints the default-build and the occupancy-capped step time of scripts/occ_ab.py for C4; the library under test is
# taken from CPPFLOW_HIP_LIB, the first line is the in-tree library)
set -e
for rep in 1 2; do
for lib in cppflow_amd/csrc/libcppflow_hip.so "$@"; do
  echo -n "$lib  "; CPPFLOW_HIP_LIB=$lib python scripts/occ_ab.py panda 1024 256 2>&1 | grep panda
done; done
