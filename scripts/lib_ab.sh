# developer A/B of alternative builds of the library on ONE box: bash scripts/lib_ab.sh build_var/lib_x.so build_var/lib_y.so ...
# Alternates the in-tree library and the given ones (CPPFLOW_HIP_LIB, which skips the build-id check) three times over and prints
# the driver-style step time and the isolated kernel time of the default bench (C4) for each.
set -e
for rep in 1 2 3; do
for lib in cppflow_amd/csrc/libcppflow_hip.so "$@"; do
  echo -n "$lib  "
  CPPFLOW_HIP_LIB=$lib python bench.py --no-cpu-baseline --no-siblings --repeats 3 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('us/step %.2f   isolated kernel %.2f us' % (d['ms_per_step']*1e3, d['roofline']['kernel_ms']*1e3))"
done; done
