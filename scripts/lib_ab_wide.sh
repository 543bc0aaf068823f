# developer A/B of an alternative library build across the side paths (one box): bash scripts/lib_ab_wide.sh build_var/lib_x.so
set -e
for lib in cppflow_amd/csrc/libcppflow_hip.so "$1"; do
  echo "== $lib"
  CPPFLOW_HIP_LIB=$lib python scripts/coupled_bench.py --robots panda --seeds 1,1024 --rounds 3 2>&1 | grep "S="  | cut -c1-110
  CPPFLOW_HIP_LIB=$lib python scripts/shard_bench.py --shapes row,quad --sizes 32,128,1024 2>&1 | grep -v amdgpu | tail -8
  CPPFLOW_HIP_LIB=$lib python bench.py --seeds 128 --steps 1000 --warmup 128 --no-cpu-baseline --no-siblings 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('shard 32768 rows  us/step %.2f' % (d['ms_per_step']*1e3))"
  CPPFLOW_HIP_LIB=$lib python bench.py --config C2 --no-cpu-baseline --no-siblings 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('C2  us/step %.2f' % (d['ms_per_step']*1e3))"
  CPPFLOW_HIP_LIB=$lib python bench.py --config C5 --steps 60 --warmup 6 --no-cpu-baseline --no-siblings 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('C5  us/step %.2f' % (d['ms_per_step']*1e3))"
  CPPFLOW_HIP_LIB=$lib python bench.py --solver f64 --no-cpu-baseline --no-siblings 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('f64 us/step %.2f' % (d['ms_per_step']*1e3))"
done
