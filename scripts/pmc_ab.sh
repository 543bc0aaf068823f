# developer A/B of executed instruction counts: bash scripts/pmc_ab.sh build_var/lib_x.so   (in-tree library vs the given one)
# One rocprofv3 --pmc pass per library over scripts/pmc_probe.py; prints SQ_INSTS_VALU / SQ_INSTS_SALU per wavefront of every variant.
set -e
ROOT="$(pwd)"; OUT="$ROOT/gpurun_out/pmc_ab"; mkdir -p "$OUT"; export TMPDIR=/tmp
k=0
for lib in cppflow_amd/csrc/libcppflow_hip.so "$@"; do
  k=$((k+1))
  export CPPFLOW_HIP_LIB="$ROOT/$lib"
  (cd /tmp && timeout -k 10 400 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVES SQ_WAVE_CYCLES --output-format csv -d "$OUT/p$k" -o pmc -- python3 "$ROOT/scripts/pmc_probe.py" panda > /dev/null 2> "$OUT/p${k}_stderr.txt")
  python3 scripts/trim_pmc.py "$OUT/p$k/pmc_counter_collection.csv" 14 > /dev/null
  echo "== $lib"
  python3 - "$OUT/p$k/pmc_counter_collection.csv" <<'PY'
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
by = collections.OrderedDict()
for r in rows:
    by.setdefault(int(r["Dispatch_Id"]), {})[r["Counter_Name"]] = float(r["Counter_Value"])
    by[int(r["Dispatch_Id"])]["k"] = r["Kernel_Name"][:60]
for i, (d, c) in enumerate(by.items()):
    w = c.get("SQ_WAVES", 1)
    print(f"  variant {'ABCDEFGHIJKLMN'[i] if i < 14 else i}  waves {w:6.0f}  VALU/wave {c['SQ_INSTS_VALU'] / w:9.1f}  SALU/wave {c['SQ_INSTS_SALU'] / w:7.1f}  wave quad-cycles/wave {c['SQ_WAVE_CYCLES'] / w:9.1f}")
PY
done
