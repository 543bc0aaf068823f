#!/usr/bin/env bash
# round 5, GPU call 18: executed instruction counts of the queue build against CPPF_COLL_QUEUE=0 (planner inputs, random configurations)
set -o pipefail
ROOT="$(pwd)"; OUT="$ROOT/gpurun_out/r5"; mkdir -p "$OUT"; export TMPDIR=/tmp
F="$OUT/pmc_ab_coll_queue.txt"; : > "$F"
for lib in cppflow_amd/csrc/libcppflow_hip.so build_var/lib_noqueue.so ${EXTRA_LIBS:-}; do
  export CPPFLOW_HIP_LIB="$ROOT/$lib"
  echo "== $lib" | tee -a "$F"
  k=0
  for set in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES" "SQ_INSTS_LDS SQ_INSTS_SMEM SQ_INSTS_BRANCH SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVES" "SQ_INSTS_VALU_TRANS_F32 SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_WAVES" "SQ_IFETCH SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_WAVES"; do
    k=$((k+1)); D="$OUT/pc_$(basename $lib .so)_$k"; rm -rf "$D"
    (cd /tmp && timeout -k 10 300 rocprofv3 --pmc $set --output-format csv -d "$D" -o pmc -- python3 "$ROOT/scripts/pmc_probe_coll.py" > /dev/null 2> "$D.stderr") || { echo "   (pass '$set' failed: $(tail -1 $D.stderr | cut -c1-160))" | tee -a "$F"; continue; }
    python3 - "$D/pmc_counter_collection.csv" <<'PY' | tee -a "$F"
import csv, sys, collections
rows = [r for r in csv.DictReader(open(sys.argv[1])) if "lm_fused_kernel" in r["Kernel_Name"] and r["Grid_Size"] == "262144"]
by = collections.OrderedDict()
for r in rows:
    by.setdefault(int(r["Dispatch_Id"]), {})[r["Counter_Name"]] = float(r["Counter_Value"])
last = list(by.values())[-2:]
for label, c in zip(("planner inputs", "random configs "), last):
    w = c.get("SQ_WAVES", 4096.0)
    print(f"   {label}: " + "  ".join(f"{k[3:]} {v / w:9.1f}" for k, v in c.items() if k != "SQ_WAVES") + "   (per wavefront)")
PY
  done
done
rm -rf "$OUT"/pc_*
