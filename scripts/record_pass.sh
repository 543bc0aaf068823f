#!/usr/bin/env bash
# Round-end measurement pass on the MI355X box (run from the repo root through gpurun); everything lands in gpurun_out/rec/.
#   bash scripts/record_pass.sh [full|quick|pmc|side|bench]
# Order matters: the PROFILER passes and scripts/summarize_profiles.py come FIRST and write profiles/<round>_issue.json,
# _traffic.json and _fused_kernel_stats.csv into the box's own copy of the tree (and into gpurun_out/rec/profiles/ for the way
# back), so that every bench.py record taken AFTERWARDS in the same pass finds the executed-instruction counters of its own launch
# shape and library build and reports roofline.frac on the executed basis (the round-3 pass ran the siblings first and
# all of them fell back to the flop model).  Counter passes are separate rocprofv3 runs with --pmc only (no trace domains), as
# MI355X_MICROARCH.md prescribes.
#   full  = tests + pmc + bench + side      quick = pmc + bench       pmc = profiler passes + summary only
#   bench = the bench records only (expects profiles/<round>_* of this build to be there already)      side = tests + slow side benches
set -eo pipefail
ROOT="$(pwd)"
OUT="$ROOT/gpurun_out/rec"
ROUND="${CPPF_ROUND:-r5}"
mkdir -p "$OUT"
export TMPDIR=/tmp
MODE="${1:-full}"
want() { case " $* " in *" $MODE "*) return 0 ;; esac; return 1; }

if want full side; then
    echo "== pytest -m gpu"
    timeout -k 10 1100 python -m pytest tests -m gpu -q 2>&1 | tail -3 | tee "$OUT/pytest_gpu.txt"
fi

if want full quick pmc; then
    cd /tmp
    echo "== rocprofv3 kernel trace of the headline command (one launch in flight)"
    timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/kt" -o kt -- python3 "$ROOT/bench.py" --steps 200 --warmup 10 --no-cpu-baseline --no-siblings --streams 1 --pace off > "$OUT/kt_stdout.txt" 2> "$OUT/kt_stderr.txt"
    echo "== rocprofv3 pmc FETCH_SIZE / WRITE_SIZE"
    timeout -k 10 600 rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$OUT/pmc_fetch" -o pmc -- python3 "$ROOT/bench.py" --steps 5 --warmup 2 --prewarm-ms 0 --no-cpu-baseline --no-siblings --streams 1 > /dev/null 2> "$OUT/pmc_fetch_stderr.txt"
    timeout -k 10 600 rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$OUT/pmc_write" -o pmc -- python3 "$ROOT/bench.py" --steps 5 --warmup 2 --prewarm-ms 0 --no-cpu-baseline --no-siblings --streams 1 > /dev/null 2> "$OUT/pmc_write_stderr.txt"
    echo "== rocprofv3 pmc SQ counters over scripts/pmc_probe.py (four passes)"
    timeout -k 10 600 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVES SQ_BUSY_CYCLES SQ_INSTS_MFMA --output-format csv -d "$OUT/pmc_p1" -o pmc -- python3 "$ROOT/scripts/pmc_probe.py" panda > /dev/null 2> "$OUT/pmc_p1_stderr.txt"
    python3 "$ROOT/scripts/trim_pmc.py" "$OUT/pmc_p1/pmc_counter_collection.csv" 18
    timeout -k 10 600 rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAVE_CYCLES --output-format csv -d "$OUT/pmc_p2" -o pmc -- python3 "$ROOT/scripts/pmc_probe.py" panda > /dev/null 2> "$OUT/pmc_p2_stderr.txt"
    python3 "$ROOT/scripts/trim_pmc.py" "$OUT/pmc_p2/pmc_counter_collection.csv" 18
    timeout -k 10 600 rocprofv3 --pmc SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 --output-format csv -d "$OUT/pmc_p3" -o pmc -- python3 "$ROOT/scripts/pmc_probe.py" panda > /dev/null 2> "$OUT/pmc_p3_stderr.txt"
    python3 "$ROOT/scripts/trim_pmc.py" "$OUT/pmc_p3/pmc_counter_collection.csv" 18
    timeout -k 10 600 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_VALU_MFMA_COEXEC_CYCLES SQ_INSTS_VALU_MFMA_F32 SQ_WAIT_ANY --output-format csv -d "$OUT/pmc_p4" -o pmc -- python3 "$ROOT/scripts/pmc_probe.py" panda > /dev/null 2> "$OUT/pmc_p4_stderr.txt"
    python3 "$ROOT/scripts/trim_pmc.py" "$OUT/pmc_p4/pmc_counter_collection.csv" 18
    echo "== rocprofv3 kernel trace of the DRIVER's command (two streams): the overlap behind ms_per_step < kernel time"
    timeout -k 10 600 rocprofv3 --kernel-trace --output-format csv -d "$OUT/overlap" -o kt -- python3 "$ROOT/bench.py" --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-siblings > "$OUT/overlap_stdout.txt" 2> "$OUT/overlap_stderr.txt"
    cd "$ROOT"
    mkdir -p "$OUT/profiles"
    python3 scripts/overlap_summary.py "$OUT/overlap" "$OUT/overlap_stdout.txt" "$OUT/profiles/${ROUND}_overlap.json" 20 | tail -22
    rm -rf "$OUT/overlap"
    echo "== counters -> profiles/${ROUND}_* (this box's tree: the bench records below read them) and gpurun_out/rec/profiles/"
    mkdir -p "$OUT/profiles"
    python3 scripts/summarize_profiles.py "$OUT" "$OUT/profiles" "$ROUND" counters > "$OUT/summarize.txt" 2>&1 || { tail -20 "$OUT/summarize.txt"; exit 1; }
    cp "$OUT/profiles/${ROUND}_"* "$ROOT/profiles/"
    tail -25 "$OUT/summarize.txt"
fi

if want full quick bench; then
    echo "== bench, default flags (N = 1, C4, 2000 steps, + one_stream / random_inputs siblings, + cpu baselines)"
    timeout -k 10 600 python bench.py > "$OUT/bench.json" 2> "$OUT/bench.err"
    cat "$OUT/bench.json"
    echo "== bench with the DRIVER's flags (--steps 20 --warmup 5)"
    timeout -k 10 600 python bench.py --gpus 1 --steps 20 --warmup 5 > "$OUT/bench_driverflags.json" 2>> "$OUT/bench.err"
    python3 -c "import json; d=json.load(open('$OUT/bench_driverflags.json')); r=d['roofline']; print('driver flags: us/step %.2f  value %.3e  frac %s  kernel %.1f us  overlapped %s  cpu agreement %s' % (d['ms_per_step']*1e3, d['value'], r['frac'], r['kernel_ms']*1e3, r['kernel_ms_overlapped'], d['cpu_baseline']['agreement']['max_abs_pos_err_diff_m']))"
    echo "== one rank's N = 8 shard (128 seeds) with the one-rank RCCL exchange on: driver flags, then 2000 steps"
    CPPF_BENCH_FORCE_DIST=1 timeout -k 10 300 python bench.py --gpus 1 --seeds 128 --steps 20 --warmup 5 --no-cpu-baseline > "$OUT/bench_shard128_driverflags.json" 2>> "$OUT/bench.err"
    CPPF_BENCH_FORCE_DIST=1 timeout -k 10 300 python bench.py --gpus 1 --seeds 128 --steps 2000 --warmup 100 --no-cpu-baseline --no-siblings > "$OUT/bench_shard128_2000steps.json" 2>> "$OUT/bench.err"
    for f in bench_shard128_driverflags bench_shard128_2000steps; do
        python3 -c "import json; d=json.load(open('$OUT/$f.json')); c=d['config']; print('$f: us/step %.2f (calibrated streams: %s)  steps/launch %s  steps/allgather %s  streams %s  frac %s  allgather %.1f us' % (d['ms_per_step']*1e3, d.get('ms_per_step_calibrated_streams'), c['steps_per_launch'], c['steps_per_allgather'], c['streams'], d['roofline']['frac'], d['rccl']['allgather_latency_us']))"
    done
    echo "== bench, one-rank RCCL group at full size (collective + seed selection on the dependency path)"
    CPPF_BENCH_FORCE_DIST=1 timeout -k 10 300 python bench.py --no-cpu-baseline --no-siblings > "$OUT/bench_dist1.json" 2>> "$OUT/bench.err"
    echo "== two ranks sharing the one GPU (host-staged gloo: choreography rehearsal, NOT a multi-GPU result), driver flags"
    CPPF_BENCH_SHARE_GPU=1 timeout -k 10 300 python bench.py --gpus 2 --steps 20 --warmup 5 --no-siblings > "$OUT/bench_2ranks_driverflags.json" 2>> "$OUT/bench.err"
    for c in C2 C3 C5; do
        echo "== bench --config $c"
        timeout -k 10 300 python bench.py --no-cpu-baseline --no-siblings --config $c > "$OUT/bench_$c.json" 2>> "$OUT/bench.err"
        python3 -c "import json; d=json.load(open('$OUT/bench_$c.json')); r=d['roofline']; print('$c: us/step %.2f  steps/launch %s  frac %s  at step rate %s' % (d['ms_per_step']*1e3, d['config']['steps_per_launch'], r['frac'], r['at_step_rate'].get('frac')))"
    done
    echo "== the other solver modes (what the gate costs)"
    timeout -k 10 300 python bench.py --no-cpu-baseline --no-siblings --solver f64 --steps 300 > "$OUT/bench_solver_f64.json" 2>> "$OUT/bench.err"
    timeout -k 10 300 python bench.py --no-cpu-baseline --solver f32 > "$OUT/bench_solver_f32.json" 2>> "$OUT/bench.err"
    echo "== strong-scaling shards on one GPU (what each of 2 / 4 / 8 GPUs runs), batched launches vs one step per launch on 4 streams"
    rm -f "$OUT/shard_streams.txt"
    for s in 512 256 128; do for b in 0 1; do
        timeout -k 10 200 python bench.py --seeds $s --steps 1000 --warmup 100 --batch $b --no-cpu-baseline --no-siblings 2>> "$OUT/bench.err" | python -c "import json,sys; d=json.loads(sys.stdin.read()); c=d['config']; print('seeds/GPU', c['seeds_per_gpu'], ' steps/launch', c['steps_per_launch'], ' streams', c['streams'], ' graphs', c['hip_graphs'][:3], ' us/step %.2f' % (d['ms_per_step']*1e3), ' isolated launch %.2f us' % (d['roofline']['kernel_ms']*1e3), ' host %.1f us/step' % c['host_enqueue_us_per_step'])" >> "$OUT/shard_streams.txt"
    done; done
    for g in 16 64; do
        CPPF_BENCH_FORCE_DIST=1 timeout -k 10 200 python bench.py --seeds 128 --steps 2048 --warmup 256 --gather-every $g --no-cpu-baseline --no-siblings 2>> "$OUT/bench.err" | python -c "import json,sys; d=json.loads(sys.stdin.read()); c=d['config']; print('seeds/GPU 128 + RCCL(1 rank) + select: steps/launch', c['steps_per_launch'], ' steps/allgather', c['steps_per_allgather'], ' streams', c['streams'], ' us/step %.2f' % (d['ms_per_step']*1e3), ' host %.1f us/step' % c['host_enqueue_us_per_step'])" >> "$OUT/shard_streams.txt"
    done
    cat "$OUT/shard_streams.txt"
    echo "== which rows the gate flags, per iteration (bench inputs and independent random configurations)"
    timeout -k 10 200 python scripts/gate_census.py > "$OUT/gate_census_problem.txt" 2>> "$OUT/bench.err"
    timeout -k 10 200 python scripts/gate_census.py --inputs random > "$OUT/gate_census_random.txt" 2>> "$OUT/bench.err"
    echo "== dp_search / coupled step"
    timeout -k 10 400 python scripts/dp_bench.py > "$OUT/dp_bench.txt" 2>&1
    timeout -k 10 300 python scripts/coupled_bench.py > "$OUT/coupled_bench.txt" 2>&1
    echo "== shard_bench (isolated latency by kernel shape), K sweep"
    timeout -k 10 300 python scripts/shard_bench.py --shapes row,quad --sizes 8,32,64,128,256,512,1024 > "$OUT/shard_bench.txt" 2>&1
    timeout -k 10 200 python scripts/ksweep.py > "$OUT/ksweep.txt" 2>&1
fi

if want full side; then
    echo "== kbench"
    timeout -k 10 600 python scripts/kbench.py > "$OUT/kbench.txt" 2>&1
    timeout -k 10 300 python scripts/kbench_small.py > "$OUT/kbench_small.txt" 2>&1
    echo "== launch model"
    timeout -k 10 300 python scripts/launch_model.py > "$OUT/launch_model.txt" 2>&1
    echo "== run-time specialisation"
    timeout -k 10 300 python scripts/rtc_bench.py > "$OUT/rtc_bench.txt" 2>&1
    echo "== MFMA variants of the quad shape"
    timeout -k 10 300 python scripts/shard_bench.py --shapes quad --mfma 1 --sizes 8,32,64 > "$OUT/shard_bench_mfma.txt" 2>&1
    { for m in 0 1; do echo "== chain12 quad, mfma=$m"; timeout -k 10 200 python scripts/shard_bench.py --robot chain12 --shapes quad --mfma $m --sizes 8,32,64 2>&1 | tail -4; done; echo "== chain12 row"; timeout -k 10 200 python scripts/shard_bench.py --robot chain12 --shapes row --sizes 8,32,64 2>&1 | tail -4; } > "$OUT/mfma_chain12.txt" 2>&1
    echo "== rocprofv3 kernel trace of the coupled step and dp_search (per kernel and grid size)"
    cd /tmp
    timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d "$OUT/kt_coupled" -o kt -- python3 "$ROOT/scripts/coupled_bench.py" --rounds 2 --seeds 1,256,1024 > /dev/null 2> "$OUT/kt_coupled_stderr.txt"
    timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d "$OUT/kt_dp" -o kt -- python3 "$ROOT/scripts/dp_bench.py" > /dev/null 2> "$OUT/kt_dp_stderr.txt"
    { python3 "$ROOT/scripts/trace_by_grid.py" "$OUT/kt_coupled" full_; python3 "$ROOT/scripts/trace_by_grid.py" "$OUT/kt_dp" dp_ | grep -v dp_step_kernel; } > "$OUT/coupled_dp_kernels.txt" 2>&1
    rm -rf "$OUT/kt_coupled" "$OUT/kt_dp"
    cd "$ROOT"
    if [ -x build_var/lone_wave ]; then timeout -k 5 200 ./build_var/lone_wave > "$OUT/lone_wave_micro.txt" 2>&1; fi
    if [ -x build_var/valu_rate ]; then timeout -k 5 120 ./build_var/valu_rate > "$OUT/valu_issue_rate_calibration.txt" 2>&1; fi
fi
# keep the merge small: drop anything big that is not a csv / json / txt summary
find "$OUT" -type f -size +30M -delete
find "$OUT" -type f | xargs ls -la | awk '{print $5, $9}'
echo "== done ($MODE)"
