#!/usr/bin/env bash
# round 5, GPU call 14: the general lean angle path / wave-independence test, the driver's smoke() hook, one default bench line
set -eo pipefail
ROOT="$(pwd)"; OUT="$ROOT/gpurun_out/r5"; mkdir -p "$OUT"; export TMPDIR=/tmp
timeout -k 10 600 python -m pytest tests/test_gpu_lean_parity.py -m gpu -q 2>&1 | tail -12 | tee "$OUT/call14_pytest.txt"
timeout -k 10 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -3
timeout -k 10 300 python bench.py --gpus 1 --steps 20 --warmup 5 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('driver flags: us/step %.2f value %.3e frac %s' % (d['ms_per_step']*1e3, d['value'], d['roofline']['frac']))"
echo "== the driver's N > 1 launcher (torch.distributed.run) on the one GPU: two and four ranks sharing it, host-staged gloo (rehearsal)"
for n in 2 4; do
  CPPF_BENCH_SHARE_GPU=1 timeout -k 10 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node $n --master-addr 127.0.0.1 --master-port 2951$n bench.py --gpus $n --steps 20 --warmup 5 --no-siblings 2>"$OUT/torchrun_$n.err" | python -c "import json,sys; lines=[l for l in sys.stdin.read().splitlines() if l.startswith('{\"metric\"')]; assert len(lines)==1, lines; d=json.loads(lines[0]); print('torchrun $n ranks on one GPU: n_gpus', d['n_gpus'], 'scaling', d['scaling'], 'seeds/GPU', d['config']['seeds_per_gpu'], 'selection equals single process:', d['selection_check']['equals_single_process'], 'identical on ranks:', d['selection_check']['identical_on_every_rank'], 'plan_search candidates', d['plan_search']['candidates'])" | tee -a "$OUT/torchrun_rehearsal.txt"
done
