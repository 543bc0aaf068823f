#!/usr/bin/env bash
# A/B of the solver modes on one box: the headline workload and a 32 768-row strong-scaling shard (what the conditioning gate costs
# in throughput and in launch latency).  bash scripts/shard_ab.sh  (on the GPU box)
set -e
for sv in auto f32; do
python bench.py --solver $sv --no-cpu-baseline --no-siblings 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$sv C4            us/step %.2f' % (d['ms_per_step']*1e3), ' isolated kernel %.2f us' % (d['roofline']['kernel_ms']*1e3))"
python bench.py --seeds 128 --steps 2000 --warmup 200 --streams 4 --solver $sv --no-cpu-baseline --no-siblings 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$sv seeds/GPU 128  streams 4  us/step %.2f' % (d['ms_per_step']*1e3), ' isolated kernel %.2f us' % (d['roofline']['kernel_ms']*1e3))"
done
