#!/bin/bash
# micro-batch sweep of the robot_nav PPO iteration: B = 16,384 samples in micro-batches of 4,096 / 8,192 / 16,384,
# then B = 4,096 in one micro-batch (the bench's nav sub-record).  usage (GPU box): bash tools/nav_sweep.sh <out file>
OUT=${1:-gpurun_out/nav_sweep.txt}
: > "$OUT"
for cap in 4096 8192 16384; do
  python3 tools/bench_nav.py 16384 $cap 3 | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('B=16384 cap=%d  ms/iter %.2f  per 4096 samples %.2f  samples/s %.0f' % ($cap, d['ms_per_ppo_iter_wall'], d['ms_per_ppo_iter_wall']/4, d['samples_per_s']))" >> "$OUT" 2>&1
done
python3 tools/bench_nav.py 4096 4096 3 >> "$OUT" 2>&1
cat "$OUT" | cut -c1-400
