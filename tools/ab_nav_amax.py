"""Same-box A/B of the robot_nav PPO iteration: per-sample magnitudes from the producers' epilogues (round 5) against a pre-pass in front
of every consumer (round 4's arrangement, nn/generic.py PRODUCER_AMAX = False).  One stream and two, per-operator times of the first.
    python tools/ab_nav_amax.py"""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench_nav  # noqa: E402
from ddrl4nav_amd.nn import generic  # noqa: E402

for rep in range(2):
    for flag in (True, False):
        generic.PRODUCER_AMAX = flag
        two = bench_nav.run(4096, 4096, 4)
        one = bench_nav.run(4096, 4096, 4, encoder_streams=False)
        ops = {k: v["ms_per_iter"] for k, v in list(one["ops"].items())[:9]}
        print(json.dumps({"producer_amax": flag, "ms_two_streams": two["ms_per_ppo_iter_wall"], "ms_one_stream": one["ms_per_ppo_iter_wall"],
                          "gemm_ops_ms": one["gemm_ops_ms_per_iter"], "ops": ops}), flush=True)
