"""Same-box A/B: one PPO iteration over 65,536 samples as ONE launch set against 65,536 / mb launch sets of mb samples
("cache-blocked iteration" -- at mb <= 2,048 the a1 + dz1 round trips of a micro-batch fit the 256 MiB
Infinity Cache between producer and consumer).  Timing only: every micro-batch is a ddrl_ppo_iter(B = mb, B_global = 65,536)
on its slice of the same frames; the fixed-order accumulation of the gradients (13.5 MB per micro-batch, ~5 us) is not run.

    python tools/ab_microbatch.py [total] [mb ...]

Prints ms per `total` samples and the per-kernel sums (HIP events around every launch in a second pass)."""
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from ddrl4nav_amd.engine import HotPath  # noqa: E402
from ddrl4nav_amd.utils.recipe import flatten, make_weights  # noqa: E402


def run(total, mb, frames, acts, old, adv, ret, reps=3):
    hp = HotPath(max_batch=mb)
    hp.set_params(flatten(make_weights(0)))
    n = total // mb

    def one_pass():
        for i in range(n):
            s = slice(i * mb, (i + 1) * mb)
            hp.ppo_iter(frames[s], acts[s], old[s], adv[s], ret[s], b_global=total)

    one_pass()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    ts = []
    for _ in range(reps):
        e0.record()
        one_pass()
        e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1))
    hp.profile(True)
    one_pass()
    torch.cuda.synchronize()
    hp.profile(False)
    prof = {k: round(v[0], 3) for k, v in sorted(hp.profile_read().items(), key=lambda kv: -kv[1][0])}
    hp.close()
    del hp
    torch.cuda.empty_cache()
    return {"mb": mb, "launch_sets": n, "ms_per_total": round(float(np.median(ts)), 3), "ms_all": [round(t, 3) for t in ts],
            "kernels_ms": prof}


def main():
    total = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
    mbs = [int(a) for a in sys.argv[2:]] or [65536, 16384, 4096, 2048, 1024, 65536]
    torch.cuda.set_device(0)
    dev = torch.device("cuda:0")
    g = torch.Generator(device=dev)
    g.manual_seed(1234)
    frames = torch.randint(0, 256, (total, 4, 84, 84), dtype=torch.uint8, device=dev, generator=g)
    acts = torch.randint(0, 6, (total,), device=dev, generator=g).to(torch.float32)
    old = torch.full((total,), -1.79, dtype=torch.float32, device=dev)
    adv = torch.randn((total,), device=dev, generator=g)
    ret = torch.randn((total,), device=dev, generator=g)
    for mb in mbs:
        print(json.dumps(run(total, mb, frames, acts, old, adv, ret)), flush=True)


if __name__ == "__main__":
    main()
