"""Acting-forward latency (ddrl_forward, us per call) over batch sizes for a given build of the library:
    python tools/act_latency.py [path/to/libddrl_hip.so]"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ddrl4nav_amd import _lib  # noqa: E402

if len(sys.argv) > 1:
    _lib.LIB_PATH = os.path.abspath(sys.argv[1])
from ddrl4nav_amd.engine import HotPath  # noqa: E402
from ddrl4nav_amd.utils.recipe import make_weights, flatten  # noqa: E402

hp = HotPath(max_batch=2048)
hp.set_params(flatten(make_weights(0)))
rng = np.random.default_rng(0)
out = []
for n in (4, 8, 32, 64, 128, 256, 512, 1024, 2048):
    frames = torch.from_numpy(rng.integers(0, 256, size=(n, 4, 84, 84), dtype=np.uint8)).cuda()
    probs = torch.empty((n, 6), device="cuda")
    val, act, lp = (torch.empty(n, device="cuda") for _ in range(3))
    for _ in range(20):
        hp.forward(frames, seed=1, stream_id=0, probs=probs, value=val, action=act, logp=lp)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(200):
        hp.forward(frames, seed=1, stream_id=0, probs=probs, value=val, action=act, logp=lp)
    torch.cuda.synchronize()
    out.append("n=%d %.1f" % (n, (time.perf_counter() - t0) / 200 * 1e6))
print(os.path.basename(_lib.LIB_PATH), " ".join(out))
