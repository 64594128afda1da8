#!/usr/bin/env python3
"""Small driver for rocprofv3: a few PPO iterations at the bench shape (B = 65,536) and a few
acting forwards (n = 256), nothing else.  Usage: rocprofv3 ... -- python3 tools/profile_iter.py [iters] [acts] [B]"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from ddrl4nav_amd import _lib  # noqa: E402
if os.environ.get("DDRL_ABL_LIB"):
    _lib.LIB_PATH = os.environ["DDRL_ABL_LIB"]  # A/B builds under tools/_scratch_abl/
from ddrl4nav_amd.engine import HotPath  # noqa: E402
from ddrl4nav_amd.utils.recipe import flatten, make_weights  # noqa: E402

iters = int(sys.argv[1]) if len(sys.argv) > 1 else 2
acts = int(sys.argv[2]) if len(sys.argv) > 2 else 4
B = int(sys.argv[3]) if len(sys.argv) > 3 else 65536
torch.cuda.set_device(0)
hp = HotPath(max_batch=B)
hp.set_params(flatten(make_weights(0)))
g = torch.Generator(device="cuda")
g.manual_seed(1)
frames = torch.randint(0, 256, (B, 4, 84, 84), dtype=torch.uint8, device="cuda", generator=g)
actions = torch.randint(0, 6, (B,), device="cuda", generator=g).float()
old = torch.full((B,), -1.79, device="cuda")
adv = torch.randn(B, device="cuda", generator=g)
ret = torch.randn(B, device="cuda", generator=g)
for _ in range(iters):
    hp.ppo_iter(frames, actions, old, adv, ret)
    hp.clip_adam_step()
for t in range(acts):
    hp.forward(frames[:256], seed=1, stream_id=t)
torch.cuda.synchronize()
print("done", hp.stats())
