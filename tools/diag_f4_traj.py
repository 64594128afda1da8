"""Per-iteration deviation of the HIP learner from the reference on the F4 learn fixture (B = 64, ten PPO iterations):
losses against the stored fp32 run and against the float64 run, beside the reference's own spread, and the parameter
deviation ratios of tests/parity_util.py after every iteration.  A/B of library builds through DDRL_ABL_LIB
(e.g. a -DDDRL_PLANES_BF16 build under tools/_scratch_abl/).  Test infrastructure: uses the oracle as the checker.
Usage: python tools/diag_f4_traj.py [tag] [mode: default|shared|smooth]"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from ddrl4nav_amd import _lib
if os.environ.get("DDRL_ABL_LIB"):
    _lib.LIB_PATH = os.environ["DDRL_ABL_LIB"]
from ddrl4nav_amd.engine import HotPath
from ddrl4nav_amd.utils.recipe import flatten, make_weights
import parity_util as P

tag = sys.argv[1] if len(sys.argv) > 1 else "head"
mode = sys.argv[2] if len(sys.argv) > 2 else "default"
fix = {"default": "f4_learn", "shared": "f10_shared", "smooth": "f11_smooth_l1"}[mode]
g4 = P._load(fix)
frames, actions, old_logps, advs, rets = P.mode_batch(mode)
shared, smooth = P.MODES[mode][3], P.MODES[mode][4]
dev = "cuda"
t = lambda a: torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32)).to(dev)
fr = torch.from_numpy(frames).to(dev)
hp = HotPath(max_batch=64, share_cnn_net=shared, smooth_l1_loss=smooth)
hp.set_params(flatten(make_weights(0, shared=shared)))
hp.reset_optimizer()
ref = g4["losses"]
f64 = g4["losses_f64"]
others = [g4[k] for k in ("losses_f64", "losses_f32t8") if k in g4.files]
env = P.mode_loss_envelope(mode, ref, *others)
traj = P.f64_trajectory(mode)
print("%s  mode %s  lib %s" % (tag, mode, _lib.LIB_PATH))
print("it | |got-ref| total actor v ent | envelope | ratio | |got-f64| | |ref-f64| | param ratios l2 max 1mcos")
for it in range(1, 11):
    hp.ppo_iter(fr, t(actions), t(old_logps), t(advs), t(rets))
    hp.clip_adam_step()
    s = hp.stats()
    got = np.asarray([s["PpoTotalLoss"], s["ActorLoss"], s["VLoss"], s["EntLoss"]], np.float64)
    d = np.abs(got - ref[it - 1])
    excess = d - (1e-5 * np.abs(ref[it - 1]) + 2e-6)
    ratio = float(np.max(excess / np.maximum(env[it - 1], 1e-12)))
    w = P.param_deviation(mode, it, hp.params.cpu().numpy()) if it in (1, 10) else {}
    print("%2d | %s | %s | %6.2f | %s | %s | %s" % (
        it, " ".join("%.2e" % x for x in d), " ".join("%.1e" % x for x in env[it - 1]), ratio,
        " ".join("%.2e" % x for x in np.abs(got - f64[it - 1])), " ".join("%.2e" % x for x in np.abs(ref[it - 1] - f64[it - 1])),
        " ".join("%s %.2f (%s)" % (k, v[0], v[1].replace("actor.", "a.").replace("critic.", "c.")) for k, v in w.items())))
hp.close()
