#!/usr/bin/env python3
"""Per-iteration deviation of the generic nets' learn() from the reference trajectory + per-tensor ratios (GPU box)."""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import parity_util as P
from test_generic_gpu import _make, _states
from ddrl4nav_amd.data import Experience
name = sys.argv[1]
g = np.load(os.path.join(ROOT, "tests/golden/%s.npz" % name)); sp = P._load(name[:3] + "b_spread")
net, _ = _make(name)
exp = Experience(states=_states(g), advs=g["advs"], actions=g["actions"], old_logps=g["old_logps"], values=g["rets"].reshape(1, -1))
env = P.loss_envelope(g["losses"], sp["losses_f64"], g["losses_f32t8"], sp["losses_perm"], *([sp["losses_noise"]] if "losses_noise" in sp.files else []))
tr = P.nav_f64_trajectory(name)
for it, (li, ut, last) in enumerate(net.learn(exp), 1):
    got = np.array([li[k] for k in ("PpoTotalLoss", "ActorLoss", "VLoss", "EntLoss")])
    print("it %2d |got-ref| %s  |got-f64| %s  env %s" % (it, np.abs(got - g["losses"][it - 1]), np.abs(got - tr["losses"][it - 1]), env[it - 1]))
    if it in (1, 10):
        cur = {k: p.detach().cpu().numpy().astype(np.float64) for k, p in net.named_parameters()}
        for k, a in cur.items():
            kk = "it%d/%s" % (it, k)
            d = a - tr["params"][it][k]
            print("    %-28s upd %.2e  |d|2 %.2e (ref %.2e)  max %.2e (ref %.2e)  frac(|d|>3*refmax) %.4f" % (
                k, sp["upd_l2/" + kk], np.linalg.norm(d), sp["ref_l2/" + kk], np.abs(d).max(), sp["ref_max/" + kk],
                (np.abs(d) > 3 * sp["ref_max/" + kk]).mean()))
