#!/usr/bin/env python3
"""How far do two EQUIVALENT fp32 evaluations of the reference's learner drift apart over ten iterations?

The full-batch losses are means over the batch, so a permutation of the samples leaves the mathematics of
`PPO.learn` (reference nn/ppo.py:77-146) unchanged and only changes the order of the fp32 summations (the batch
means, the weight-gradient reductions).  This script runs the REFERENCE on the f11 fixture (SMOOTH_L1_LOSS=True,
the most rounding-sensitive sequence: gradients of +-1/B per sample, Adam steps of O(lr) on noise-floor elements)
with the batch in three other orders and stores the loss trajectories:

  f11b_smooth_l1_reorder.npz   losses_perm [3, 10, 4]; per parameter tensor and for iterations 1 and 10, how the
                               strided parameter samples of the permuted runs differ from the stored run:
                               pbad/it<k>/<name> = largest fraction of elements further than 0.05 lr k + 1e-6 |w| away,
                               pmax/it<k>/<name> = largest absolute difference

tests/test_gpu_parity.py uses max(|fp32 - f64|, |fp32 - permuted fp32|) as the envelope of what "the same fp32
computation" means for that sequence.  Runs only in the build container (needs /root/reference).

Usage:  python tests/golden/make_golden_reorder.py
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
sys.path.insert(0, HERE)
import make_golden_shared as S  # noqa: E402


def main():
    from make_golden import _install_stubs
    _install_stubs()
    sys.path.insert(0, S.REF)
    from ddrl4nav_amd.utils.recipe import make_weights
    from USTC_lab.data import Experience
    torch.set_num_threads(1)
    f3 = np.load(os.path.join(HERE, "f3_loss.npz"))
    f11 = np.load(os.path.join(HERE, "f11_smooth_l1.npz"))
    B = f3["frames"].shape[0]
    x = (f3["frames"] / 255.0).astype(np.float32)
    runs, stats = [], {}
    for seed in (101, 102, 103):
        perm = np.random.default_rng(seed).permutation(B)
        net, _ = S.build_smooth(make_weights(seed=0))
        exp = Experience(states=[x[perm]], advs=f3["advs"][perm], actions=f3["actions"][perm],
                         old_logps=f3["old_logps"][perm], values=f11["rets"][perm].reshape(1, B))
        exp.to_tensor(dtype=torch.float32, device="cpu")
        po = {}
        runs.append(S.run_learn(net, exp, po, True))
        for key, val in po.items():
            if "/stride/" not in key:
                continue
            it, name = int(key.split("/")[0][2:]), key.split("/stride/")[1]
            lr = 5e-5 if name.startswith("actor.") else 1e-3
            want = f11[key]
            d = np.abs(val - want)
            bad = float((d > 0.05 * lr * it + 1e-6 * np.abs(want)).mean())
            stats["pbad/it%d/%s" % (it, name)] = max(stats.get("pbad/it%d/%s" % (it, name), 0.0), bad)
            stats["pmax/it%d/%s" % (it, name)] = max(stats.get("pmax/it%d/%s" % (it, name), 0.0), float(d.max()))
    runs = np.stack(runs)
    # sanity: the unpermuted order reproduces the committed trajectory bit for bit
    net, _ = S.build_smooth(make_weights(seed=0))
    exp = Experience(states=[x], advs=f3["advs"], actions=f3["actions"], old_logps=f3["old_logps"],
                     values=f11["rets"].reshape(1, B))
    exp.to_tensor(dtype=torch.float32, device="cpu")
    same = S.run_learn(net, exp, {}, False)
    assert np.array_equal(same, f11["losses"]), np.abs(same - f11["losses"]).max()
    np.savez(os.path.join(HERE, "f11b_smooth_l1_reorder.npz"), losses_perm=runs, **{k: np.float64(v) for k, v in stats.items()})
    for k in sorted(stats):
        if k.startswith("pbad/it10") and stats[k] > 0.02:
            print(k, round(stats[k], 3), "max", stats["pmax" + k[4:]])
    print("max |perm - ref| per iteration:", np.abs(runs - f11["losses"][None]).max(axis=(0, 2)))
    print("max |f64 - ref| per iteration :", np.abs(f11["losses_f64"] - f11["losses"]).max(axis=1))


if __name__ == "__main__":
    main()
