#!/usr/bin/env python3
"""Which tensor drives the F21 (and F11 / F12 / F4) deviation ratios?  Test infrastructure: uses the oracle as the checker.

Per parameter tensor, for one PPO iteration on a fixture batch:
  * gradient error of the HIP path against the float64 oracle (under the kernels' leaky-ReLU decisions), rms and max, beside
    the same error of the oracle's own fp32 evaluation (= what the reference computes): ratio > 1 means "less accurate than an
    fp32 evaluation" for that tensor;
  * after the optimiser step: the three deviation ratios of tests/parity_util.py (L2, max-abs, direction) per tensor.
usage: python tools/diag_f21.py [pong|default|smooth]      (on the GPU box)
"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    mode = sys.argv[1] if len(sys.argv) > 1 else "pong"
    import parity_util as P
    from ddrl4nav_amd.engine import HotPath
    from ddrl4nav_amd.utils.recipe import flatten, make_weights
    from oracle import ddrl_oracle as O
    from test_gpu_parity import _adopt_kernel_decisions, _grad_views, _release_decisions, dev
    _, _, _, shared, smooth = P.MODES[mode]
    assert not shared
    frames, actions, old_logps, advs, rets = P.mode_batch(mode)
    n = frames.shape[0]
    hp = HotPath(max_batch=n, smooth_l1_loss=int(smooth)).keep_activations()
    hp.set_params(flatten(make_weights(0)))
    args = (dev(frames), dev(actions), dev(old_logps), dev(advs), dev(rets))
    hp.ppo_iter(*args)
    x = O.frames_to_f32(frames)
    net32 = O.OraclePPO()
    net32.load_weights(make_weights(0))
    _adopt_kernel_decisions(hp, net32, n, x)
    encs32 = [m for m in net32.modules() if isinstance(m, O.Encoder)]
    grads = {}
    for tag, dt in (("f64", torch.float64), ("f32", torch.float32)):
        net = O.OraclePPO()
        net.load_weights(make_weights(0))
        net.to(dt)
        for a, b in zip([m for m in net.modules() if isinstance(m, O.Encoder)], encs32):
            a.forced = b.forced
        t = lambda a: torch.from_numpy(np.asarray(a)).to(dt)
        net.zero_grad()
        _, al, vl, _ = O.ppo_losses(net, x.to(dt), t(actions), t(old_logps), t(advs), t(rets), smooth_l1=smooth)
        al.backward()
        vl.backward()
        grads[tag] = {k: p.grad.double().numpy().copy() for k, p in net.named_parameters()}
    _release_decisions(net32)
    got = _grad_views(hp)
    KEEP = ("actor.actor_linear.weight", "critic.pre.conv2.weight", "actor.pre.conv3.weight", "critic.critic_linear.weight")
    dump = {"ghip/" + k: got[k].copy() for k in KEEP}
    print("%-34s %10s %10s %8s | %10s %8s" % ("gradient vs float64", "rms hip", "rms fp32", "ratio", "max hip/|g|", "ratio"))
    for k, g64 in grads["f64"].items():
        eh = got[k].astype(np.float64) - g64
        e32 = grads["f32"][k] - g64
        rh, r32 = np.sqrt((eh ** 2).mean()), np.sqrt((e32 ** 2).mean())
        mh, m32 = np.abs(eh).max(), np.abs(e32).max()
        print("%-34s %10.3e %10.3e %8.2f | %10.3e %8.2f" % (k, rh, r32, rh / max(r32, 1e-300), mh / np.abs(g64).max(), mh / max(m32, 1e-300)))
    hp.clip_adam_step()
    traj = P.f64_trajectory(mode)
    flat = hp.params.cpu().numpy()
    gp = P.split_flat(flat, False)
    dump.update(tail=hp.grads[hp.n_params:].cpu().numpy().copy(), **{"p1/" + k: gp[k].copy() for k in KEEP},
                **{"g64/" + k: grads["f64"][k] for k in KEEP}, **{"g32/" + k: grads["f32"][k] for k in KEEP})
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    np.savez_compressed(os.path.join(ROOT, "gpurun_out", "diag_%s.npz" % mode), **dump)
    gotp = P.split_flat(np.asarray(flat, np.float64), False)
    ref = P.spread(mode)
    floor = {}
    for name in gotp:
        if float(ref["upd_l2/it1/" + name]) > 0:
            grp = name.split(".")[0]
            floor[grp] = max(floor.get(grp, 0.0), float(ref["ref_max/it1/" + name]))
    print("\n%-34s %8s %8s %8s   (after iteration 1; floor %s)" % ("deviation ratios", "l2", "max", "1-cos", floor))
    p64, p0 = traj["params"][1], traj["p0"]
    for name, a in gotp.items():
        k = "it1/" + name
        upd = float(ref["upd_l2/" + k])
        fl = floor[name.split(".")[0]]
        d, u, u64 = (a - p64[name]).ravel(), (a - p0[name]).ravel(), (p64[name] - p0[name]).ravel()
        cos = float(u @ u64 / (np.linalg.norm(u) * np.linalg.norm(u64) + 1e-300))
        print("%-34s %8.3f %8.3f %8.3f   flips %d of %d (ref 1-cos %.2e, ours %.2e)" % (
            name, np.sqrt(d @ d) / max(float(ref["ref_l2/" + k]), fl), np.abs(d).max() / fl,
            (1 - cos) / max(float(ref["ref_1mcos/" + k]), 0.5 * (fl / upd) ** 2), int((np.sign(u) != np.sign(u64)).sum()), u.size,
            float(ref["ref_1mcos/" + k]), 1 - cos))
    hp.close()


if __name__ == "__main__":
    main()
