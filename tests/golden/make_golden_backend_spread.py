#!/usr/bin/env python3
"""F23: the reference's fp32 learner under PyTorch's OTHER CPU convolution backend, by IMPORTING THE REFERENCE.

Why.  The learn-sequence bounds are stated in the reference's own currency: how far do ITS fp32 evaluations sit from its float64
run (tests/golden/make_golden_spread*.py: thread counts, batch orders)?  Those variants all run torch's oneDNN convolutions, whose
per-sample arithmetic does not depend on threads or batch order -- fresh variants of that kind score 1.00 - 1.08 against the stored
ones (tests/parity_util.py) -- so they describe ONE fp32 implementation, not "an fp32 evaluation of the reference".  This script
runs the same reference objects (USTC_lab.nn.PPO.learn, nn/ppo.py:77-146) with `torch.backends.mkldnn.flags(enabled=False)`:
torch's native convolution path, i.e. what the reference computes on a torch build without oneDNN -- a second, independent fp32
implementation of the same mathematics.  Against the oneDNN variants it scores up to 2.7 on F21 (measured 2.69 L2 / 2.22 max /
2.69 angle on actor.pre.linear.weight), which is the distance between two fp32 implementations of the reference itself.

Stored per learner mode (default = F4 batch, shared = F10, smooth = F11, pong = F21, actions18 = F12), for the native backend at
1 and 8 threads, exactly the keys of the other spread fixtures under a "<mode>/" prefix:
    <mode>/ref_l2|ref_max|ref_1mcos/it<k>/<name>   largest deviation of the runs from the float64 run (k = 1, 10)
    <mode>/losses_variants [2 + 16, 10, 4]          their loss trajectories: native backend at 1 and 8 threads, then eight batch
                                                    orders under oneDNN and under the native backend each (one thread)
tests/parity_util.py merges them into a mode's spread with max() (as it does for the wide F4 fixture).
-> tests/golden/f23_backend_spread.npz.   Usage: python tests/golden/make_golden_backend_spread.py
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
sys.path.insert(0, HERE)
import make_golden_shared as S  # noqa: E402


N_ORDERS = 8


def _snap(net):
    return {k: p.detach().double().numpy().copy() for k, p in net.named_parameters()}


def _run(net, exp):
    losses, snaps = [], {}
    for it, (ld, _, _) in enumerate(net.learn(exp), 1):
        losses.append([ld["PpoTotalLoss"], ld["ActorLoss"], ld["VLoss"], ld["EntLoss"]])
        if it in (1, 10):
            snaps[it] = _snap(net)
    return np.asarray(losses, np.float64), snaps


def main():
    from make_golden import _install_stubs
    _install_stubs()
    sys.path.insert(0, S.REF)
    from ddrl4nav_amd.utils.recipe import make_weights, pong_frames
    from USTC_lab.data import Experience
    from USTC_lab.nn import AtariPreNet, CategoricalActor, Critic, PPO
    g = lambda n: np.load(os.path.join(HERE, n + ".npz"))
    f3, f4, f10, f11, f12, f21 = g("f3_loss"), g("f4_learn"), g("f10_shared"), g("f11_smooth_l1"), g("f12_actions18"), g("f21_pong_wide")
    x3 = (f3["frames"] / 255.0).astype(np.float32)                                   # f64 divide -> f32, as forward.py:102-104
    xp = (pong_frames(int(f21["frame_seed"]), f21["actions"].size) / 255.0).astype(np.float32)

    def two_adams(net, cfg_nn):
        net.actor_optim = torch.optim.Adam(net.actor.parameters(), cfg_nn.ACTOR_LEARNING_RATE)
        net.critic_optim = torch.optim.Adam(net.critic.parameters(), cfg_nn.CRITIC_LEARNING_RATE)

    def one_adam(net, cfg_nn):
        net.optim = torch.optim.Adam(net.parameters(), cfg_nn.LEARNING_RATE)

    def build_default(weights, n_actions=6):
        cfg, cfg_nn = S._cfg()
        actor = CategoricalActor(action_output_dim=n_actions, device="cpu", soft_max_grid=True, last_input_dim=512,
                                 pre=AtariPreNet(4, last_output_dim=512, device="cpu"), nn_dtype=torch.float32)
        critic = Critic(device="cpu", last_input_dim=512, pre=AtariPreNet(4, last_output_dim=512, device="cpu"))
        net = PPO(actor, critic, None, None, cfg, cfg_nn).to("cpu")
        net.load_state_dict({k: torch.from_numpy(v.copy()) for k, v in weights.items()}, strict=True)
        return net, cfg_nn

    modes = (  # mode, builder, weights, states, batch fixture, rets, reset, stored fp32 losses (oneDNN, one thread)
        ("default", build_default, make_weights(seed=0), x3, f4, f4["rets"], two_adams, f4["losses"]),
        ("shared", S.build_shared, make_weights(seed=0, shared=True), x3, f10, f10["rets"], one_adam, f10["losses"]),
        ("smooth", S.build_smooth, make_weights(seed=0), x3, f3, f11["rets"], two_adams, f11["losses"]),
        ("actions18", lambda w: build_default(w, 18), make_weights(seed=0, n_actions=18), x3, f12, f12["rets"], two_adams, f12["losses"]),
        ("pong", build_default, make_weights(seed=0), xp, f21, f21["rets"], two_adams, f21["losses"]),
    )
    out = {}
    for mode, build, weights, x, src, rets, reset, stored in modes:
        B = x.shape[0]

        def run(dtype, threads, native, perm=None):
            torch.set_num_threads(threads)
            net, cfg_nn = build(weights)
            net.to(dtype)
            reset(net, cfg_nn)
            net.update_time = 0
            idx = np.arange(B) if perm is None else perm
            e = Experience(states=[x[idx]], advs=src["advs"][idx], actions=src["actions"][idx], old_logps=src["old_logps"][idx],
                           values=rets[idx].reshape(1, B))
            e.to_tensor(dtype=dtype, device="cpu")
            with torch.backends.mkldnn.flags(enabled=not native):
                return _run(net, e)

        l32, _ = run(torch.float32, 1, False)
        assert np.array_equal(l32, stored), (mode, np.abs(l32 - stored).max())     # the committed fixture IS the oneDNN run
        l64, s64 = run(torch.float64, 1, False)
        variants = [run(torch.float32, 1, True), run(torch.float32, 8, True)]
        # ... and batch orders under BOTH backends (the wide-spread idea of make_golden_spread_wide.py: from the fourth iteration on
        # two fp32 evaluations drift apart, and a handful of variants under-states how far): N_ORDERS orders x 2 backends
        for seed in range(N_ORDERS):
            perm = np.random.default_rng(2300 + seed).permutation(B)
            variants.append(run(torch.float32, 1, False, perm))
            variants.append(run(torch.float32, 1, True, perm))
        torch.set_num_threads(1)
        p0 = {k: np.asarray(v, np.float64) for k, v in weights.items()}
        out[mode + "/losses_variants"] = np.stack([l for l, _ in variants])
        worst = (0.0, "")
        for it in (1, 10):
            for name in p0:
                a64 = s64[it][name]
                u64 = (a64 - p0[name]).ravel()
                l2 = mx = omc = 0.0
                for _, sn in variants:
                    d = (sn[it][name] - a64).ravel()
                    l2, mx = max(l2, float(np.sqrt(d @ d))), max(mx, float(np.abs(d).max()))
                    uv = (sn[it][name] - p0[name]).ravel()
                    den = np.linalg.norm(uv) * np.linalg.norm(u64)
                    omc = max(omc, 1.0 - float(uv @ u64 / den) if den > 0 else 0.0)
                kk = "it%d/%s" % (it, name)
                out["%s/ref_l2/%s" % (mode, kk)], out["%s/ref_max/%s" % (mode, kk)] = np.float64(l2), np.float64(mx)
                out["%s/ref_1mcos/%s" % (mode, kk)] = np.float64(omc)
        print("  %-10s native-backend runs: max |loss - oneDNN fp32| per iteration %s" % (
            mode, " ".join("%.1e" % v for v in np.abs(out[mode + "/losses_variants"] - stored[None]).max((0, 2)))), flush=True)
    f = os.path.join(HERE, "f23_backend_spread.npz")
    np.savez_compressed(f, **out)
    print("  f23_backend_spread.npz %d B" % os.path.getsize(f))


if __name__ == "__main__":
    main()
