#!/usr/bin/env python3
"""How far does the REFERENCE's own fp32 learner sit from its float64 evaluation, per parameter tensor?

The parameter-parity bound of the 10-step learn sequences has to be one that can fail, and it has to be
stated in the reference's own currency: after k Adam steps, two fp32 evaluations of the same mathematics
differ because an element whose gradient sits at the fp32 summation-noise floor moves by O(lr) either
way.  This script measures that with the reference itself (IMPORTED from /root/reference, build
container only): `PPO.learn` (nn/ppo.py:77-146) is run

  * in float64                       -> the "true" trajectory p64
  * in float32, 1 intra-op thread    -> must reproduce the committed fixture's losses bit for bit
  * in float32, 8 intra-op threads   -> another summation order
  * in float32 with the batch permuted (3 orders) -> the same full-batch means in yet another order

and for every parameter tensor and k in (1, 10) stores the LARGEST deviation of any fp32 variant from p64:

  ref_l2/it<k>/<name>     max_v || p_v - p64 ||_2
  ref_max/it<k>/<name>    max_v max | p_v - p64 |
  ref_1mcos/it<k>/<name>  max_v 1 - cos(p_v - p0, p64 - p0)      (direction of the accumulated update)
  upd_l2/it<k>/<name>     || p64 - p0 ||_2
  f64_sum / f64_l2 / f64_head/it<k>/<name>   float64 checksums of p64: they pin the oracle's float64 run
                          (tests/test_oracle_golden.py), which the GPU tests then use as p64 in full.

One file per learner mode: f4b_spread_default.npz (inputs of f3/f4), f10b_spread_shared.npz (f10),
f11c_spread_smooth.npz (f11).  The GPU tests assert  dev(HIP, p64) <= c * ref_dev  per tensor with the
constants c recorded in tests/golden/margins.json.

Usage:  python tests/golden/make_golden_spread.py
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
sys.path.insert(0, HERE)
import make_golden_shared as S  # noqa: E402


def _params(net):
    return {k: p.detach().double().numpy().copy() for k, p in net.named_parameters()}


def _run(net, exp, want_its=(1, 10)):
    losses, snaps = [], {}
    for it, (ld, _, _) in enumerate(net.learn(exp), 1):
        losses.append([ld["PpoTotalLoss"], ld["ActorLoss"], ld["VLoss"], ld["EntLoss"]])
        if it in want_its:
            snaps[it] = _params(net)
    return np.asarray(losses, np.float64), snaps


def spread(build, weights, make_exp, stored_losses, reset_optim, B):
    """build() -> (net, cfg_nn) fresh reference net; make_exp(dtype, perm) -> Experience."""
    def fresh(dtype):
        net, cfg_nn = build(weights)
        net.to(dtype)
        reset_optim(net, cfg_nn)
        net.update_time = 0
        return net

    p0 = {k: np.asarray(v, np.float64) for k, v in weights.items()}
    torch.set_num_threads(1)
    l32, s32 = _run(fresh(torch.float32), make_exp(torch.float32, None))
    assert np.array_equal(l32, stored_losses), np.abs(l32 - stored_losses).max()
    l64, s64 = _run(fresh(torch.float64), make_exp(torch.float64, None))
    variants = [s32]
    torch.set_num_threads(8)
    variants.append(_run(fresh(torch.float32), make_exp(torch.float32, None))[1])
    torch.set_num_threads(1)
    for seed in (101, 102, 103):
        perm = np.random.default_rng(seed).permutation(B)
        variants.append(_run(fresh(torch.float32), make_exp(torch.float32, perm))[1])
    out = {"losses_f64": l64}
    for it in (1, 10):
        for name in p0:
            a64 = s64[it][name]
            u64 = (a64 - p0[name]).ravel()
            l2 = mx = omc = 0.0
            for v in variants:
                d = (v[it][name] - a64).ravel()
                l2 = max(l2, float(np.sqrt(d @ d)))
                mx = max(mx, float(np.abs(d).max()))
                uv = (v[it][name] - p0[name]).ravel()
                c = float(uv @ u64 / (np.linalg.norm(uv) * np.linalg.norm(u64) + 1e-300))
                omc = max(omc, 1.0 - c)
            key = "it%d/%s" % (it, name)
            out["ref_l2/" + key] = np.float64(l2)
            out["ref_max/" + key] = np.float64(mx)
            out["ref_1mcos/" + key] = np.float64(omc)
            out["upd_l2/" + key] = np.float64(np.linalg.norm(u64))
            out["f64_sum/" + key] = np.float64(a64.sum())
            out["f64_l2/" + key] = np.float64(np.sqrt((a64 ** 2).sum()))
            out["f64_head/" + key] = a64.ravel()[:8].copy()
    return out


def main():
    from make_golden import _install_stubs
    _install_stubs()
    sys.path.insert(0, S.REF)
    from ddrl4nav_amd.utils.recipe import make_weights
    from USTC_lab.data import Experience
    f3 = np.load(os.path.join(HERE, "f3_loss.npz"))
    f4 = np.load(os.path.join(HERE, "f4_learn.npz"))
    f10 = np.load(os.path.join(HERE, "f10_shared.npz"))
    f11 = np.load(os.path.join(HERE, "f11_smooth_l1.npz"))
    B = f3["frames"].shape[0]
    x = (f3["frames"] / 255.0).astype(np.float32)  # f64 divide -> f32, as forward.py:102-104

    def exp_maker(src, rets):
        def make(dtype, perm):
            idx = np.arange(B) if perm is None else perm
            e = Experience(states=[x[idx]], advs=src["advs"][idx], actions=src["actions"][idx],
                           old_logps=src["old_logps"][idx], values=rets[idx].reshape(1, B))
            e.to_tensor(dtype=dtype, device="cpu")
            return e
        return make

    def two_adams(net, cfg_nn):
        net.actor_optim = torch.optim.Adam(net.actor.parameters(), cfg_nn.ACTOR_LEARNING_RATE)
        net.critic_optim = torch.optim.Adam(net.critic.parameters(), cfg_nn.CRITIC_LEARNING_RATE)

    def one_adam(net, cfg_nn):
        net.optim = torch.optim.Adam(net.parameters(), cfg_nn.LEARNING_RATE)

    def build_default(weights):
        """create_net's atari / SHARE_CNN_NET=False branch (runner/utils.py:122-134) with the default value loss."""
        from USTC_lab.nn import AtariPreNet, CategoricalActor, Critic, PPO
        cfg, cfg_nn = S._cfg()
        actor = CategoricalActor(action_output_dim=6, device="cpu", soft_max_grid=True, last_input_dim=512,
                                 pre=AtariPreNet(4, last_output_dim=512, device="cpu"), nn_dtype=torch.float32)
        critic = Critic(device="cpu", last_input_dim=512, pre=AtariPreNet(4, last_output_dim=512, device="cpu"))
        net = PPO(actor, critic, None, None, cfg, cfg_nn).to("cpu")
        net.load_state_dict({k: torch.from_numpy(v.copy()) for k, v in weights.items()}, strict=True)
        return net, cfg_nn

    jobs = (("f4b_spread_default.npz", build_default, make_weights(seed=0), exp_maker(f4, f4["rets"]), f4["losses"], two_adams),
            ("f10b_spread_shared.npz", S.build_shared, make_weights(seed=0, shared=True), exp_maker(f10, f10["rets"]),
             f10["losses"], one_adam),
            ("f11c_spread_smooth.npz", S.build_smooth, make_weights(seed=0), exp_maker(f3, f11["rets"]), f11["losses"], two_adams))
    for fname, build, weights, make_exp, stored, reset in jobs:
        out = spread(build, weights, make_exp, stored, reset, B)
        np.savez(os.path.join(HERE, fname), **out)
        worst = max((float(out[k]) / max(float(out["upd_l2/" + k[7:]]), 1e-300), k) for k in out if k.startswith("ref_l2/it10"))
        print("%-26s %7d B   largest ref_l2 / upd_l2 at it10: %.3g (%s)" % (fname, os.path.getsize(os.path.join(HERE, fname)),
                                                                          worst[0], worst[1]))


if __name__ == "__main__":
    main()
