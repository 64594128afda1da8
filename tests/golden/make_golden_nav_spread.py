#!/usr/bin/env python3
"""The reference's own fp32 spread around its float64 run for the non-Atari nets (F13 / F14 / F15), per parameter
tensor -- the denominators of the falsifiable parameter bounds in tests/test_generic_gpu.py (same measures as
tests/golden/make_golden_spread.py, which documents them).  The nets, inputs and batches are those of
make_golden_nav.py (rebuilt here from the committed fixtures); every run is the REFERENCE's PPO.learn
(/root/reference/USTC_lab/nn/ppo.py:77-146), imported.

  f13b_spread.npz / f14b_spread.npz / f15b_spread.npz

Variants (round 3).  Thread counts and batch orders barely change what torch computes for a few dozen samples (the same
kernels run in the same order; only the final means differ), so five such variants under-state what "another fp32 evaluation of
the reference" is -- any implementation with another summation order inside its convolutions differs from the reference by a
few ulps in every activation.  The spread therefore also holds NOISE variants: the reference's own learn(), with every parameter
multiplied by (1 +- 4 x 2^-24) (independent random signs, seeded) before each iteration -- at most four fp32 roundings, the
error scale of one convolution's summation order.  On the round-2 fixtures (B = 18 / 20) ONE of two such variants left the stored
trajectory by 2e-4 in VLoss from the fifth iteration on (a discrete branch of the critic's lr = 1e-3 dynamics, like F4's), which
is what the HIP path's fitted limits of 190 - 56,000 had been absorbing; the fixtures now hold 64 samples and the envelope holds
those variants.

Usage: python tests/golden/make_golden_nav_spread.py
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
sys.path.insert(0, HERE)
import make_golden_nav as NV  # noqa: E402


def builders():
    from USTC_lab.nn import CategoricalActor, Critic, GaussionActor, PPO
    from USTC_lab.nn.mlp_encoder import MLPPreNet
    from USTC_lab.nn.nav_encoder import NavPedPreNet, NavPreNet1D

    def two(cfg_nn):
        def f(net):
            net.actor_optim = torch.optim.Adam(net.actor.parameters(), cfg_nn.ACTOR_LEARNING_RATE)
            net.critic_optim = torch.optim.Adam(net.critic.parameters(), cfg_nn.CRITIC_LEARNING_RATE)
        return f

    def f13():
        cfg, cfg_nn = NV.cfgs({"discrete_action": False, "act_dim": 2})
        actor = GaussionActor(action_output_dim=2, device="cpu", soft_max_grid=True, last_input_dim=512, nn_dtype=torch.float32,
                              pre=NavPreNet1D(image_channel=3, last_output_dim=512))
        critic = Critic(device="cpu", last_input_dim=512, pre=NavPreNet1D(image_channel=3, last_output_dim=512))
        return PPO(actor, critic, None, None, cfg, cfg_nn).to("cpu"), two(cfg_nn), 13

    def f14():
        cfg, cfg_nn = NV.cfgs({"discrete_action": True, "discrete_actions": list(range(5))})
        cfg_nn.SHARE_CNN_NET = True
        actor = CategoricalActor(action_output_dim=5, device="cpu", soft_max_grid=True, last_input_dim=512, nn_dtype=torch.float32)
        net = PPO(actor, Critic(device="cpu", last_input_dim=512), NavPedPreNet(image_channel=4, last_output_dim=512), None, cfg,
                  cfg_nn).to("cpu")

        def one(n):
            n.optim = torch.optim.Adam(n.parameters(), cfg_nn.LEARNING_RATE)
        return net, one, 14

    def f15():
        cfg, cfg_nn = NV.cfgs({"discrete_action": True, "discrete_actions": [0, 1]})
        actor = CategoricalActor(action_output_dim=2, device="cpu", soft_max_grid=True, last_input_dim=512, nn_dtype=torch.float32,
                                 pre=MLPPreNet(4, 512))
        critic = Critic(device="cpu", last_input_dim=512, pre=MLPPreNet(4, 512))
        return PPO(actor, critic, None, None, cfg, cfg_nn).to("cpu"), two(cfg_nn), 15

    return {"f13_nav1d_gauss": f13, "f14_navped_shared": f14, "f15_mlp_classical": f15}


def main():
    from make_golden import _install_stubs
    _install_stubs()
    sys.path.insert(0, NV.REF)
    spread_for(builders())


def spread_for(makers):
    """The spread fixtures <fNN>b_spread.npz of {fixture name: builder -> (net, reopt, recipe seed)} (make_golden_navpre.py
    calls this for F25)."""
    from ddrl4nav_amd.utils.recipe import hash_weights
    from USTC_lab.data import Experience
    for name, make in makers.items():
        g = np.load(os.path.join(HERE, name + ".npz"))
        net, reopt, seed = make()
        weights = hash_weights([(k, tuple(p.shape)) for k, p in net.named_parameters()], seed)
        states = [g["state%d" % i] for i in range(len([k for k in g.files if k.startswith("state")]))]
        B = len(g["advs"])

        def run(dtype, threads, order=None, noise_seed=None, noise_ulps=4.0):
            torch.set_num_threads(threads)
            net.to(torch.float32)
            net.load_state_dict({k: torch.from_numpy(v.copy()) for k, v in weights.items()}, strict=True)
            net.to(dtype)
            net.update_time = 0
            reopt(net)
            idx = np.arange(B) if order is None else order
            e = Experience(states=[s[idx].copy() for s in states], advs=g["advs"][idx], actions=g["actions"][idx],
                           old_logps=g["old_logps"][idx], values=g["rets"][idx].reshape(1, B))
            e.to_tensor(dtype=dtype, device="cpu")
            rows, snaps = [], {}
            gen_noise = None if noise_seed is None else torch.Generator().manual_seed(noise_seed)
            learner = net.learn(e)
            for it in range(1, 11):
                if gen_noise is not None:   # <= 4 fp32 roundings on every parameter before the iteration reads it
                    with torch.no_grad():
                        for p in net.parameters():
                            sign = torch.randint(0, 2, p.shape, generator=gen_noise).to(p.dtype) * 2 - 1
                            p.mul_(1 + sign * noise_ulps * 2.0 ** -24)
                ld, _, _ = next(learner)
                rows.append([ld["PpoTotalLoss"], ld["ActorLoss"], ld["VLoss"], ld["EntLoss"]])
                if it in (1, 10):
                    snaps[it] = {k: p.detach().double().numpy().copy() for k, p in net.named_parameters()}
            assert next(learner, None) is None
            torch.set_num_threads(1)
            return np.asarray(rows, np.float64), snaps

        l32, s32 = run(torch.float32, 1)
        assert np.array_equal(l32, g["losses"]), np.abs(l32 - g["losses"]).max()
        l64, s64 = run(torch.float64, 1)
        variants = [s32, run(torch.float32, 8)[1]]
        perms = []
        for ps in (301, 302, 303):
            lp, sp = run(torch.float32, 1, np.random.default_rng(ps).permutation(B))
            variants.append(sp)
            perms.append(lp)
        noisy = []
        for ns in (401, 402, 403, 404, 405, 406):
            ln, sn = run(torch.float32, 8, noise_seed=ns)
            variants.append(sn)
            noisy.append(ln)
            print("  %s noise seed %d: max |loss - f64| per iteration %s" % (name, ns, " ".join("%.1e" % v for v in np.abs(ln - l64).max(1))),
                  flush=True)
        out = {"losses_f64": l64, "losses_perm": np.stack(perms), "losses_noise": np.stack(noisy)}
        p0 = {k: np.asarray(v, np.float64) for k, v in weights.items()}
        for it in (1, 10):
            for k in p0:
                a64 = s64[it][k]
                u64 = (a64 - p0[k]).ravel()
                l2 = mx = omc = 0.0
                for v in variants:
                    d = (v[it][k] - a64).ravel()
                    l2, mx = max(l2, float(np.sqrt(d @ d))), max(mx, float(np.abs(d).max()))
                    uv = (v[it][k] - p0[k]).ravel()
                    den = np.linalg.norm(uv) * np.linalg.norm(u64)
                    omc = max(omc, 1.0 - float(uv @ u64 / den) if den > 0 else 0.0)
                kk = "it%d/%s" % (it, k)
                out["ref_l2/" + kk], out["ref_max/" + kk], out["ref_1mcos/" + kk] = np.float64(l2), np.float64(mx), np.float64(omc)
                out["upd_l2/" + kk] = np.float64(np.linalg.norm(u64))
                out["f64_l2/" + kk] = np.float64(np.sqrt((a64 ** 2).sum()))
                out["f64_head/" + kk] = a64.ravel()[:8].copy()
        fn = name[:3] + "b_spread.npz"
        np.savez(os.path.join(HERE, fn), **out)
        print("  %-18s %7d B" % (fn, os.path.getsize(os.path.join(HERE, fn))))


if __name__ == "__main__":
    main()
