#!/usr/bin/env python3
"""F24: the reference's fp32 learners of the NON-ATARI nets under PyTorch's other CPU convolution backend, by IMPORTING THE REFERENCE
(the nav counterpart of make_golden_backend_spread.py, which says why: thread counts and batch orders leave oneDNN's per-sample
arithmetic untouched, so the stored spreads describe ONE fp32 implementation; torch's native convolution path -- oneDNN off -- is a
second one).

  f13_nav1d_gauss / f14_navped_shared   the reference's PPO.learn (nn/ppo.py:77-146) on the fixtures of make_golden_nav.py
  f22_gail_navped                        the reference's GAIL.learn (nn/GAIL.py:149-158) on the fixture of make_golden_gail_nav.py

For every net: the native backend at 1 and 8 threads and four batch orders under BOTH backends; stored under "<fixture>/" with the
keys of the fixtures' own spreads (ref_l2 / ref_max / ref_1mcos per tensor and snapshot, losses_variants [, d_loss_spread]);
tests/parity_util.py merges them with max().   -> tests/golden/f24_nav_backend_spread.npz
Usage: python tests/golden/make_golden_backend_spread_nav.py
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
sys.path.insert(0, HERE)
import make_golden_nav as NV  # noqa: E402
import make_golden_nav_spread as NS  # noqa: E402

N_ORDERS = 4


def _spread(out, prefix, p0, s64, variants, tags):
    for tag, key in tags:
        for name in p0:
            a64 = s64[key][name]
            u64 = (a64 - p0[name]).ravel()
            l2 = mx = omc = 0.0
            for sn in variants:
                d = (sn[key][name] - a64).ravel()
                l2, mx = max(l2, float(np.sqrt(d @ d))), max(mx, float(np.abs(d).max()))
                uv = (sn[key][name] - p0[name]).ravel()
                den = np.linalg.norm(uv) * np.linalg.norm(u64)
                omc = max(omc, 1.0 - float(uv @ u64 / den) if den > 0 else 0.0)
            kk = "%s/%s" % (tag, name)
            out["%s/ref_l2/%s" % (prefix, kk)], out["%s/ref_max/%s" % (prefix, kk)] = np.float64(l2), np.float64(mx)
            out["%s/ref_1mcos/%s" % (prefix, kk)] = np.float64(omc)


def ppo_nets(out, makers=None, names=("f13_nav1d_gauss", "f14_navped_shared")):
    from ddrl4nav_amd.utils.recipe import hash_weights
    from USTC_lab.data import Experience
    makers = NS.builders() if makers is None else makers
    for name in names:
        g = np.load(os.path.join(HERE, name + ".npz"))
        net, reopt, seed = makers[name]()
        weights = hash_weights([(k, tuple(p.shape)) for k, p in net.named_parameters()], seed)
        states = [g["state%d" % i] for i in range(len([k for k in g.files if k.startswith("state")]))]
        B = len(g["advs"])

        def run(dtype, threads, native, order=None):
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
            with torch.backends.mkldnn.flags(enabled=not native):
                for it, (ld, _, _) in enumerate(net.learn(e), 1):
                    rows.append([ld["PpoTotalLoss"], ld["ActorLoss"], ld["VLoss"], ld["EntLoss"]])
                    if it in (1, 10):
                        snaps[it] = {k: p.detach().double().numpy().copy() for k, p in net.named_parameters()}
            torch.set_num_threads(1)
            return np.asarray(rows, np.float64), snaps

        l32, _ = run(torch.float32, 1, False)
        assert np.array_equal(l32, g["losses"]), np.abs(l32 - g["losses"]).max()     # the committed fixture IS the oneDNN run
        _, s64 = run(torch.float64, 1, False)
        variants = [run(torch.float32, 1, True), run(torch.float32, 8, True)]
        for seed_o in range(N_ORDERS):
            perm = np.random.default_rng(2400 + seed_o).permutation(B)
            variants.append(run(torch.float32, 1, False, perm))
            variants.append(run(torch.float32, 1, True, perm))
        out[name + "/losses_variants"] = np.stack([l for l, _ in variants])
        _spread(out, name, {k: np.asarray(v, np.float64) for k, v in weights.items()}, s64, [s for _, s in variants],
                (("it1", 1), ("it10", 10)))
        print("  %-18s max |loss - oneDNN fp32| per iteration %s" % (
            name, " ".join("%.1e" % v for v in np.abs(out[name + "/losses_variants"] - g["losses"][None]).max((0, 2)))), flush=True)


def gail_navped(out):
    """F22 through the objects make_golden_gail_nav.py builds (same seeds, same inputs: taken from the committed fixture)."""
    import copy
    import make_golden_gail as G
    import make_golden_gail_nav as GN
    from ddrl4nav_amd.utils.recipe import hash_weights
    from USTC_lab.data import Experience
    from USTC_lab.nn import CategoricalActor, Critic, Discriminator, GAIL, PPO
    from USTC_lab.nn.nav_encoder import NavPedPreNet
    name = "f22_gail_navped"
    g = np.load(os.path.join(HERE, name + ".npz"))
    torch.set_num_threads(1)
    mimic_dir = G._write_mimic_dir(4)
    cfg, cfg_nn = G._configs({"discrete_action": True, "discrete_actions": list(range(GN.A))}, "classical", mimic_dir)
    prenet = NavPedPreNet(image_channel=1 + 3, last_output_dim=512)
    actor = CategoricalActor(action_output_dim=GN.A, device="cpu", last_input_dim=512, soft_max_grid=True, nn_dtype=torch.float32)
    critic = Critic(device="cpu", last_input_dim=512)
    gail_critic = copy.deepcopy(critic)
    ppo_net = PPO(actor, critic, prenet, None, cfg, cfg_nn).to("cpu")
    d_net = Discriminator(pre=copy.deepcopy(prenet), config=cfg, config_nn=cfg_nn).to("cpu")
    net = GAIL(generator=ppo_net, discriminator=d_net, gail_critic=gail_critic).to("cpu")
    names = [k for k, _ in net.named_parameters()]
    assert names == list(g["names"])
    weights = hash_weights([(k, tuple(p.shape)) for k, p in net.named_parameters()], GN.SEED)
    states_np = [g["state0"], g["state1"], g["state2"]]
    ex_np = [g["expert_state0"], g["expert_state1"], g["expert_state2"]]
    B = len(g["actions"])

    def run(dtype, threads, native, order=None):
        torch.set_num_threads(threads)
        G.load_weights(net, weights, dtype)
        net.discriminator.expert_data = [(GN.StateList([torch.from_numpy(s).to(dtype) for s in ex_np]), torch.from_numpy(g["expert_actions"]).to(dtype))]
        G.reset_optims(net, cfg_nn)
        idx = np.arange(B) if order is None else order
        e = Experience(states=[s[idx] for s in states_np], advs=g["advs"][idx], actions=g["actions"][idx], old_logps=g["old_logps"][idx],
                       values=g["rets"][:, idx])
        e.to_tensor(dtype=dtype, device="cpu")
        with torch.backends.mkldnn.flags(enabled=not native):
            res = G.run_gail_learn(net, e)
        torch.set_num_threads(1)
        return res

    d32, l32, _ = run(torch.float32, 1, False)
    assert np.array_equal(l32, g["losses"]) and np.array_equal(d32, g["d_loss"])
    d64, _, s64 = run(torch.float64, 1, False)
    variants = [run(torch.float32, 1, True), run(torch.float32, 8, True)]
    for seed_o in range(N_ORDERS):
        perm = np.random.default_rng(2420 + seed_o).permutation(B)
        variants.append(run(torch.float32, 1, False, perm))
        variants.append(run(torch.float32, 1, True, perm))
    out[name + "/losses_variants"] = np.stack([l for _, l, _ in variants])
    out[name + "/d_loss_spread"] = np.float64(max(np.abs(np.asarray([d for d, _, _ in variants]) - g["d_loss"][None]).max(), np.abs(d64 - g["d_loss"]).max()))
    _spread(out, name, {k: np.asarray(v, np.float64) for k, v in weights.items()}, s64, [s for _, _, s in variants],
            (("D1", "D1"), ("it1", 1), ("it10", 10)))
    print("  %-18s max |loss - oneDNN fp32| per iteration %s" % (
        name, " ".join("%.1e" % v for v in np.abs(out[name + "/losses_variants"] - g["losses"][None]).max((0, 2)))), flush=True)


def main():
    from make_golden import _install_stubs
    _install_stubs()
    sys.path.insert(0, NV.REF)
    out = {}
    ppo_nets(out)
    gail_navped(out)
    f = os.path.join(HERE, "f24_nav_backend_spread.npz")
    np.savez_compressed(f, **out)
    print("  f24_nav_backend_spread.npz %d B" % os.path.getsize(f))


if __name__ == "__main__":
    main()
