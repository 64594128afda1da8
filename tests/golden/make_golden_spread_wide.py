#!/usr/bin/env python3
"""The REFERENCE's own fp32 spread on the F4 learn fixture over MANY of its evaluation orders (imports /root/reference,
build container only; stores outputs, copies nothing).

make_golden_spread.py samples five fp32 variants (1 / 8 threads, three batch orders).  F4 is a chaotic case -- the critic
steps with lr 1e-3 on 64 samples and differences between two fp32 evaluations grow about tenfold per iteration from the
fourth on -- so the largest deviation of FIVE variants is a noisy estimate of what "another fp32 evaluation of the reference"
looks like.  This script runs the reference's `PPO.learn` in fp32 under V = 2 thread counts x 16 batch orders and stores

  losses_variants   [V, 10, 4]   PpoTotalLoss / ActorLoss / VLoss / EntLoss per iteration of every variant
  losses_f64        [10, 4]      its float64 run (the yardstick)
  ref_l2 / ref_max / ref_1mcos / upd_l2 / it<k> / <tensor>   as make_golden_spread.py, maximum over the V variants
  var_l2/it10       [V, n_tensors]  per-variant L2 deviation of every tensor from the float64 run (distribution, not only max)

-> tests/golden/f4d_spread_wide.npz.  Usage: python tests/golden/make_golden_spread_wide.py [n_orders]
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
sys.path.insert(0, HERE)
import make_golden_shared as S  # noqa: E402
import make_golden_spread as SP  # noqa: E402


def main():
    n_orders = int(sys.argv[1]) if len(sys.argv) > 1 else 16
    from make_golden import _install_stubs
    _install_stubs()
    sys.path.insert(0, S.REF)
    from ddrl4nav_amd.utils.recipe import make_weights
    from USTC_lab.data import Experience
    from USTC_lab.nn import AtariPreNet, CategoricalActor, Critic, PPO
    f3 = np.load(os.path.join(HERE, "f3_loss.npz"))
    f4 = np.load(os.path.join(HERE, "f4_learn.npz"))
    B = f3["frames"].shape[0]
    x = (f3["frames"] / 255.0).astype(np.float32)
    weights = make_weights(seed=0)
    p0 = {k: np.asarray(v, np.float64) for k, v in weights.items()}

    def make_exp(dtype, perm):
        idx = np.arange(B) if perm is None else perm
        e = Experience(states=[x[idx]], advs=f4["advs"][idx], actions=f4["actions"][idx], old_logps=f4["old_logps"][idx],
                       values=f4["rets"][idx].reshape(1, B))
        e.to_tensor(dtype=dtype, device="cpu")
        return e

    def fresh(dtype):
        cfg, cfg_nn = S._cfg()
        actor = CategoricalActor(action_output_dim=6, device="cpu", soft_max_grid=True, last_input_dim=512,
                                 pre=AtariPreNet(4, last_output_dim=512, device="cpu"), nn_dtype=torch.float32)
        critic = Critic(device="cpu", last_input_dim=512, pre=AtariPreNet(4, last_output_dim=512, device="cpu"))
        net = PPO(actor, critic, None, None, cfg, cfg_nn).to("cpu")
        net.load_state_dict({k: torch.from_numpy(v.copy()) for k, v in weights.items()}, strict=True)
        net.to(dtype)
        net.actor_optim = torch.optim.Adam(net.actor.parameters(), cfg_nn.ACTOR_LEARNING_RATE)
        net.critic_optim = torch.optim.Adam(net.critic.parameters(), cfg_nn.CRITIC_LEARNING_RATE)
        net.update_time = 0
        return net

    torch.set_num_threads(1)
    l32, _ = SP._run(fresh(torch.float32), make_exp(torch.float32, None))
    assert np.array_equal(l32, f4["losses"])
    l64, s64 = SP._run(fresh(torch.float64), make_exp(torch.float64, None))
    variants, losses = [], []
    for threads in (1, 8):
        torch.set_num_threads(threads)
        for o in range(n_orders):
            perm = None if o == 0 else np.random.default_rng(1000 + o).permutation(B)
            lv, sv = SP._run(fresh(torch.float32), make_exp(torch.float32, perm))
            variants.append(sv)
            losses.append(lv)
            print("threads %d order %2d  max |VLoss - f64| per iteration: %s" % (
                threads, o, " ".join("%.1e" % v for v in np.abs(lv[:, 2] - l64[:, 2]))), flush=True)
    torch.set_num_threads(1)
    out = {"losses_f64": l64, "losses_variants": np.stack(losses)}
    names = list(p0)
    for it in (1, 10):
        var_l2 = np.zeros((len(variants), len(names)))
        for ti, name in enumerate(names):
            a64 = s64[it][name]
            u64 = (a64 - p0[name]).ravel()
            l2 = mx = omc = 0.0
            for vi, v in enumerate(variants):
                d = (v[it][name] - a64).ravel()
                var_l2[vi, ti] = float(np.sqrt(d @ d))
                l2 = max(l2, var_l2[vi, ti])
                mx = max(mx, float(np.abs(d).max()))
                uv = (v[it][name] - p0[name]).ravel()
                c = float(uv @ u64 / (np.linalg.norm(uv) * np.linalg.norm(u64) + 1e-300))
                omc = max(omc, 1.0 - c)
            key = "it%d/%s" % (it, name)
            out["ref_l2/" + key] = np.float64(l2)
            out["ref_max/" + key] = np.float64(mx)
            out["ref_1mcos/" + key] = np.float64(omc)
            out["upd_l2/" + key] = np.float64(np.linalg.norm(u64))
        out["var_l2/it%d" % it] = var_l2
    np.savez(os.path.join(HERE, "f4d_spread_wide.npz"), **out)
    env = np.abs(out["losses_variants"] - f4["losses"][None]).max(axis=0)
    print("largest |variant - stored fp32 run| per iteration (total, actor, v, ent):")
    for it in range(10):
        print(it + 1, " ".join("%.2e" % v for v in env[it]))


if __name__ == "__main__":
    main()
