#!/usr/bin/env python3
"""Golden vectors for GAIL over a NAV encoder (BASELINE config 5 = "nav env + discriminator"), by IMPORTING THE REFERENCE:

  f22_gail_navped.npz   NavPedPreNet(4) shared + CategoricalActor(5) + Critic + GAIL critic + Discriminator whose `pre` is
                        deepcopy(prenet) (runner/utils.py:88-102 robot_nav branch with a pedestrian map + the gail branch
                        :161-168), through the reference's own GAIL / Discriminator / PPO objects (nn/GAIL.py:19-158,
                        nn/ppo.py:61-62,72-75,95-129, nn/nav_encoder.py:50-79)

Same content as F16 / F17 (tests/golden/make_golden_gail.py, whose helpers this script imports): the two-critic forward, the
discriminator reward D((s, a)), ONE GAIL.learn (a WGAN discriminator step, then ten PPO iterations with the GAIL critic's value
loss), and the reference's own fp32 spread around its float64 run (8 threads, three batch orders).

What the reference leaves undefined and this script supplies (stated, not guessed silently), in addition to make_golden_gail's list:
  * the reference has NO expert-data reader for robot_nav (data/mimic_exp.py:268-285 registers atari / mujoco / classical only), and
    Discriminator.learn calls `.to(device).to(dtype)` on the expert batch's states (GAIL.py:79) although a nav PreNet indexes a LIST
    of three tensors (nav_encoder.py:72).  The expert batch is therefore handed over as a list subclass with a `.to()` that maps
    over its members -- the smallest adapter under which GAIL.py:76-91 runs on a nav encoder at all; every arithmetic line executed
    is the reference's.  Discriminator.__init__ opens MIMIC_START_LOAD_PATH through the classical reader (as for F17).
Usage: python tests/golden/make_golden_gail_nav.py
"""
import copy
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)
REF = os.environ.get("DDRL_REFERENCE", "/root/reference")

SEED, B, N_EXPERT, A = 22, 64, 32, 5


class StateList(list):
    """[image, vector, pedestrian map] with the `.to()` GAIL.py:79 calls on an expert batch's states."""

    def to(self, x):
        return StateList([t.to(x) for t in self])


def main():
    from make_golden import _install_stubs
    _install_stubs()
    sys.path.insert(0, REF)
    import make_golden_gail as G
    from ddrl4nav_amd.utils.recipe import hash_weights
    from USTC_lab.data import Experience
    from USTC_lab.nn import CategoricalActor, Critic, Discriminator, GAIL, PPO
    from USTC_lab.nn.nav_encoder import NavPedPreNet

    torch.set_num_threads(1)
    mimic_dir = G._write_mimic_dir(4)
    cfg, cfg_nn = G._configs({"discrete_action": True, "discrete_actions": list(range(A))}, "classical", mimic_dir)
    prenet = NavPedPreNet(image_channel=1 + 3, last_output_dim=512)
    actor = CategoricalActor(action_output_dim=A, device="cpu", last_input_dim=512, soft_max_grid=True, nn_dtype=torch.float32)
    critic = Critic(device="cpu", last_input_dim=512)
    gail_critic = copy.deepcopy(critic)
    ppo_net = PPO(actor, critic, prenet, None, cfg, cfg_nn).to("cpu")
    d_net = Discriminator(pre=copy.deepcopy(prenet), config=cfg, config_nn=cfg_nn).to("cpu")
    net = GAIL(generator=ppo_net, discriminator=d_net, gail_critic=gail_critic).to("cpu")
    names = [k for k, _ in net.named_parameters()]
    weights = hash_weights([(k, tuple(p.shape)) for k, p in net.named_parameters()], SEED)
    G.load_weights(net, weights)

    rng = np.random.default_rng(SEED)
    img = (rng.random((B, 1, 48, 48)) < 0.3).astype(np.float32)
    vec = rng.normal(0, 1, size=(B, 9)).astype(np.float32)
    ped = (rng.random((B, 3, 48, 48)) < 0.1).astype(np.float32)
    states_np = [img, vec, ped]
    x = [torch.from_numpy(s) for s in states_np]
    out = {"names": np.array(names), "state0": img, "state1": vec, "state2": ped}
    with torch.no_grad():
        (dist, _), values = net(x)
        torch.manual_seed(SEED * 10 + 1)
        actions = dist.sample().to(torch.float32)
        old_logps = net.actor.log_prob_from_distribution(dist, actions)
        assert len(values) == 2 and values[1].shape == (B, 1)
        v0, v1 = values[0][:, 0], values[1][:, 0]
        (play, _), _ = net(x, None, True)
        d_reward = net((x, actions.reshape(B, cfg.ACTIONS_DIM)))      # forward.py:159-165
        assert d_reward.shape == (B, 1)
    old_logps = (old_logps + torch.from_numpy(rng.normal(0, 0.25, B).astype(np.float32))).contiguous()
    advs = torch.from_numpy(rng.normal(0, 1, B).astype(np.float32))
    advs[1] = 0.0
    rets = torch.stack([v0 + advs, v1 + torch.from_numpy(rng.normal(0, 0.7, B).astype(np.float32))]).contiguous()  # [2, B]
    # the expert batch: OTHER seeded observations of the same kind (so that the two WGAN terms do not nearly cancel in the
    # encoder, unlike F17) + "expert" actions
    ex_np = [(rng.random((N_EXPERT, 1, 48, 48)) < 0.3).astype(np.float32), rng.normal(0, 1, size=(N_EXPERT, 9)).astype(np.float32),
             (rng.random((N_EXPERT, 3, 48, 48)) < 0.1).astype(np.float32)]
    ex_actions = rng.integers(0, A, size=(N_EXPERT, 1)).astype(np.float32)

    def expert_batch(dtype):
        return [(StateList([torch.from_numpy(s).to(dtype) for s in ex_np]), torch.from_numpy(ex_actions).to(dtype))]

    def make_exp(dtype=torch.float32, order=None):
        idx = np.arange(B) if order is None else order
        e = Experience(states=[s[idx] for s in states_np], advs=advs.numpy()[idx], actions=actions.numpy()[idx],
                       old_logps=old_logps.numpy()[idx], values=rets.numpy()[:, idx])
        e.to_tensor(dtype=dtype, device="cpu")
        return e

    with torch.no_grad():
        (_, lp), _ = net(x, actions)
    out.update({"actions": actions.numpy(), "old_logps": old_logps.numpy(), "advs": advs.numpy(), "rets": rets.numpy(),
                "expert_state0": ex_np[0], "expert_state1": ex_np[1], "expert_state2": ex_np[2], "expert_actions": ex_actions,
                "value0": v0.numpy(), "value1": v1.numpy(), "probs": play.numpy(), "logp": lp.numpy(),
                "d_reward": d_reward.numpy()[:, 0], "d_mlp_hidden": np.int64(G.D_HIDDEN), "n_expert": np.int64(N_EXPERT)})

    # ---- one GAIL.learn: D step then 10 PPO iterations with the GAIL critic --------------------------------
    net.discriminator.expert_data = expert_batch(torch.float32)
    G.reset_optims(net, cfg_nn)
    d_loss, p_loss, s32 = G.run_gail_learn(net, make_exp())
    out["d_loss"], out["losses"] = d_loss, p_loss
    out["d_lr_after"] = np.float64(net.discriminator.optim.param_groups[0]["lr"])
    for tag, snap in (("D1", s32["D1"]), ("it1", s32[1]), ("it10", s32[10])):
        for k in names:
            a = snap[k].astype(np.float32).reshape(-1)
            out["%s/stride/%s" % (tag, k)] = a[::max(1, a.size // 129)][:129].copy()
    for k in names:   # the GAIL critic is in no optimiser: it must not have moved
        if k.startswith("gail_critic."):
            assert np.array_equal(s32[10][k], np.asarray(weights[k], np.float64)), k

    # ---- the reference's own spread: float64 run + fp32 variants (8 threads, 3 batch orders) ---------------
    p0 = {k: np.asarray(v, np.float64) for k, v in weights.items()}
    G.load_weights(net, weights, torch.float64)
    net.discriminator.expert_data = expert_batch(torch.float64)
    G.reset_optims(net, cfg_nn)
    d64, l64, s64 = G.run_gail_learn(net, make_exp(torch.float64))
    out["d_loss_f64"], out["losses_f64"] = d64, l64
    variants, lvars, dvars = [s32], [], []
    torch.set_num_threads(8)
    G.load_weights(net, weights)
    net.discriminator.expert_data = expert_batch(torch.float32)
    G.reset_optims(net, cfg_nn)
    dv, lv, sv = G.run_gail_learn(net, make_exp())
    variants.append(sv); lvars.append(lv); dvars.append(dv)
    out["losses_f32t8"] = lv
    torch.set_num_threads(1)
    for ps in (221, 222, 223):
        order = np.random.default_rng(ps).permutation(B)
        G.load_weights(net, weights)
        G.reset_optims(net, cfg_nn)
        dv, lv, sv = G.run_gail_learn(net, make_exp(order=order))
        variants.append(sv); lvars.append(lv); dvars.append(dv)
    out["losses_perm"] = np.stack(lvars[1:])
    out["d_loss_spread"] = np.float64(max(np.abs(np.asarray(dvars) - d_loss[None]).max(), np.abs(d64 - d_loss).max()))
    for tag, key in (("D1", "D1"), ("it1", 1), ("it10", 10)):
        for name in names:
            a64 = s64[key][name]
            u64 = (a64 - p0[name]).ravel()
            l2 = mx = omc = 0.0
            for v in variants:
                d = (v[key][name] - a64).ravel()
                l2, mx = max(l2, float(np.sqrt(d @ d))), max(mx, float(np.abs(d).max()))
                uv = (v[key][name] - p0[name]).ravel()
                den = np.linalg.norm(uv) * np.linalg.norm(u64)
                omc = max(omc, 1.0 - float(uv @ u64 / den) if den > 0 else 0.0)
            kk = "%s/%s" % (tag, name)
            out["ref_l2/" + kk], out["ref_max/" + kk], out["ref_1mcos/" + kk] = np.float64(l2), np.float64(mx), np.float64(omc)
            out["upd_l2/" + kk] = np.float64(np.linalg.norm(u64))
            out["f64_l2/" + kk] = np.float64(np.sqrt((a64 ** 2).sum()))
            out["f64_head/" + kk] = a64.ravel()[:8].copy()
    f = os.path.join(HERE, "f22_gail_navped.npz")
    np.savez_compressed(f, **out)
    print("  f22_gail_navped.npz %8d B  D loss %.6f  PPO losses[0] %s  names %d" % (os.path.getsize(f), d_loss[0], p_loss[0], len(names)))


if __name__ == "__main__":
    main()
