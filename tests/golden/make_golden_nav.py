#!/usr/bin/env python3
"""Golden vectors for the non-Atari nets, by IMPORTING THE REFERENCE:

  f13_nav1d_gauss.npz   NavPreNet1D x2 (non-shared) + GaussionActor(act_dim=2) + Critic -- the robot_nav
                        branch of create_net (runner/utils.py:110-121), PPO.forward / learn
  f14_navped_shared.npz NavPedPreNet shared + CategoricalActor(5) -- the SHARE_CNN_NET branch with a
                        pedestrian map (runner/utils.py:88-102)
  f15_mlp_classical.npz MLPPreNet x2 + CategoricalActor(2) -- the classical / mujoco branch (utils.py:61-86)

Inputs are seeded and stored in the fixtures; weights come from utils/recipe.py:hash_weights so that
nothing but data is committed.  The reference's env (img_env, ROS) is absent: parity beyond the NN
is unpinned.  Usage: python tests/golden/make_golden_nav.py
"""
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)
REF = os.environ.get("DDRL_REFERENCE", "/root/reference")


def cfgs(env):
    from USTC_lab.config.config_nn import ConfigNN
    cfg = types.SimpleNamespace(MIDDLE_REDIS_HOST="127.0.0.1", MIDDLE_REDIS_PORT=0, TASK_NAME="golden", MODULE_KEY="MODEL",
                                DEVICE="cpu")
    cfg_nn = ConfigNN(env)
    cfg_nn.DEVICE = "cpu"
    return cfg, cfg_nn


def load_recipe(net, seed):
    from ddrl4nav_amd.utils.recipe import hash_weights
    shapes = [(k, tuple(p.shape)) for k, p in net.named_parameters()]
    w = hash_weights(shapes, seed)
    net.load_state_dict({k: torch.from_numpy(v.copy()) for k, v in w.items()}, strict=True)
    return w, shapes


def run_case(net, cfg_nn, states, actions_from, B, rng, out, reopt):
    """forward, one loss/grad evaluation and the reference's learn() on a fixed batch."""
    from USTC_lab.data import Experience
    weights = {k: p.detach().numpy().copy() for k, p in net.named_parameters()}
    st = [torch.from_numpy(s) for s in states]
    with torch.no_grad():
        (dist, _), values = net(st)
        actions = actions_from(dist)
        old_logps = net.actor.log_prob_from_distribution(dist, actions)
        v0 = values[0][:, 0]
        (play, _), _ = net(st, None, True)
    old_logps = (old_logps + torch.from_numpy(rng.normal(0, 0.25, B).astype(np.float32))).contiguous()
    advs = torch.from_numpy(rng.normal(0, 1, B).astype(np.float32))
    advs[1] = 0.0
    rets = (v0 + advs).contiguous()

    def make_exp(dtype=torch.float32):
        e = Experience(states=[s.copy() for s in states], advs=advs.numpy(), actions=actions.numpy(),
                       old_logps=old_logps.numpy(), values=rets.numpy().reshape(1, B))
        e.to_tensor(dtype=dtype, device="cpu")
        return e

    exp = make_exp()
    with torch.no_grad():
        (d2, lp), _ = net(exp.states, exp.actions)
        ent_el = d2.entropy()
    out.update({"actions": actions.numpy(), "old_logps": old_logps.numpy(), "advs": advs.numpy(), "rets": rets.numpy(),
                "dist_out": play.numpy(), "value": v0.numpy(), "logp": lp.numpy(), "entropy": ent_el.numpy()})
    for i, s in enumerate(states):
        out["state%d" % i] = s
    # one loss evaluation + gradients, exactly as the two branches of ppo.py:110-129 differentiate
    net.zero_grad()
    pi, values = net(exp.states, exp.actions)
    dist, log_p = pi
    ratio = torch.exp(log_p - exp.old_logps)
    m = torch.min(ratio * exp.advs, torch.clamp(ratio, 1.0 - net.ppo_clip, 1.0 + net.ppo_clip) * exp.advs)
    actor_loss = -torch.mean(torch.where(exp.advs > 0, m, torch.max(m, net.duel_ppo_clip * exp.advs)))
    v_loss = net.vlossf(exp.values[0, :], values[0].squeeze())
    ent = torch.mean(dist.entropy())
    total = actor_loss + v_loss * net.v_loss_theta - ent * net.ent_loss_theta
    if net.share_cnn_net:
        total.backward()
    else:
        actor_loss.backward()
        v_loss.backward()
    out["loss4"] = np.array([total.item(), actor_loss.item(), v_loss.item(), ent.item()], np.float64)
    for k, p in net.named_parameters():
        g = p.grad.detach().numpy()
        out["gl2/" + k] = np.float64(np.sqrt((g.astype(np.float64) ** 2).sum()))
        out["gsum/" + k] = np.float64(g.astype(np.float64).sum())
        flat = g.reshape(-1)
        out["gstride/" + k] = flat[::max(1, flat.size // 129)][:129].copy()
    net.zero_grad()

    def learn_losses(e):
        rows = []
        for it, (ld, ut, last) in enumerate(net.learn(e), 1):
            rows.append([ld["PpoTotalLoss"], ld["ActorLoss"], ld["VLoss"], ld["EntLoss"]])
            assert ut == it and last is True
        return np.asarray(rows, np.float64)

    out["losses"] = learn_losses(exp)
    for k, p in net.named_parameters():
        flat = p.detach().numpy().reshape(-1)
        out["it10/stride/" + k] = flat[::max(1, flat.size // 129)][:129].copy()
    for tag, dtype, threads in (("f64", torch.float64, 1), ("f32t8", torch.float32, 8)):
        torch.set_num_threads(threads)
        net.to(torch.float32)
        net.load_state_dict({k: torch.from_numpy(v.copy()) for k, v in weights.items()}, strict=True)
        net.to(dtype)
        net.update_time = 0
        reopt(net)
        out["losses_" + tag] = learn_losses(make_exp(dtype))
    torch.set_num_threads(1)
    net.to(torch.float32)


def main():
    from make_golden import _install_stubs
    _install_stubs()
    sys.path.insert(0, REF)
    from USTC_lab.nn import CategoricalActor, Critic, GaussionActor, PPO
    from USTC_lab.nn.mlp_encoder import MLPPreNet
    from USTC_lab.nn.nav_encoder import NavPedPreNet, NavPreNet1D
    torch.set_num_threads(1)

    def reopt_two(cfg_nn):
        def f(net):
            net.actor_optim = torch.optim.Adam(net.actor.parameters(), cfg_nn.ACTOR_LEARNING_RATE)
            net.critic_optim = torch.optim.Adam(net.critic.parameters(), cfg_nn.CRITIC_LEARNING_RATE)
        return f

    # ---------------- F13: NavPreNet1D x2 + Gaussian actor (robot_nav, non-shared) -------------
    torch.manual_seed(0)
    cfg, cfg_nn = cfgs({"discrete_action": False, "act_dim": 2})
    actor = GaussionActor(action_output_dim=2, device="cpu", soft_max_grid=True, last_input_dim=512,
                          nn_dtype=torch.float32, pre=NavPreNet1D(image_channel=3, last_output_dim=512))
    critic = Critic(device="cpu", last_input_dim=512, pre=NavPreNet1D(image_channel=3, last_output_dim=512))
    net = PPO(actor, critic, None, None, cfg, cfg_nn).to("cpu")
    load_recipe(net, 13)
    rng = np.random.default_rng(13)
    B = 64   # round 3: 20 samples made the critic's lr = 1e-3 trajectory a knife edge (see make_golden_nav_spread.py)
    laser = rng.uniform(0.05, 1.0, size=(B, 1, 960)).astype(np.float32)
    vec = rng.normal(0, 1, size=(B, 5)).astype(np.float32)
    ped = (rng.random((B, 3, 48, 48)) < 0.15).astype(np.float32) * rng.uniform(0.5, 1.0, size=(B, 3, 48, 48)).astype(np.float32)
    out = {"names": np.array([k for k, _ in net.named_parameters()])}

    def sample_normal(dist):
        torch.manual_seed(131)
        return dist.sample().to(torch.float32)

    run_case(net, cfg_nn, [laser, vec, ped], sample_normal, B, rng, out, reopt_two(cfg_nn))
    np.savez_compressed(os.path.join(HERE, "f13_nav1d_gauss.npz"), **out)

    # ---------------- F14: NavPedPreNet shared + Categorical(5) ---------------------------------
    cfg, cfg_nn = cfgs({"discrete_action": True, "discrete_actions": list(range(5))})
    cfg_nn.SHARE_CNN_NET = True
    actor = CategoricalActor(action_output_dim=5, device="cpu", soft_max_grid=True, last_input_dim=512,
                             nn_dtype=torch.float32)
    critic = Critic(device="cpu", last_input_dim=512)
    prenet = NavPedPreNet(image_channel=1 + 3, last_output_dim=512)
    net = PPO(actor, critic, prenet, None, cfg, cfg_nn).to("cpu")
    load_recipe(net, 14)
    rng = np.random.default_rng(14)
    B = 64
    img = (rng.random((B, 1, 48, 48)) < 0.3).astype(np.float32)
    vec = rng.normal(0, 1, size=(B, 9)).astype(np.float32)
    ped = (rng.random((B, 3, 48, 48)) < 0.1).astype(np.float32)
    out = {"names": np.array([k for k, _ in net.named_parameters()])}

    def sample_cat(dist):
        torch.manual_seed(141)
        return dist.sample().to(torch.float32)

    def reopt_shared(n):
        n.optim = torch.optim.Adam(n.parameters(), cfg_nn.LEARNING_RATE)

    run_case(net, cfg_nn, [img, vec, ped], sample_cat, B, rng, out, reopt_shared)
    np.savez_compressed(os.path.join(HERE, "f14_navped_shared.npz"), **out)

    # ---------------- F15: MLPPreNet x2 + Categorical(2) (classical control) ------------------
    cfg, cfg_nn = cfgs({"discrete_action": True, "discrete_actions": [0, 1]})
    actor = CategoricalActor(action_output_dim=2, device="cpu", soft_max_grid=True, last_input_dim=512,
                             nn_dtype=torch.float32, pre=MLPPreNet(4, 512))
    critic = Critic(device="cpu", last_input_dim=512, pre=MLPPreNet(4, 512))
    net = PPO(actor, critic, None, None, cfg, cfg_nn).to("cpu")
    load_recipe(net, 15)
    rng = np.random.default_rng(15)
    B = 200
    obs = rng.normal(0, 1, size=(B, 4)).astype(np.float32)
    out = {"names": np.array([k for k, _ in net.named_parameters()])}
    run_case(net, cfg_nn, [obs], sample_cat, B, rng, out, reopt_two(cfg_nn))
    np.savez_compressed(os.path.join(HERE, "f15_mlp_classical.npz"), **out)
    for f in ("f13_nav1d_gauss.npz", "f14_navped_shared.npz", "f15_mlp_classical.npz"):
        print("  %-24s %8d B" % (f, os.path.getsize(os.path.join(HERE, f))))


if __name__ == "__main__":
    main()
