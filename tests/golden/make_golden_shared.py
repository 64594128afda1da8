#!/usr/bin/env python3
"""Golden vectors for the two non-default learner modes, again by IMPORTING THE REFERENCE:

  f10_shared.npz     SHARE_CNN_NET=True: one AtariPreNet shared by actor and critic, one Adam
                     (LEARNING_RATE) on total_loss (reference nn/ppo.py:39,110-117,
                     runner/utils.py:136-143)
  f11_smooth_l1.npz  SMOOTH_L1_LOSS=True on the default two-encoder net (ppo.py:53-54)
  f12_actions18.npz  ACTION_OUTPUT_DIM=18 (the full Atari action set) on the default net

Inputs are the frames of f3_loss.npz (stored once, there).  Runs only in the build container
(needs /root/reference); nothing of the reference is copied, only its outputs are stored.

Usage:  python tests/golden/make_golden_shared.py
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


def _cfg():
    from USTC_lab.config.config_nn import ConfigNN
    cfg = types.SimpleNamespace(MIDDLE_REDIS_HOST="127.0.0.1", MIDDLE_REDIS_PORT=0, TASK_NAME="golden",
                                MODULE_KEY="MODEL", DEVICE="cpu")
    cfg_nn = ConfigNN({"discrete_action": True, "discrete_actions": list(range(6))})
    cfg_nn.DEVICE = "cpu"
    return cfg, cfg_nn


def build_shared(weights):
    """create_net's atari / SHARE_CNN_NET=True branch (runner/utils.py:136-143), by hand because
    USTC_lab.runner imports gym (absent)."""
    from USTC_lab.nn import AtariPreNet, CategoricalActor, Critic, PPO
    from ddrl4nav_amd.utils.recipe import param_specs
    cfg, cfg_nn = _cfg()
    cfg_nn.SHARE_CNN_NET = True
    actor = CategoricalActor(action_output_dim=6, device="cpu", last_input_dim=512, soft_max_grid=True,
                             nn_dtype=torch.float32)
    critic = Critic(device="cpu")
    prenet = AtariPreNet(4, last_output_dim=512, device="cpu")
    net = PPO(actor, critic, prenet, None, cfg, cfg_nn).to("cpu")
    names = [k for k, _ in net.named_parameters()]
    assert names == [n for n, _, _ in param_specs(shared=True)], names
    net.load_state_dict({k: torch.from_numpy(v.copy()) for k, v in weights.items()}, strict=True)
    return net, cfg_nn


def build_smooth(weights):
    from USTC_lab.nn import AtariPreNet, CategoricalActor, Critic, PPO
    cfg, cfg_nn = _cfg()
    cfg_nn.SMOOTH_L1_LOSS = True
    pre_a = AtariPreNet(4, last_output_dim=512, device="cpu")
    pre_c = AtariPreNet(4, last_output_dim=512, device="cpu")
    actor = CategoricalActor(action_output_dim=6, device="cpu", soft_max_grid=True, last_input_dim=512, pre=pre_a,
                             nn_dtype=torch.float32)
    critic = Critic(device="cpu", last_input_dim=512, pre=pre_c)
    net = PPO(actor, critic, None, None, cfg, cfg_nn).to("cpu")
    net.load_state_dict({k: torch.from_numpy(v.copy()) for k, v in weights.items()}, strict=True)
    return net, cfg_nn


def loss_terms(net, exp):
    """The loss block of the reference's learn() (ppo.py:82-108) evaluated once, through the
    reference's own forward / vlossf / Categorical."""
    pi, values = net(exp.states, exp.actions)
    dist, log_p = pi
    ratio = torch.exp(log_p - exp.old_logps)
    m = torch.min(ratio * exp.advs, torch.clamp(ratio, 1.0 - net.ppo_clip, 1.0 + net.ppo_clip) * exp.advs)
    actor_loss = -torch.mean(torch.where(exp.advs > 0, m, torch.max(m, net.duel_ppo_clip * exp.advs)))
    v_loss = net.vlossf(exp.values[0, :], values[0].squeeze())
    ent = torch.mean(dist.entropy())
    total = actor_loss + v_loss * net.v_loss_theta - ent * net.ent_loss_theta
    return total, actor_loss, v_loss, ent, values[0]


def grad_summary(net, out, prefix=""):
    for k, p in net.named_parameters():
        g = p.grad.detach().numpy()
        out[prefix + "gsum/" + k] = np.float64(g.astype(np.float64).sum())
        out[prefix + "gl2/" + k] = np.float64(np.sqrt((g.astype(np.float64) ** 2).sum()))
        out[prefix + "ghead/" + k] = g.reshape(-1)[:64].copy()


def run_learn(net, exp, out, keep_params):
    losses = []
    for it, (ld, update_time, last) in enumerate(net.learn(exp), 1):
        losses.append([ld["PpoTotalLoss"], ld["ActorLoss"], ld["VLoss"], ld["EntLoss"]])
        assert update_time == it and last is True
        if keep_params and it in (1, 10):
            for k, p in net.named_parameters():
                a = p.detach().numpy()
                out["it%d/sum/%s" % (it, k)] = np.float64(a.astype(np.float64).sum())
                out["it%d/l2/%s" % (it, k)] = np.float64(np.sqrt((a.astype(np.float64) ** 2).sum()))
                flat = a.reshape(-1)
                out["it%d/stride/%s" % (it, k)] = flat[::max(1, flat.size // 257)][:257].copy()
    return np.asarray(losses, np.float64)


def main():
    from make_golden import _install_stubs
    _install_stubs()
    sys.path.insert(0, REF)
    from ddrl4nav_amd.utils.recipe import make_weights
    from USTC_lab.data import Experience
    torch.set_num_threads(1)
    f3 = np.load(os.path.join(HERE, "f3_loss.npz"))
    frames = f3["frames"]
    B = frames.shape[0]
    xb = torch.tensor(frames / 255.0, dtype=torch.float32)

    # ---------------- F10: shared prenet ----------------------------------------------------
    torch.manual_seed(0)
    weights = make_weights(seed=0, shared=True)
    net, cfg_nn = build_shared(weights)
    rng = np.random.default_rng(10)
    with torch.no_grad():
        (dist, _), values = net([xb])
        torch.manual_seed(11)
        actions = dist.sample().to(torch.float32)
        old_logps = net.actor.log_prob_from_distribution(dist, actions)
        v0 = values[0][:, 0]
        h = net.prenet([xb])
        probs_play, _ = net.actor(h, None, True)
    old_logps = (old_logps + torch.from_numpy(rng.normal(0, 0.25, B).astype(np.float32))).contiguous()
    advs = torch.from_numpy(rng.normal(0, 1, B).astype(np.float32))
    advs[9] = 0.0
    rets = (v0 + advs * 1.5).contiguous()

    def make_exp(dtype=torch.float32):
        e = Experience(states=[xb.numpy()], advs=advs.numpy(), actions=actions.numpy(), old_logps=old_logps.numpy(),
                       values=rets.numpy().reshape(1, B))
        e.to_tensor(dtype=dtype, device="cpu")
        return e

    exp = make_exp()
    out = {"actions": actions.numpy(), "old_logps": old_logps.numpy(), "advs": advs.numpy(), "rets": rets.numpy(),
           "probs": probs_play.numpy(), "value": v0.numpy(), "h": h.numpy()[:, :16]}
    with torch.no_grad():
        (_, lp), _ = net([xb], actions)
    out["logp"] = lp.numpy()
    net.zero_grad()
    total, actor_loss, v_loss, ent, _ = loss_terms(net, exp)
    total.backward()  # the shared branch differentiates total_loss (ppo.py:111-112)
    out["loss4"] = np.array([total.item(), actor_loss.item(), v_loss.item(), ent.item()], np.float64)
    grad_summary(net, out)
    out["gnorm"] = np.float64(np.sqrt(sum((p.grad.double() ** 2).sum().item() for p in net.parameters())))
    net.zero_grad()
    out["losses"] = run_learn(net, exp, out, True)
    for tag, dtype, threads in (("f64", torch.float64, 1), ("f32t8", torch.float32, 8)):
        torch.set_num_threads(threads)
        net.to(torch.float32)
        net.load_state_dict({k: torch.from_numpy(v.copy()) for k, v in weights.items()}, strict=True)
        net.to(dtype)
        net.update_time = 0
        net.optim = torch.optim.Adam(net.parameters(), cfg_nn.LEARNING_RATE)
        out["losses_" + tag] = run_learn(net, make_exp(dtype), {}, False)
    torch.set_num_threads(1)
    out["learning_rate"] = np.float64(cfg_nn.LEARNING_RATE)
    np.savez(os.path.join(HERE, "f10_shared.npz"), **out)

    # ---------------- F11: smooth-L1 value loss, default two-encoder net ---------------------
    weights2 = make_weights(seed=0)
    net2, cfg_nn2 = build_smooth(weights2)
    rets2 = torch.from_numpy(f3["rets"] + 1.5 * np.sign(f3["advs"]) * (np.arange(B) % 3 == 0)).to(torch.float32)

    def make_exp2(dtype=torch.float32):
        e = Experience(states=[xb.numpy()], advs=f3["advs"], actions=f3["actions"], old_logps=f3["old_logps"],
                       values=rets2.numpy().reshape(1, B))
        e.to_tensor(dtype=dtype, device="cpu")
        return e

    exp2 = make_exp2()
    out2 = {"rets": rets2.numpy()}
    net2.zero_grad()
    total, actor_loss, v_loss, ent, values = loss_terms(net2, exp2)
    values.retain_grad()
    actor_loss.backward()
    v_loss.backward()
    out2["loss4"] = np.array([total.item(), actor_loss.item(), v_loss.item(), ent.item()], np.float64)
    out2["dvalue"] = values.grad.numpy()[:, 0].copy()
    grad_summary(net2, out2)
    net2.zero_grad()
    out2["losses"] = run_learn(net2, exp2, out2, True)
    net2.to(torch.float32)
    net2.load_state_dict({k: torch.from_numpy(v.copy()) for k, v in weights2.items()}, strict=True)
    net2.to(torch.float64)
    net2.update_time = 0
    net2.actor_optim = torch.optim.Adam(net2.actor.parameters(), cfg_nn2.ACTOR_LEARNING_RATE)
    net2.critic_optim = torch.optim.Adam(net2.critic.parameters(), cfg_nn2.CRITIC_LEARNING_RATE)
    out2["losses_f64"] = run_learn(net2, make_exp2(torch.float64), {}, False)
    np.savez(os.path.join(HERE, "f11_smooth_l1.npz"), **out2)
    # ---------------- F12: 18 actions (full Atari action set), default two-encoder net --------
    from USTC_lab.nn import AtariPreNet, CategoricalActor, Critic, PPO
    from ddrl4nav_amd.utils.recipe import param_specs
    cfg, cfg_nn3 = _cfg()
    w18 = make_weights(seed=0, n_actions=18)
    actor = CategoricalActor(action_output_dim=18, device="cpu", soft_max_grid=True, last_input_dim=512,
                             pre=AtariPreNet(4, last_output_dim=512, device="cpu"), nn_dtype=torch.float32)
    critic = Critic(device="cpu", last_input_dim=512, pre=AtariPreNet(4, last_output_dim=512, device="cpu"))
    net3 = PPO(actor, critic, None, None, cfg, cfg_nn3).to("cpu")
    assert [k for k, _ in net3.named_parameters()] == [n for n, _, _ in param_specs(n_actions=18)]
    net3.load_state_dict({k: torch.from_numpy(v.copy()) for k, v in w18.items()}, strict=True)
    rng = np.random.default_rng(12)
    with torch.no_grad():
        (dist, _), values = net3([xb])
        torch.manual_seed(13)
        act18 = dist.sample().to(torch.float32)
        act18[:18] = torch.arange(18, dtype=torch.float32)  # every action index occurs
        old18 = net3.actor.log_prob_from_distribution(dist, act18)
        probs18, _ = net3.actor([xb], None, True)
        ent18 = dist.entropy()
    old18 = (old18 + torch.from_numpy(rng.normal(0, 0.25, B).astype(np.float32))).contiguous()
    adv18 = torch.from_numpy(rng.normal(0, 1, B).astype(np.float32))
    ret18 = (values[0][:, 0] + adv18).contiguous()
    exp3 = Experience(states=[xb.numpy()], advs=adv18.numpy(), actions=act18.numpy(), old_logps=old18.numpy(),
                      values=ret18.numpy().reshape(1, B))
    exp3.to_tensor(dtype=torch.float32, device="cpu")
    out3 = {"actions": act18.numpy(), "old_logps": old18.numpy(), "advs": adv18.numpy(), "rets": ret18.numpy(),
            "probs": probs18.numpy(), "value": values[0].numpy()[:, 0], "entropy": ent18.numpy()}
    with torch.no_grad():
        (_, lp), _ = net3([xb], act18)
    out3["logp"] = lp.numpy()
    net3.zero_grad()
    total, actor_loss, v_loss, ent, _ = loss_terms(net3, exp3)
    actor_loss.backward()
    v_loss.backward()
    out3["loss4"] = np.array([total.item(), actor_loss.item(), v_loss.item(), ent.item()], np.float64)
    grad_summary(net3, out3)
    out3["grad_actor_linear_w"] = net3.actor.actor_linear.weight.grad.numpy().copy()
    out3["grad_actor_linear_b"] = net3.actor.actor_linear.bias.grad.numpy().copy()
    net3.zero_grad()
    out3["losses"] = run_learn(net3, exp3, out3, False)
    # the reference's own sensitivity to summation order (see make_golden.py, F4)
    for tag, dtype, threads in (("f64", torch.float64, 1), ("f32t8", torch.float32, 8)):
        torch.set_num_threads(threads)
        net3.to(torch.float32)
        net3.load_state_dict({k: torch.from_numpy(v.copy()) for k, v in w18.items()}, strict=True)
        net3.to(dtype)
        net3.update_time = 0
        net3.actor_optim = torch.optim.Adam(net3.actor.parameters(), cfg_nn3.ACTOR_LEARNING_RATE)
        net3.critic_optim = torch.optim.Adam(net3.critic.parameters(), cfg_nn3.CRITIC_LEARNING_RATE)
        e = Experience(states=[xb.numpy()], advs=adv18.numpy(), actions=act18.numpy(), old_logps=old18.numpy(),
                       values=ret18.numpy().reshape(1, B))
        e.to_tensor(dtype=dtype, device="cpu")
        out3["losses_" + tag] = run_learn(net3, e, {}, False)
    torch.set_num_threads(1)
    np.savez(os.path.join(HERE, "f12_actions18.npz"), **out3)
    for f in ("f10_shared.npz", "f11_smooth_l1.npz", "f12_actions18.npz"):
        print("  %-20s %8d B" % (f, os.path.getsize(os.path.join(HERE, f))))


if __name__ == "__main__":
    main()
