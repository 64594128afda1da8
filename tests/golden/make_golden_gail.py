#!/usr/bin/env python3
"""Golden vectors for the GAIL path (BASELINE config 5), by IMPORTING THE REFERENCE:

  f16_gail_classical.npz  MLPPreNet(4) shared + CategoricalActor(2) + Critic + GAIL critic + Discriminator
                          (classical branch of create_net + the gail branch, runner/utils.py:61-74,161-168)
  f17_gail_atari.npz      AtariPreNet(4) shared + CategoricalActor(6) ... (atari branch, utils.py:136-143,161-168);
                          frames are those of f3_loss.npz (stored once, there)

What is exercised, all through the reference's own objects (nn/GAIL.py, nn/ppo.py):
  * GAIL(generator, discriminator, gail_critic): named_parameters() order, the two-critic forward
    (`values` = [critic(h), gail_critic(h)], ppo.py:72-75), the discriminator reward D((s, a)) (GAIL.py:63-71,145-147)
  * GAIL.learn(data) = Discriminator.learn (one WGAN step: mean D(generator batch) - mean D(expert batch), grad-norm
    clip WGAN_CLIP_GRAD_NUM, RMSprop(lr GAN_D_LEARNING_RATE, alpha 0.9), StepLR(250, 0.95); yields last=False) followed by
    PPO.learn with the extra value loss on data.values[-1] (ppo.py:101-107; yields last=True) -- GAIL.py:149-158
  * a Discriminator-only run of 260 steps (classical case) so that the StepLR boundary at step 250 is pinned
  * Agents._accumulate_rewards with TWO value / reward rows and per-row discounts (agent.py:97-101,124-140): the
    GAIL-critic GAE (the shipped Agents hard-codes network_type='ppo', agent.py:95; the arithmetic is pinned by calling
    the reference method unbound, as make_golden.py does for F2)

What the reference leaves undefined and this script therefore supplies (stated, not guessed silently):
  * config.GAN_D_MLP_LIST (read at GAIL.py:23, defined nowhere): [(512 + ACTIONS_DIM, 64, "relu"), (64, 1, None)]
  * the expert batch: Discriminator.__init__ builds a DataLoader over MIMIC_START_LOAD_PATH (a developer's home
    directory, base_config.py:49).  The script lets the reference's own MimicExpClassicalWriter write a small data set
    into a temp dir for the constructor, then replaces `expert_data` by ONE fixed batch so the run is reproducible.
    The reference's readers hand back states as a bare tensor although every PreNet indexes `state[0]`
    (mlp_encoder.py:28, atari_encoder.py:26); the batch is therefore shaped [1, n, ...] so that `state[0]` is the
    [n, ...] batch -- the only shape with which GAIL.py:78 runs at all.
  * the GAIL critic is `copy.deepcopy(critic)` (utils.py:162) and sits in NO optimiser (PPO.add_critic appends to a
    plain list, ppo.py:61-62; the Adam of ppo.py:39 was built before): it is never trained, but its value loss is part of
    VLoss / PpoTotalLoss and its gradient flows into the shared prenet.  The fixtures pin exactly that behaviour.

Also stored: the reference's own fp32 spread against its float64 run (as tests/golden/make_golden_spread.py) so that the
GPU bounds are stated relative to it.  Usage: python tests/golden/make_golden_gail.py
"""
import copy
import os
import sys
import tempfile
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)
REF = os.environ.get("DDRL_REFERENCE", "/root/reference")

D_HIDDEN = 64


def _configs(env, task_type, mimic_dir):
    from USTC_lab.config.config_nn import ConfigNN
    cfg_nn = ConfigNN(env)
    cfg_nn.DEVICE = "cpu"
    cfg_nn.SHARE_CNN_NET = True
    cfg_nn.NETWORK_TYPE = "gail"
    cfg = types.SimpleNamespace(MIDDLE_REDIS_HOST="127.0.0.1", MIDDLE_REDIS_PORT=0, TASK_NAME="golden", MODULE_KEY="MODEL",
                                DEVICE="cpu", TASK_TYPE=task_type, MIMIC_START_LOAD_PATH=mimic_dir,
                                ACTIONS_DIM=cfg_nn.ACTIONS_DIM,
                                GAN_D_MLP_LIST=[(512 + cfg_nn.ACTIONS_DIM, D_HIDDEN, "relu"), (D_HIDDEN, 1, None)])
    return cfg, cfg_nn


def _write_mimic_dir(obs_dim):
    """A tiny data set written by the reference's own writer (data/mimic_exp.py:17-140), only so that
    Discriminator.__init__ (GAIL.py:43) finds something to open."""
    from USTC_lab.data import MimicExpFactory
    d = tempfile.mkdtemp(prefix="ddrl_mimic_") + "/"
    w = MimicExpFactory().mimic_writer("classical", "golden", d, 1, 1)
    rng = np.random.default_rng(5)
    for _ in range(3):
        w.put(rng.normal(size=(4, obs_dim)).astype(np.float32), rng.integers(0, 2, size=(4, 1)).astype(np.float32), 0)
    w.write()
    w.sf.close()
    return d


def build(kind, mimic_dir):
    """create_net's shared branch + its gail branch (runner/utils.py:61-74 / 136-143, 161-168), by hand because
    USTC_lab.runner imports gym."""
    from USTC_lab.nn import AtariPreNet, CategoricalActor, Critic, Discriminator, GAIL, PPO
    from USTC_lab.nn.mlp_encoder import MLPPreNet
    if kind == "classical":
        cfg, cfg_nn = _configs({"discrete_action": True, "discrete_actions": [0, 1]}, "classical", mimic_dir)
        prenet, A = MLPPreNet(4, 512), 2
    else:
        cfg, cfg_nn = _configs({"discrete_action": True, "discrete_actions": list(range(6))}, "classical", mimic_dir)
        prenet, A = AtariPreNet(4, last_output_dim=512, device="cpu"), 6
    actor = CategoricalActor(action_output_dim=A, device="cpu", last_input_dim=512, soft_max_grid=True, nn_dtype=torch.float32)
    critic = Critic(device="cpu", last_input_dim=512)
    gail_critic = copy.deepcopy(critic)
    ppo_net = PPO(actor, critic, prenet, None, cfg, cfg_nn).to("cpu")
    d_net = Discriminator(pre=copy.deepcopy(prenet), config=cfg, config_nn=cfg_nn).to("cpu")
    net = GAIL(generator=ppo_net, discriminator=d_net, gail_critic=gail_critic).to("cpu")
    return net, cfg, cfg_nn


def load_weights(net, weights, dtype=torch.float32):
    net.to(torch.float32)
    # state_dict() lists GAIL.actor a second time (alias of generator.actor, GAIL.py:113); named_parameters() does not
    res = net.load_state_dict({k: torch.from_numpy(v.copy()) for k, v in weights.items()}, strict=False)
    assert not res.unexpected_keys and all(k.startswith("actor.") for k in res.missing_keys), res
    net.to(dtype)
    net.discriminator.dtype = dtype   # GAIL.py:78 casts the expert batch to this


def reset_optims(net, cfg_nn):
    g, d = net.generator, net.discriminator
    g.optim = torch.optim.Adam(g.parameters(), cfg_nn.LEARNING_RATE)
    d.optim = torch.optim.RMSprop(d.parameters(), lr=cfg_nn.GAN_D_LEARNING_RATE, alpha=0.9)
    d.optim_decay = torch.optim.lr_scheduler.StepLR(d.optim, step_size=250, gamma=0.95)
    g.update_time = d.update_time = 0


def snapshot(net):
    return {k: p.detach().double().numpy().copy() for k, p in net.named_parameters()}


def run_gail_learn(net, exp):
    """One GAIL.learn(data): rows of (D loss | 4 PPO losses), with the yielded (update_time, last) checked."""
    d_rows, p_rows, snaps = [], [], {}
    for loss_item, update_time, last in net.learn(exp):
        if "Gail[D]Loss" in loss_item:
            assert last is False and update_time == len(d_rows) + 1
            d_rows.append(loss_item["Gail[D]Loss"])
            snaps["D1"] = snapshot(net)
        else:
            assert last is True and update_time == len(p_rows) + 1
            p_rows.append([loss_item[k] for k in ("PpoTotalLoss", "ActorLoss", "VLoss", "EntLoss")])
            if len(p_rows) in (1, 10):
                snaps[len(p_rows)] = snapshot(net)
    return np.asarray(d_rows, np.float64), np.asarray(p_rows, np.float64), snaps


def case(kind, states_np, seed, out_name, n_expert, d_only_steps):
    from ddrl4nav_amd.utils.recipe import hash_weights
    from USTC_lab.data import Experience
    torch.set_num_threads(1)
    mimic_dir = _write_mimic_dir(4)
    net, cfg, cfg_nn = build(kind, mimic_dir)
    names = [k for k, _ in net.named_parameters()]
    weights = hash_weights([(k, tuple(p.shape)) for k, p in net.named_parameters()], seed)
    load_weights(net, weights)
    rng = np.random.default_rng(seed)
    B = states_np.shape[0]
    x = torch.from_numpy(states_np)
    out = {"names": np.array(names)}
    with torch.no_grad():
        (dist, _), values = net([x])
        torch.manual_seed(seed * 10 + 1)
        actions = dist.sample().to(torch.float32)
        old_logps = net.actor.log_prob_from_distribution(dist, actions)
        assert len(values) == 2 and values[1].shape == (B, 1)
        v0, v1 = values[0][:, 0], values[1][:, 0]
        (play, _), _ = net([x], None, True)
        d_reward = net(([x], actions.reshape(B, cfg.ACTIONS_DIM)))      # forward.py:159-165
        assert d_reward.shape == (B, 1)
    old_logps = (old_logps + torch.from_numpy(rng.normal(0, 0.25, B).astype(np.float32))).contiguous()
    advs = torch.from_numpy(rng.normal(0, 1, B).astype(np.float32))
    advs[1] = 0.0
    rets = torch.stack([v0 + advs, v1 + torch.from_numpy(rng.normal(0, 0.7, B).astype(np.float32))]).contiguous()  # [2, B]
    # expert batch: other states of the same kind + "expert" actions; shaped [1, n, ...] (see module docstring)
    perm = rng.permutation(B)[:n_expert]
    ex_states = states_np[perm][::-1].copy()
    ex_actions = rng.integers(0, int(cfg_nn.ACTION_OUTPUT_DIM), size=(n_expert, 1)).astype(np.float32)

    def expert_batch(dtype):
        return [(torch.from_numpy(ex_states[None]).to(dtype), torch.from_numpy(ex_actions).to(dtype))]

    def make_exp(dtype=torch.float32, order=None):
        idx = np.arange(B) if order is None else order
        e = Experience(states=[states_np[idx]], advs=advs.numpy()[idx], actions=actions.numpy()[idx],
                       old_logps=old_logps.numpy()[idx], values=rets.numpy()[:, idx])
        e.to_tensor(dtype=dtype, device="cpu")
        return e

    with torch.no_grad():
        (_, lp), _ = net([x], actions)
    out.update({"actions": actions.numpy(), "old_logps": old_logps.numpy(), "advs": advs.numpy(), "rets": rets.numpy(),
                "expert_index": perm.astype(np.int64), "expert_actions": ex_actions, "value0": v0.numpy(), "value1": v1.numpy(),
                "probs": play.numpy(), "logp": lp.numpy(), "d_reward": d_reward.numpy()[:, 0],
                "d_mlp_hidden": np.int64(D_HIDDEN), "n_expert": np.int64(n_expert)})
    if kind == "classical":
        out["states"] = states_np

    # ---- one GAIL.learn: D step then 10 PPO iterations with the GAIL critic --------------------------------
    net.discriminator.expert_data = expert_batch(torch.float32)
    reset_optims(net, cfg_nn)
    d_loss, p_loss, s32 = run_gail_learn(net, make_exp())
    out["d_loss"], out["losses"] = d_loss, p_loss
    out["d_lr_after"] = np.float64(net.discriminator.optim.param_groups[0]["lr"])
    for tag, snap in (("D1", s32["D1"]), ("it1", s32[1]), ("it10", s32[10])):
        for k in names:
            a = snap[k].astype(np.float32).reshape(-1)
            out["%s/stride/%s" % (tag, k)] = a[::max(1, a.size // 129)][:129].copy()
    # the GAIL critic is in no optimiser: it must not have moved
    for k in names:
        if k.startswith("gail_critic."):
            assert np.array_equal(s32[10][k], np.asarray(weights[k], np.float64)), k

    # ---- the reference's own spread: float64 run + fp32 variants (8 threads, 3 batch orders) ---------------
    p0 = {k: np.asarray(v, np.float64) for k, v in weights.items()}
    load_weights(net, weights, torch.float64)
    net.discriminator.expert_data = expert_batch(torch.float64)
    reset_optims(net, cfg_nn)
    d64, l64, s64 = run_gail_learn(net, make_exp(torch.float64))
    out["d_loss_f64"], out["losses_f64"] = d64, l64
    variants, lvars, dvars = [s32], [], []
    torch.set_num_threads(8)
    load_weights(net, weights)
    net.discriminator.expert_data = expert_batch(torch.float32)
    reset_optims(net, cfg_nn)
    dv, lv, sv = run_gail_learn(net, make_exp())
    variants.append(sv); lvars.append(lv); dvars.append(dv)
    out["losses_f32t8"] = lv
    torch.set_num_threads(1)
    for ps in (201, 202, 203):
        order = np.random.default_rng(ps).permutation(B)
        load_weights(net, weights)
        reset_optims(net, cfg_nn)
        dv, lv, sv = run_gail_learn(net, make_exp(order=order))
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

    # ---- Discriminator alone for many steps: the StepLR boundary (GAIL.py:31,84) ---------------------------
    if d_only_steps:
        load_weights(net, weights)
        net.discriminator.expert_data = expert_batch(torch.float32)
        reset_optims(net, cfg_nn)
        exp = make_exp()
        rows, lrs = [], []
        for step in range(1, d_only_steps + 1):
            for loss_item, update_time, last in net.discriminator.learn(exp):
                assert update_time == step
                rows.append(loss_item["Gail[D]Loss"])
            lrs.append(net.discriminator.optim.param_groups[0]["lr"])
        out["d_only_loss"], out["d_only_lr"] = np.asarray(rows, np.float64), np.asarray(lrs, np.float64)
        snap = snapshot(net)
        for k in names:
            if k.startswith("discriminator."):
                a = snap[k].astype(np.float32).reshape(-1)
                out["Dend/stride/%s" % k] = a[::max(1, a.size // 129)][:129].copy()
        load_weights(net, weights, torch.float64)
        net.discriminator.expert_data = expert_batch(torch.float64)
        reset_optims(net, cfg_nn)
        exp64 = make_exp(torch.float64)
        rows = [li["Gail[D]Loss"] for _ in range(d_only_steps) for li, _, _ in net.discriminator.learn(exp64)]
        out["d_only_loss_f64"] = np.asarray(rows, np.float64)
    np.savez_compressed(os.path.join(HERE, out_name), **out)
    print("  %-26s %8d B  D loss %.6f  PPO VLoss[0] %.6f  names %d" % (out_name, os.path.getsize(os.path.join(HERE, out_name)),
                                                                       d_loss[0], p_loss[0][2], len(names)))


def gae_two_rows():
    """Agents._accumulate_rewards (agent.py:124-140) with value_dim_num = reward_dim_num = 2 (agent.py:97-101):
    discounts [[EXTRINSIC_DISCOUNT], [GAN_DISCOUNT]], row 1 of `dones` stays zero ("follow rnd trick", agent.py:261-264)."""
    from USTC_lab.agent.agent import Agents
    from USTC_lab.data import Experience
    rng = np.random.default_rng(18)
    T, N = 40, 6
    values = rng.normal(0, 1, size=(T + 1, 2, N)).astype(np.float32)
    rewards = np.zeros((T + 1, 2, N), np.float32)
    rewards[:, 0] = rng.choice(np.array([-1, 0, 1], np.float32), p=[0.1, 0.8, 0.1], size=(T + 1, N))
    rewards[:, 1] = rng.normal(0, 0.3, size=(T + 1, N)).astype(np.float32)           # discriminator rewards
    dones = np.zeros((T + 1, 2, N), np.uint8)
    dones[:, 0] = rng.random((T + 1, N)) < 0.08
    me = types.SimpleNamespace(model_dtype=np.float32, landa=0.95,
                               discounts=np.array([0.99, 0.97], np.float32).reshape(2, 1))
    exps = [Experience(states=None, values=values[t].copy(), dones=dones[t].copy()) for t in range(T + 1)]
    res = Agents._accumulate_rewards(me, exps, rewards)
    assert len(res) == T
    np.savez(os.path.join(HERE, "f18_gae_two_rows.npz"), values=values, rewards=rewards, dones=dones,
             discounts=me.discounts[:, 0], landa=np.float32(0.95), adv=np.stack([e.advs for e in res]),
             ret=np.stack([e.values for e in res]))
    print("  f18_gae_two_rows.npz written")


def main():
    from make_golden import _install_stubs
    _install_stubs()
    sys.path.insert(0, REF)
    rng = np.random.default_rng(16)
    case("classical", rng.normal(0, 1, size=(200, 4)).astype(np.float32), 16, "f16_gail_classical.npz", 128, 260)
    f3 = np.load(os.path.join(HERE, "f3_loss.npz"))
    x = (f3["frames"] / 255.0).astype(np.float32)
    case("atari", x, 17, "f17_gail_atari.npz", 32, 0)
    gae_two_rows()


if __name__ == "__main__":
    main()
