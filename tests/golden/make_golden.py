#!/usr/bin/env python3
"""Generate the golden vectors in tests/golden/*.npz by IMPORTING THE REFERENCE ITSELF.

Runs only in the build container (needs /root/reference, read-only).  Nothing from the
reference is copied: the script drives the reference's own classes on seeded inputs and stores
inputs + outputs as data.  ``redis`` and ``torch.utils.tensorboard`` are absent from the image
and are stubbed in-process (they are only touched by constructors / annotations on this path,
SURVEY.md section 8c).

Usage:  python tests/golden/make_golden.py
"""
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
REF = os.environ.get("DDRL_REFERENCE", "/root/reference")


def _install_stubs():
    redis = types.ModuleType("redis")

    class _Pipe:
        def __init__(self):
            self.store = {}

        def set(self, k, v):
            self.store[k] = v

        def incr(self, k):
            self.store[k] = int(self.store.get(k, 0)) + 1

        def execute(self):
            return []

    class Redis:
        def __init__(self, *a, **k):
            self._pipe = _Pipe()

        def pipeline(self):
            return self._pipe

        def get(self, k):
            return self._pipe.store.get(k)

        def hset(self, *a, **k):
            pass

    redis.Redis = Redis
    client = types.ModuleType("redis.client")
    client.Pipeline = _Pipe
    redis.client = client
    sys.modules["redis"] = redis
    sys.modules["redis.client"] = client
    tb = types.ModuleType("torch.utils.tensorboard")

    class SummaryWriter:
        def __init__(self, *a, **k):
            pass

    tb.SummaryWriter = SummaryWriter
    sys.modules["torch.utils.tensorboard"] = tb
    if not hasattr(np, "bool"):
        np.bool = bool  # reference uses the removed alias (experience.py:141)


def build_reference_net(weights):
    """create_net's atari / non-shared branch (runner/utils.py:122-134,159-160) re-stated by
    hand because USTC_lab.runner imports gym (absent)."""
    from USTC_lab.nn import AtariPreNet, CategoricalActor, Critic, PPO
    cfg = types.SimpleNamespace(MIDDLE_REDIS_HOST="127.0.0.1", MIDDLE_REDIS_PORT=0, TASK_NAME="golden",
                                MODULE_KEY="MODEL", DEVICE="cpu")
    from USTC_lab.config.config_nn import ConfigNN
    cfg_nn = ConfigNN({"discrete_action": True, "discrete_actions": list(range(6))})
    cfg_nn.DEVICE = "cpu"
    pre_a = AtariPreNet(4, last_output_dim=512, device="cpu")
    pre_c = AtariPreNet(4, last_output_dim=512, device="cpu")
    actor = CategoricalActor(action_output_dim=6, device="cpu", soft_max_grid=True, last_input_dim=512,
                             pre=pre_a, nn_dtype=torch.float32)
    critic = Critic(device="cpu", last_input_dim=512, pre=pre_c)
    net = PPO(actor, critic, None, None, cfg, cfg_nn).to("cpu")
    names = [k for k, _ in net.named_parameters()]
    from ddrl4nav_amd.utils.recipe import param_specs
    assert names == [n for n, _, _ in param_specs()], names
    net.load_state_dict({k: torch.from_numpy(v.copy()) for k, v in weights.items()}, strict=True)
    return net, cfg_nn


def main():
    _install_stubs()
    sys.path.insert(0, REF)
    from ddrl4nav_amd.utils.recipe import make_weights, param_specs
    from USTC_lab.data import Experience, EasyBytes
    torch.set_num_threads(1)  # fixed summation order for the committed vectors
    torch.manual_seed(0)
    weights = make_weights(seed=0)
    net, cfg_nn = build_reference_net(weights)
    lut = (np.arange(256, dtype=np.uint8) / 255.0).astype(np.float32)  # warputils.py:300 + forward.py:102-104

    # ---------------- F5: u8 -> f32 --------------------------------------------------------
    f64 = np.arange(256, dtype=np.uint8) / 255.0  # what WarpFrameWrapper emits
    as_tensor = torch.tensor(f64, dtype=torch.float32).numpy()  # what state2tensor / to_tensor do
    assert np.array_equal(as_tensor, lut)
    np.savez(os.path.join(HERE, "f5_u8_lut.npz"), lut=lut)

    # ---------------- F1: forward ----------------------------------------------------------
    rng = np.random.default_rng(1234)
    n = 8
    frames = rng.integers(0, 256, size=(n, 4, 84, 84), dtype=np.uint8)
    # make two samples Pong-like (mostly flat background) so that the leaky branch is exercised
    frames[6] = 87
    frames[6, :, 10:26, 4:8] = 147
    frames[6, :, 40:44, 40:42] = 236
    frames[7] = 0
    acts = rng.integers(0, 6, size=n).astype(np.float32)
    x = torch.tensor(frames / 255.0, dtype=torch.float32)  # f64 -> f32 exactly as forward.py:102-104
    with torch.no_grad():
        (dist, logp), values = net([x], torch.from_numpy(acts))
        h_a = net.actor.pre([x])
        h_c = net.critic.pre([x])
        probs_play, _ = net.actor([x], None, True)  # play_mode -> raw softmax
        ent = dist.entropy()
    np.savez(os.path.join(HERE, "f1_forward.npz"), frames=frames, actions=acts,
             probs=probs_play.numpy(), p_hat=dist.probs.numpy(), logits=dist.logits.numpy(),
             logp=logp.numpy(), value=values[0].numpy()[:, 0], entropy=ent.numpy(),
             h_actor=h_a.numpy()[:, :16], h_critic=h_c.numpy()[:, :16])

    # ---------------- F2: GAE --------------------------------------------------------------
    from USTC_lab.agent.agent import Agents
    T, N = 256, 8
    rng = np.random.default_rng(2)
    vals = rng.normal(0, 1, size=(2 * T + 1, N)).astype(np.float32)
    rew = rng.choice(np.array([-1, 0, 1], np.float32), p=[0.05, 0.9, 0.05], size=(2 * T + 1, N)).astype(np.float32)
    dones = (rng.random((2 * T + 1, N)) < 1 / 50).astype(np.uint8)
    me = types.SimpleNamespace(model_dtype=np.float32,
                               discounts=np.array([cfg_nn.EXTRINSIC_DISCOUNT], dtype=np.float32).reshape(1, 1),
                               landa=cfg_nn.LANDA)

    def run_ref_gae(v, r, d):
        exps = [Experience(states=[np.zeros((N, 1), np.float32)], values=v[t:t + 1].copy(),
                           dones=d[t:t + 1].copy()) for t in range(T + 1)]
        rewards_step = r[:T + 1].reshape(T + 1, 1, N).copy()
        out = Agents._accumulate_rewards(me, exps, rewards_step)
        assert len(out) == T
        return (np.stack([e.advs for e in out]).astype(np.float32),
                np.stack([e.values[0] for e in out]).astype(np.float32))

    adv1, ret1 = run_ref_gae(vals[:T + 1], rew[:T + 1], dones[:T + 1])
    # second rollout: the (T+1)-th step is carried over as step 0 (agent.py:286-291)
    adv2, ret2 = run_ref_gae(vals[T:2 * T + 1], rew[T:2 * T + 1], dones[T:2 * T + 1])
    np.savez(os.path.join(HERE, "f2_gae.npz"), values=vals, rewards=rew, dones=dones,
             adv1=adv1, ret1=ret1, adv2=adv2, ret2=ret2, gamma=np.float64(cfg_nn.EXTRINSIC_DISCOUNT),
             landa=np.float64(cfg_nn.LANDA))

    # ---------------- F3 / F4: loss + learn ------------------------------------------------
    B = 64
    rng = np.random.default_rng(3)
    frames_b = rng.integers(0, 256, size=(B, 4, 84, 84), dtype=np.uint8)
    frames_b[::7] = (frames_b[::7] // 64) * 64  # some low-entropy frames
    xb = torch.tensor(frames_b / 255.0, dtype=torch.float32)
    with torch.no_grad():
        (dist, _), values = net([xb])
        torch.manual_seed(7)
        actions = dist.sample().to(torch.float32)
        old_logps = net.actor.log_prob_from_distribution(dist, actions)
        v0 = values[0][:, 0]
    # perturb old_logps so ratios straddle the clip range, with advantages of both signs
    old_logps = (old_logps + torch.from_numpy(rng.normal(0, 0.25, B).astype(np.float32))).contiguous()
    advs = torch.from_numpy(rng.normal(0, 1, B).astype(np.float32))
    advs[5] = 0.0
    rets = (v0 + advs).contiguous()
    exp = Experience(states=[xb.numpy()], advs=advs.numpy(), actions=actions.numpy(),
                     old_logps=old_logps.numpy(), values=rets.numpy().reshape(1, B))
    exp.to_tensor(dtype=torch.float32, device="cpu")

    # F3: one loss evaluation with autograd gradients at the head inputs
    net.zero_grad()
    pi, values = net(exp.states, exp.actions)
    distribution, log_p = pi
    log_p.retain_grad()
    values[0].retain_grad()
    ratio = torch.exp(log_p - exp.old_logps)
    actor_loss = -torch.mean(torch.where(exp.advs > 0,
                                         torch.min(ratio * exp.advs, torch.clamp(ratio, 0.8, 1.2) * exp.advs),
                                         torch.max(torch.min(ratio * exp.advs, torch.clamp(ratio, 0.8, 1.2) * exp.advs),
                                                   3 * exp.advs)))
    v_loss = net.vlossf(exp.values[0, :], values[0].squeeze())
    ent = torch.mean(distribution.entropy())
    actor_loss.backward()
    v_loss.backward()
    grads = {k: p.grad.detach().numpy().copy() for k, p in net.named_parameters()}
    gsel = {}
    for k, g in grads.items():
        gsel["gsum/" + k] = np.float64(g.astype(np.float64).sum())
        gsel["gl2/" + k] = np.float64(np.sqrt((g.astype(np.float64) ** 2).sum()))
        gsel["ghead/" + k] = g.reshape(-1)[:64].copy()
    np.savez(os.path.join(HERE, "f3_loss.npz"), frames=frames_b, actions=actions.numpy(),
             old_logps=old_logps.numpy(), advs=advs.numpy(), rets=rets.numpy(),
             actor_loss=np.float32(actor_loss.item()), v_loss=np.float32(v_loss.item()),
             ent=np.float32(ent.item()),
             total=np.float32((actor_loss + v_loss * 1.0 - ent * 0.05).item()),
             dlogp=log_p.grad.numpy(), dvalue=values[0].grad.numpy()[:, 0],
             grad_actor_linear_w=grads["actor.actor_linear.weight"], grad_actor_linear_b=grads["actor.actor_linear.bias"],
             grad_critic_linear_w=grads["critic.critic_linear.weight"], grad_critic_linear_b=grads["critic.critic_linear.bias"],
             grad_actor_conv1_b=grads["actor.pre.conv1.bias"], grad_critic_conv3_b=grads["critic.pre.conv3.bias"],
             **gsel)

    # F4: the reference's own learn() generator, 10 iterations
    net.load_state_dict({k: torch.from_numpy(v.copy()) for k, v in weights.items()}, strict=True)
    net.zero_grad()
    # inputs are those of f3_loss.npz (frames are stored once, there)
    out = {"actions": actions.numpy(), "old_logps": old_logps.numpy(),
           "advs": advs.numpy(), "rets": rets.numpy()}
    losses = []
    for it, (ld, update_time, last) in enumerate(net.learn(exp), 1):
        losses.append([ld["PpoTotalLoss"], ld["ActorLoss"], ld["VLoss"], ld["EntLoss"]])
        assert update_time == it and last is True
        if it in (1, 10):
            for k, p in net.named_parameters():
                a = p.detach().numpy()
                out["it%d/sum/%s" % (it, k)] = np.float64(a.astype(np.float64).sum())
                out["it%d/l2/%s" % (it, k)] = np.float64(np.sqrt((a.astype(np.float64) ** 2).sum()))
                out["it%d/head/%s" % (it, k)] = a.reshape(-1)[:8].copy()
            # Adam is sign-like on its first steps, so also keep a strided sample of every tensor
            for k, p in net.named_parameters():
                a = p.detach().numpy().reshape(-1)
                out["it%d/stride/%s" % (it, k)] = a[::max(1, a.size // 257)][:257].copy()
    out["losses"] = np.asarray(losses, np.float64)
    # The same reference code in float64 and with another intra-op thread count: the spread
    # between these runs is the reference's OWN sensitivity to summation order over the 10
    # Adam steps, and is what bounds the tolerance of the sequence test.
    for tag, dtype, threads in (("f64", torch.float64, 1), ("f32t8", torch.float32, 8)):
        torch.set_num_threads(threads)
        net.to(torch.float32)
        net.load_state_dict({k: torch.from_numpy(v.copy()) for k, v in weights.items()}, strict=True)
        net.to(dtype)
        net.update_time = 0
        net.actor_optim = torch.optim.Adam(net.actor.parameters(), cfg_nn.ACTOR_LEARNING_RATE)
        net.critic_optim = torch.optim.Adam(net.critic.parameters(), cfg_nn.CRITIC_LEARNING_RATE)
        exp2 = Experience(states=[xb.numpy()], advs=advs.numpy(), actions=actions.numpy(),
                          old_logps=old_logps.numpy(), values=rets.numpy().reshape(1, B))
        exp2.to_tensor(dtype=dtype, device="cpu")
        rows = [[ld["PpoTotalLoss"], ld["ActorLoss"], ld["VLoss"], ld["EntLoss"]] for ld, _, _ in net.learn(exp2)]
        out["losses_" + tag] = np.asarray(rows, np.float64)
    torch.set_num_threads(1)
    net.to(torch.float32)
    net.load_state_dict({k: torch.from_numpy(v.copy()) for k, v in weights.items()}, strict=True)
    np.savez(os.path.join(HERE, "f4_learn.npz"), **out)

    # ---------------- F6: episode-return accumulator ---------------------------------------
    from USTC_lab.agent.statistics import Status
    rng = np.random.default_rng(6)
    Tn, Nn = 400, 4
    r6 = rng.choice(np.array([-1, 0, 1], np.float32), p=[0.1, 0.8, 0.1], size=(Tn, Nn)).astype(np.float32)
    d6 = (rng.random((Tn, Nn)) < 1 / 40).astype(np.float32)
    st = types.SimpleNamespace(rewards_sum=np.zeros(Nn, np.float32), rewards_episode=np.zeros(Nn, np.float32))
    trace = np.empty((Tn, Nn), np.float32)
    for t in range(Tn):
        Status.update_reward_status(st, d6[t], r6[t])
        trace[t] = st.rewards_episode
    np.savez(os.path.join(HERE, "f6_returns.npz"), rewards=r6, dones=d6, trace=trace, final_sum=st.rewards_sum)

    # ---------------- F7: EasyBytes + weight-blob known answers ("next" rows 1,2) ----------
    eb = EasyBytes()
    kat_arrays = [np.array([[1, 2, 3], [4, 5, 6]], dtype=np.uint8), np.array([4, 6], dtype=np.float32)]
    enc = eb.encode_data(kat_arrays)  # literals of easybytes.py:177-178
    fwd = eb.encode_forward_states(1, [np.array([[1, 2, 3, 4]], dtype=np.float32)])  # easybytes.py:191-192
    blob = b"".join(net._encode_wb(v.detach().numpy()) for k, v in list(net.named_parameters())[8:12])
    np.savez(os.path.join(HERE, "f7_codec.npz"), enc=np.frombuffer(enc, np.uint8),
             fwd=np.frombuffer(fwd, np.uint8), blob_heads=np.frombuffer(blob, np.uint8))
    # ---- F8: drop-in surface: ConfigNN contract values + Experience.batch_data -------------
    import json
    contract = {}
    for k in ("NETWORK_TYPE", "USE_RND", "AC_INPUT_DIM", "EXTRINSIC_DISCOUNT", "LANDA", "LEARNING_RATE",
              "ACTOR_LEARNING_RATE", "CRITIC_LEARNING_RATE", "V_LOSS_THETA", "ENTROPY_LOSS_THETA", "PPO_CLIP",
              "DUEL_PPO_CLIP", "TRAINING_ITER_TIME", "TRAINING_MIN_BATCH", "SOFT_MAX_GRID", "CLIP_GRID", "CLIP_GRID_NUM",
              "SMOOTH_L1_LOSS", "SHARE_CNN_NET", "HALF", "MODULE_BITS", "MODEL_TO_REDIS_FREQUENCY", "ACTION_OUTPUT_DIM",
              "ACTIONS_DIM"):
        contract[k] = getattr(cfg_nn, k)
    with open(os.path.join(HERE, "f8_config_nn.json"), "w") as f:
        json.dump(contract, f, indent=1, sort_keys=True)
    rng = np.random.default_rng(8)
    steps = []
    for t in range(5):
        m = 3
        steps.append(dict(states=[rng.integers(0, 256, size=(m, 2, 4, 4), dtype=np.uint8)],
                          advs=rng.normal(size=m).astype(np.float32), actions=rng.integers(0, 6, size=m).astype(np.float32),
                          old_logps=rng.normal(size=m).astype(np.float32), values=rng.normal(size=(1, m)).astype(np.float32),
                          is_clean=rng.random(m) < 0.7))
    ref_exps = [Experience(**s) for s in steps]
    b_all = Experience.batch_data(ref_exps)
    b_clean = Experience.batch_data(ref_exps, clean=False)
    f8 = {}
    for i, s in enumerate(steps):
        for k, v in s.items():
            f8["in%d/%s" % (i, k)] = v[0] if k == "states" else v
    for tag, b in (("all", b_all), ("clean", b_clean)):
        f8[tag + "/states"] = b.states[0]
        f8[tag + "/advs"], f8[tag + "/actions"], f8[tag + "/old_logps"], f8[tag + "/values"] = b.advs, b.actions, b.old_logps, b.values
    np.savez(os.path.join(HERE, "f8_experience.npz"), **f8)

    # ---- F9: EasyBytes message framings driven through the reference's own codec ---------------
    rng = np.random.default_rng(9)
    eb2 = EasyBytes("10.2.3.4")
    fr_a = rng.integers(0, 256, size=(2, 4, 6, 6), dtype=np.uint8)
    fr_b = rng.integers(0, 256, size=(3, 4, 6, 6), dtype=np.uint8)
    vec_a = rng.normal(size=(2, 5)).astype(np.float32)
    vec_b = rng.normal(size=(3, 5)).astype(np.float32)
    # frames travel as float64 = uint8/255.0 (warputils.py:300), the second state as float32
    msg = eb2.encode_forward_states(3, [fr_a / 255.0, vec_a]) + eb2.encode_forward_states(70000, [fr_b / 255.0, vec_b])
    ids, states = eb2.decode_forward_states(msg)
    actions = rng.integers(0, 6, size=5).astype(np.float32)
    logps = rng.normal(size=5).astype(np.float32)
    values = rng.normal(size=(1, 5, 1)).astype(np.float32)
    replies = eb2.encode_forward_return_data([actions, logps, values], [2, 3])
    st_list = [rng.integers(0, 256, size=(4, 2, 3, 3), dtype=np.uint8), rng.normal(size=(4, 2)).astype(np.float16)]
    other = [rng.normal(size=4).astype(np.float32), rng.integers(0, 6, size=4).astype(np.float32),
             rng.normal(size=4).astype(np.float32), rng.normal(size=(1, 4)).astype(np.float32)]
    stats = {"RewardEpisode": 1.5, "steps": 7}
    blob = eb2.encode_backward_data([st_list] + other, stats)
    d_states, d_other, d_stats = eb2.decode_backward_data(blob)
    import marshal
    tail = len(marshal.dumps(stats))
    np.savez(os.path.join(HERE, "f9_easybytes.npz"), fr_a=fr_a, fr_b=fr_b, vec_a=vec_a, vec_b=vec_b,
             msg=np.frombuffer(msg, np.uint8), ids=np.array(ids), dec_frames=states[0], dec_vec=states[1],
             actions=actions, logps=logps, values=values, reply0=np.frombuffer(replies[0], np.uint8),
             reply1=np.frombuffer(replies[1], np.uint8), st0=st_list[0], st1=st_list[1], o0=other[0], o1=other[1],
             o2=other[2], o3=other[3], blob_head=np.frombuffer(blob[:-tail], np.uint8), tail_len=np.int64(tail),
             machine_bytes=np.frombuffer(eb2.machine_bytes, np.uint8))
    assert d_stats == stats and np.array_equal(d_states[0], st_list[0]) and np.array_equal(d_other[3], other[3])

    print("golden vectors written to", HERE)
    for f in sorted(os.listdir(HERE)):
        if f.endswith(".npz"):
            print("  %-20s %8d B" % (f, os.path.getsize(os.path.join(HERE, f))))


if __name__ == "__main__":
    main()
