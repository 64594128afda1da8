"""GPU tests of the GAIL path (BASELINE config 5) against golden vectors produced by the
reference's own GAIL / Discriminator / PPO objects (tests/golden/make_golden_gail.py: F16 classical, F17 Atari, F18 GAE;
make_golden_gail_nav.py: F22 = config 5's "nav env + discriminator", a shared NavPedPreNet under generator and discriminator)
and the pinned oracle (oracle/ddrl_oracle_gail.py).  Everything goes through the drop-in surface:
create_net(NETWORK_TYPE="gail") -> GAIL.forward / GAIL.learn -> the HIP operators behind include/ddrl.h."""
import types

import numpy as np
import pytest
import torch

from ddrl4nav_amd.utils.recipe import hash_weights

pytestmark = pytest.mark.gpu

CASES = ["f16_gail_classical", "f17_gail_atari", "f22_gail_navped"]


def _net(name, golden, max_batch=256):
    from ddrl4nav_amd.config import BaseConfig, ConfigNN
    from ddrl4nav_amd.runner import create_net
    g = golden(name)
    task = None
    if name == "f16_gail_classical":
        env = {"env_type": "gym", "env_name": "CartPole-v1", "env_num": 8, "discrete_action": True, "discrete_actions": [0, 1],
               "input_dim": 4}
        states = [g["states"]]
        seed = 16
    elif name == "f22_gail_navped":   # robot_nav with a pedestrian map (runner/utils.py:88-102) + the gail branch (:161-168)
        env = {"env_type": "robot_nav", "env_name": "robot_nav", "env_num": 8, "discrete_action": True, "discrete_actions": list(range(5)),
               "image_batch": 1, "ped_sim": {"total": 3}}
        states = [g["state0"], g["state1"], g["state2"]]
        seed, task = 22, "robot_nav"
    else:
        env = {"env_type": "gym", "env_name": "PongNoFrameskip-v4", "env_num": 8, "int_frame_stack": 4, "discrete_action": True,
               "discrete_actions": list(range(6))}
        states = [golden("f3_loss")["frames"]]      # uint8; the reference fixture saw float32(u8 / 255.0)
        seed = 17
    cfg = BaseConfig(types.SimpleNamespace(task="gail", ip="127.0.0.1"), env)
    if task:
        cfg.TASK_TYPE = task
    cfg_nn = ConfigNN(env)
    cfg_nn.NETWORK_TYPE, cfg_nn.SHARE_CNN_NET = "gail", True
    hidden = int(g["d_mlp_hidden"])
    cfg.GAN_D_MLP_LIST = [(512 + cfg.ACTIONS_DIM, hidden, "relu"), (hidden, 1, None)]
    if name == "f22_gail_navped":
        expert = [([g["expert_state0"], g["expert_state1"], g["expert_state2"]], g["expert_actions"])]   # a LIST of components
    else:
        ex_states = states[0][g["expert_index"]][::-1].copy()
        expert = [(ex_states[None], g["expert_actions"])]       # the reference-shaped batch: states [1, n, ...]
    net = create_net({"config": cfg, "config_nn": cfg_nn, "config_env": env}, max_batch=max_batch, expert_data=expert)
    assert [k for k, _ in net.named_parameters()] == list(g["names"])   # reference module tree / blob order
    w = hash_weights([(k, tuple(p.shape)) for k, p in net.named_parameters()], seed)
    res = net.load_state_dict({k: torch.from_numpy(v.copy()) for k, v in w.items()}, strict=False)
    assert not res.unexpected_keys and all(k.startswith("actor.") for k in res.missing_keys)   # alias of generator.actor
    return g, net, states, w


def _params(net):
    return {k: p.detach().cpu().numpy().copy() for k, p in net.named_parameters()}


@pytest.mark.parametrize("name", CASES)
def test_gail_forward_two_critics_and_discriminator_reward(golden, name):
    g, net, states, _ = _net(name, golden)
    B = len(g["actions"])
    acts = torch.from_numpy(g["actions"])
    (dist, logp), values = net(states, acts)
    assert len(values) == 2 and values[0].shape == (B, 1) and values[1].shape == (B, 1)
    np.testing.assert_allclose(values[0].cpu().numpy()[:, 0], g["value0"], rtol=2e-5, atol=2e-6)
    np.testing.assert_allclose(values[1].cpu().numpy()[:, 0], g["value1"], rtol=2e-5, atol=2e-6)
    np.testing.assert_allclose(logp.cpu().numpy(), g["logp"], rtol=2e-5, atol=2e-6)
    (probs, _), _ = net(states, None, True)
    np.testing.assert_allclose(probs.cpu().numpy(), g["probs"], rtol=2e-5, atol=2e-6)
    # forward.py:159-165: D_rewards = net((batch_states, actions.reshape(n, action_dim)))[:, 0]
    d = net((states, acts.reshape(B, 1)))
    assert d.shape == (B, 1)
    np.testing.assert_allclose(d.cpu().numpy()[:, 0], g["d_reward"], rtol=2e-5, atol=2e-6)
    # micro-batched evaluation gives the same scores: to the bit while every launch takes the same kernels, to fp32 rounding when
    # the 48-row launches fall below the 128 rows from which dense layers run on the 16-bit plane kernels (csrc/plin.hip)
    net.discriminator.cap, cap = 48, net.discriminator.cap
    d2 = net((states, acts.reshape(B, 1)))
    net.discriminator.cap = cap
    if B < 128:
        assert torch.equal(d, d2)
    else:
        np.testing.assert_allclose(d2.cpu().numpy(), d.cpu().numpy(), rtol=2e-6, atol=2e-7)


def _enc_signs(pre, n):
    """[a1 > 0, a2 > 0, a3 > 0] of the latest forward of an AtariPreNet (its own encoder-only context)."""
    from ctypes import byref, c_int64, c_void_p
    out = []
    for which, shp in ((0, (32, 20, 20)), (1, (64, 9, 9)), (2, (64, 7, 7))):
        p, es = c_void_p(), c_int64()
        assert pre._lib.ddrl_debug_buffer(pre._ctx, which, byref(p), byref(es)) == 0
        off = (p.value - pre._workspace.data_ptr()) // 4
        cnt = int(np.prod(shp)) * n
        out.append((pre._workspace.view(torch.float32)[off:off + cnt].reshape((n,) + shp) > 0).cpu())
    return out


def _d_decisions(net, name, g, states, w):
    """ReLU / leaky-ReLU / max-pool decisions of the discriminator's encoder on the policy batch and on the expert batch (the two
    forwards of one discriminator step), read from the kernels' activations: sign triples for the Atari encoder, {site: ReLU
    output} dicts for a nav encoder; None for an MLP encoder.  Checked against the fp32 oracle's pre-activations: at most 8
    decisions per layer differ, all with |z| < 2e-5 (fp32 noise of zero)."""
    import parity_util as P
    D = net.discriminator
    atari, nav = hasattr(D.pre, "_ctx"), hasattr(D.pre, "cat")
    if not (atari or nav):
        return None
    acts = torch.from_numpy(g["actions"]).reshape(-1, 1)
    _, onet, states_np, _ = P.gail_oracle(name)
    onet.load_weights(w)
    enc = onet.discriminator.pre
    st_np, ex_np = P.gail_state_lists(g, states_np)
    _, ex_dev = P.gail_state_lists(g, states if nav else states[0])       # what the HIP path is fed (uint8 frames for Atari)
    seq = []
    for st, st_o, a in ((states, st_np, acts), (ex_dev, ex_np, torch.from_numpy(g["expert_actions"]))):
        net((st, a))                         # D forward with the step's weights: same kernels, same decisions as in learn()
        n = st[0].shape[0]
        if atari:
            signs = _enc_signs(D.pre, n)
            with torch.no_grad():
                enc([torch.from_numpy(x) for x in st_o])
            pairs = list(zip(signs, enc.last_z))
        else:
            signs = P.relu_outputs_of(D.pre, n)
            enc.record = {}
            with torch.no_grad():
                enc([torch.from_numpy(x) for x in st_o])
            pairs = [(signs[k].reshape(z.shape) > 0, z) for k, z in enc.record.items()]
            fused = [getattr(getattr(D.pre, "c" + k[4:]), "fused", False) if k.startswith("conv") else False for k in enc.record]
            enc.record = None
            assert len(pairs) == 5   # conv1-3, fc0, fc1
        for k, (pos, z) in enumerate(pairs):
            if not atari and fused[k]:
                # a layer that pools in its epilogue keeps ONE decision per window (which element is the first maximum, and whether
                # it is positive): compare those with the oracle's relu + max-pool decisions on the same window
                H, W = z.shape[2:]
                win = lambda t: t.reshape(-1, H // 2, 2, W // 2, 2).permute(0, 2, 4, 1, 3).reshape(-1, 4, (H // 2) * (W // 2))
                zw, pw = win(torch.relu(z)), win(pos.float())                               # [planes, 4 (scan order), windows]
                pos_o, pos_k = zw.amax(1) > 0, pw.amax(1) > 0
                am_o, am_k = zw.argmax(1), pw.argmax(1)
                differ = (pos_o != pos_k) | (pos_o & pos_k & (am_o != am_k))
                assert int(differ.sum()) <= 8, (k, int(differ.sum()))
                if differ.any():   # within fp32 noise of the decision boundary: the two candidates (or the maximum and zero) are that close
                    gap = torch.where(pos_o & pos_k, (zw.gather(1, am_o[:, None]) - zw.gather(1, am_k[:, None]))[:, 0].abs(), zw.amax(1))
                    assert float(gap[differ].max()) < 2e-5, (k, float(gap[differ].max()))
                continue
            differ = pos != (z > 0)
            assert int(differ.sum()) <= 8, (k, int(differ.sum()))
            if differ.any():
                assert float(z[differ].abs().max()) < 2e-5, (k, float(z[differ].abs().max()))
        seq.append(signs)
    return seq


@pytest.mark.parametrize("name", CASES)
def test_discriminator_gradient_vs_float64_oracle(golden, name):
    """One WGAN term pair of Discriminator.learn (GAIL.py:76-81): the full gradient mean(D(policy)) - mean(D(expert)) before the
    clip, every tensor within 1e-5 max|g| of the float64 oracle (under the kernel's own leaky-ReLU decisions for the Atari
    encoder, which are checked to be within fp32 noise of the oracle's).  The two terms nearly cancel in the encoder's layers
    (the expert batch holds the same frames): the bound is on the DIFFERENCE."""
    import parity_util as P
    from ddrl4nav_amd.data import Experience
    g, net, states, w = _net(name, golden)
    forced = _d_decisions(net, name, g, states, w)
    exp = Experience(states=states, advs=g["advs"], actions=g["actions"], old_logps=g["old_logps"], values=g["rets"])
    D = net.discriminator
    next(D.learn(exp))
    st = D.stats()
    _, onet, states_np, _ = P.gail_oracle(name)
    onet.load_weights(w)
    onet.double()
    Dn = onet.discriminator
    if forced is not None and isinstance(forced[0], dict):
        Dn.sub_seq = list(forced)
    elif forced is not None:
        Dn.pre.forced_seq = forced
    t = lambda k: torch.from_numpy(g[k]).double()
    st_np, ex_np = P.gail_state_lists(g, states_np)
    s, ex = [torch.from_numpy(a).double() for a in st_np], [torch.from_numpy(a).double() for a in ex_np]
    loss = torch.mean(Dn((s, t("actions").reshape(-1, 1)))) - torch.mean(Dn((ex, t("expert_actions"))))
    loss.backward()
    gn = float(torch.sqrt(sum((p.grad ** 2).sum() for p in Dn.parameters())))
    np.testing.assert_allclose(st["GradNorm"], gn, rtol=1e-5)
    worst = {}
    for k, p in D.named_parameters():
        off = (p.data_ptr() - D.params.data_ptr()) // 4
        mine = (D.grads[off:off + p.numel()].cpu().numpy().reshape(p.shape) / st["ClipCoef"]).astype(np.float64)
        want = dict(Dn.named_parameters())[k].grad.numpy()
        scale = float(np.abs(want).max())
        if scale < 1e-12:           # the score layer's bias: sum(1/n) - sum(1/m), zero up to float64 rounding
            assert float(np.abs(mine).max()) <= 1e-12, k
            continue
        worst[k] = float(np.abs(mine - want).max()) / scale
        assert worst[k] <= 1e-5, (k, worst[k])
    print(name, "D gradient, max |dg| / max |g| per tensor:", {k: "%.1e" % v for k, v in worst.items()})


@pytest.mark.parametrize("name", CASES)
def test_gail_learn_matches_reference(golden, name):
    """GAIL.learn = one discriminator step (last=False) then ten PPO iterations with the GAIL critic (last=True):
    protocol, losses, and every parameter tensor relative to the reference's own fp32 spread around its float64 run."""
    import parity_util as P
    from ddrl4nav_amd.data import Experience
    g, net, states, w = _net(name, golden)
    # The float64 yardstick advances step by step and takes the kernels' leaky-ReLU decisions for the Atari encoders (a
    # pre-activation within fp32 noise of zero comes out on either side depending on the summation order; one such flip moves
    # hundreds of conv weight-gradient elements by a few 1e-3, and the sign-like first RMSprop / Adam steps turn that into
    # flipped updates): _d_decisions checks that those decisions are within 2e-5 of the oracle's own.
    forced = _d_decisions(net, name, g, states, w)
    atari, nav = forced is not None and not isinstance(forced[0], dict), forced is not None and isinstance(forced[0], dict)
    ora = P.GailStepper(name)
    exp = Experience(states=states, advs=g["advs"], actions=g["actions"], old_logps=g["old_logps"], values=g["rets"])
    tag = "gail_" + name[:3]
    env = P.loss_envelope(g["losses"], g["losses_f64"], g["losses_f32t8"], g["losses_perm"], *P.backend_losses(name))
    B = len(g["actions"])
    rows, seen_d = [], 0
    for loss_item, update_time, last in net.learn(exp):
        if not last:
            seen_d += 1
            assert set(loss_item) == {"Gail[D]BackUpTime", "Gail[D]Loss"} and update_time == seen_d
            excess = abs(loss_item["Gail[D]Loss"] - g["d_loss"][0]) - (2e-5 * abs(g["d_loss"][0]) + 2e-7)
            d_spread = max(float(g["d_loss_spread"]), float(P._backend(name, P.NAV_BACKEND).get("d_loss_spread", 0.0)))
            P.MARGINS.check(tag, "d_loss", max(0.0, excess / max(d_spread, 1e-9)))
            ora.d_step(forced)
            worst = P.gail_deviation_from(name, "D1", _params(net), ora.params(), ora.p0)
            for k, (v, pname) in worst.items():
                P.MARGINS.check(tag, "D1_param_" + k, v, "(%s)" % pname)
            P.MARGINS.record_onednn_only(tag, worst, "D1_param_%s")
            continue
        rows.append([loss_item[k] for k in ("PpoTotalLoss", "ActorLoss", "VLoss", "EntLoss")])
        it = len(rows)
        assert update_time == it
        # the generator's shared prenet holds this iteration's activations: its decisions for the yardstick's iteration
        l64 = np.asarray(ora.g_step(_enc_signs(net.generator.prenet, B) if atari else
                                    P.relu_outputs_of(net.generator.prenet, B) if nav else None))
        row = g["losses"][it - 1] if it == 1 else l64     # iteration 1 (nothing stepped yet): the reference's stored value
        excess = np.abs(np.asarray(rows[-1]) - row) - (1e-5 * np.abs(row) + 2e-6)
        P.MARGINS.check(tag, "loss_env", max(0.0, float(np.max(excess / np.maximum(env[it - 1], 1e-12)))), "(iteration %d)" % it)
        if it in (1, 10):
            worst = P.gail_deviation_from(name, "it%d" % it, _params(net), ora.params(), ora.p0)
            for k, (v, pname) in worst.items():
                P.MARGINS.check(tag, "param_%s_it%d" % (k, it), v, "(%s)" % pname)
            P.MARGINS.record_onednn_only(tag, worst, "param_%%s_it%d" % it)
    assert seen_d == 1 and len(rows) == 10
    got = _params(net)
    for k in got:   # in no optimiser (ppo.py:39,61-62): the GAIL critic has not moved
        if k.startswith("gail_critic."):
            assert np.array_equal(got[k], w[k])
    assert net.discriminator.lr == float(g["d_lr_after"])


def test_discriminator_260_steps_cross_the_steplr_boundary(golden):
    """RMSprop + grad-norm clip + StepLR(250, 0.95): loss trajectory, learning rate and final parameters of a
    discriminator-only run (reference: f16 `d_only_*`)."""
    import parity_util as P
    from ddrl4nav_amd.data import Experience
    g, net, states, _ = _net("f16_gail_classical", golden)
    exp = Experience(states=states, advs=g["advs"], actions=g["actions"], old_logps=g["old_logps"], values=g["rets"])
    D = net.discriminator
    ref, ref64 = g["d_only_loss"], g["d_only_loss_f64"]
    env = np.maximum.accumulate(np.abs(ref - ref64)) + 1e-9
    worst = 0.0
    for step in range(1, 261):
        items = list(D.learn(exp))
        assert len(items) == 1 and items[0][1] == step and items[0][2] is True
        assert D.lr == float(g["d_only_lr"][step - 1])
        excess = abs(items[0][0]["Gail[D]Loss"] - ref[step - 1]) - (2e-5 * abs(ref[step - 1]) + 2e-7)
        worst = max(worst, excess / env[step - 1])
    P.MARGINS.check("gail_d_only", "loss_env", max(0.0, worst))
    assert D.lr == 5e-5 * 0.95
    w0 = hash_weights([(k, tuple(p.shape)) for k, p in net.named_parameters()], 16)
    for k, p in D.named_parameters():
        a = p.detach().cpu().numpy().reshape(-1)
        want = g["Dend/stride/discriminator." + k]
        got = a[::max(1, a.size // 129)][:129]
        start = w0["discriminator." + k].reshape(-1)[::max(1, a.size // 129)][:129]
        moved = np.abs(want - start).max()     # 260 RMSprop steps: ~1e-2
        if moved < 1e-5:
            # the score layer's bias: its gradient is sum(1/n) - sum(1/m) = 0; the reference's sums cancel exactly and so
            # must ours (ddrl_op_colsum), otherwise RMSprop turns the residue into a random walk of ~lr per step
            assert np.abs(got - want).max() <= 1e-6, (k, np.abs(got - want).max())
            continue
        # RMSprop normalises by sqrt(E[g^2]): like Adam it amplifies summation-order noise on elements whose gradient is
        # tiny; the limit (parity_util.PARAM_LIMIT) is a fraction of the distance the tensor travelled in 260 steps
        P.MARGINS.check("gail_d_only", "param_end", float(np.abs(got - want).max() / moved), "(%s)" % k)


def test_agents_two_row_gae_bit_exact(golden):
    """Agents._accumulate_rewards with the GAIL value row (agent.py:97-101,124-140) == the reference, bit for bit."""
    from ddrl4nav_amd.agent import Agents
    from ddrl4nav_amd.config import BaseConfig, ConfigNN
    from ddrl4nav_amd.data import Experience
    g = golden("f18_gae_two_rows")
    env = {"env_type": "gym", "env_name": "PongNoFrameskip-v4", "env_num": 6, "discrete_action": True,
           "discrete_actions": list(range(6)), "agent_num_per_env": 1, "batch_num_per_env": 6}
    cfg_nn = ConfigNN(env)
    cfg_nn.NETWORK_TYPE = "gail"
    # reference default: network_type is hard-coded to 'ppo' (agent.py:95) -> one row, even for a GAIL net
    ref_like = Agents(config=BaseConfig(types.SimpleNamespace(task="t", ip="127.0.0.1"), env), config_nn=cfg_nn, config_env=env)
    assert ref_like.network_type == "ppo" and ref_like.value_dim_num == 1 and ref_like.reward_dim_num == 1
    cfg_nn.GAIL_TWO_ROW_VALUES = True        # explicit opt-in to the branch the reference leaves dead (agent.py:97-101)
    cfg_nn.GAN_DISCOUNT = float(g["discounts"][1])
    ag = Agents(config=BaseConfig(types.SimpleNamespace(task="t", ip="127.0.0.1"), env), config_nn=cfg_nn, config_env=env)
    assert ag.value_dim_num == 2 and ag.reward_dim_num == 2 and ag.discounts.shape == (2, 1)
    T = g["values"].shape[0] - 1
    exps = [Experience(states=None, values=g["values"][t].copy(), dones=g["dones"][t].copy()) for t in range(T + 1)]
    out = ag._accumulate_rewards(exps, g["rewards"])
    assert len(out) == T
    assert np.array_equal(np.stack([e.advs for e in out]), g["adv"])
    assert np.array_equal(np.stack([e.values for e in out]), g["ret"])


def test_gail_config5_full_size_properties(golden):
    """BASELINE config 5 at the full per-GPU batch (256 envs x 256 steps = 65,536 samples through the Atari encoder): size-independent
    properties of the discriminator step and of the generator's PPO iteration with the GAIL critic.
    (a) data-parallel additivity of the discriminator: the gradient of mean(D(policy)) - mean(D(expert)) over the whole batch equals
        the sum of two uneven shards' gradients when every shard divides by the TOTAL sizes -- what the all-reduce of nn/gail.py sums;
    (b) duplicating policy and expert batch leaves the discriminator's mean gradient and loss unchanged;
    (c) the generator's gradient over micro-batches (max_batch 20,000: three full chunks and a ragged one) equals the one-chunk one."""
    g, net, _, _ = _net("f17_gail_atari", golden, max_batch=65536)
    D = net.discriminator
    B = 65536
    gen = torch.Generator(device="cuda")
    gen.manual_seed(505)
    base = torch.randint(0, 256, (B // 2, 4, 84, 84), dtype=torch.uint8, device="cuda", generator=gen)
    frames = torch.cat([base, base])
    half = lambda t: torch.cat([t, t]).contiguous()
    acts = half(torch.randint(0, 6, (B // 2,), device="cuda", generator=gen).float())
    ex_frames = frames.flip(0).contiguous()
    ex_acts = half(torch.randint(0, 6, (B // 2,), device="cuda", generator=gen).float()).reshape(B, 1)
    n = D.n_params
    D._ensure_packed()

    def d_grad(pol, pol_a, ex, ex_a, n_pol, n_ex):
        D._pass([pol], pol_a, +1.0, True, n_total=n_pol)
        D._pass([ex], ex_a, -1.0, False, n_total=n_ex)
        return D.grads[:n].clone(), float(D._loss.item())

    full, loss_full = d_grad(frames, acts, ex_frames, ex_acts, B, B)
    scale = full.abs().max().item()
    assert scale > 0
    cut = 40000
    acc, loss_acc = torch.zeros_like(full), 0.0
    for sl in (slice(0, cut), slice(cut, B)):
        gsh, lsh = d_grad(frames[sl], acts[sl], ex_frames[sl], ex_acts[sl], B, B)
        acc += gsh
        loss_acc += lsh
    assert (acc - full).abs().max().item() <= 1e-4 * scale
    np.testing.assert_allclose(loss_acc, loss_full, rtol=1e-4, atol=1e-7)
    one, loss_one = d_grad(frames[:B // 2], acts[:B // 2], ex_frames[B // 2:], ex_acts[B // 2:], B // 2, B // 2)   # (b)
    assert (one - full).abs().max().item() <= 1e-4 * scale
    np.testing.assert_allclose(loss_one, loss_full, rtol=1e-4, atol=1e-7)
    # (c) generator: micro-batched PPO iteration with the GAIL critic == one chunk
    G = net.generator
    old = half(torch.full((B // 2,), -1.79, device="cuda") + 0.2 * torch.randn(B // 2, device="cuda", generator=gen))
    adv = half(torch.randn(B // 2, device="cuda", generator=gen))
    rets = torch.stack([half(torch.randn(B // 2, device="cuda", generator=gen)) for _ in range(2)])
    G._ensure_packed()
    G._iter_chunk([frames], B, acts, old, adv, rets[0].contiguous(), B, [rets[1].contiguous()])
    g_full = G.gtmp[:G.n_params + 3].clone()
    acc = torch.zeros_like(g_full)
    cap = 20000
    for lo in range(0, B, cap):
        hi = min(B, lo + cap)
        G._iter_chunk([frames[lo:hi]], hi - lo, acts[lo:hi], old[lo:hi], adv[lo:hi], rets[0, lo:hi].contiguous(), B, [rets[1, lo:hi].contiguous()])
        acc += G.gtmp[:G.n_params + 3]
    gs = g_full[:G.n_params].abs().max().item()
    assert (acc[:G.n_params] - g_full[:G.n_params]).abs().max().item() <= 1e-4 * gs
    np.testing.assert_allclose(acc[G.n_params:].cpu().numpy(), g_full[G.n_params:].cpu().numpy(), rtol=1e-4, atol=1e-7)


def test_gail_config5_full_size_properties_on_the_nav_encoder(golden):
    """BASELINE config 5 on ITS OWN encoder at size: GAIL over the shared NavPedPreNet(1 + 3 channels) (runner/utils.py:98-102,161-168;
    reference arithmetic GAIL.py:73-94), 32,768 samples in micro-batches of 4,096 -- the properties of
    test_gail_config5_full_size_properties, on the nav kernels (csrc/fconv.hip, pconv.hip, plin.hip):
    (a) data-parallel additivity of the discriminator step over two uneven shards that divide by the TOTAL sizes (what the all-reduce
        of nn/gail.py sums), gradient and loss;
    (b) duplicating policy and expert batch leaves the discriminator's mean gradient and loss unchanged;
    (c) the generator's PPO gradient with the GAIL critic over 8 micro-batches equals the sum over two uneven shards of it.
    Tolerance 1e-4 of the largest gradient element: sums over 32,768 samples in different association orders (fp32)."""
    g, net, _, _ = _net("f22_gail_navped", golden, max_batch=4096)
    D = net.discriminator
    B = 32768
    gen = torch.Generator(device="cuda")
    gen.manual_seed(522)
    half = lambda t: torch.cat([t, t]).contiguous()
    img = half((torch.rand((B // 2, 1, 48, 48), device="cuda", generator=gen) < 0.3).float())
    vec = half(torch.randn((B // 2, 9), device="cuda", generator=gen))
    ped = half((torch.rand((B // 2, 3, 48, 48), device="cuda", generator=gen) < 0.1).float())
    states = [img, vec, ped]
    ex_states = [s.flip(0).contiguous() for s in states]
    acts = half(torch.randint(0, 5, (B // 2,), device="cuda", generator=gen).float())
    ex_acts = half(torch.randint(0, 5, (B // 2,), device="cuda", generator=gen).float()).reshape(B, 1)
    n = D.n_params
    D._ensure_packed()
    cut_of = lambda ts, sl: [t[sl] for t in ts]

    def d_grad(pol, pol_a, ex, ex_a, n_pol, n_ex):
        D._pass(pol, pol_a, +1.0, True, n_total=n_pol)
        D._pass(ex, ex_a, -1.0, False, n_total=n_ex)
        return D.grads[:n].clone(), float(D._loss.item())

    full, loss_full = d_grad(states, acts, ex_states, ex_acts, B, B)
    scale = full.abs().max().item()
    assert scale > 0
    cut = 20000                      # 4 full micro-batches + a ragged one | 3 full + a ragged one
    acc, loss_acc = torch.zeros_like(full), 0.0
    for sl in (slice(0, cut), slice(cut, B)):
        gsh, lsh = d_grad(cut_of(states, sl), acts[sl], cut_of(ex_states, sl), ex_acts[sl], B, B)
        acc += gsh
        loss_acc += lsh
    assert (acc - full).abs().max().item() <= 1e-4 * scale
    np.testing.assert_allclose(loss_acc, loss_full, rtol=1e-4, atol=1e-7)
    h = slice(0, B // 2)
    one, loss_one = d_grad(cut_of(states, h), acts[h], cut_of(ex_states, slice(B // 2, B)), ex_acts[B // 2:], B // 2, B // 2)   # (b)
    assert (one - full).abs().max().item() <= 1e-4 * scale
    np.testing.assert_allclose(loss_one, loss_full, rtol=1e-4, atol=1e-7)
    # (c) generator with the GAIL critic: gradient of the whole batch (8 micro-batches, accumulated by learn's rule) = sum of two shards'
    G = net.generator
    old = half(torch.full((B // 2,), -1.6, device="cuda") + 0.2 * torch.randn(B // 2, device="cuda", generator=gen))
    adv = half(torch.randn(B // 2, device="cuda", generator=gen))
    rets = torch.stack([half(torch.randn(B // 2, device="cuda", generator=gen)) for _ in range(2)])
    G._ensure_packed()

    def g_grad(sl):
        tot = torch.zeros(G.n_params + 3, device="cuda")
        lo0, hi0 = sl.start, sl.stop
        for lo in range(lo0, hi0, G.cap):
            hi = min(hi0, lo + G.cap)
            G._iter_chunk([t[lo:hi] for t in states], hi - lo, acts[lo:hi], old[lo:hi], adv[lo:hi], rets[0, lo:hi].contiguous(), B,
                          [rets[1, lo:hi].contiguous()])
            tot += G.gtmp[:G.n_params + 3]
        return tot

    g_full = g_grad(slice(0, B))
    acc = g_grad(slice(0, cut)) + g_grad(slice(cut, B))
    gs = g_full[:G.n_params].abs().max().item()
    assert gs > 0 and (acc[:G.n_params] - g_full[:G.n_params]).abs().max().item() <= 1e-4 * gs
    np.testing.assert_allclose(acc[G.n_params:].cpu().numpy(), g_full[G.n_params:].cpu().numpy(), rtol=1e-4, atol=1e-7)
    # duplicated batch: the first half alone (means over B / 2) gives the same mean gradient
    acc2 = torch.zeros_like(g_full)
    for lo in range(0, B // 2, G.cap):
        hi = lo + G.cap
        G._iter_chunk([t[lo:hi] for t in states], hi - lo, acts[lo:hi], old[lo:hi], adv[lo:hi], rets[0, lo:hi].contiguous(), B // 2,
                      [rets[1, lo:hi].contiguous()])
        acc2 += G.gtmp[:G.n_params + 3]
    assert (acc2[:G.n_params] - g_full[:G.n_params]).abs().max().item() <= 1e-4 * gs
