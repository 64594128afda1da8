"""GPU parity of the operator-composed nets (nn/generic.py: nav / MLP encoders, Gaussian and
categorical actors, both PPO optimise branches) against golden vectors made by importing the
reference (tests/golden/make_golden_nav.py) and against the CPU oracle (oracle/ddrl_oracle_nav.py,
itself pinned to the same fixtures).  Tolerances as in test_gpu_parity.py."""
import types

import numpy as np
import pytest
import torch

from ddrl4nav_amd.utils.recipe import hash_weights, sample_uniform

pytestmark = pytest.mark.gpu


def _configs(env, task="robot_nav", shared=False):
    from ddrl4nav_amd.config import BaseConfig, ConfigNN
    parse = types.SimpleNamespace(task="test", ip="127.0.0.1")
    base = {"env_type": "gym", "env_name": "PongNoFrameskip-v4", "env_num": 8}
    cfg = BaseConfig(parse, dict(base, **env))
    cfg.TASK_TYPE = task
    cfg_nn = ConfigNN(env)
    cfg_nn.SHARE_CNN_NET = shared
    return {"config": cfg, "config_nn": cfg_nn, "config_env": env}


def _make(name, max_batch=64):
    from ddrl4nav_amd.runner import create_net
    if name in ("f13_nav1d_gauss", "f27_nav1d_unaligned"):
        c = _configs({"discrete_action": False, "act_dim": 2, "image_batch": 1, "ped_sim": {"total": 3}})
        seed = 13 if name == "f13_nav1d_gauss" else 27
    elif name == "f14_navped_shared":
        c = _configs({"discrete_action": True, "discrete_actions": list(range(5)), "image_batch": 1, "ped_sim": {"total": 3}},
                     shared=True)
        seed = 14
    elif name in ("f25_navpre_shared", "f26_navpre_unaligned"):
        # no pedestrian map: create_net picks the image-only NavPreNet(image_batch) as the shared encoder (runner/utils.py:104)
        c = _configs({"discrete_action": True, "discrete_actions": list(range(5)), "image_batch": 1, "ped_sim": {"total": 0}},
                     shared=True)
        seed = 25
    else:
        c = _configs({"discrete_action": True, "discrete_actions": [0, 1], "input_dim": 4}, task="classical")
        seed = 15
    net = create_net(c, max_batch=max_batch)
    w = hash_weights([(k, tuple(p.shape)) for k, p in net.named_parameters()], seed)
    net.load_state_dict({k: torch.from_numpy(v.copy()) for k, v in w.items()})
    return net, w


def _states(g):
    n = len([k for k in g.files if k.startswith("state")])
    return [g["state%d" % i] for i in range(n)]


CASES = ["f13_nav1d_gauss", "f14_navped_shared", "f15_mlp_classical", "f25_navpre_shared"]


@pytest.mark.parametrize("name", CASES)
def test_generic_net_forward_loss_gradients_golden(golden, name):
    g = golden(name)
    net, _ = _make(name, max_batch=256)  # one micro-batch: the test inspects the chunk gradient arena directly
    assert [k for k, _ in net.named_parameters()] == list(g["names"])          # reference parameter order
    assert list(net.state_dict().keys()) == list(g["names"])                    # .pt checkpoints interchange
    states = _states(g)
    acts = torch.from_numpy(g["actions"])
    (dist, logp), values = net(states, acts)
    np.testing.assert_allclose(values[0].cpu().numpy()[:, 0], g["value"], rtol=2e-5, atol=2e-6)
    np.testing.assert_allclose(logp.cpu().numpy(), g["logp"], rtol=2e-5, atol=2e-5)
    np.testing.assert_allclose(dist.entropy().cpu().numpy(), g["entropy"], rtol=1e-5, atol=1e-6)
    (play, _), _ = net(states, None, True)
    np.testing.assert_allclose(play.cpu().numpy(), g["dist_out"], rtol=2e-5, atol=2e-6)
    # one PPO iteration's loss terms and gradients (no optimiser step yet: inspect the grad arena)
    B = len(g["advs"])
    dev = lambda k: torch.from_numpy(g[k]).cuda()
    net._ensure_packed()
    net._iter_chunk(net._stage(states, 0, B), B, dev("actions"), dev("old_logps"), dev("advs"), dev("rets"), B)
    tail = net.gtmp[net.n_params:net.n_params + 3].cpu().numpy()
    np.testing.assert_allclose(tail, g["loss4"][1:], rtol=2e-5, atol=2e-6)
    flat = net.gtmp[:net.n_params].cpu().numpy()
    off = 0
    for k, p in net.named_parameters():
        n = p.numel()
        got = flat[off:off + n]
        off += n
        l2 = float(g["gl2/" + k])
        # (the norm carries the same decision sensitivity as the samples below: 2e-2 here, 2e-5 per tensor under aligned decisions)
        np.testing.assert_allclose(np.sqrt((got.astype(np.float64) ** 2).sum()), l2, rtol=2e-2, atol=1e-9, err_msg=k)
        scale = max(np.abs(got).max(), l2 / np.sqrt(n))
        d = np.abs(got[::max(1, n // 129)][:129] - g["gstride/" + k])
        # beside the reference's stored samples: a ReLU / max-pool decision within rounding of its boundary flips with the
        # summation order and moves every gradient element upstream of it by ~1e-3 -- bounded here, pinned below
        assert d.max() <= 2e-2 * scale + 1e-9, (k, d.max(), scale)   # coarse: ONE flipped decision shows up here at up to ~1e-2 of an element
    # ... and the FULL gradient against the float64 oracle under the kernels' own ReLU / max-pool decisions: 2e-5 max|g| per tensor
    import parity_util as P
    ora = P.NavStepper(name)
    for e, sub in zip(ora.encoders, _relu_outputs(net, B)):
        e.sub = sub
    total, al, vl, _ = ora.N.losses(ora.net, ora.states, *ora.args)
    if ora.net.shared:
        total.backward()
    else:
        al.backward()
        vl.backward()
    off, worst = 0, {}
    for k, p in ora.net.named_parameters():
        n = p.numel()
        want, got = p.grad.numpy().ravel(), flat[off:off + n].astype(np.float64)
        off += n
        scale = float(np.abs(want).max())
        if scale == 0.0:
            continue
        worst[k] = float(np.abs(got - want).max()) / scale
        assert worst[k] <= 2e-5, (k, worst[k])
    print(name, "gradient vs float64 under the kernels' decisions, worst tensor: %.2e" % max(worst.values()))


@pytest.mark.parametrize("name", ["f26_navpre_unaligned", "f27_nav1d_unaligned"])
def test_nav_gradient_unaligned(golden, name):
    """The nav gradient checks with NO transfer of decisions.  F26 = the shared NavPreNet(1), F27 = BASELINE config
    4's own net (NavPreNet1D x2 + GaussionActor(2): the 7x7 / 5x5 / 3x3 pooled stack, the laser branch, both optimiser groups), each on
    32 samples selected (by the reference in float64, tests/golden/make_golden_navpre.py) so that every ReLU / max-pool decision of
    the forward is at least 8.6e-6 / ~5e-6 of its site's largest pre-activation away from a tie.  The kernels' pre-activations are
    accurate to ~1e-7 of that scale, so they must take the reference's decisions ON THEIR OWN: asserted through the decision digests
    (per sample and site: how many windows / units are active, and which window entry won), then the FULL gradient of every tensor is
    held to 2e-5 max|g| against the float64 oracle running with ITS OWN decisions (pinned to the reference on these batches by
    test_oracle_golden.py::test_no_tie_batch_...) and against the reference's stored fp32 gradient."""
    import parity_util as P
    g = golden(name)
    net, w = _make(name, max_batch=64)
    assert [k for k, _ in net.named_parameters()] == list(g["names"])
    states = _states(g)
    B = len(g["advs"])
    (dist, logp), values = net(states, torch.from_numpy(g["actions"]))
    np.testing.assert_allclose(values[0].cpu().numpy()[:, 0], g["value"], rtol=2e-5, atol=2e-6)
    np.testing.assert_allclose(logp.cpu().numpy(), g["logp"], rtol=2e-5, atol=2e-5)
    dev = lambda k: torch.from_numpy(g[k]).cuda()
    net._ensure_packed()
    net._iter_chunk(net._stage(states, 0, B), B, dev("actions"), dev("old_logps"), dev("advs"), dev("rets"), B)
    np.testing.assert_allclose(net.gtmp[net.n_params:net.n_params + 3].cpu().numpy(), g["loss4"][1:], rtol=2e-5, atol=2e-6)
    # ---- the kernels' own decisions, as digests: identical to the reference's
    prefixes = [""] if net.share_cnn_net else ["actor.pre/", "critic.pre/"]
    for pre, e in zip(prefixes, net._encs):
        for site, blk in (("conv1", e.c1), ("conv2", e.c2), ("conv3", e.c3)):
            a = blk.relu_output(B).cpu()
            n, c, h, wd = a.shape
            win = a.view(n, c, h // 2, 2, wd // 2, 2).permute(0, 1, 2, 4, 3, 5).reshape(n, c, h // 2, wd // 2, 4)
            mx, arg = win.max(dim=-1)
            pos = mx > 0
            assert np.array_equal(pos.reshape(n, -1).sum(1).numpy(), g["digest/%s%s_positive" % (pre, site)]), pre + site
            assert np.array_equal((arg * pos).reshape(n, -1).sum(1).numpy(), g["digest/%s%s_argsum" % (pre, site)]), pre + site
        assert np.array_equal((e.cat[:B, e.extra:e.extra + 512] > 0).sum(1).cpu().numpy(), g["digest/%sfc0_positive" % pre])
        assert np.array_equal((e.f1[:B] > 0).sum(1).cpu().numpy(), g["digest/%sfc1_positive" % pre])
        if e.extra:
            assert np.array_equal((e.cat[:B, :e.extra] > 0).sum(1).cpu().numpy(), g["digest/%sfc_1d_positive" % pre])
    # ---- the full gradient: float64 oracle with its OWN decisions (no `sub`), and the reference's stored fp32 gradient
    flat = net.gtmp[:net.n_params].cpu().numpy().astype(np.float64)
    ora = P.NavStepper(name)
    total, al, vl, _ = ora.N.losses(ora.net, ora.states, *ora.args)
    if ora.net.shared:
        total.backward()
    else:
        al.backward()
        vl.backward()
    off, worst = 0, {}
    for k, p in ora.net.named_parameters():
        n = p.numel()
        got = flat[off:off + n]
        off += n
        if p.grad is None:                 # actor.log_std has a gradient only through the actor loss; nothing else is ever None here
            continue
        want = p.grad.numpy().ravel()
        scale = float(np.abs(want).max())
        if scale == 0.0:
            assert np.abs(got).max() == 0.0, k
            continue
        worst[k] = float(np.abs(got - want).max()) / scale
        assert worst[k] <= 2e-5, (k, worst[k])
        ref = g["gfull/" + k] if "gfull/" + k in g.files else g["gstride/" + k]
        sub = got if "gfull/" + k in g.files else got[::max(1, n // 4097)][:4097]
        assert np.abs(sub - ref).max() <= 2e-5 * float(g["gmax/" + k]), (k, "vs the reference's fp32 gradient")
    print(name, "un-aligned gradient vs float64, worst tensor: %.2e (%s)" % (max(worst.values()), max(worst, key=worst.get)))


def _relu_outputs(net, n):
    """[{site: ReLU output of the latest forward}] per encoder of a GenericPPO, in the oracle's encoder order."""
    import parity_util as P
    return [P.relu_outputs_of(e, n) for e in net._encs]


@pytest.mark.parametrize("name", CASES)
def test_generic_net_learn_sequence_golden(golden, name):
    """Ten PPO iterations through the drop-in surface against the reference's stored trajectory.  Losses: single-step
    tolerance + c x the reference's own spread (float64 / 8-thread / permuted-batch runs); parameters after iterations
    1 and 10: every tensor, all elements, relative to the reference's own fp32 spread around its float64 run
    (tests/parity_util.py:deviation_ratios; c from tests/golden/margins.json)."""
    import parity_util as P
    from ddrl4nav_amd.data import Experience
    g = golden(name)
    sp = golden(name[:3] + "b_spread")
    net, _ = _make(name, max_batch=256)   # one micro-batch: the activation buffers then hold the whole batch's decisions
    exp = Experience(states=_states(g), advs=g["advs"], actions=g["actions"], old_logps=g["old_logps"],
                     values=g["rets"].reshape(1, -1))
    ref = g["losses"]
    env = P.loss_envelope(ref, sp["losses_f64"], g["losses_f32t8"], sp["losses_perm"], *([sp["losses_noise"]] if "losses_noise" in sp.files else []),
                          *P.backend_losses(name))   # incl. the reference on torch's native convolution backend (parity_util.NAV_BACKEND)
    tag = "generic_" + name[:3]
    seen = 0
    # the float64 yardstick advances one iteration behind the kernels and takes THEIR ReLU / max-pool decisions (one flipped
    # decision moves every conv-weight gradient upstream of it by ~1e-3, and Adam's sign-like first steps turn that into
    # hundreds of flipped updates); its arithmetic, weights and optimiser state stay its own
    ora = P.NavStepper(name)
    B = len(g["advs"])
    for loss_items, update_time, last in net.learn(exp):
        seen += 1
        assert update_time == seen and last is True
        got = np.array([loss_items[k] for k in ("PpoTotalLoss", "ActorLoss", "VLoss", "EntLoss")])
        l64 = np.array(ora.step(_relu_outputs(net, B)))
        # iteration 1 (nothing stepped yet): the reference's stored losses at the single-step tolerance of the forward / loss tests
        # (1e-4 relative: conv sums over thousands of signed terms); later iterations: the decision-aligned float64 yardstick
        # within c x the reference's own envelope
        row = ref[0] if seen == 1 else l64
        excess = np.abs(got - row) - (1e-4 * np.abs(row) + 1e-5)
        P.MARGINS.check(tag, "loss_env", max(0.0, float(np.max(excess / np.maximum(env[seen - 1], 1e-12)))), "(iteration %d)" % seen)
        if seen in (1, 10):
            worst = P.nav_param_deviation(name, seen, {k: p.detach().cpu().numpy() for k, p in net.named_parameters()},
                                          p64=ora.params(), p0=ora.p0)
            for k, (v, pname) in worst.items():
                P.MARGINS.check(tag, "param_%s_it%d" % (k, seen), v, "(%s)" % pname)
            P.MARGINS.record_onednn_only(tag, worst, "param_%%s_it%d" % seen)
    assert seen == 10


@pytest.mark.parametrize("name", ["f15_mlp_classical", "f13_nav1d_gauss"])
def test_generic_deferred_loss_readback_yields_the_same_values(golden, name):
    """config_nn.DEFERRED_LOSS_READBACK on the operator-composed PPO (nn/generic.py:learn): all iterations enqueued, every iteration's
    loss terms copied to its own pinned row, ONE synchronisation, then the same items -- keys, values, update_time -- and the same
    parameters as the per-iteration protocol of the reference (ppo.py:131-146), bit for bit."""
    from ddrl4nav_amd.data import Experience
    g = golden(name)
    runs = []
    for deferred in (False, True):
        net, _ = _make(name, max_batch=48)      # two micro-batches (one of them ragged) per iteration
        net.deferred_stats = deferred
        exp = Experience(states=_states(g), advs=g["advs"], actions=g["actions"], old_logps=g["old_logps"], values=g["rets"].reshape(1, -1))
        items = [(dict(l), ut, last) for l, ut, last in net.learn(exp)]
        runs.append((items, net.params.clone()))
    (a, pa), (b, pb) = runs
    assert len(a) == len(b) == 10 and torch.equal(pa, pb)
    for (la, ua, fa), (lb, ub, fb) in zip(a, b):
        assert ua == ub and fa is fb is True and set(la) == set(lb)
        assert all(la[k] == lb[k] for k in ("PpoTotalLoss", "ActorLoss", "VLoss", "EntLoss"))


def test_micro_batching_matches_one_shot(golden):
    """max_batch smaller than B: gradients are accumulated over micro-batches (ddrl_op_accumulate)."""
    from ddrl4nav_amd.data import Experience
    g = golden("f15_mlp_classical")
    outs = []
    for cap in (256, 64):
        net, _ = _make("f15_mlp_classical", max_batch=cap)
        net.training_iter_time = 1
        exp = Experience(states=_states(g), advs=g["advs"], actions=g["actions"], old_logps=g["old_logps"],
                         values=g["rets"].reshape(1, -1))
        losses = [l for l, _, _ in net.learn(exp)][0]
        outs.append((net.grads[:net.n_params + 6].cpu().numpy().copy(), losses))
        (dist, _), values = net(_states(g))
        outs[-1] += (values[0].cpu().numpy().copy(),)
    a, b = outs
    scale = np.abs(a[0][:-6]).max()
    assert np.abs(a[0][:-6] - b[0][:-6]).max() <= 2e-6 * scale
    np.testing.assert_allclose(a[0][-6:], b[0][-6:], rtol=1e-5, atol=1e-7)
    np.testing.assert_allclose(a[2], b[2], rtol=1e-6, atol=1e-7)


def test_gaussian_sampler_contract(golden):
    """Acting draw of the Gaussian head: mu + std * Box-Muller(counter-based uniforms); the fused
    log-prob equals the summed Normal log-density of that draw."""
    from oracle import ddrl_oracle_nav as N
    g = golden("f13_nav1d_gauss")
    net, w = _make("f13_nav1d_gauss")
    states = _states(g)
    (dist, none), values = net(states)
    assert none is None
    a = dist.sample()
    n, D = a.shape
    seed, stream = net._seed, net._calls * 4096
    u = sample_uniform(seed, stream, 2 * n * D)
    z = N.box_muller(u[0::2], u[1::2]).reshape(n, D)
    mu, std = dist.mean.cpu().numpy(), np.exp(w["actor.log_std"])
    np.testing.assert_allclose(a.cpu().numpy(), mu + std * z, rtol=1e-5, atol=1e-5)
    lp = net.actor.log_prob_from_distribution(dist, a)
    want = dist.log_prob(a).sum(-1)
    np.testing.assert_allclose(lp.cpu().numpy(), want.cpu().numpy(), rtol=1e-5, atol=1e-5)
    # log-prob of OTHER actions re-evaluates the head on the retained features
    other = torch.from_numpy(g["actions"]).cuda()
    lp2 = net.actor.log_prob_from_distribution(dist, other)
    np.testing.assert_allclose(lp2.cpu().numpy(), g["logp"], rtol=2e-5, atol=2e-5)


def test_generic_ppo_dispatch_and_blob_roundtrip(golden):
    from ddrl4nav_amd.nn import GenericPPO, PPO
    net, w = _make("f15_mlp_classical")
    assert isinstance(net, GenericPPO)
    blob = net.model_bytes()
    net2, _ = _make("f15_mlp_classical")
    for p in net2.parameters():
        p.data.zero_()
    net2.load_model_bytes(blob)
    for (k, a), (_, b) in zip(net.named_parameters(), net2.named_parameters()):
        assert torch.equal(a.data, b.data), k
    g = golden("f15_mlp_classical")
    (_, lp), _ = net2(_states(g), torch.from_numpy(g["actions"]))
    np.testing.assert_allclose(lp.cpu().numpy(), g["logp"], rtol=2e-5, atol=2e-6)


def test_nav_config4_batch_properties():
    """BASELINE config 4 scale (512 envs): size-independent properties of the robot_nav net at
    B = 2,048 -- (a) the micro-batched gradient (4 x 512) equals the one-shot gradient, (b) duplicating
    the batch leaves mean losses and the mean gradient unchanged, (c) forward is independent of the
    position of a sample in the batch."""
    from ddrl4nav_amd.data import Experience
    B = 2048
    g = torch.Generator(device="cuda").manual_seed(404)
    half = B // 2
    lz = torch.rand((half, 1, 960), device="cuda", generator=g)
    vec = torch.randn((half, 5), device="cuda", generator=g)
    ped = (torch.rand((half, 3, 48, 48), device="cuda", generator=g) < 0.15).float()
    dup = lambda t: torch.cat([t, t]).contiguous()
    states = [dup(lz), dup(vec), dup(ped)]
    acts = dup(torch.randn((half, 2), device="cuda", generator=g))
    old = dup(torch.full((half,), -2.0, device="cuda") + 0.2 * torch.randn(half, device="cuda", generator=g))
    adv = dup(torch.randn(half, device="cuda", generator=g))
    ret = dup(torch.randn(half, device="cuda", generator=g))
    grads, losses = [], []
    for cap, sl in ((2048, slice(0, B)), (512, slice(0, B)), (1024, slice(0, half))):
        net, _ = _make("f13_nav1d_gauss", max_batch=cap)
        net.training_iter_time = 1
        exp = Experience(states=[s[sl] for s in states], advs=adv[sl], actions=acts[sl], old_logps=old[sl],
                         values=ret[sl].reshape(1, -1))
        losses.append([l for l, _, _ in net.learn(exp)][0])
        grads.append(net.grads[:net.n_params].clone())
        if cap == 2048:
            (d1, _), v1 = net([s[:300] for s in states], acts[:300])
            (d2, _), v2 = net([s[half:half + 300] for s in states], acts[half:half + 300])
            assert torch.equal(v1[0], v2[0]) and torch.equal(d1.mean, d2.mean)       # (c)
    # clip_adam scales .grad in place by the clip coefficient: compare directions and the logged losses
    unit = [x / x.norm() for x in grads]
    assert (unit[0] - unit[1]).abs().max().item() <= 2e-4 * unit[0].abs().max().item()   # (a)
    assert (unit[0] - unit[2]).abs().max().item() <= 2e-4 * unit[0].abs().max().item()   # (b)
    for k in ("ActorLoss", "VLoss", "EntLoss"):
        assert abs(losses[0][k] - losses[1][k]) <= 1e-5 * max(1.0, abs(losses[0][k]))
        assert abs(losses[0][k] - losses[2][k]) <= 1e-5 * max(1.0, abs(losses[0][k]))


@pytest.mark.parametrize("name", ["f13_nav1d_gauss", "f14_navped_shared"])
def test_producer_magnitudes_equal_the_prepass_bit_for_bit(golden, name):
    """Round 5: the per-sample magnitudes behind the fp16 plane scales come from the PRODUCER of a tensor (pooling epilogues of
    fconv.hip / pconv.hip, the data gradients of pconv.hip / plin.hip, the laser branch's Conv1d) instead of a pre-pass over the
    tensor in front of every consumer (include/ddrl.h "per-sample magnitudes").  Both are the exact maximum of the same values, so
    three PPO iterations leave the same parameters and losses bit for bit (nn/generic.py PRODUCER_AMAX = False restores the
    pre-passes: tools/ab_nav_amax.py)."""
    from ddrl4nav_amd.data import Experience
    from ddrl4nav_amd.nn import generic
    g = golden(name)
    B = 256
    gen = torch.Generator(device="cuda").manual_seed(91)
    if name == "f13_nav1d_gauss":
        states = [torch.rand((B, 1, 960), device="cuda", generator=gen), torch.randn((B, 5), device="cuda", generator=gen),
                  (torch.rand((B, 3, 48, 48), device="cuda", generator=gen) < 0.15).float()]
        actions = torch.randn((B, 2), device="cuda", generator=gen)
    else:
        states = [(torch.rand((B, 1, 48, 48), device="cuda", generator=gen) < 0.3).float(), torch.randn((B, 9), device="cuda", generator=gen),
                  (torch.rand((B, 3, 48, 48), device="cuda", generator=gen) < 0.1).float()]
        actions = torch.randint(0, 5, (B,), device="cuda", generator=gen).float()
    states[0][7] *= 1e-4                 # a sample whose magnitudes are decades below the batch's: its scale must follow it in both forms
    adv = torch.randn(B, device="cuda", generator=gen)
    adv[3] = 0.0                         # an all-zero gradient row: the largest scale, in both forms
    exp = Experience(states=states, advs=adv, actions=actions, old_logps=torch.full((B,), -2.0, device="cuda"),
                     values=torch.randn((1, B), device="cuda", generator=gen))
    outs = []
    try:
        for flag in (True, False):
            generic.PRODUCER_AMAX = flag
            net, _ = _make(name, max_batch=160)       # two micro-batches, the second one ragged
            net.training_iter_time = 3
            losses = [l for l, _, _ in net.learn(exp)]
            assert all(np.isfinite(l["PpoTotalLoss"]) for l in losses)
            outs.append((net.params.clone(), losses))
    finally:
        generic.PRODUCER_AMAX = True
    assert torch.equal(outs[0][0], outs[1][0])
    for la, lb in zip(outs[0][1], outs[1][1]):
        assert all(la[k] == lb[k] for k in ("PpoTotalLoss", "ActorLoss", "VLoss", "EntLoss"))


@pytest.mark.parametrize("ped", [0, 3])
def test_stacked_image_batch_takes_the_unfused_first_layer(ped):
    """`image_batch` = 2 gives conv1 an input-channel count no specialised first-layer kernel takes (fconv.hip: 1 or 4
    channels for 3x3, 3 for 7x7), so the first block runs conv + stand-alone pool -- and must still leave the per-sample magnitudes the
    fused second block scales its fp16 planes by (a zero row means scale 2^60 and Inf / NaN planes).  NavPreNet(2) (runner/utils.py:104)
    and NavPedPreNet(5) (:100): forward and one learn() against the CPU oracle, and bit-identical to the pre-pass arrangement."""
    from ddrl4nav_amd.data import Experience
    from ddrl4nav_amd.nn import generic
    from ddrl4nav_amd.runner import create_net
    from oracle import ddrl_oracle_nav as N
    env = {"discrete_action": True, "discrete_actions": list(range(5)), "image_batch": 2, "ped_sim": {"total": ped}}
    B = 96
    rng = np.random.default_rng(60 + ped)
    states = [(rng.random((B, 2, 48, 48)) < 0.3).astype(np.float32), rng.normal(size=(B, 9)).astype(np.float32)]
    if ped:
        states.append((rng.random((B, 3, 48, 48)) < 0.1).astype(np.float32))
    states[0][5] *= 1e-4                 # one sample decades below the others: its scale must follow it
    acts = rng.integers(0, 5, B).astype(np.float32)
    adv, ret = rng.normal(size=B).astype(np.float32), rng.normal(size=B).astype(np.float32)
    old = np.full(B, -1.7, np.float32)
    outs = []
    try:
        for flag in (True, False):
            generic.PRODUCER_AMAX = flag
            net = create_net(_configs(env, shared=True), max_batch=64)     # two micro-batches, the second one ragged
            enc = net.prenet
            assert type(enc).__name__ == ("NavPedPreNet" if ped else "NavPreNet") and enc.image_channel == (5 if ped else 2)
            assert enc.producer_amax is flag and not enc.c1.fused and enc.c2.fused
            w = hash_weights([(k, tuple(p.shape)) for k, p in net.named_parameters()], 61)
            net.load_state_dict({k: torch.from_numpy(v.copy()) for k, v in w.items()})
            (dist, logp), values = net([torch.from_numpy(s) for s in states], torch.from_numpy(acts))
            fwd = (values[0].cpu().numpy()[:, 0].copy(), logp.cpu().numpy().copy())
            exp = Experience(states=[torch.from_numpy(s).cuda() for s in states], advs=torch.from_numpy(adv).cuda(),
                             actions=torch.from_numpy(acts).cuda(), old_logps=torch.from_numpy(old).cuda(),
                             values=torch.from_numpy(ret[None]).cuda())
            net.training_iter_time = 2
            losses = [l for l, _, _ in net.learn(exp)]
            outs.append((net.params.clone(), losses, fwd, w))
    finally:
        generic.PRODUCER_AMAX = True
    assert torch.equal(outs[0][0], outs[1][0]) and bool(torch.isfinite(outs[0][0]).all())
    for la, lb in zip(outs[0][1], outs[1][1]):
        assert all(la[k] == lb[k] for k in ("PpoTotalLoss", "ActorLoss", "VLoss", "EntLoss"))
    ora = N.OracleNet((lambda: N.NavPedPreNet(5)) if ped else (lambda: N.NavPreNet(2)), 5, False, True)
    ora.load_weights(outs[0][3])
    t = [torch.from_numpy(s) for s in states]
    with torch.no_grad():
        _, ologp, _, ov = ora(t, torch.from_numpy(acts))
    np.testing.assert_allclose(outs[0][2][0], ov.numpy()[:, 0], rtol=2e-5, atol=2e-6)
    np.testing.assert_allclose(outs[0][2][1], ologp.numpy(), rtol=2e-5, atol=2e-5)
    ol = [l for l, _, _ in N.learn(ora, ora.make_optims(), t, torch.from_numpy(acts), torch.from_numpy(old), torch.from_numpy(adv),
                                   torch.from_numpy(ret), iters=2)]
    for got, want in zip(outs[0][1], ol):
        for k in ("ActorLoss", "VLoss", "EntLoss", "PpoTotalLoss"):
            np.testing.assert_allclose(got[k], want[k], rtol=2e-4, atol=2e-5, err_msg=k)


def test_nav_two_encoder_streams_are_bit_identical():
    """The robot_nav learner with the critic's encoder on its own stream equals the one-stream run (`net.encoder_streams = False`)
    bit for bit: independent buffers and gradient slices, no atomics between them."""
    from ddrl4nav_amd.data import Experience
    B = 384
    g = torch.Generator(device="cuda").manual_seed(77)
    states = [torch.rand((B, 1, 960), device="cuda", generator=g), torch.randn((B, 5), device="cuda", generator=g),
              (torch.rand((B, 3, 48, 48), device="cuda", generator=g) < 0.15).float()]
    exp = Experience(states=states, advs=torch.randn(B, device="cuda", generator=g), actions=torch.randn((B, 2), device="cuda", generator=g),
                     old_logps=torch.full((B,), -2.0, device="cuda"), values=torch.randn((1, B), device="cuda", generator=g))
    outs = []
    for two in (True, False):
        net, _ = _make("f13_nav1d_gauss", max_batch=256)       # two micro-batches: the streams join before the accumulation
        net.encoder_streams = two
        net.training_iter_time = 3
        losses = [l for l, _, _ in net.learn(exp)]
        assert (net._side() is not None) == two
        outs.append((net.params.clone(), losses))
    assert torch.equal(outs[0][0], outs[1][0])
    for la, lb in zip(outs[0][1], outs[1][1]):
        assert all(la[k] == lb[k] for k in ("PpoTotalLoss", "ActorLoss", "VLoss", "EntLoss"))


def test_nav_config4_at_size_512_envs_x_256_steps():
    """BASELINE config 4 AT ITS SIZE (robot_nav, 512 envs, TIME_MAX 256: 131,072 samples, 4.1 GB of fp32 observations in the device
    pool): StateRollout acting over all 256 steps + bootstrap + GAE, then net.learn on the whole rollout in micro-batches.
    (1) acting at n = 512 against the nav ORACLE on one 512-env step (values, mu, log-prob of the drawn actions);
    (2) the stored values / log-probs equal a fresh evaluation of the stored actions (a late and an early step);
    (3) GAE over [256, 512] equals the oracle scan bit for bit;
    (4) learner on the full batch: micro-batched in 32 chunks of 4,096 == in 16 chunks of 8,192 (loss terms and gradient direction);
    (5) duplication: the first 65,536 samples twice == once; (6) position independence of the forward."""
    from ddrl4nav_amd.agent import StateRollout
    from ddrl4nav_amd.data import Experience
    from oracle import ddrl_oracle as O
    from oracle import ddrl_oracle_nav as N
    NE, T = 512, 256
    B = NE * T
    net, w = _make("f13_nav1d_gauss", max_batch=4096)
    shapes = [(1, 960), (5,), (3, 48, 48)]
    gen = torch.Generator(device="cuda").manual_seed(4004)
    ro = StateRollout(net, NE, shapes, horizon=T)
    # observations of all 257 steps straight into the pool (the synthetic env costs nothing, as in bench.py)
    for t0 in range(0, T + 1, 32):
        t1 = min(T + 1, t0 + 32)
        ro.states[0][t0:t1] = torch.rand((t1 - t0, NE) + shapes[0], device="cuda", generator=gen)
        ro.states[1][t0:t1] = torch.randn((t1 - t0, NE) + shapes[1], device="cuda", generator=gen)
        ro.states[2][t0:t1] = (torch.rand((t1 - t0, NE) + shapes[2], device="cuda", generator=gen) < 0.15).float()
    rew = (torch.rand((T, NE), device="cuda", generator=gen) < 0.02).float() - (torch.rand((T, NE), device="cuda", generator=gen) < 0.02).float()
    done = (torch.rand((T, NE), device="cuda", generator=gen) < 1 / 200).to(torch.uint8)
    for t in range(T):
        a = ro.act(t)
        ro.record(t, rew[t], done[t])
    assert a.shape == (NE, 2)
    ro.bootstrap()
    ro.finish()
    assert bool(torch.isfinite(ro.values).all()) and bool(torch.isfinite(ro.logps).all())
    # (1) one 512-env acting step against the oracle
    t_chk = 131
    onet = N.OracleNet(lambda: N.NavPreNet1D(3), 2, True, False)
    onet.load_weights(w)
    st = [p[t_chk].cpu() for p in ro.states]
    with torch.no_grad():
        mu, logp, _, ov = onet(st, ro.actions[t_chk].cpu())
    np.testing.assert_allclose(ro.values[t_chk].cpu().numpy(), ov.numpy()[:, 0], rtol=2e-5, atol=2e-6)
    np.testing.assert_allclose(ro.logps[t_chk].cpu().numpy(), logp.numpy(), rtol=2e-5, atol=2e-5)
    # (2) stored values / log-probs == a fresh evaluation of the stored actions
    for t in (3, 250):
        (_, lp), v = net([p[t] for p in ro.states], ro.actions[t])
        assert torch.allclose(lp, ro.logps[t], rtol=1e-5, atol=1e-5) and torch.equal(v[0][:, 0], ro.values[t])
    # (3) GAE
    adv, ret = O.gae(ro.values.cpu().numpy(), rew.cpu().numpy(), done.cpu().numpy())
    assert np.array_equal(ro.adv.cpu().numpy(), adv) and np.array_equal(ro.ret.cpu().numpy(), ret)
    exp = ro.batch()
    assert len(exp) == B and exp.states[2].shape == (B, 3, 48, 48) and exp.actions.shape == (B, 2)
    # (6) position independence of the forward: the same 300 observations at two places of a 4,096-sample forward
    big = [torch.cat([s[:300], s[5000:5000 + 3496], s[:300]]) for s in exp.states]
    (d_big, _), v_big = net(big, None, True)
    assert torch.equal(v_big[0][:300], v_big[0][-300:]) and torch.equal(d_big[:300], d_big[-300:])
    # (4) micro-batching and (5) duplication, one iteration each from the same weights
    runs = {}
    half = B // 2
    dup = lambda x: torch.cat([x[:half], x[:half]])
    cases = (("32x4096", 4096, exp),
             ("16x8192", 8192, exp),
             ("half", 4096, Experience(states=[s[:half] for s in exp.states], advs=exp.advs[:half], actions=exp.actions[:half],
                                       old_logps=exp.old_logps[:half], values=exp.values[:, :half])),
             ("half_twice", 4096, Experience(states=[dup(s) for s in exp.states], advs=dup(exp.advs), actions=dup(exp.actions),
                                             old_logps=dup(exp.old_logps), values=torch.cat([exp.values[:, :half]] * 2, dim=1))))
    for tag, cap, e in cases:
        n2, _ = (net, None) if cap == 4096 and tag == "32x4096" else _make("f13_nav1d_gauss", max_batch=cap)
        n2.training_iter_time = 1
        losses = [l for l, _, _ in n2.learn(e)][0]
        gflat = n2.grads[:n2.n_params].clone()
        runs[tag] = (losses, gflat / gflat.norm())     # clip_adam scales .grad in place by the clip coefficient: compare directions
        assert all(np.isfinite(losses[k]) for k in ("ActorLoss", "VLoss", "EntLoss", "PpoTotalLoss"))
    for a, b in (("32x4096", "16x8192"), ("half", "half_twice")):
        la, ga = runs[a]
        lb, gb = runs[b]
        assert (ga - gb).abs().max().item() <= 2e-4 * ga.abs().max().item(), (a, b)
        for k in ("ActorLoss", "VLoss", "EntLoss"):
            assert abs(la[k] - lb[k]) <= 1e-5 * max(1.0, abs(la[k])), (a, b, k)


def test_state_rollout_loop_matches_manual_sequence(golden):
    """StateRollout (device-resident pool for list-of-tensor observations): acting, bootstrap, GAE and the
    learner batch equal the same steps done by hand; one learn() on its batch runs."""
    from ddrl4nav_amd.agent import StateRollout
    from oracle import ddrl_oracle as O
    net, _ = _make("f13_nav1d_gauss", max_batch=64)
    N, T = 6, 5
    g = torch.Generator(device="cuda").manual_seed(5)
    ro = StateRollout(net, N, [(1, 960), (5,), (3, 48, 48)], horizon=T)
    for t in range(T + 1):
        ro.put_states(t, [torch.rand((N, 1, 960), device="cuda", generator=g), torch.randn((N, 5), device="cuda", generator=g),
                          (torch.rand((N, 3, 48, 48), device="cuda", generator=g) < 0.2).float()])
    rew = torch.randn((T, N), device="cuda", generator=g)
    done = (torch.rand((T, N), device="cuda", generator=g) < 0.2).to(torch.uint8)
    for t in range(T):
        a = ro.act(t)
        assert a.shape == (N, 2)
        ro.record(t, rew[t], done[t])
    ro.bootstrap()
    ro.finish()
    # values / logps equal a fresh evaluation of the stored actions; GAE equals the oracle scan
    for t in range(T):
        (_, lp), v = net([p[t] for p in ro.states], ro.actions[t])
        assert torch.allclose(lp, ro.logps[t], rtol=1e-5, atol=1e-5) and torch.equal(v[0][:, 0], ro.values[t])
    adv, ret = O.gae(ro.values.cpu().numpy(), rew.cpu().numpy(), done.cpu().numpy())
    assert np.array_equal(ro.adv.cpu().numpy(), adv) and np.array_equal(ro.ret.cpu().numpy(), ret)
    exp = ro.batch()
    assert exp.states[2].shape == (N * T, 3, 48, 48) and exp.actions.shape == (N * T, 2) and exp.values.shape == (1, N * T)
    losses = [l for l, _, _ in net.learn(exp)]
    assert len(losses) == 10 and all(np.isfinite(l["PpoTotalLoss"]) for l in losses)
    ro.carry_over()
    assert torch.equal(ro.states[0][0], ro.states[0][T])
