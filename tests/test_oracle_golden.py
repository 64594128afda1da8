"""Pin the CPU oracle (oracle/ddrl_oracle.py) to golden vectors produced by importing the
reference itself (tests/golden/make_golden.py).  CPU only."""
import numpy as np
import pytest
import torch

from ddrl4nav_amd.utils.recipe import make_weights, param_specs, flatten
from oracle import ddrl_oracle as O


@pytest.fixture(scope="module")
def net():
    torch.set_num_threads(1)
    n = O.OraclePPO()
    n.load_weights(make_weights(0))
    return n


def test_param_order_and_size(net):
    names = [k for k, _ in net.named_parameters()]
    assert names == [n for n, _, _ in param_specs()]
    assert flatten(make_weights(0)).size == 3371847


def test_f5_lut(golden):
    assert np.array_equal(O.u8_lut(), golden("f5_u8_lut")["lut"])


def test_f1_forward(net, golden):
    g = golden("f1_forward")
    x = O.frames_to_f32(g["frames"])
    with torch.no_grad():
        probs, p_hat, logits, v = net(x)
        logp = O.categorical_log_prob(logits, torch.from_numpy(g["actions"]))
        ent = O.categorical_entropy(p_hat, logits)
        h_a = net.actor.pre(x)
    # same torch ops, same thread count -> bit-exact
    assert np.array_equal(probs.numpy(), g["probs"])
    assert np.array_equal(p_hat.numpy(), g["p_hat"])
    assert np.array_equal(logits.numpy(), g["logits"])
    assert np.array_equal(logp.numpy(), g["logp"])
    assert np.array_equal(v.numpy()[:, 0], g["value"])
    assert np.array_equal(ent.numpy(), g["entropy"])
    assert np.array_equal(h_a.numpy()[:, :16], g["h_actor"])


def test_f2_gae_bit_exact(golden):
    g = golden("f2_gae")
    T = 256
    adv1, ret1 = O.gae(g["values"][:T + 1], g["rewards"][:T + 1], g["dones"][:T + 1])
    assert np.array_equal(adv1, g["adv1"]) and np.array_equal(ret1, g["ret1"])
    # carry-over of the last stored step into the next rollout (agent.py:289-291)
    adv2, ret2 = O.gae(g["values"][T:], g["rewards"][T:], g["dones"][T:])
    assert np.array_equal(adv2, g["adv2"]) and np.array_equal(ret2, g["ret2"])


def _batch(golden):
    g3 = golden("f3_loss")
    x = O.frames_to_f32(g3["frames"])
    t = lambda k: torch.from_numpy(g3[k])
    return g3, x, t("actions"), t("old_logps"), t("advs"), t("rets")


def test_f3_losses_and_grads(net, golden):
    g, x, a, ol, adv, ret = _batch(golden)
    net.load_weights(make_weights(0))
    net.zero_grad()
    total, al, vl, ent = O.ppo_losses(net, x, a, ol, adv, ret)
    assert np.float32(al.item()) == g["actor_loss"]
    assert np.float32(vl.item()) == g["v_loss"]
    assert np.float32(ent.item()) == g["ent"]
    assert np.float32(total.item()) == g["total"]
    al.backward()
    vl.backward()
    sd = dict(net.named_parameters())
    assert np.array_equal(sd["actor.actor_linear.weight"].grad.numpy(), g["grad_actor_linear_w"])
    assert np.array_equal(sd["critic.critic_linear.bias"].grad.numpy(), g["grad_critic_linear_b"])
    for k, p in sd.items():
        np.testing.assert_allclose(p.grad.numpy().reshape(-1)[:64], g["ghead/" + k], rtol=0, atol=0)


def test_f4_learn_sequence(golden):
    g, x, a, ol, adv, ret = _batch(golden)
    g4 = golden("f4_learn")
    torch.set_num_threads(1)
    net = O.OraclePPO()
    net.load_weights(make_weights(0))
    opt = net.make_optims()
    for it, (ld, ut, last) in enumerate(O.learn(net, opt, x, a, ol, adv, ret), 1):
        row = g4["losses"][it - 1]
        assert ut == it and last
        got = [ld["PpoTotalLoss"], ld["ActorLoss"], ld["VLoss"], ld["EntLoss"]]
        np.testing.assert_allclose(got, row, rtol=1e-6, atol=1e-7)
        if it in (1, 10):
            for k, p in net.named_parameters():
                np.testing.assert_allclose(p.detach().numpy().reshape(-1)[:8], g4["it%d/head/%s" % (it, k)],
                                           rtol=1e-6, atol=2e-8)
                arr = p.detach().numpy().reshape(-1)
                np.testing.assert_allclose(arr[::max(1, arr.size // 257)][:257], g4["it%d/stride/%s" % (it, k)],
                                           rtol=1e-6, atol=2e-8)
    assert it == 10


def test_f6_episode_returns(golden):
    g = golden("f6_returns")
    trace, rsum = O.episode_returns(g["rewards"], g["dones"])
    assert np.array_equal(trace, g["trace"]) and np.array_equal(rsum, g["final_sum"])


def test_inverse_cdf_sampler_contract():
    p = np.array([[0.1, 0.2, 0.3, 0.4], [0.25, 0.25, 0.25, 0.25]], np.float32)
    assert list(O.inverse_cdf_sample(p, np.array([0.05, 0.99], np.float32))) == [0, 3]
    assert list(O.inverse_cdf_sample(p, np.array([0.1, 0.5], np.float32))) == [1, 2]


# ---- non-default learner modes (tests/golden/make_golden_shared.py) ------------------------------
def _check_learn(gen, g):
    it = 0
    for it, (ld, ut, last) in enumerate(gen, 1):
        got = [ld["PpoTotalLoss"], ld["ActorLoss"], ld["VLoss"], ld["EntLoss"]]
        np.testing.assert_allclose(got, g["losses"][it - 1], rtol=1e-6, atol=1e-7)
    assert it == 10


def test_f10_shared_prenet(golden):
    g3, g = golden("f3_loss"), golden("f10_shared")
    torch.set_num_threads(1)
    w = make_weights(0, shared=True)
    assert flatten(w).size == 1684128 + 6 * 512 + 6 + 513
    net = O.OracleSharedPPO()
    assert [k for k, _ in net.named_parameters()] == [n for n, _, _ in param_specs(shared=True)]
    net.load_weights(w)
    x = O.frames_to_f32(g3["frames"])
    t = lambda k: torch.from_numpy(g[k])
    with torch.no_grad():
        probs, _, logits, v = net(x)
        assert np.array_equal(probs.numpy(), g["probs"]) and np.array_equal(v.numpy()[:, 0], g["value"])
        assert np.array_equal(O.categorical_log_prob(logits, t("actions")).numpy(), g["logp"])
    total, al, vl, ent = O.ppo_losses(net, x, t("actions"), t("old_logps"), t("advs"), t("rets"))
    np.testing.assert_allclose([total.item(), al.item(), vl.item(), ent.item()], g["loss4"], rtol=1e-7, atol=0)
    total.backward()
    for k, p in net.named_parameters():
        np.testing.assert_allclose(p.grad.numpy().reshape(-1)[:64], g["ghead/" + k], rtol=1e-6, atol=1e-9)
    net.zero_grad()
    _check_learn(O.learn(net, net.make_optims(), x, t("actions"), t("old_logps"), t("advs"), t("rets")), g)
    for k, p in net.named_parameters():
        arr = p.detach().numpy().reshape(-1)
        np.testing.assert_allclose(arr[::max(1, arr.size // 257)][:257], g["it10/stride/" + k], rtol=1e-6, atol=2e-8)


def test_f11_smooth_l1(golden):
    g3, g = golden("f3_loss"), golden("f11_smooth_l1")
    torch.set_num_threads(1)
    net = O.OraclePPO()
    net.load_weights(make_weights(0))
    x = O.frames_to_f32(g3["frames"])
    t3 = lambda k: torch.from_numpy(g3[k])
    rets = torch.from_numpy(g["rets"])
    total, al, vl, ent = O.ppo_losses(net, x, t3("actions"), t3("old_logps"), t3("advs"), rets, smooth_l1=True)
    np.testing.assert_allclose([total.item(), al.item(), vl.item(), ent.item()], g["loss4"], rtol=1e-7, atol=0)
    _check_learn(O.learn(net, net.make_optims(), x, t3("actions"), t3("old_logps"), t3("advs"), rets, smooth_l1=True), g)


def test_f12_eighteen_actions(golden):
    g3, g = golden("f3_loss"), golden("f12_actions18")
    torch.set_num_threads(1)
    net = O.OraclePPO(n_actions=18)
    net.load_weights(make_weights(0, n_actions=18))
    x = O.frames_to_f32(g3["frames"])
    t = lambda k: torch.from_numpy(g[k])
    with torch.no_grad():
        probs, p_hat, logits, v = net(x)
    assert np.array_equal(probs.numpy(), g["probs"]) and np.array_equal(v.numpy()[:, 0], g["value"])
    assert np.array_equal(O.categorical_log_prob(logits, t("actions")).numpy(), g["logp"])
    assert np.array_equal(O.categorical_entropy(p_hat, logits).numpy(), g["entropy"])
    total, al, vl, ent = O.ppo_losses(net, x, t("actions"), t("old_logps"), t("advs"), t("rets"))
    np.testing.assert_allclose([total.item(), al.item(), vl.item(), ent.item()], g["loss4"], rtol=1e-7, atol=0)
    al.backward()
    vl.backward()
    assert np.array_equal(net.actor.actor_linear.weight.grad.numpy(), g["grad_actor_linear_w"])
    net.zero_grad()
    _check_learn(O.learn(net, net.make_optims(), x, t("actions"), t("old_logps"), t("advs"), t("rets")), g)


# ---- non-Atari nets (tests/golden/make_golden_nav.py) ---------------------------------------------
def _nav_case(name):
    from oracle import ddrl_oracle_nav as N
    return {"f13_nav1d_gauss": (lambda: N.NavPreNet1D(3), 2, True, False, 13),
            "f14_navped_shared": (lambda: N.NavPedPreNet(4), 5, False, True, 14),
            "f15_mlp_classical": (lambda: N.MLPPreNet(4, 512), 2, False, False, 15),
            "f25_navpre_shared": (lambda: N.NavPreNet(1), 5, False, True, 25),
            "f26_navpre_unaligned": (lambda: N.NavPreNet(1), 5, False, True, 25),
            "f27_nav1d_unaligned": (lambda: N.NavPreNet1D(3), 2, True, False, 27)}[name]


@pytest.mark.parametrize("name", ["f13_nav1d_gauss", "f14_navped_shared", "f15_mlp_classical", "f25_navpre_shared"])
def test_nav_oracle_pinned_to_reference(golden, name):
    from ddrl4nav_amd.utils.recipe import hash_weights
    from oracle import ddrl_oracle_nav as N
    g = golden(name)
    make_pre, n_out, gaussian, shared, seed = _nav_case(name)
    torch.set_num_threads(1)
    net = N.OracleNet(make_pre, n_out, gaussian, shared)
    names = [k for k, _ in net.named_parameters()]
    assert names == list(g["names"])  # same module tree / parameter order as the reference
    net.load_weights(hash_weights([(k, tuple(p.shape)) for k, p in net.named_parameters()], seed))
    states = [torch.from_numpy(g["state%d" % i]) for i in range(len([k for k in g.files if k.startswith("state")]))]
    t = lambda k: torch.from_numpy(g[k])
    with torch.no_grad():
        dist_out, logp, ent_el, v = net(states, t("actions"))
    assert np.array_equal(dist_out.numpy(), g["dist_out"]) and np.array_equal(v.numpy()[:, 0], g["value"])
    assert np.array_equal(logp.numpy(), g["logp"]) and np.array_equal(ent_el.numpy(), g["entropy"])
    total, al, vl, ent = N.losses(net, states, t("actions"), t("old_logps"), t("advs"), t("rets"))
    np.testing.assert_allclose([total.item(), al.item(), vl.item(), ent.item()], g["loss4"], rtol=1e-7, atol=0)
    if shared:
        total.backward()
    else:
        al.backward()
        vl.backward()
    for k, p in net.named_parameters():
        flat = p.grad.numpy().reshape(-1)
        np.testing.assert_allclose(flat[::max(1, flat.size // 129)][:129], g["gstride/" + k], rtol=1e-6, atol=1e-10)
    net.zero_grad()
    it = 0
    for it, (ld, ut, last) in enumerate(N.learn(net, net.make_optims(), states, t("actions"), t("old_logps"), t("advs"),
                                                t("rets")), 1):
        got = [ld["PpoTotalLoss"], ld["ActorLoss"], ld["VLoss"], ld["EntLoss"]]
        np.testing.assert_allclose(got, g["losses"][it - 1], rtol=2e-6, atol=2e-7)
    assert it == 10


@pytest.mark.parametrize("mode", ["default", "shared", "smooth"])
def test_oracle_float64_trajectory_pinned_to_reference_float64(mode):
    """The GPU sequence tests measure every parameter against the oracle's FLOAT64 trajectory
    (tests/parity_util.py); that trajectory is pinned here against the reference's own float64 run
    (tests/golden/make_golden_spread.py): losses and per-tensor checksums after iterations 1 and 10."""
    import parity_util as P
    sp = P._load(P.MODES[mode][0])
    tr = P.f64_trajectory(mode)
    np.testing.assert_allclose(tr["losses"], sp["losses_f64"], rtol=1e-11, atol=1e-13)
    for it in (1, 10):
        for name, a in tr["params"][it].items():
            k = "it%d/%s" % (it, name)
            np.testing.assert_allclose(np.sqrt((a ** 2).sum()), sp["f64_l2/" + k], rtol=1e-12)
            np.testing.assert_allclose(a.ravel()[:8], sp["f64_head/" + k], rtol=0, atol=1e-13)
            np.testing.assert_allclose(a.sum(), sp["f64_sum/" + k], rtol=0, atol=1e-11 * np.sqrt(a.size))
            # the stored reference spread is a positive number every bound divides by
            assert sp["ref_l2/" + k] > 0 and sp["ref_max/" + k] > 0


def test_wide_spread_fixture_is_the_reference_under_other_batch_orders(golden):
    """tests/golden/f4d_spread_wide.npz (make_golden_spread_wide.py): 2 thread counts x 16 batch orders of the reference's own
    fp32 `learn` on F4.  Variant 0 is the stored F4 run itself; the oracle run in the batch order of variant 1 (one thread)
    reproduces that variant's losses, i.e. the variants are the same mathematics summed in another order -- and they drift
    from the float64 run by up to several 1e-3 in VLoss from the sixth iteration on (the envelope the GPU tests use)."""
    import parity_util as P
    w = golden("f4d_spread_wide")
    g4 = golden("f4_learn")
    lv = w["losses_variants"]
    assert lv.shape == (32, 10, 4) and np.array_equal(lv[0], g4["losses"])
    np.testing.assert_allclose(w["losses_f64"], P._load("f4b_spread_default")["losses_f64"], rtol=0, atol=0)
    frames, actions, old_logps, advs, rets = P.mode_batch("default")
    perm = np.random.default_rng(1001).permutation(64)
    net = O.OraclePPO()
    net.load_weights(make_weights(0))
    t = lambda a: torch.from_numpy(np.ascontiguousarray(np.asarray(a)[perm]))
    threads = torch.get_num_threads()
    torch.set_num_threads(1)
    try:
        got = [[ld[k] for k in ("PpoTotalLoss", "ActorLoss", "VLoss", "EntLoss")]
               for ld, _, _ in O.learn(net, net.make_optims(), O.frames_to_f32(frames[perm]), t(actions), t(old_logps), t(advs), t(rets))]
    finally:
        torch.set_num_threads(threads)
    np.testing.assert_allclose(np.asarray(got), lv[1], rtol=2e-6, atol=2e-7)
    dev = np.abs(lv[:, :, 2] - w["losses_f64"][None, :, 2])
    assert dev[:, 3].max() > 5e-5 and dev[:, 5].max() > 3e-3 and np.median(dev[:, 5]) < 2e-4   # discrete jumps, not a smooth drift
    for k in w.files:
        if k.startswith("upd_l2/"):
            np.testing.assert_allclose(w[k], P._load("f4b_spread_default")[k], rtol=1e-12)


@pytest.mark.parametrize("C", [1, 3])
def test_oracle_other_frame_stacks_golden_f20(golden, C):
    """The oracle with `num_inputs` = 1 / 3 against the reference's own AtariPreNet(num_inputs) net (F20,
    tests/golden/make_golden_channels.py): forward bit-exact, three learn iterations to 1e-6."""
    g = golden("f20_channels")
    p = "c%d/" % C
    net = O.OraclePPO(num_inputs=C)
    net.load_weights(make_weights(seed=20 + C, num_inputs=C))
    x = O.frames_to_f32(g[p + "frames"])
    t = lambda k: torch.from_numpy(g[p + k])
    threads = torch.get_num_threads()
    torch.set_num_threads(1)
    try:
        with torch.no_grad():
            probs, p_hat, logits, v = net(x)
        assert np.array_equal(probs.numpy(), g[p + "probs"]) and np.array_equal(v.numpy()[:, 0], g[p + "value"])
        assert np.array_equal(O.categorical_log_prob(logits, t("actions")).numpy(), g[p + "logp"])
        got = [[ld[k] for k in ("PpoTotalLoss", "ActorLoss", "VLoss", "EntLoss")]
               for ld, _, _ in O.learn(net, net.make_optims(), x, t("actions"), t("old_logps"), t("advs"), t("rets"), iters=3)]
    finally:
        torch.set_num_threads(threads)
    np.testing.assert_allclose(np.asarray(got), g[p + "losses"], rtol=2e-6, atol=2e-7)
    for name, q in net.named_parameters():
        a = q.detach().double().numpy()
        np.testing.assert_allclose(np.sqrt((a ** 2).sum()), g[p + "it3/l2/" + name], rtol=1e-6)


def _gail_case(name, golden):
    from oracle import ddrl_oracle_gail as G
    from oracle import ddrl_oracle_nav as N
    g = golden(name)
    hidden = int(g["d_mlp_hidden"])
    if name == "f22_gail_navped":
        net = G.OracleGAIL(lambda: N.NavPedPreNet(4), 5, False, [(513, hidden, "relu"), (hidden, 1, None)])
        states_np, seed = [g["state0"], g["state1"], g["state2"]], 22
    elif name == "f16_gail_classical":
        net = G.OracleGAIL(lambda: N.MLPPreNet(4, 512), 2, False, [(513, hidden, "relu"), (hidden, 1, None)])
        states_np, seed = g["states"], 16
    else:
        net = G.OracleGAIL(lambda: G.AtariPre(4), 6, False, [(513, hidden, "relu"), (hidden, 1, None)])
        states_np, seed = O.u8_lut()[golden("f3_loss")["frames"]], 17
    return g, net, states_np, seed


@pytest.mark.parametrize("name", ["f16_gail_classical", "f17_gail_atari", "f22_gail_navped"])
def test_gail_oracle_pinned_to_reference(golden, name):
    """Discriminator forward / WGAN step / StepLR and the PPO update with the extra GAIL critic, against the
    reference's own GAIL.learn (tests/golden/make_golden_gail.py; F22, a shared NavPedPreNet: make_golden_gail_nav.py)."""
    import parity_util as P
    from ddrl4nav_amd.utils.recipe import hash_weights
    from oracle import ddrl_oracle_gail as G
    g, net, states_np, seed = _gail_case(name, golden)
    torch.set_num_threads(1)
    assert [k for k, _ in net.named_parameters()] == list(g["names"])
    w = hash_weights([(k, tuple(p.shape)) for k, p in net.named_parameters()], seed)
    net.load_weights(w)
    t = lambda k: torch.from_numpy(g[k])
    st_np, ex_np = P.gail_state_lists(g, states_np)
    states, ex_states = [torch.from_numpy(a) for a in st_np], [torch.from_numpy(a) for a in ex_np]
    with torch.no_grad():
        probs, logp, _, values = net(states, t("actions"))
        dr = net.d_reward(states, t("actions"))
    assert np.array_equal(probs.numpy(), g["probs"]) and np.array_equal(logp.numpy(), g["logp"])
    assert np.array_equal(values[0].numpy()[:, 0], g["value0"]) and np.array_equal(values[1].numpy()[:, 0], g["value1"])
    assert np.array_equal(dr.numpy()[:, 0], g["d_reward"])
    optims = net.make_optims()
    rows = list(G.learn(net, optims, states, t("actions"), t("old_logps"), t("advs"), t("rets"), ex_states, t("expert_actions")))
    assert [last for _, _, last in rows] == [False] + [True] * 10 and [ut for _, ut, _ in rows] == [1] + list(range(1, 11))
    np.testing.assert_allclose(rows[0][0]["Gail[D]Loss"], g["d_loss"][0], rtol=1e-6, atol=1e-9)
    got = np.array([[r[0][k] for k in ("PpoTotalLoss", "ActorLoss", "VLoss", "EntLoss")] for r in rows[1:]])
    np.testing.assert_allclose(got, g["losses"], rtol=2e-6, atol=2e-7)
    assert optims[1].param_groups[0]["lr"] == float(g["d_lr_after"])
    for k, p in net.named_parameters():
        a = p.detach().numpy().reshape(-1)
        np.testing.assert_allclose(a[::max(1, a.size // 129)][:129], g["it10/stride/" + k], rtol=1e-5, atol=1e-7, err_msg=k)
        if k.startswith("gail_critic."):
            assert np.array_equal(p.detach().numpy(), w[k])   # in no optimiser: never trained (ppo.py:39,61-62)


def test_gail_discriminator_steplr_boundary(golden):
    """260 discriminator-only steps: the loss trajectory and the learning rate across the StepLR(250, 0.95) boundary."""
    from ddrl4nav_amd.utils.recipe import hash_weights
    from oracle import ddrl_oracle_gail as G
    g, net, states_np, seed = _gail_case("f16_gail_classical", golden)
    torch.set_num_threads(1)
    net.load_weights(hash_weights([(k, tuple(p.shape)) for k, p in net.named_parameters()], seed))
    t = lambda k: torch.from_numpy(g[k])
    states = [torch.from_numpy(states_np)]
    ex_states = [torch.from_numpy(states_np[g["expert_index"]][::-1].copy())]
    _, d_optim, d_sched = net.make_optims()
    rows, lrs = [], []
    for _ in range(260):
        item, _, _ = G.d_step(net, d_optim, d_sched, states, t("actions"), ex_states, t("expert_actions"))
        rows.append(item["Gail[D]Loss"])
        lrs.append(d_optim.param_groups[0]["lr"])
    assert lrs == list(g["d_only_lr"]) and lrs[248] == 5e-5 and abs(lrs[249] - 4.75e-5) < 1e-18
    np.testing.assert_allclose(rows, g["d_only_loss"], rtol=1e-5, atol=1e-8)
    for k, p in net.discriminator.named_parameters():
        a = p.detach().numpy().reshape(-1)
        np.testing.assert_allclose(a[::max(1, a.size // 129)][:129], g["Dend/stride/discriminator." + k], rtol=1e-4, atol=1e-7)


def test_gae_two_value_rows_bit_exact(golden):
    from oracle import ddrl_oracle_gail as G
    g = golden("f18_gae_two_rows")
    adv, ret = G.gae_rows(g["values"], g["rewards"], g["dones"], g["discounts"], float(g["landa"]))
    assert np.array_equal(adv, g["adv"]) and np.array_equal(ret, g["ret"])


@pytest.mark.parametrize("name", ["f16_gail_classical", "f17_gail_atari", "f22_gail_navped"])
def test_gail_oracle_float64_trajectory_pinned(name):
    """The float64 trajectory the GPU GAIL tests measure against == the reference's own float64 GAIL.learn."""
    import parity_util as P
    g = P._load(name)
    tr = P.gail_f64_trajectory(name)
    np.testing.assert_allclose(tr["losses"], g["losses_f64"], rtol=1e-11, atol=1e-13)
    np.testing.assert_allclose(tr["d_loss"], g["d_loss_f64"], rtol=1e-10, atol=1e-14)
    for tag in ("D1", "it1", "it10"):
        for k, a in tr["params"][tag].items():
            np.testing.assert_allclose(np.sqrt((a ** 2).sum()), g["f64_l2/%s/%s" % (tag, k)], rtol=1e-12)
            np.testing.assert_allclose(a.ravel()[:8], g["f64_head/%s/%s" % (tag, k)], rtol=0, atol=1e-13)


@pytest.mark.parametrize("name", ["f13_nav1d_gauss", "f14_navped_shared", "f15_mlp_classical", "f25_navpre_shared"])
def test_nav_oracle_float64_trajectory_pinned(name):
    import parity_util as P
    sp = P._load(name[:3] + "b_spread")
    tr = P.nav_f64_trajectory(name)
    np.testing.assert_allclose(tr["losses"], sp["losses_f64"], rtol=1e-10, atol=1e-12)
    for it in (1, 10):
        for k, a in tr["params"][it].items():
            np.testing.assert_allclose(np.sqrt((a ** 2).sum()), sp["f64_l2/it%d/%s" % (it, k)], rtol=1e-11)
            np.testing.assert_allclose(a.ravel()[:8], sp["f64_head/it%d/%s" % (it, k)], rtol=0, atol=1e-12)


@pytest.mark.parametrize("name", ["f26_navpre_unaligned", "f27_nav1d_unaligned"])
def test_no_tie_batch_oracle_decisions_and_gradient_are_the_references(golden, name):
    """F26 / F27 (tests/golden/make_golden_navpre.py): the shared NavPreNet(1) net and config 4's NavPreNet1D x2 + Gaussian net, each
    on a batch selected so that no ReLU / max-pool decision lies within ~1e-5 / ~5e-6 (of the site's largest pre-activation) of a tie.
    The oracle, run here in fp32 AND float64 with its own decisions, reproduces the reference's decision digests, margins, losses and
    its whole stored gradient -- which makes the oracle's float64 gradient the yardstick of the un-aligned GPU test
    (test_generic_gpu.py::test_nav_gradient_unaligned)."""
    import parity_util as P
    from ddrl4nav_amd.utils.recipe import hash_weights
    from oracle import ddrl_oracle_nav as N
    g = golden(name)
    make_pre, n_out, gaussian, shared, seed = _nav_case(name)
    n_states = len([k for k in g.files if k.startswith("state")])
    torch.set_num_threads(1)
    for dtype in (torch.float32, torch.float64):
        net = N.OracleNet(make_pre, n_out, gaussian, shared)
        assert [k for k, _ in net.named_parameters()] == list(g["names"])
        net.load_weights(hash_weights([(k, tuple(p.shape)) for k, p in net.named_parameters()], seed))
        net.to(dtype)
        states = [torch.from_numpy(g["state%d" % i]).to(dtype) for i in range(n_states)]
        t = lambda k: torch.from_numpy(g[k]).to(dtype)
        encs = [("", net.prenet)] if shared else [("actor.pre/", net.actor.pre), ("critic.pre/", net.critic.pre)]
        for _, e in encs:
            e.record = {}
        total, al, vl, ent = N.losses(net, states, t("actions"), t("old_logps"), t("advs"), t("rets"))
        margins = []
        for pre, e in encs:
            dig, margin = P.decision_digest(e.record)
            e.record = None
            margins.append(margin)
            for k, v in dig.items():
                assert np.array_equal(v, g["digest/" + pre + k]), (str(dtype), pre + k)
        margin = np.concatenate(margins, 1)
        if dtype == torch.float64:
            np.testing.assert_allclose(margin, g["margin_f64"], rtol=1e-9)
            assert margin.min() > (8e-6 if shared else 3e-6)      # the property the fixture was selected for
        tol = 1e-6 if dtype == torch.float32 else 2e-6     # fp32: the same arithmetic; float64 against the reference's fp32 figures
        np.testing.assert_allclose([total.item(), al.item(), vl.item(), ent.item()], g["loss4"], rtol=tol, atol=1e-7)
        if shared:
            total.backward()
        else:
            al.backward()
            vl.backward()
        for k, p in net.named_parameters():
            if p.grad is None:      # the non-shared branch leaves log_std without a gradient when ... (never for these nets)
                continue
            flat = p.grad.double().numpy().reshape(-1)
            gmax = float(g["gmax/" + k])
            if "gfull/" + k in g.files:
                want = g["gfull/" + k]
            else:
                want, flat = g["gstride/" + k], flat[::max(1, flat.size // 4097)][:4097]
            lim = 1e-6 if dtype == torch.float32 else 2e-5      # float64 vs the reference's fp32 gradient: fp32 rounding of the latter
            assert np.abs(flat - want).max() <= lim * max(gmax, 1e-30), (str(dtype), k, np.abs(flat - want).max() / max(gmax, 1e-30))


# ---- F21: Pong-like frames, advantages over eight decades (tests/golden/make_golden_pong.py) ---------------------------
def _f21(golden):
    import hashlib
    from ddrl4nav_amd.utils.recipe import pong_frames
    g = golden("f21_pong_wide")
    frames = pong_frames(int(g["frame_seed"]), g["actions"].size)
    assert hashlib.sha256(frames.tobytes()).digest() == g["frames_sha256"].tobytes(), "pong_frames recipe changed"
    t = lambda k: torch.from_numpy(g[k])
    return g, frames, O.frames_to_f32(frames), t("actions"), t("old_logps"), t("advs"), t("rets")


def test_f21_pong_forward_losses_gradients_and_per_sample_norms(golden):
    """The oracle on the realistic / wide-range fixture: forward and losses bit-exact, the full gradient's checksums and
    samples, and the PER-SAMPLE L2 norms of d loss / d conv1-pre-activation and d loss / d encoder output -- the quantities
    the GPU tests' per-sample relative bounds are scaled by -- against the reference's own autograd."""
    g, frames, x, a, ol, adv, ret = _f21(golden)
    assert (frames == 87).mean() > 0.98 and float(np.abs(g["advs"]).max() / np.abs(g["advs"]).min()) > 1e7
    torch.set_num_threads(1)
    net = O.OraclePPO()
    net.load_weights(make_weights(0))
    with torch.no_grad():
        probs, _, logits, v = net(x)
    assert np.array_equal(probs.numpy(), g["probs"]) and np.array_equal(v.numpy()[:, 0], g["value"])
    assert np.array_equal(O.categorical_log_prob(logits, a).numpy(), g["logp"])
    net.actor.pre.tap, net.critic.pre.tap = {}, {}
    total, al, vl, ent = O.ppo_losses(net, x, a, ol, adv, ret)
    assert (np.float32(al.item()), np.float32(vl.item()), np.float32(ent.item()), np.float32(total.item())) == (
        g["actor_loss"], g["v_loss"], g["ent"], g["total"])
    al.backward()
    vl.backward()
    B = a.numel()
    for e, enc in enumerate((net.actor.pre, net.critic.pre)):
        for key, name in (("z1", "dz1_l2"), ("h", "dh_l2")):
            got = enc.tap[key].double().reshape(B, -1).norm(dim=1).numpy()
            np.testing.assert_allclose(got, g[name][e], rtol=1e-12, atol=0)
        enc.tap = None
    assert float(g["dz1_l2"].max() / g["dz1_l2"][g["dz1_l2"] > 0].min()) > 1e6  # per-sample gradients span > 20 binades
    for k, p in net.named_parameters():
        gr = p.grad.numpy()
        assert np.array_equal(gr.reshape(-1)[:64], g["ghead/" + k]), k
        assert np.array_equal(gr.reshape(-1)[::max(1, gr.size // 257)][:257], g["gstride/" + k]), k
        np.testing.assert_allclose(np.sqrt((gr.astype(np.float64) ** 2).sum()), g["gl2/" + k], rtol=1e-12)


def test_f21_pong_learn_sequence_fp32_and_float64(golden):
    g, _, x, a, ol, adv, ret = _f21(golden)
    torch.set_num_threads(1)
    net = O.OraclePPO()
    net.load_weights(make_weights(0))
    for it, (ld, ut, last) in enumerate(O.learn(net, net.make_optims(), x, a, ol, adv, ret), 1):
        got = [ld["PpoTotalLoss"], ld["ActorLoss"], ld["VLoss"], ld["EntLoss"]]
        np.testing.assert_allclose(got, g["losses"][it - 1], rtol=1e-6, atol=1e-7)
        if it in (1, 10):
            for k, p in net.named_parameters():
                arr = p.detach().numpy().reshape(-1)
                np.testing.assert_allclose(arr[::max(1, arr.size // 257)][:257], g["it%d/stride/%s" % (it, k)], rtol=1e-6, atol=2e-8)
    assert it == 10
    # the float64 run = the yardstick of the GPU sequence test (parity_util.f64_trajectory("pong"))
    import parity_util as P
    traj = P.f64_trajectory("pong")
    np.testing.assert_allclose(traj["losses"], g["losses_f64"], rtol=1e-11, atol=1e-13)
    for it in (1, 10):
        for k, arr in traj["params"][it].items():
            key = "it%d/%s" % (it, k)
            np.testing.assert_allclose(arr.sum(), g["f64_sum/" + key], rtol=1e-10, atol=1e-12)
            np.testing.assert_allclose(arr.ravel()[:8], g["f64_head/" + key], rtol=1e-10, atol=1e-14)
