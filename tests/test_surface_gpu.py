"""GPU tests of the drop-in surface: create_net / PPO.forward / PPO.learn protocol, state_dict
and Redis-blob interchange, Agents._accumulate_rewards, the device-resident rollout and the
pinned-host ring.  Run with `-m gpu`."""
import types

import numpy as np
import pytest
import torch

from ddrl4nav_amd.utils.recipe import flatten, make_weights, param_specs
from oracle import ddrl_oracle as O

pytestmark = pytest.mark.gpu


def _configs(n_actions=6):
    from ddrl4nav_amd.config import BaseConfig, ConfigNN
    env = {"env_type": "gym", "env_name": "PongNoFrameskip-v4", "env_num": 8, "int_frame_stack": 4,
           "discrete_action": True, "discrete_actions": list(range(n_actions)), "agent_num_per_env": 1, "batch_num_per_env": 8}
    parse = types.SimpleNamespace(task="test", ip="127.0.0.1")
    return {"config": BaseConfig(parse, env), "config_nn": ConfigNN(env), "config_env": env}


@pytest.fixture(scope="module")
def net():
    from ddrl4nav_amd.runner import create_net
    n = create_net(_configs(), max_batch=128)
    n.load_state_dict({k: torch.from_numpy(v.copy()) for k, v in make_weights(0).items()})
    return n


def test_create_net_param_names_and_arena_aliasing(net):
    names = [k for k, _ in net.named_parameters()]
    assert names == [n for n, _, _ in param_specs()]
    assert list(net.state_dict().keys()) == names  # .pt checkpoints interchange with the reference
    flat = net.hot_path.params
    assert flat.numel() == 3371847
    np.testing.assert_array_equal(flat.cpu().numpy(), flatten(make_weights(0)))
    # parameters are views of the flat arena
    p = dict(net.named_parameters())["critic.critic_linear.bias"]
    p.data.fill_(0.25)
    assert float(flat[1687206 + 512].item()) == 0.25
    net.load_state_dict({k: torch.from_numpy(v.copy()) for k, v in make_weights(0).items()})
    assert net.rnd is None


def test_forward_protocol_matches_reference_shapes(net, golden):
    g = golden("f1_forward")
    # the reference callers pass float32 frames = uint8/255 (forward.py:102-104)
    x = torch.from_numpy(O.u8_lut()[g["frames"]])
    (dist, logp), values = net([x], torch.from_numpy(g["actions"]))
    assert isinstance(values, list) and values[0].shape == (8, 1)
    np.testing.assert_allclose(logp.cpu().numpy(), g["logp"], rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(values[0].cpu().numpy()[:, 0], g["value"], rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(dist.probs.cpu().numpy(), g["p_hat"], rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(dist.logits.cpu().numpy(), g["logits"], rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(dist.entropy().cpu().numpy(), g["entropy"], rtol=1e-5, atol=1e-6)
    # acting call pattern of ForwardThread.run (forward.py:132-138)
    (dist, none), values = net([torch.from_numpy(g["frames"])])
    assert none is None
    actions = dist.sample().to(torch.float32)
    logps = net.actor.log_prob_from_distribution(dist, actions)
    want = dist.logits.gather(-1, actions.long().unsqueeze(-1)).squeeze(-1)
    np.testing.assert_allclose(logps.cpu().numpy(), want.cpu().numpy(), rtol=1e-6, atol=1e-7)
    assert actions.shape == (8,) and float(actions.min()) >= 0 and float(actions.max()) <= 5
    again = dist.sample()
    assert again.shape == (8,)
    # play_mode returns the raw softmax (actor.py:94-96)
    (probs, _), _ = net([torch.from_numpy(g["frames"])], play_mode=True)
    np.testing.assert_allclose(probs.cpu().numpy(), g["probs"], rtol=1e-5, atol=1e-6)
    assert torch.argmax(probs, dim=1).shape == (8,)


def test_actor_and_critic_are_callable_like_the_reference(net, golden):
    """net.actor(states, act) / net.critic(states) (reference actor.py:27-40, critic.py:14-21; ppo.py:75 composes exactly these)
    return what the fused forward returns, pinned to golden F1."""
    g = golden("f1_forward")
    x = [torch.from_numpy(g["frames"])]
    dist, logp = net.actor(x, torch.from_numpy(g["actions"]))
    np.testing.assert_allclose(logp.cpu().numpy(), g["logp"], rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(dist.probs.cpu().numpy(), g["p_hat"], rtol=1e-5, atol=1e-6)
    v = net.critic(x)
    assert v.shape == (8, 1)
    np.testing.assert_allclose(v.cpu().numpy()[:, 0], g["value"], rtol=1e-5, atol=1e-6)
    probs, none = net.actor(x, play_mode=True)
    assert none is None
    np.testing.assert_allclose(probs.cpu().numpy(), g["probs"], rtol=1e-5, atol=1e-6)


def test_learn_generator_protocol_golden_f4(net, golden):
    from ddrl4nav_amd.data import Experience
    g3, g4 = golden("f3_loss"), golden("f4_learn")
    net.load_state_dict({k: torch.from_numpy(v.copy()) for k, v in make_weights(0).items()})
    net.hot_path.reset_optimizer()
    net.update_time = 0
    exp = Experience(states=[g3["frames"]], advs=g3["advs"], actions=g3["actions"], old_logps=g3["old_logps"],
                     values=g3["rets"].reshape(1, -1))
    exp.to_tensor(dtype=torch.float32, device="cuda")
    import parity_util as P
    ref = g4["losses"]
    env = P.mode_loss_envelope("default", ref, g4["losses_f64"], g4["losses_f32t8"])
    gen = net.learn(exp)
    seen = [0]

    def step():
        loss_items, update_time, last = next(gen)
        seen[0] += 1
        assert update_time == seen[0] and last is True
        assert set(loss_items) == {"PpoTotalLoss", "ActorLoss", "VLoss", "EntLoss", "PpoBackUpTime"}
        return [loss_items[k] for k in ("PpoTotalLoss", "ActorLoss", "VLoss", "EntLoss")]

    # same bounds as tests/test_gpu_parity.py::test_learn_sequence_golden_f4, through the drop-in surface
    P.check_sequence("learn_f4", "default", step, lambda: net.hot_path.params.cpu().numpy(), ref, env)
    assert next(gen, None) is None and seen[0] == 10 and net.update_time == 10


def test_redis_blob_and_checkpoint_roundtrip(net, tmp_path):
    store = {}

    class Pipe:
        def set(self, k, v):
            store[k] = v

        def incr(self, k):
            store[k] = store.get(k, 0) + 1

        def execute(self):
            pass

    class Conn:
        def get(self, k):
            return store[k]

    net.load_state_dict({k: torch.from_numpy(v.copy()) for k, v in make_weights(0).items()})
    net.nn2redis(Pipe(), "TAG")
    assert store["TAG"] == 1 and len(store[net.model_key]) == 13487388 + sum(4 + 4 * len(s) for _, s, _ in param_specs())
    before = net.hot_path.params.clone()
    net.hot_path.params.zero_()
    net.updatenn_by_redis(Conn())
    assert torch.equal(net.hot_path.params, before)
    path = str(tmp_path / "m_10.pt")
    torch.save(net.state_dict(), path)
    net.hot_path.params.zero_()
    net.updatenn("file://" + path)
    assert torch.equal(net.hot_path.params, before)
    # derived weight layouts follow the reload: forward still matches the oracle
    frames = np.random.default_rng(0).integers(0, 256, size=(4, 4, 84, 84), dtype=np.uint8)
    (probs, _), values = net([torch.from_numpy(frames)], play_mode=True)
    onet = O.OraclePPO()
    onet.load_weights(make_weights(0))
    with torch.no_grad():
        oprobs, _, _, ov = onet(O.frames_to_f32(frames))
    np.testing.assert_allclose(probs.cpu().numpy(), oprobs.numpy(), rtol=1e-5, atol=1e-6)


def test_agents_accumulate_rewards_golden_f2(golden):
    from ddrl4nav_amd.agent import Agents
    from ddrl4nav_amd.data import Experience
    cfgs = _configs()
    ag = Agents(config=cfgs["config"], config_nn=cfgs["config_nn"], config_env=cfgs["config_env"])
    g = golden("f2_gae")
    T, N = 256, 8
    exps = [Experience(states=[np.zeros((N, 1), np.float32)], values=g["values"][t:t + 1].copy(),
                       dones=g["dones"][t:t + 1].copy()) for t in range(T + 1)]
    out = ag._accumulate_rewards(exps, g["rewards"][:T + 1].reshape(T + 1, 1, N).copy())
    assert len(out) == T and out[0] is exps[0]
    assert np.array_equal(np.stack([e.advs for e in out]), g["adv1"])
    assert np.array_equal(np.stack([e.values[0] for e in out]), g["ret1"])
    assert ag._accumulate_rewards([], None) == []


def test_device_rollout_end_to_end_vs_oracle(net):
    """T acting steps + bootstrap + GAE + one learn() on the pool, against the oracle run on the
    same frames with the actions the HIP sampler drew."""
    from ddrl4nav_amd.agent import DeviceRollout
    net.load_state_dict({k: torch.from_numpy(v.copy()) for k, v in make_weights(0).items()})
    net.hot_path.reset_optimizer()
    N, T = 6, 5
    rng = np.random.default_rng(42)
    frames = rng.integers(0, 256, size=(T + 1, N, 4, 84, 84), dtype=np.uint8)
    rewards = rng.choice(np.array([-1, 0, 1], np.float32), size=(T, N)).astype(np.float32)
    dones = (rng.random((T, N)) < 0.2).astype(np.uint8)
    ro = DeviceRollout(net, N, horizon=T, seed=3)
    for t in range(T):
        ro.put_frames(t, torch.from_numpy(frames[t]).cuda())
        ro.act(t)
        ro.record(t, torch.from_numpy(rewards[t]).cuda(), torch.from_numpy(dones[t]).cuda())
    ro.put_frames(T, torch.from_numpy(frames[T]).cuda())
    ro.bootstrap()
    ro.finish()
    onet = O.OraclePPO()
    onet.load_weights(make_weights(0))
    x_all = O.frames_to_f32(frames.reshape(-1, 4, 84, 84))
    with torch.no_grad():
        _, _, ologits, ov = onet(x_all)
    ov = ov.numpy()[:, 0].reshape(T + 1, N)
    np.testing.assert_allclose(ro.values.cpu().numpy(), ov, rtol=1e-5, atol=1e-6)
    acts = ro.actions.cpu().numpy()
    ologp = O.categorical_log_prob(ologits[:T * N], torch.from_numpy(acts.reshape(-1))).numpy().reshape(T, N)
    np.testing.assert_allclose(ro.logps.cpu().numpy(), ologp, rtol=1e-5, atol=1e-6)
    oadv, oret = O.gae(ro.values.cpu().numpy(), rewards, dones)  # same values in -> bit-exact scan
    assert np.array_equal(ro.adv.cpu().numpy(), oadv) and np.array_equal(ro.ret.cpu().numpy(), oret)
    batch = ro.batch()
    assert len(batch) == N * T and batch.values.shape == (1, N * T)
    first = next(net.learn(batch))[0]
    opt = onet.make_optims()
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a.reshape(-1)))
    old = t(ro.logps.cpu().numpy())
    ld = next(O.learn(onet, opt, x_all[:T * N], t(acts), old, t(oadv), t(oret), iters=1))[0]
    for k in ("ActorLoss", "VLoss", "EntLoss", "PpoTotalLoss"):
        np.testing.assert_allclose(first[k], ld[k], rtol=1e-4, atol=1e-5)
    ro.carry_over()
    assert torch.equal(ro.frames[0], ro.frames[T])


def test_carry_over_keep_step_is_the_references_carry(net, golden):
    """agent.py:286-292: the whole (T+1)-th step -- frame, action already sent to the env, old log-prob, value under the
    weights of that moment, reward, done -- becomes step 0 of the next rollout, which then acts from step 1 on.
    (a) GAE side through the POOL on F2 (two chained rollouts of the reference's own _accumulate_rewards): bit-exact;
    (b) acting side: act(0) of the next rollout returns the kept action and evaluates nothing."""
    from ddrl4nav_amd.agent import DeviceRollout
    g = golden("f2_gae")
    T, N = 256, 8
    dev = "cuda"
    ro = DeviceRollout(net, N, horizon=T, seed=5, track_returns=True)
    vals, rew, dones = (torch.from_numpy(g[k]).to(dev) for k in ("values", "rewards", "dones"))
    ro.values.copy_(vals[:T + 1])
    for t in range(T + 1):               # rows 0..T: the reference stores T+1 steps before it scans (agent.py:272-276)
        ro.record(t, rew[t], dones[t])
    ro.finish()
    assert np.array_equal(ro.adv.cpu().numpy(), g["adv1"]) and np.array_equal(ro.ret.cpu().numpy(), g["ret1"])
    ro.carry_over(keep_step=True)
    assert ro.t0 == 1
    assert torch.equal(ro.values[0], vals[T]) and torch.equal(ro.rewards[0], rew[T]) and torch.equal(ro.dones[0], dones[T])
    with pytest.raises(ValueError):
        ro.record(0, rew[T], dones[T])   # slot 0 came with its reward and done
    ro.values[1:].copy_(vals[T + 1:2 * T + 1])
    for t in range(1, T + 1):
        ro.record(t, rew[T + t], dones[T + t])
    ro.finish()
    assert np.array_equal(ro.adv.cpu().numpy(), g["adv2"]) and np.array_equal(ro.ret.cpu().numpy(), g["ret2"])
    # every step entered the episode-return accumulator exactly once (rows 0..T-1 of both rollouts = fixture rows 0..2T-1)
    tr, rsum = O.episode_returns(g["rewards"][:2 * T], g["dones"][:2 * T])
    assert np.array_equal(ro.returns.rewards_episode.cpu().numpy(), tr[-1])
    assert np.array_equal(ro.returns.rewards_sum.cpu().numpy(), rsum)

    # (b) acting side on a short rollout
    net.load_state_dict({k: torch.from_numpy(v.copy()) for k, v in make_weights(0).items()})
    N, T = 6, 4
    rng = np.random.default_rng(8)
    frames = torch.from_numpy(rng.integers(0, 256, size=(2 * T + 1, N, 4, 84, 84), dtype=np.uint8)).to(dev)
    r = torch.from_numpy(rng.choice(np.array([-1, 0, 1], np.float32), size=(2 * T + 1, N)).astype(np.float32)).to(dev)
    d = torch.from_numpy((rng.random((2 * T + 1, N)) < 0.3).astype(np.uint8)).to(dev)
    ro = DeviceRollout(net, N, horizon=T, seed=9)
    for t in range(T):
        ro.put_frames(t, frames[t])
        ro.act(t)
        ro.record(t, r[t], d[t])
    ro.put_frames(T, frames[T])
    a_T = ro.bootstrap().clone()          # the action the host steps the env with
    ro.record(T, r[T], d[T])
    ro.finish()
    kept = [x.clone() for x in (ro.frames[T], ro.values[T], ro._actions[T], ro._logps[T], ro._rewards[T], ro._dones[T])]
    assert torch.equal(kept[2], a_T)
    # weights move between the rollouts (an update happened): the kept step must NOT be re-evaluated
    net.hot_path.params.mul_(1.001)
    net.hot_path.params_changed()
    ro.carry_over(keep_step=True)
    assert torch.equal(ro.act(0), a_T)
    now = (ro.frames[0], ro.values[0], ro.actions[0], ro.logps[0], ro.rewards[0], ro.dones[0])
    for a, b in zip(kept, now):
        assert torch.equal(a, b)
    for t in range(ro.t0, T):
        ro.put_frames(t, frames[T + t])
        ro.act(t)
        ro.record(t, r[T + t], d[T + t])
    ro.put_frames(T, frames[2 * T])
    ro.bootstrap()
    ro.finish()
    oadv, oret = O.gae(ro.values.cpu().numpy(), ro.rewards.cpu().numpy(), ro.dones.cpu().numpy())
    assert np.array_equal(ro.adv.cpu().numpy(), oadv) and np.array_equal(ro.ret.cpu().numpy(), oret)
    # default carry-over: the frame only, slot 0 is evaluated again under the current weights
    v0 = ro.values[T].clone()
    net.hot_path.params.mul_(1.001)
    net.hot_path.params_changed()
    ro.carry_over()
    assert ro.t0 == 0
    ro.act(0)
    assert not torch.equal(ro.values[0], v0)
    net.load_state_dict({k: torch.from_numpy(v.copy()) for k, v in make_weights(0).items()})


def test_episode_returns_device_accumulator_golden_f6(golden):
    """Status.update_reward_status (statistics.py:118-123) on the device, fed in chunks the way rollouts arrive: the trace and
    the final running sums of the reference's own Status run (F6), bit for bit."""
    from ddrl4nav_amd.agent import EpisodeReturns
    g = golden("f6_returns")
    r = torch.from_numpy(g["rewards"]).cuda()
    d = torch.from_numpy(g["dones"].astype(np.uint8)).cuda()
    Tn, Nn = r.shape
    acc = EpisodeReturns(Nn, "cuda")
    trace = torch.empty_like(r)
    for lo, hi in ((0, 1), (1, 130), (130, 257), (257, Tn)):   # ragged chunks: the state carries across calls
        acc.update(r[lo:hi].contiguous(), d[lo:hi].contiguous(), trace=trace[lo:hi])
    assert np.array_equal(trace.cpu().numpy(), g["trace"])
    assert np.array_equal(acc.rewards_sum.cpu().numpy(), g["final_sum"])
    assert np.array_equal(acc.rewards_episode.cpu().numpy(), g["trace"][-1])
    assert np.array_equal(acc.episodes_finished.cpu().numpy(), g["dones"].sum(0).astype(np.int32))
    # non-integer rewards: against the oracle's restatement (same fp32 operation order)
    rng = np.random.default_rng(66)
    r2 = rng.normal(size=(300, 70)).astype(np.float32)
    d2 = (rng.random((300, 70)) < 0.05).astype(np.uint8)
    acc2 = EpisodeReturns(70, "cuda")
    tr2 = torch.empty((300, 70), dtype=torch.float32, device="cuda")
    acc2.update(torch.from_numpy(r2).cuda(), torch.from_numpy(d2).cuda(), trace=tr2)
    otr, osum = O.episode_returns(r2, d2.astype(np.float32))
    assert np.array_equal(tr2.cpu().numpy(), otr) and np.array_equal(acc2.rewards_sum.cpu().numpy(), osum)
    with pytest.raises(ValueError):
        acc.update(r[:, :2].contiguous(), d[:, :2].contiguous())


def test_chained_rollout_gae_learn_config1_shape():
    """BASELINE config 1's shape (8 envs, TIME_MAX 256, TRAINING_ITER_TIME 10) END TO END: 256 acting steps + bootstrap on the device
    pool -> GAE -> ten PPO iterations on the 2,048 samples, against the oracle running the SAME chain on the same frames with the
    actions the HIP sampler drew (forward.py:128-149 -> agent.py:124-140 -> ppo.py:77-146).
    Acting and GAE: single-step tolerances.  The ten-iteration trajectory: deviation from the oracle's float64 chain as a ratio to
    the spread of the oracle's own fp32 chains around it (in order, another batch order, two runs with every parameter perturbed
    by <= 4 fp32 roundings before each forward, and one on torch's native convolution backend instead of oneDNN -- what "another
    fp32 evaluation of the reference" looks like: tests/parity_util.py BACKEND).  Fixed limits:
    losses 2.0 x that envelope beyond the single-step tolerance, parameters 1.5 x (L2 / max / direction)."""
    import parity_util as P
    from ddrl4nav_amd.agent import DeviceRollout
    from ddrl4nav_amd.runner import create_net
    N, T, ITERS = 8, 256, 10
    B = N * T
    net = create_net(_configs(), max_batch=B)
    w = make_weights(0)
    net.load_state_dict({k: torch.from_numpy(v.copy()) for k, v in w.items()})
    rng = np.random.default_rng(256)
    frames = rng.integers(0, 256, size=(T + 1, N, 4, 84, 84), dtype=np.uint8)
    frames[::5] = (frames[::5] // 64) * 64                      # some low-entropy frames
    rewards = rng.choice(np.array([-1, 0, 1], np.float32), p=[0.05, 0.9, 0.05], size=(T, N)).astype(np.float32)
    dones = (rng.random((T, N)) < 1 / 60).astype(np.uint8)
    ro = DeviceRollout(net, N, horizon=T, seed=77, track_returns=True)
    fr = torch.from_numpy(frames).cuda()
    r_dev, d_dev = torch.from_numpy(rewards).cuda(), torch.from_numpy(dones).cuda()
    for t in range(T):
        ro.put_frames(t, fr[t])
        ro.act(t)
        ro.record(t, r_dev[t], d_dev[t])
    ro.put_frames(T, fr[T])
    ro.bootstrap()
    ro.finish()
    acts = ro.actions.cpu().numpy()
    x_all = O.frames_to_f32(frames.reshape(-1, 4, 84, 84))
    a_t = torch.from_numpy(acts.reshape(-1))
    threads = torch.get_num_threads()
    torch.set_num_threads(min(16, max(1, (__import__("os").cpu_count() or 1))))

    def chain(dtype, order=None, noise_seed=None, native=False):
        """The reference's chain in `dtype`: values / log-probs of the given actions, GAE, ten learn iterations."""
        with torch.backends.mkldnn.flags(enabled=not native):
            return _chain(dtype, order, noise_seed)

    def _chain(dtype, order, noise_seed):
        onet = O.OraclePPO()
        onet.load_weights(w)
        onet.to(dtype)
        gen = None if noise_seed is None else torch.Generator().manual_seed(noise_seed)

        def perturb():
            if gen is not None:
                with torch.no_grad():
                    for p in onet.parameters():
                        p.mul_(1 + (torch.randint(0, 2, p.shape, generator=gen).to(p.dtype) * 2 - 1) * 4.0 * 2.0 ** -24)
        perturb()
        with torch.no_grad():
            outs = [onet(x_all[i:i + 512].to(dtype)) for i in range(0, x_all.shape[0], 512)]
        logits = torch.cat([o[2] for o in outs])
        v = torch.cat([o[3] for o in outs])[:, 0].reshape(T + 1, N)
        logp = O.categorical_log_prob(logits[:B], a_t.to(dtype))
        if dtype == torch.float64:                                # agent.py:124-140 in float64 (the yardstick's own GAE)
            vv, g, nv = v.numpy(), np.zeros(N), v.numpy()[T]
            adv, ret = np.empty((T, N)), np.empty((T, N))
            for t in reversed(range(T)):
                k = 1.0 - dones[t]
                g = g * k
                g = np.float64(np.float32(0.99)) * 0.95 * g + (np.float64(np.float32(0.99)) * nv * k - vv[t] + rewards[t])
                nv = vv[t]
                ret[t], adv[t] = vv[t] + g, g
        else:
            adv, ret = O.gae(v.numpy(), rewards, dones)
        idx = np.arange(B) if order is None else order
        tt = lambda a: torch.from_numpy(np.ascontiguousarray(np.asarray(a).reshape(-1)[idx])).to(dtype)
        xs = x_all[:B][idx].to(dtype)
        rows, snaps = [], {}
        learner = O.learn(onet, onet.make_optims(), xs, tt(acts), tt(logp.numpy()), tt(adv), tt(ret), iters=ITERS)
        for it in range(1, ITERS + 1):
            if it > 1:
                perturb()
            ld, _, _ = next(learner)
            rows.append([ld[k] for k in ("PpoTotalLoss", "ActorLoss", "VLoss", "EntLoss")])
            if it in (1, ITERS):
                snaps[it] = {k: p.detach().double().numpy().copy() for k, p in onet.named_parameters()}
        return {"v": v.double().numpy(), "logp": logp.double().numpy().reshape(T, N), "adv": np.asarray(adv, np.float64),
                "ret": np.asarray(ret, np.float64), "losses": np.asarray(rows, np.float64), "params": snaps}

    try:
        c64 = chain(torch.float64)
        variants = [chain(torch.float32), chain(torch.float32, order=np.random.default_rng(1).permutation(B)),
                    chain(torch.float32, noise_seed=501), chain(torch.float32, noise_seed=502), chain(torch.float32, native=True)]
    finally:
        torch.set_num_threads(threads)
    c32 = variants[0]
    # ---- acting + GAE: single-step tolerances against the reference-order fp32 chain ------------------------------------
    np.testing.assert_allclose(ro.values.cpu().numpy(), c32["v"], rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(ro.logps.cpu().numpy(), c32["logp"], rtol=1e-5, atol=1e-6)
    oadv, oret = O.gae(ro.values.cpu().numpy(), rewards, dones)                 # same values in -> bit-exact scan
    assert np.array_equal(ro.adv.cpu().numpy(), oadv) and np.array_equal(ro.ret.cpu().numpy(), oret)
    np.testing.assert_allclose(ro.adv.cpu().numpy(), c32["adv"], rtol=1e-4, atol=3e-5)   # value noise x sum of (gamma landa)^k <= 17
    tr, _ = O.episode_returns(rewards, dones)
    assert np.array_equal(ro.returns.rewards_episode.cpu().numpy(), tr[-1])
    # ---- ten PPO iterations on the pool's batch ---------------------------------------------------------------------------
    net.hot_path.reset_optimizer()
    env = P.loss_envelope(c64["losses"], *[v["losses"] for v in variants])
    p0 = {k: np.asarray(v, np.float64) for k, v in w.items()}
    ref = {}
    for it in (1, ITERS):
        for k in p0:
            a64 = c64["params"][it][k]
            u64 = (a64 - p0[k]).ravel()
            l2 = mx = omc = 0.0
            for v in variants:
                dd = (v["params"][it][k] - a64).ravel()
                l2, mx = max(l2, float(np.sqrt(dd @ dd))), max(mx, float(np.abs(dd).max()))
                uv = (v["params"][it][k] - p0[k]).ravel()
                den = np.linalg.norm(uv) * np.linalg.norm(u64)
                omc = max(omc, 1.0 - float(uv @ u64 / den) if den > 0 else 0.0)
            kk = "it%d/%s" % (it, k)
            ref["ref_l2/" + kk], ref["ref_max/" + kk], ref["ref_1mcos/" + kk], ref["upd_l2/" + kk] = l2, mx, omc, float(np.linalg.norm(u64))
    seen = 0
    for ld, update_time, last in net.learn(ro.batch()):
        seen += 1
        assert update_time == seen and last is True
        got = np.array([ld[k] for k in ("PpoTotalLoss", "ActorLoss", "VLoss", "EntLoss")])
        row = c64["losses"][seen - 1]
        excess = np.abs(got - row) - (1e-5 * np.abs(row) + 2e-6)
        P.MARGINS.check("chain_config1", "loss_env", max(0.0, float(np.max(excess / np.maximum(env[seen - 1], 1e-12)))), "(iteration %d)" % seen)
        if seen in (1, ITERS):
            gotp = P.split_flat(net.hot_path.params.cpu().numpy().astype(np.float64))
            worst = P.deviation_ratios(gotp, c64["params"][seen], p0, ref, lambda n, it=seen: "it%d/%s" % (it, n), lambda n: n.split(".")[0])
            for k, (v, pname) in worst.items():
                P.MARGINS.check("chain_config1", "param_%s_it%d" % (k, seen), v, "(%s)" % pname)
    assert seen == ITERS


def test_pinned_ring_feeds_pool():
    from ddrl4nav_amd.data import PinnedRing
    from ddrl4nav_amd._lib import DdrlError
    slot = 8 * 4 * 84 * 84
    ring = PinnedRing(slot, n_slots=3)
    rng = np.random.default_rng(1)
    dst = torch.zeros((5, 8, 4, 84, 84), dtype=torch.uint8, device="cuda")
    sent = []
    copy_stream = torch.cuda.Stream()
    # the zero fill of `dst` runs on the CURRENT stream: order it in front of the copies (data/ring.py: with a stream of its own the
    # caller orders; without this line the fill can land after a copy -- seen once in round 6 as an all-zero slot)
    copy_stream.wait_stream(torch.cuda.current_stream())
    for i in range(5):  # more messages than slots: exercises recycling
        buf = ring.acquire(timeout_ms=2000)
        data = rng.integers(0, 256, size=slot, dtype=np.uint8)
        buf[:slot] = data
        sent.append(data)
        ring.commit()
        assert ring.pending() == 1
        ring.pop_to(dst[i], stream=copy_stream)
    copy_stream.synchronize()
    for i in range(5):
        assert np.array_equal(dst[i].cpu().numpy().reshape(-1), sent[i])
    with pytest.raises(DdrlError):
        ring.pop_to(dst[0], timeout_ms=10)  # nothing committed -> timeout, not a hang
    ring.close()


def test_redis_message_to_device_pool_via_codec_and_ring(net):
    """A reference env worker's forward-states message (float64 frames) -> C codec -> pinned ring
    -> hipMemcpyAsync -> forward: the drop-in ingest path of INTEGRATION.md."""
    from ddrl4nav_amd.data import EasyBytes, PinnedRing
    net.load_state_dict({k: torch.from_numpy(v.copy()) for k, v in make_weights(0).items()})
    rng = np.random.default_rng(21)
    fa = rng.integers(0, 256, size=(3, 4, 84, 84), dtype=np.uint8)
    fb = rng.integers(0, 256, size=(2, 4, 84, 84), dtype=np.uint8)
    eb = EasyBytes("127.0.0.1")
    item = eb.encode_forward_states(0, [fa / 255.0]) + eb.encode_forward_states(1, [fb / 255.0])
    ring = PinnedRing(5 * 4 * 84 * 84, n_slots=2)
    slot = ring.acquire()
    n, per = eb.frames_to_u8(item, slot)
    assert (n, per) == (5, 4 * 84 * 84)
    ring.commit()
    dst = torch.empty((5, 4, 84, 84), dtype=torch.uint8, device="cuda")
    ring.pop_to(dst)
    torch.cuda.synchronize()
    assert np.array_equal(dst.cpu().numpy(), np.concatenate([fa, fb]))
    (probs, _), values = net([dst], play_mode=True)
    # same result as feeding the reference's decoded float64 states
    ids, states = eb.decode_forward_states(item)
    (probs2, _), values2 = net([torch.from_numpy(states[0])], play_mode=True)
    assert torch.equal(probs, probs2) and torch.equal(values[0], values2[0]) and ids == ["127.0.0.1_0", "127.0.0.1_1"]
    replies = eb.encode_forward_return_data([torch.argmax(probs, 1).float(), torch.zeros(5), torch.stack(values, 0)], [3, 2])
    a, lp, v = eb.decode_data(replies[1])
    assert a.shape == (2,) and v.shape == (1, 2, 1)
    ring.close()


def test_create_net_shared_prenet_branch(golden):
    """SHARE_CNN_NET=True / SMOOTH_L1_LOSS=True through the reference's own surface
    (runner/utils.py:136-143, ppo.py:110-117): names, forward, learn protocol."""
    from ddrl4nav_amd.data import Experience
    from ddrl4nav_amd.runner import create_net
    cfgs = _configs()
    cfgs["config_nn"].SHARE_CNN_NET = True
    n = create_net(cfgs, max_batch=64)
    names = [k for k, _ in n.named_parameters()]
    assert names == [k for k, _, _ in param_specs(shared=True)] and list(n.state_dict().keys()) == names
    assert n.prenet is not None and n.actor.pre is None and n.critic.pre is None and n.share_cnn_net
    n.load_state_dict({k: torch.from_numpy(v.copy()) for k, v in make_weights(0, shared=True).items()})
    g3, g = golden("f3_loss"), golden("f10_shared")
    (dist, logp), values = n([torch.from_numpy(g3["frames"])], torch.from_numpy(g["actions"]))
    np.testing.assert_allclose(logp.cpu().numpy(), g["logp"], rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(values[0].cpu().numpy()[:, 0], g["value"], rtol=1e-5, atol=1e-6)
    exp = Experience(states=[g3["frames"]], advs=g["advs"], actions=g["actions"], old_logps=g["old_logps"],
                     values=g["rets"].reshape(1, -1))
    exp.to_tensor(dtype=torch.float32, device="cuda")
    ref = g["losses"]
    env = np.maximum.accumulate(np.maximum(np.abs(ref - g["losses_f64"]), np.abs(ref - g["losses_f32t8"])), axis=0)
    seen = 0
    for loss_items, update_time, last in n.learn(exp):
        seen += 1
        got = np.array([loss_items[k] for k in ("PpoTotalLoss", "ActorLoss", "VLoss", "EntLoss")])
        assert np.all(np.abs(got - ref[seen - 1]) <= 10.0 * env[seen - 1] + 1e-5 * np.abs(ref[seen - 1]) + 2e-6)
    assert seen == 10
    # mismatched prenet / SHARE_CNN_NET combinations are rejected instead of silently ignored
    from ddrl4nav_amd.nn import PPO
    cfgs2 = _configs()
    with pytest.raises(ValueError):
        PPO(n.actor, n.critic, n.prenet, None, cfgs2["config"], cfgs2["config_nn"], max_batch=8)


def test_c_abi_from_plain_c(tmp_path):
    """tests/c/abi_smoke.c: a C program (no Python, no torch) drives forward / GAE / PPO iteration /
    Adam through include/ddrl.h."""
    import os
    import shutil
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    cc = shutil.which("gcc") or "cc"
    lib_dir = os.path.join(root, "ddrl4nav_amd", "csrc")
    exe = str(tmp_path / "abi_smoke")
    subprocess.run([cc, "-std=c11", "-O2", "-D__HIP_PLATFORM_AMD__", os.path.join(root, "tests", "c", "abi_smoke.c"),
                    "-I", os.path.join(root, "include"), "-I", "/opt/rocm/include", "-L", lib_dir, "-L", "/opt/rocm/lib",
                    "-lddrl_hip", "-lamdhip64", "-lm", "-o", exe], check=True, timeout=300)
    env = dict(os.environ, LD_LIBRARY_PATH=lib_dir + ":/opt/rocm/lib:" + os.environ.get("LD_LIBRARY_PATH", ""))
    try:
        out = subprocess.run([exe], env=env, capture_output=True, text=True, timeout=120)
    except subprocess.TimeoutExpired as e:  # the stage markers on stderr say where it stopped
        raise AssertionError("abi_smoke hung; stderr so far:\n%s" % (e.stderr.decode() if isinstance(e.stderr, bytes) else e.stderr))
    assert out.returncode == 0, out.stderr
    assert "C ABI OK" in out.stdout


def test_pinned_ring_producer_consumer_threads():
    """Producer and consumer on different threads, more messages than slots: order is preserved, no
    slot is overwritten before its copy finished, back-pressure blocks instead of dropping."""
    import threading
    from ddrl4nav_amd.data import PinnedRing
    slot, n_msgs = 256 * 1024, 64
    ring = PinnedRing(slot, n_slots=4)
    dst = torch.zeros((n_msgs, slot), dtype=torch.uint8, device="cuda")
    errors = []

    def producer():
        try:
            for i in range(n_msgs):
                buf = ring.acquire(timeout_ms=5000)
                buf[:] = (i * 7 + np.arange(slot) % 251) % 256   # content depends on the message index
                ring.commit()
        except Exception as e:  # pragma: no cover
            errors.append(e)

    t = threading.Thread(target=producer)
    t.start()
    stream = torch.cuda.Stream()
    stream.wait_stream(torch.cuda.current_stream())   # dst's zero fill (current stream) in front of the copies
    for i in range(n_msgs):
        ring.pop_to(dst[i], stream=stream, timeout_ms=5000)
    t.join(timeout=30)
    stream.synchronize()
    assert not errors and not t.is_alive() and ring.pending() == 0
    got = dst.cpu().numpy()
    base = np.arange(slot) % 251
    for i in range(n_msgs):
        assert np.array_equal(got[i], ((i * 7 + base) % 256).astype(np.uint8)), i
    ring.close()


def test_sampler_streams_use_all_64_bits_and_successive_draws_differ(net):
    """Stream ids carry the forward counter / draw index in their HIGH bits: ids that differ only
    above bit 24 (or 32, or 48) must give independent draws, a second dist.sample() is a new draw,
    and the draws follow utils.recipe.sample_uniform (the documented counter stream)."""
    from ddrl4nav_amd.utils.recipe import sample_uniform
    hp = net.hot_path
    n = 128
    probs = torch.full((n, 6), 1.0 / 6, device="cuda")
    base, _ = hp.categorical_sample(probs, 7, 3)
    for shift in (24, 32, 40, 48, 63):
        other, _ = hp.categorical_sample(probs, 7, 3 + (1 << shift))
        assert not torch.equal(base, other), shift
        u = sample_uniform(7, 3 + (1 << shift), n)
        want = np.minimum((u.astype(np.float64) * 6).astype(np.int64), 5)
        # inverse CDF over six equal bins (a draw within 1e-6 of a bin edge may land on either side)
        got = other.cpu().numpy()
        edge = np.abs(u.astype(np.float64) * 6 - np.round(u.astype(np.float64) * 6)) < 1e-5
        assert np.array_equal(got[~edge], want[~edge].astype(np.float32)), shift
    again, _ = hp.categorical_sample(probs, 7, 3)
    assert torch.equal(base, again)  # same (seed, stream) -> same draw
    rng = np.random.default_rng(5)
    frames = torch.from_numpy(rng.integers(0, 256, size=(n, 4, 84, 84), dtype=np.uint8))
    (dist, _), _ = net([frames])
    a1 = dist.sample().clone()
    a2 = dist.sample().clone()
    a3 = dist.sample().clone()
    assert not torch.equal(a1, a2) and not torch.equal(a2, a3) and not torch.equal(a1, a3)
    lp = net.actor.log_prob_from_distribution(dist, a3)
    want = dist.logits.gather(-1, a3.long().unsqueeze(-1)).squeeze(-1)
    np.testing.assert_allclose(lp.cpu().numpy(), want.cpu().numpy(), rtol=1e-6, atol=1e-7)


def test_init_weight_refreshes_packed_weights(net):
    """Basenn.init_weight writes the parameter views in place; the kernels must see the new weights."""
    w0 = {k: torch.from_numpy(v.copy()) for k, v in make_weights(0).items()}
    rng = np.random.default_rng(6)
    frames = torch.from_numpy(rng.integers(0, 256, size=(16, 4, 84, 84), dtype=np.uint8))
    (p0, _), v0 = net([frames], play_mode=True)
    p0, v0 = p0.clone(), v0[0].clone()
    torch.manual_seed(3)
    net.init_weight()
    (p1, _), v1 = net([frames], play_mode=True)
    assert not torch.allclose(p0, p1) and not torch.allclose(v0, v1[0])
    onet = O.OraclePPO()
    onet.load_state_dict({k: v.detach().cpu() for k, v in net.state_dict().items()})
    with torch.no_grad():
        oprobs, _, _, ov = onet(O.frames_to_f32(frames.numpy()))
    np.testing.assert_allclose(p1.cpu().numpy(), oprobs.numpy(), rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(v1[0].cpu().numpy(), ov.numpy(), rtol=1e-5, atol=2e-6)
    net.load_state_dict(w0)


def test_rccl_communicator_through_the_c_abi(net):
    """ddrl_comm_* / ddrl_grad_allreduce / ddrl_params_broadcast with a communicator of one rank (a one-GPU box cannot hold
    two RCCL ranks; the N > 1 arithmetic is covered by tests/test_dist_gpu.py over gloo): SUM over one rank is the identity,
    bit for bit, and the learner step runs with the collective between ppo_iter and clip_adam."""
    from ctypes import c_void_p
    from ddrl4nav_amd._lib import check
    from ddrl4nav_amd.dist import RcclComm
    hp = net.hot_path
    comm = RcclComm(0, 1)
    x = torch.randn(100003, device="cuda")
    want = x.clone()
    comm.allreduce(x)
    comm.broadcast(x, 0)
    torch.cuda.synchronize()
    assert torch.equal(x, want)
    rng = np.random.default_rng(9)
    n = 32
    frames = torch.from_numpy(rng.integers(0, 256, size=(n, 4, 84, 84), dtype=np.uint8)).cuda()
    f = lambda a: torch.from_numpy(a.astype(np.float32)).cuda()
    args = (frames, f(rng.integers(0, 6, size=n)), f(np.full(n, -1.79)), f(rng.normal(size=n)), f(rng.normal(size=n)))
    hp.ppo_iter(*args)
    before = hp.grads.clone()
    s = c_void_p(torch.cuda.current_stream().cuda_stream)
    check(hp.lib.ddrl_grad_allreduce(hp.ctx, comm.h, s))
    check(hp.lib.ddrl_params_broadcast(hp.ctx, comm.h, 0, s))
    torch.cuda.synchronize()
    assert torch.equal(hp.grads, before)
    comm.close()
    net.load_state_dict({k: torch.from_numpy(v.copy()) for k, v in make_weights(0).items()})


def test_backward_trainer_publish_save_and_checkpoint_offsets(net, golden, tmp_path):
    """The consumer loop of BackwardTrainThread.run (backward.py:168-217): publish at start and every
    MODEL_TO_REDIS_FREQUENCY updates, checkpoint every SAVE_FREQUENCY, LOAD_CHECKPOINT restores + offsets the counter."""
    from ddrl4nav_amd.data import Experience
    from ddrl4nav_amd.server import BackwardTrainer
    c = _configs()
    cfg, cfg_nn = c["config"], c["config_nn"]
    cfg.SAVE_MODEL_PATH, cfg.SAVE_FREQUENCY, cfg.LOG_LOSS_FREQUENCY = str(tmp_path / "pong"), 5, 2
    assert cfg_nn.MODEL_TO_REDIS_FREQUENCY == 10 and cfg.UPDATE_TAG_KEY == "UPDATE_TAG"
    store, ops = {}, []

    class Pipe:
        def set(self, k, v):
            store[k] = v
            ops.append(("set", k))

        def incr(self, k):
            store[k] = store.get(k, 0) + 1

        def execute(self):
            ops.append(("exec", None))

    g3 = golden("f3_loss")
    w0 = {k: torch.from_numpy(v.copy()) for k, v in make_weights(0).items()}
    net.load_state_dict(w0)
    net.hot_path.reset_optimizer()
    net.update_time = 0
    logged = []
    tr = BackwardTrainer(net, cfg, cfg_nn, pipe=Pipe(), log=lambda k, v, t: logged.append((k, t)))
    batch = lambda: Experience(states=[g3["frames"]], advs=g3["advs"], actions=g3["actions"], old_logps=g3["old_logps"],
                               values=g3["rets"].reshape(1, -1))
    assert tr.consume(batch(), {"RewardEpisode": 1.0}) == 10
    tag = cfg.TASK_NAME + cfg.UPDATE_TAG_KEY
    assert tr.published == 2 and store[tag] == 2                 # at start + after update 10 (backward.py:179,196-197)
    assert store[net.model_key] == net.model_bytes()            # the published blob is the current arena
    assert sorted(p.name for p in tmp_path.iterdir()) == ["pong_10.pt", "pong_5.pt"] and tr.saved == 2
    assert store[cfg.TASK_NAME + cfg.TRAIN_LOCK_KEY] == 0 and tr.data_len == 64
    assert [t for k, t in logged if k == "VLoss"] == [2, 4, 6, 8, 10] and ("RewardEpisode", 64) in logged
    # resume: LOAD_CHECKPOINT restores pong_5.pt and offsets every update_time by LOAD_EPISODE
    after10 = net.hot_path.params.clone()
    cfg.LOAD_CHECKPOINT, cfg.LOAD_CHECKPOINT_PATH, cfg.LOAD_EPISODE = True, str(tmp_path / "pong_5.pt"), 5
    net.update_time = 0
    tr2 = BackwardTrainer(net, cfg, cfg_nn, pipe=Pipe())
    tr2.start()
    assert not torch.equal(net.hot_path.params, after10)          # the 5-update checkpoint is back
    sd5 = torch.load(str(tmp_path / "pong_5.pt"))
    assert torch.equal(dict(net.named_parameters())["critic.pre.linear.bias"].detach().cpu(), sd5["critic.pre.linear.bias"].cpu())
    assert tr2.consume(batch()) == 15
    assert (tmp_path / "pong_15.pt").exists() and tr2.published == 2   # start + update 10 (5 + 5)
    cfg.LOAD_CHECKPOINT = False
    net.load_state_dict(w0)


def test_learner_side_redis_chain_blobs_gather_learn_equals_the_direct_path():
    """The learner's side of the Redis drop-in END TO END: what reference env workers push --
    ``encode_backward_data`` blobs of >= 128 samples each (agent/multiqueue.py:83-105) whose frames are FLOAT64 ``u8 / 255.0``
    (warputils.py:300) -- through ``decode_backward_data`` (C codec) -> ``BackwardQueue.get(TRAINING_MIN_BATCH)`` (backward.py:48-62: eight
    ragged pieces = 1,152 samples, the ninth stays queued) -> ``BackwardTrainer.consume`` (``to_tensor`` float64 -> float32 -> device,
    backward.py:185) -> ``net.learn``: the ten yielded loss dicts and the final parameters are BIT-IDENTICAL to handing the same 1,152
    samples to ``learn`` as uint8 frames, and follow the CPU oracle's fp32 ``learn`` on them (first iteration at the single-step
    tolerance, the trajectory at 1e-3: its pinned bounds are F4 / F21's, which the direct path carries)."""
    from ddrl4nav_amd.data import EasyBytes, Experience
    from ddrl4nav_amd.runner import create_net
    from ddrl4nav_amd.server import BackwardQueue, BackwardTrainer
    c = _configs()
    cfg, cfg_nn = c["config"], c["config_nn"]
    assert cfg_nn.TRAINING_MIN_BATCH == 1024 and cfg_nn.TRAINING_ITER_TIME == 10
    cfg.LOG_LOSS_FREQUENCY, cfg.SAVE_MODELS = 1, False
    sizes = [128, 150, 131, 160, 129, 144, 128, 182, 133]                 # the first eight hold 1,152 >= 1,024; seven hold 970
    B = sum(sizes[:8])
    rng = np.random.default_rng(64)
    frames = rng.integers(0, 256, size=(sum(sizes), 4, 84, 84), dtype=np.uint8)
    frames[::3] = (frames[::3] // 32) * 32
    n_all = frames.shape[0]
    acts = rng.integers(0, 6, n_all).astype(np.float32)
    old = (np.log(1 / 6) + 0.05 * rng.normal(size=n_all)).astype(np.float32)
    advs, rets = rng.normal(size=n_all).astype(np.float32), rng.normal(size=n_all).astype(np.float32)
    w = make_weights(0)
    n = create_net(c, max_batch=B)
    keys = ("PpoTotalLoss", "ActorLoss", "VLoss", "EntLoss")

    def fresh():
        n.load_state_dict({k: torch.from_numpy(v.copy()) for k, v in w.items()})
        n.hot_path.reset_optimizer()
        n.update_time = 0

    # (a) the direct path: the same 1,152 samples as uint8 frames
    fresh()
    direct = [l for l, _, _ in n.learn(Experience(states=[frames[:B]], advs=advs[:B], actions=acts[:B], old_logps=old[:B],
                                                  values=rets[None, :B]))]
    p_direct = n.hot_path.params.clone()
    # (b) the chain: blobs -> decode -> gather -> consume -> learn
    fresh()
    eb, q = EasyBytes("127.0.0.1"), BackwardQueue()
    at = 0
    for i, m in enumerate(sizes):
        sl = slice(at, at + m)
        at += m
        xrapv = [[frames[sl] / 255.0], advs[sl], acts[sl], old[sl], rets[None, sl]]
        assert xrapv[0][0].dtype == np.float64
        q.put_blob(eb, eb.encode_backward_data(xrapv, {"RewardEpisode": float(i)}))
    logged = []
    tr = BackwardTrainer(n, cfg, cfg_nn, pipe=None, log=lambda k, v, t: logged.append((k, v, t)))
    assert tr.train_from_queue(q) == 10 and tr.data_len == B and q.q.qsize() == 1
    chain = [{k: v for k, v, t in logged if t == it and k in keys} for it in range(1, 11)]
    assert ("RewardEpisode", {"mean": 3.5}, B) in logged                 # batch_logger over the eight gathered dicts, at data_len
    for a, b in zip(direct, chain):
        assert all(a[k] == b[k] for k in keys), (a, b)
    assert torch.equal(n.hot_path.params, p_direct)
    # (c) the oracle's learn on the same samples
    threads = torch.get_num_threads()
    torch.set_num_threads(min(16, max(1, (__import__("os").cpu_count() or 1))))
    try:
        onet = O.OraclePPO()
        onet.load_weights(w)
        t = torch.from_numpy
        ora = [l for l, _, _ in O.learn(onet, onet.make_optims(), O.frames_to_f32(frames[:B]), t(acts[:B]), t(old[:B]), t(advs[:B]),
                                        t(rets[:B]), iters=10)]
    finally:
        torch.set_num_threads(threads)
    for it, (got, want) in enumerate(zip(chain, ora)):
        for k in keys:
            np.testing.assert_allclose(got[k], want[k], rtol=1e-5 if it == 0 else 1e-3, atol=1e-6 if it == 0 else 1e-5, err_msg="%s it%d" % (k, it))
    n.hot_path.close()


def test_deferred_loss_readback_yields_the_same_values(golden):
    """net.deferred_stats = True (bench.py, DEFERRED_LOSS_READBACK): all iterations enqueued, one host sync, then the yields --
    bit-identical loss dicts and parameters to the per-iteration protocol."""
    from ddrl4nav_amd.data import Experience
    import parity_util as P
    frames, actions, old_logps, advs, rets = P.mode_batch("default")
    out = []
    for deferred in (False, True):
        from ddrl4nav_amd.runner import create_net
        from ddrl4nav_amd.utils.recipe import make_weights
        net = create_net(_configs(), max_batch=64)
        net.load_state_dict({k: torch.from_numpy(v.copy()) for k, v in make_weights(0).items()})
        net.deferred_stats = deferred
        exp = Experience(states=[frames], advs=advs, actions=actions, old_logps=old_logps, values=rets.reshape(1, -1))
        rows = [(tuple(l[k] for k in ("PpoTotalLoss", "ActorLoss", "VLoss", "EntLoss")), ut, last) for l, ut, last in net.learn(exp)]
        out.append((rows, net.hot_path.params.cpu().numpy().copy()))
    assert out[0][0] == out[1][0] and [r[1] for r in out[1][0]] == list(range(1, 11))
    assert np.array_equal(out[0][1], out[1][1])


def test_gradient_buckets_cover_the_arena_and_the_overlapped_allreduce_is_the_flat_one():
    """the all-reduce runs in layer buckets, in the order the backward completes them.  The buckets' ranges
    tile [0, n_params + 8) exactly once; with a one-rank RCCL communicator (all this one-GPU box can hold) the bucketed reduction
    on the communication stream leaves the arena exactly as the flat one does, and clip + Adam on the compute stream see it."""
    from ddrl4nav_amd.dist import RcclComm
    from ddrl4nav_amd.engine import HotPath
    from ddrl4nav_amd.utils.recipe import flatten, make_weights
    import parity_util as P
    frames, actions, old_logps, advs, rets = P.mode_batch("default")
    d = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
    args = (d(frames), d(actions), d(old_logps), d(advs), d(rets))
    res = []
    for overlap in (False, True):
        h = HotPath(max_batch=64)
        h.set_params(flatten(make_weights(0)))
        try:
            h.comm = RcclComm(0, 1)
        except Exception as e:  # no librccl on this host
            h.close()
            pytest.skip("RCCL not available: %r" % (e,))
        if overlap:
            h.enable_overlap()
            cover = np.zeros(h.n_params + 8, np.int32)
            for ranges in h.grad_buckets():
                for off, cnt in ranges:
                    cover[off:off + cnt] += 1
            assert (cover == 1).all()
            assert sum(c for o, c in h.grad_buckets()[2]) >= 0.9 * h.n_params     # the dense layer's bucket: ready before conv3 / conv2
        for _ in range(3):
            h.ppo_iter(*args)
            h.allreduce_grads()
            h.clip_adam_step()
        res.append((h.params.cpu().numpy().copy(), h.stats()))
        h.close()
    assert np.array_equal(res[0][0], res[1][0]) and res[0][1] == res[1][1]
