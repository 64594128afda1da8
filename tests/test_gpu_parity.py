"""GPU parity tests: the HIP path (through the C ABI) against the CPU oracle and the golden
vectors generated from the reference.  Run with `-m gpu` on an MI355X.

Stated tolerances (fp32 path; summation order differs from torch-CPU):
  forward probs/value/logp     rtol 1e-5, atol 1e-6
  encoder activations          rtol 1e-5, atol 2e-6
  GAE                          bit-exact
  losses                       rtol 1e-5, atol 1e-6
  gradients                    |d| <= 2e-5 * max|g| per tensor  (+ cosine > 1 - 1e-9)
  parameters after k Adam steps  |d| <= 0.02 * lr * k  (Adam's first steps are sign-like:
                               an element whose gradient is at the fp32 summation-noise
                               floor moves by O(lr) either way -- see DESIGN.md)
"""
import numpy as np
import pytest
import torch

from ddrl4nav_amd.utils.recipe import flatten, make_weights, param_specs, sample_uniform
from oracle import ddrl_oracle as O
import parity_util as P

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def hp():
    from ddrl4nav_amd.engine import HotPath
    h = HotPath(max_batch=512).keep_activations()  # tests read a1 / a2 after acting forwards too
    h.set_params(flatten(make_weights(0)))
    yield h
    h.close()


@pytest.fixture(scope="module")
def onet():
    torch.set_num_threads(max(1, torch.get_num_threads()))
    n = O.OraclePPO()
    n.load_weights(make_weights(0))
    return n


def dev(a, dtype=None):
    t = torch.from_numpy(np.ascontiguousarray(a))
    if dtype is not None:
        t = t.to(dtype)
    return t.cuda()


def test_u8_table_matches_reference_division(hp, golden):
    assert np.array_equal(hp.u8_table().cpu().numpy(), golden("f5_u8_lut")["lut"])


def test_forward_golden_f1(hp, golden):
    g = golden("f1_forward")
    frames = dev(g["frames"])
    probs, value, _, logp = hp.forward(frames, act=dev(g["actions"]))
    np.testing.assert_allclose(probs.cpu().numpy(), g["probs"], rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(value.cpu().numpy(), g["value"], rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(logp.cpu().numpy(), g["logp"], rtol=1e-5, atol=1e-6)
    ha, hc = hp.last_features(8)
    np.testing.assert_allclose(ha.cpu().numpy()[:, :16], g["h_actor"], rtol=1e-5, atol=2e-6)
    np.testing.assert_allclose(hc.cpu().numpy()[:, :16], g["h_critic"], rtol=1e-5, atol=2e-6)
    p_hat, logits, ent = hp.categorical_stats(probs)
    np.testing.assert_allclose(p_hat.cpu().numpy(), g["p_hat"], rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(logits.cpu().numpy(), g["logits"], rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(ent.cpu().numpy(), g["entropy"], rtol=1e-5, atol=1e-6)


@pytest.mark.parametrize("n", [1, 5, 33, 300])
def test_forward_and_activations_vs_oracle(hp, onet, n):
    rng = np.random.default_rng(100 + n)
    frames = rng.integers(0, 256, size=(n, 4, 84, 84), dtype=np.uint8)
    if n > 4:
        frames[3] = 0
        frames[4] = 255
    acts = rng.integers(0, 6, size=n).astype(np.float32)
    probs, value, _, logp = hp.forward(dev(frames), act=dev(acts))
    x = O.frames_to_f32(frames)
    with torch.no_grad():
        oprobs, _, ologits, ov = onet(x)
        ologp = O.categorical_log_prob(ologits, torch.from_numpy(acts))
        enc = onet.actor.pre
        a1 = torch.nn.functional.leaky_relu(enc.conv1(x))
        a2 = torch.nn.functional.leaky_relu(enc.conv2(a1))
        a3 = torch.nn.functional.leaky_relu(enc.conv3(a2))
        encc = onet.critic.pre
        c1 = torch.nn.functional.leaky_relu(encc.conv1(x))
    np.testing.assert_allclose(hp.debug_buffer(0, (32, 20, 20), n, 0).cpu().numpy(), a1.numpy(), rtol=1e-5, atol=2e-6)
    np.testing.assert_allclose(hp.debug_buffer(0, (32, 20, 20), n, 1).cpu().numpy(), c1.numpy(), rtol=1e-5, atol=2e-6)
    np.testing.assert_allclose(hp.debug_buffer(1, (64, 9, 9), n, 0).cpu().numpy(), a2.numpy(), rtol=1e-5, atol=2e-6)
    np.testing.assert_allclose(hp.debug_buffer(2, (64, 7, 7), n, 0).cpu().numpy(), a3.numpy(), rtol=1e-5, atol=2e-6)
    np.testing.assert_allclose(probs.cpu().numpy(), oprobs.numpy(), rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(value.cpu().numpy(), ov.numpy()[:, 0], rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(logp.cpu().numpy(), ologp.numpy(), rtol=1e-5, atol=1e-6)


def test_forward_is_batch_independent(hp):
    rng = np.random.default_rng(5)
    frames = rng.integers(0, 256, size=(40, 4, 84, 84), dtype=np.uint8)
    acts = dev(rng.integers(0, 6, size=40).astype(np.float32))
    p_all, v_all, _, l_all = [t.clone() for t in hp.forward(dev(frames), act=acts)]
    perm = rng.permutation(40)
    p_p, v_p, _, l_p = hp.forward(dev(frames[perm]), act=acts[torch.from_numpy(perm).cuda()].contiguous())
    assert torch.equal(p_all[perm], p_p) and torch.equal(v_all[perm], v_p) and torch.equal(l_all[perm], l_p)


def test_sampler_inverse_cdf_contract(hp):
    rng = np.random.default_rng(11)
    n = 257
    frames = dev(rng.integers(0, 256, size=(n, 4, 84, 84), dtype=np.uint8))
    probs, value, action, logp = hp.forward(frames, act=None, seed=1234, stream_id=77)
    p_hat, logits, _ = hp.categorical_stats(probs)
    u = sample_uniform(1234, 77, n)
    want = O.inverse_cdf_sample(p_hat.cpu().numpy(), u)
    got = action.cpu().numpy()
    assert np.array_equal(got, want.astype(np.float32))
    np.testing.assert_allclose(logp.cpu().numpy(), logits.cpu().numpy()[np.arange(n), want], rtol=1e-6, atol=1e-7)
    # statistical check of the stream itself: many draws from one distribution
    one = frames[:1].expand(200, -1, -1, -1).contiguous()
    counts = np.zeros(6)
    for s in range(20):
        _, _, a, _ = hp.forward(one, act=None, seed=5, stream_id=s)
        counts += np.bincount(a.cpu().numpy().astype(int), minlength=6)
    p = p_hat.cpu().numpy()[0]
    expected = p * counts.sum()
    chi2 = ((counts - expected) ** 2 / np.maximum(expected, 1e-9)).sum()
    assert chi2 < 40.0, (counts, expected)


def test_gae_golden_f2_bit_exact(hp, golden):
    g = golden("f2_gae")
    T = 256
    for lo, ka, kr in ((0, "adv1", "ret1"), (T, "adv2", "ret2")):
        adv, ret = hp.gae(dev(g["values"][lo:lo + T + 1]), dev(g["rewards"][lo:lo + T]), dev(g["dones"][lo:lo + T]))
        assert np.array_equal(adv.cpu().numpy(), g[ka])
        assert np.array_equal(ret.cpu().numpy(), g[kr])


@pytest.mark.parametrize("n", [1, 3, 130, 257, 512])
def test_fused_acting_forward(onet, n):
    """ddrl_forward of at most 512 samples runs conv1-conv3 in one kernel that keeps a1 / a2 on chip (csrc/act.hip), one workgroup
    per (sample, encoder).  (a) The variant that also stores a1 / a2 (ddrl_debug_keep_activations) gives bit-identical outputs;
    (b) without it ddrl_debug_buffer refuses a1 / a2 (nothing of this call is there) and still serves a3 / h; (c) the same samples
    inside a forward of more than 512 (batch-tiled kernels, batch-wide plane scales) agree to fp32 rounding; (d) a3 against the
    oracle."""
    from ddrl4nav_amd.engine import HotPath
    h = HotPath(max_batch=1024)
    try:
        h.set_params(flatten(make_weights(0)))
        rng = np.random.default_rng(900 + n)
        frames = rng.integers(0, 256, size=(n, 4, 84, 84), dtype=np.uint8)
        frames[0] = 0
        if n > 1:
            frames[1] = 255
        acts = rng.integers(0, 6, size=n).astype(np.float32)
        fd, ad = dev(frames), dev(acts)
        p0, v0, _, l0 = (t.clone() for t in h.forward(fd, act=ad))
        with pytest.raises(RuntimeError):
            h.debug_buffer(0, (32, 20, 20), n, 0)
        with pytest.raises(RuntimeError):
            h.debug_buffer(1, (64, 9, 9), n, 1)
        a3 = h.debug_buffer(2, (64, 7, 7), n, 0).cpu().numpy()
        h.keep_activations(True)
        p1, v1, _, l1 = h.forward(fd, act=ad)
        assert torch.equal(p0, p1) and torch.equal(v0, v1) and torch.equal(l0, l1)
        a1k = h.debug_buffer(0, (32, 20, 20), n, 1).cpu().numpy()
        a1a = h.debug_buffer(0, (32, 20, 20), n, 0).cpu().numpy()
        a2k = [h.debug_buffer(1, (64, 9, 9), n, e).cpu().numpy() for e in (0, 1)]
        a3c = h.debug_buffer(2, (64, 7, 7), n, 1).cpu().numpy()
        h.keep_activations(False)
        x = O.frames_to_f32(frames)
        with torch.no_grad():
            oprobs, _, _, ov = onet(x)
            enc = onet.actor.pre
            r3 = torch.nn.functional.leaky_relu(enc.conv3(torch.nn.functional.leaky_relu(enc.conv2(torch.nn.functional.leaky_relu(enc.conv1(x))))))
            lr = torch.nn.functional.leaky_relu
            c1 = lr(onet.critic.pre.conv1(x))
            r1 = lr(enc.conv1(x))
            r2, c2 = lr(enc.conv2(r1)), lr(onet.critic.pre.conv2(c1))
            c3 = lr(onet.critic.pre.conv3(c2))
        np.testing.assert_allclose(a3, r3.numpy(), rtol=1e-5, atol=2e-6)
        np.testing.assert_allclose(a1k, c1.numpy(), rtol=1e-5, atol=2e-6)
        # every stage of both encoders (round 6: the kernel's tiles, K halves and hand-over buffers are per stage)
        np.testing.assert_allclose(a1a, r1.numpy(), rtol=1e-5, atol=2e-6)
        np.testing.assert_allclose(a2k[0], r2.numpy(), rtol=1e-5, atol=2e-6)
        np.testing.assert_allclose(a2k[1], c2.numpy(), rtol=1e-5, atol=2e-6)
        np.testing.assert_allclose(a3c, c3.numpy(), rtol=1e-5, atol=2e-6)
        np.testing.assert_allclose(p0.cpu().numpy(), oprobs.numpy(), rtol=1e-5, atol=1e-6)
        np.testing.assert_allclose(v0.cpu().numpy(), ov.numpy().reshape(-1), rtol=1e-5, atol=2e-6)
        # (c): the batch-tiled kernels on the same samples (padded to 600 with copies)
        reps = -(-600 // n)
        bigf = dev(np.concatenate([frames] * reps)[:600].copy() if n < 600 else frames)
        biga = dev(np.concatenate([acts] * reps)[:600].copy())
        pb, vb, _, lb = h.forward(bigf, act=biga)
        m = min(n, 600)
        np.testing.assert_allclose(pb[:m].cpu().numpy(), p0[:m].cpu().numpy(), rtol=2e-6, atol=1e-7)
        np.testing.assert_allclose(vb[:m].cpu().numpy(), v0[:m].cpu().numpy(), rtol=1e-5, atol=2e-6)
        np.testing.assert_allclose(lb[:m].cpu().numpy(), l0[:m].cpu().numpy(), rtol=1e-5, atol=2e-6)
    finally:
        h.close()


@pytest.mark.parametrize("T,N", [(256, 256), (1, 3), (7, 65), (256, 2048)])
def test_gae_vs_oracle_bit_exact(hp, T, N):
    rng = np.random.default_rng(T * 1000 + N)
    v = rng.normal(0, 2, size=(T + 1, N)).astype(np.float32)
    r = rng.choice(np.array([-1, 0, 1], np.float32), size=(T, N)).astype(np.float32)
    d = (rng.random((T, N)) < 0.05).astype(np.uint8)
    adv, ret = hp.gae(dev(v), dev(r), dev(d))
    oadv, oret = O.gae(v, r, d)
    assert np.array_equal(adv.cpu().numpy(), oadv) and np.array_equal(ret.cpu().numpy(), oret)


def test_gae_all_done_property(hp):
    # every step terminal -> advantage = r - V, return = r   (size-independent property)
    T, N = 256, 4096
    rng = np.random.default_rng(9)
    v = dev(rng.normal(size=(T + 1, N)).astype(np.float32))
    r = dev(rng.normal(size=(T, N)).astype(np.float32))
    d = torch.ones((T, N), dtype=torch.uint8, device="cuda")
    adv, ret = hp.gae(v, r, d)
    assert torch.equal(adv, (0.0 - v[:T]) + r)
    assert torch.equal(ret, v[:T] + adv)


def _load_batch(golden):
    g = golden("f3_loss")
    return g, dev(g["frames"]), dev(g["actions"]), dev(g["old_logps"]), dev(g["advs"]), dev(g["rets"])


def _grad_views(hp):
    flat = hp.grads[:hp.n_params].cpu().numpy()
    out, off = {}, 0
    for name, shape, _ in param_specs():
        n = int(np.prod(shape))
        out[name] = flat[off:off + n].reshape(shape)
        off += n
    return out


def test_loss_and_gradients_golden_f3(hp, onet, golden):
    g, frames, actions, old_logps, advs, rets = _load_batch(golden)
    hp.set_params(flatten(make_weights(0)))
    hp.ppo_iter(frames, actions, old_logps, advs, rets)
    tail = hp.grads[hp.n_params:hp.n_params + 3].cpu().numpy()
    np.testing.assert_allclose(tail[0], g["actor_loss"], rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(tail[1], g["v_loss"], rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(tail[2], g["ent"], rtol=1e-5, atol=1e-6)
    # d(loss)/d(value) straight from the reference's autograd
    dval = hp.debug_buffer(9, (), 64, 0).cpu().numpy()
    np.testing.assert_allclose(dval, g["dvalue"], rtol=1e-5, atol=1e-8)
    # full gradient vs oracle autograd (oracle == reference bit-for-bit, test_oracle_golden.py)
    onet.load_weights(make_weights(0))
    onet.zero_grad()
    x = O.frames_to_f32(g["frames"])
    t = lambda k: torch.from_numpy(g[k])
    _, al, vl, _ = O.ppo_losses(onet, x, t("actions"), t("old_logps"), t("advs"), t("rets"))
    al.backward()
    vl.backward()
    got = _grad_views(hp)
    for name, p in onet.named_parameters():
        want = p.grad.numpy()
        scale = np.abs(want).max()
        assert scale > 0
        err = np.abs(got[name] - want).max()
        assert err <= 2e-5 * scale, (name, err, scale)
        cos = (got[name].astype(np.float64) * want).sum() / (
            np.linalg.norm(got[name].astype(np.float64)) * np.linalg.norm(want.astype(np.float64)))
        assert cos > 1 - 1e-9, (name, cos)
        # golden (reference) spot values
        np.testing.assert_allclose(got[name].reshape(-1)[:64], g["ghead/" + name], rtol=0, atol=2e-5 * scale)


def test_gradient_is_additive_over_shards(hp, golden):
    """Data-parallel property: grads(full batch) == grads(shard 0) + grads(shard 1) when every
    shard scales by 1/B_global -- exactly what the RCCL all-reduce sums."""
    g, frames, actions, old_logps, advs, rets = _load_batch(golden)
    hp.set_params(flatten(make_weights(0)))
    hp.ppo_iter(frames, actions, old_logps, advs, rets)
    full = hp.grads.clone()
    acc = torch.zeros_like(full)
    for sl in (slice(0, 40), slice(40, 64)):
        hp.ppo_iter(frames[sl].contiguous(), actions[sl].contiguous(), old_logps[sl].contiguous(),
                    advs[sl].contiguous(), rets[sl].contiguous(), b_global=64)
        acc += hp.grads
    n = hp.n_params
    scale = full[:n].abs().max().item()
    assert (acc[:n] - full[:n]).abs().max().item() <= 2e-6 * scale
    np.testing.assert_allclose(acc[n:n + 3].cpu().numpy(), full[n:n + 3].cpu().numpy(), rtol=1e-5, atol=1e-7)


def test_full_size_properties_B65536():
    """BASELINE config 2 size (256 envs x T=256 = 65,536 samples): size-independent properties.
    (a) data-parallel additivity: grads(B) == grads(shard 0) + grads(shard 1) with 1/B_global;
    (b) a sample's forward does not depend on its position in the batch;
    (c) duplicating the batch leaves the mean losses and the mean gradient unchanged."""
    from ddrl4nav_amd.engine import HotPath
    B = 65536
    big = HotPath(max_batch=B)
    big.set_params(flatten(make_weights(0)))
    g = torch.Generator(device="cuda")
    g.manual_seed(77)
    base = torch.randint(0, 256, (B // 2, 4, 84, 84), dtype=torch.uint8, device="cuda", generator=g)
    frames = torch.cat([base, base])  # second half duplicates the first
    half = lambda t: torch.cat([t, t]).contiguous()
    acts = half(torch.randint(0, 6, (B // 2,), device="cuda", generator=g).float())
    old = half(torch.full((B // 2,), -1.79, device="cuda") + 0.2 * torch.randn(B // 2, device="cuda", generator=g))
    adv = half(torch.randn(B // 2, device="cuda", generator=g))
    ret = half(torch.randn(B // 2, device="cuda", generator=g))
    n = big.n_params
    big.ppo_iter(frames, acts, old, adv, ret)
    full = big.grads.clone()
    acc = torch.zeros_like(full)
    cut = 40000  # uneven shards
    for sl in (slice(0, cut), slice(cut, B)):
        big.ppo_iter(frames[sl], acts[sl], old[sl], adv[sl], ret[sl], b_global=B)
        acc += big.grads
    scale = full[:n].abs().max().item()
    # bound 2e-5 max|g| (the B = 64 twin above: 2e-6): three evaluations of 65,536-sample sums in different groupings; the achieved
    # figure goes on record (gpurun_out/margins_measured.json, "full_size")
    add_err = (acc[:n] - full[:n]).abs().max().item() / scale
    assert add_err <= 2e-5, add_err
    np.testing.assert_allclose(acc[n:n + 3].cpu().numpy(), full[n:n + 3].cpu().numpy(), rtol=1e-4, atol=1e-7)
    # (c): the duplicated batch has the same mean gradient as one copy
    big.ppo_iter(frames[:B // 2], acts[:B // 2], old[:B // 2], adv[:B // 2], ret[:B // 2])
    one = big.grads.clone()
    dup_err = (one[:n] - full[:n]).abs().max().item() / scale
    assert dup_err <= 2e-5, dup_err
    import parity_util as P
    P.MARGINS.measured.setdefault("full_size", {}).update(additivity_over_2e5=add_err / 2e-5, duplication_over_2e5=dup_err / 2e-5)
    P.MARGINS._flush()
    print("B = 65,536: additivity error %.2e max|g|, duplication %.2e" % (add_err, dup_err))
    np.testing.assert_allclose(one[n:n + 3].cpu().numpy(), full[n:n + 3].cpu().numpy(), rtol=1e-4, atol=1e-7)
    # (b): two copies of the same samples inside ONE forward come out bit-identical although they sit in different tiles / rows --
    # 2 x 150 samples run the fused acting kernel (csrc/act.hip) + the split dense layer, 2 x 300 the batch-tiled kernels
    for half in (150, 300):
        both = torch.cat([frames[:half], frames[:half]]).contiguous()
        a2x = torch.cat([acts[:half], acts[:half]]).contiguous()
        probs, value, _, logp = big.forward(both, act=a2x)
        assert torch.equal(probs[half:], probs[:half]) and torch.equal(value[half:], value[:half]) and torch.equal(logp[half:], logp[:half])
    big.close()


def test_learn_sequence_golden_f4(hp, golden):
    """Ten PPO iterations against the reference's stored trajectory (F4).  Bounds that can fail, in the
    reference's own currency (tests/parity_util.py): losses within single-step tolerance + c_loss x the
    reference's own loss spread (float64 / 8-thread fp32 runs); every parameter tensor within c x the deviation
    that the reference's own fp32 evaluations (1 / 8 threads, three batch orders) show from its float64 run --
    L2, max-abs and direction of the accumulated update, over ALL elements; fixed limits: tests/parity_util.py PARAM_LIMIT / LOSS_LIMIT."""
    import parity_util as P
    g, frames, actions, old_logps, advs, rets = _load_batch(golden)
    g4 = golden("f4_learn")
    hp.set_params(flatten(make_weights(0)))
    hp.reset_optimizer()
    ref = g4["losses"]
    env = P.mode_loss_envelope("default", ref, g4["losses_f64"], g4["losses_f32t8"])

    def step():
        hp.ppo_iter(frames, actions, old_logps, advs, rets)
        hp.clip_adam_step()
        s = hp.stats()
        return [s["PpoTotalLoss"], s["ActorLoss"], s["VLoss"], s["EntLoss"]]

    P.check_sequence("learn_f4", "default", step, lambda: hp.params.cpu().numpy(), ref, env)
    assert hp.step == 10


def test_clip_coefficient_and_norm(hp, onet, golden):
    g, frames, actions, old_logps, advs, rets = _load_batch(golden)
    hp.set_params(flatten(make_weights(0)))
    hp.reset_optimizer()
    hp.ppo_iter(frames, actions, old_logps, advs, rets)
    gn = float(torch.linalg.vector_norm(hp.grads[:hp.n_params].double()).item())
    hp.clip_adam_step()
    s = hp.stats()
    np.testing.assert_allclose(s["GradNorm"], gn, rtol=1e-6)
    np.testing.assert_allclose(s["ClipCoef"], min(1.0, 0.5 / (gn + 1e-6)), rtol=1e-6)


@pytest.mark.parametrize("launch", ["acting", "training"])
@pytest.mark.parametrize("kind", ["uniform", "pong_wide"])
def test_conv1_forward_is_at_least_fp32_accurate(hp, kind, launch):
    """The conv1 forward runs on the 16-bit matrix pipe (pixels 0..255, exact in fp16, x two scaled fp16 planes of the fp32 weights,
    fp32 accumulation).  That is not a precision trade: against a float64 evaluation of the reference arithmetic its
    error must not exceed that of torch's own fp32 convolution (measured: about half of it)."""
    n = 96
    frames, acts, old, adv, ret = _inputs(n, 31, kind)
    w = make_weights(0)
    hp.set_params(flatten(w))
    if launch == "acting":  # conv1 inside the fused kernel of csrc/act.hip (pixel tiles as the MFMA's rows)
        probs = torch.empty((n, 6), device="cuda")
        val, act, lp = (torch.empty(n, device="cuda") for _ in range(3))
        hp.forward(dev(frames), seed=1, stream_id=0, probs=probs, value=val, action=act, logp=lp)
    else:                   # conv_fwd1_planes_kernel (both encoders' output channels as the rows)
        hp.ppo_iter(dev(frames), dev(acts), dev(old), dev(adv), dev(ret))
    for enc, pre in ((0, "actor.pre"), (1, "critic.pre")):
        a1 = hp.debug_buffer(0, (32, 20, 20), n, enc).cpu().numpy().astype(np.float64)
        W, b = torch.from_numpy(w[pre + ".conv1.weight"]), torch.from_numpy(w[pre + ".conv1.bias"])
        x64 = torch.from_numpy(frames.astype(np.float64) / 255.0)
        ref = torch.nn.functional.leaky_relu(torch.nn.functional.conv2d(x64, W.double(), b.double(), stride=4), 0.01).numpy()
        x32 = O.frames_to_f32(frames)
        t32 = torch.nn.functional.leaky_relu(torch.nn.functional.conv2d(x32, W, b, stride=4), 0.01).numpy().astype(np.float64)
        err_kernel, err_torch32 = np.abs(a1 - ref).max(), np.abs(t32 - ref).max()
        assert err_kernel <= 1.25 * err_torch32 + 1e-9, (pre, err_kernel, err_torch32)
        assert np.abs(a1 - ref).mean() <= 1.25 * np.abs(t32 - ref).mean() + 1e-12, pre


@pytest.mark.parametrize("kind", ["uniform", "pong_wide"])
def test_conv1_weight_gradient_is_at_least_fp32_accurate(hp, kind):
    """Same statement for the conv1 weight gradient (dz1 split into two scaled fp16 planes, per-sample power-of-two scales, x exact-fp16 pixels): given
    the kernel's own dz1 and the frames, its dW1 must be as close to the float64 sum as an fp32 evaluation is."""
    n = 64
    frames = _inputs(n, 32, kind)[0]
    _bwd_setup(hp, n, 32, kind)
    got = _grad_views(hp)
    x64 = torch.from_numpy(frames.astype(np.float64) / 255.0)
    cols64 = torch.nn.functional.unfold(x64, kernel_size=8, stride=4)          # [n, 256, 400]
    cols32 = torch.nn.functional.unfold(O.frames_to_f32(frames), kernel_size=8, stride=4)
    for enc, pre in ((0, "actor.pre"), (1, "critic.pre")):
        a1 = hp.debug_buffer(0, (32, 20, 20), n, enc).cpu()
        da1 = hp.debug_buffer(4, (32, 20, 20), n, enc).cpu()
        dz = torch.where(a1 > 0, da1, da1 * np.float32(0.01)).reshape(n, 32, 400)   # the kernel's fp32 mask arithmetic
        ref = torch.einsum("bop,btp->ot", dz.double(), cols64).numpy().reshape(32, 4, 8, 8)
        f32 = torch.einsum("bop,btp->ot", dz, cols32).numpy().astype(np.float64).reshape(32, 4, 8, 8)
        k = got[pre + ".conv1.weight"].astype(np.float64)
        err_kernel, err_f32 = np.abs(k - ref).max(), np.abs(f32 - ref).max()
        assert err_kernel <= 1.5 * err_f32 + 1e-12, (pre, err_kernel, err_f32)


@pytest.mark.parametrize("kind", ["uniform", "pong_wide"])
def test_dense_forward_is_at_least_fp32_accurate(hp, kind):
    """The dense layer's forward in a training launch splits BOTH operands into two scaled fp16 planes and sums three plane
    products in fp32 (fc_fwd_planes_kernel).  Given the kernel's own a3, the error of h against the float64 product must
    stay within a few rounding units (2^-24) of sum_k |a_k w_k| (stated limit: 8 units, tests/parity_util.py ACCURACY_CAP) -- far below the n * eps bound of a sequential fp32
    chain over K = 3,136 (torch's blocked CPU matmul, whose error is reported alongside, is closer still)."""
    n = 200
    w = _bwd_setup(hp, n, 33, kind)
    for enc, pre in ((0, "actor.pre"), (1, "critic.pre")):
        a3 = hp.debug_buffer(2, (3136,), n, enc).cpu()
        h = hp.debug_buffer(3, (512,), n, enc).cpu().numpy().astype(np.float64)
        W, b = torch.from_numpy(w[pre + ".linear.weight"]), torch.from_numpy(w[pre + ".linear.bias"])
        ref = (a3.double() @ W.double().T + b.double()).numpy()
        f32 = (a3 @ W.T + b).numpy().astype(np.float64)
        err_kernel, err_f32 = np.abs(h - ref).max(), np.abs(f32 - ref).max()
        mass = float((a3.double().abs() @ W.double().abs().T).max())  # largest sum_k |a_k w_k|
        P.MARGINS.check("accuracy", "dense_fwd_units", err_kernel / (2.0 ** -24 * mass), "(%s: kernel %.3e, fp32 %.3e)" % (pre, err_kernel, err_f32))


@pytest.mark.parametrize("kind", ["uniform", "pong_wide"])
def test_conv2_forward_is_at_least_fp32_accurate(hp, kind):
    """conv2's forward in a training launch is a plane-product kernel too (conv_fwd2_planes_kernel: weights pre-split, a1 split
    while staged, one MFMA k-group = the 16 taps of one input channel).  Given the kernel's own a1, a2 must be as close
    to the float64 convolution as torch's fp32 convolution is, and within a few rounding units of sum |a w| (fixed limits: parity_util.ACCURACY_CAP, VS_TORCH_LIMIT)."""
    n = 100  # 34 tiles of 3 samples, the last one holds a single sample
    w = _bwd_setup(hp, n, 34, kind)
    for enc, pre in ((0, "actor.pre"), (1, "critic.pre")):
        a1 = hp.debug_buffer(0, (32, 20, 20), n, enc).cpu()
        a2 = hp.debug_buffer(1, (64, 9, 9), n, enc).cpu().numpy().astype(np.float64)
        W, b = torch.from_numpy(w[pre + ".conv2.weight"]), torch.from_numpy(w[pre + ".conv2.bias"])
        z64 = torch.nn.functional.conv2d(a1.double(), W.double(), b.double(), stride=2)
        ref = torch.nn.functional.leaky_relu(z64, 0.01).numpy()
        f32 = torch.nn.functional.leaky_relu(torch.nn.functional.conv2d(a1, W, b, stride=2), 0.01).numpy().astype(np.float64)
        err_kernel, err_f32 = np.abs(a2 - ref).max(), np.abs(f32 - ref).max()
        mass = float(torch.nn.functional.conv2d(a1.double().abs(), W.double().abs(), stride=2).max())
        P.MARGINS.check("accuracy", "conv2_fwd_units", err_kernel / (2.0 ** -24 * mass), "(%s: kernel %.3e, fp32 %.3e)" % (pre, err_kernel, err_f32))
        assert np.abs(a2 - ref).mean() <= 1.5 * np.abs(f32 - ref).mean() + 1e-12, (pre, np.abs(a2 - ref).mean(), np.abs(f32 - ref).mean())
        print(pre, "conv2 fwd err", err_kernel, "torch f32", err_f32, "units of mass", err_kernel / (2.0 ** -24 * mass))


@pytest.mark.parametrize("kind", ["uniform", "pong_wide"])
def test_conv3_forward_is_at_least_fp32_accurate(hp, kind):
    """Same statement for conv3's training-launch forward (conv_fwd3_planes_kernel: a2 staged channel-innermost, one
    MFMA k-group = two taps x eight channels, the tenth tap padded with zero weights)."""
    n = 101  # 21 tiles of 5 samples, the last one holds a single sample
    w = _bwd_setup(hp, n, 35, kind)
    for enc, pre in ((0, "actor.pre"), (1, "critic.pre")):
        a2 = hp.debug_buffer(1, (64, 9, 9), n, enc).cpu()
        a3 = hp.debug_buffer(2, (64, 7, 7), n, enc).cpu().numpy().astype(np.float64)
        W, b = torch.from_numpy(w[pre + ".conv3.weight"]), torch.from_numpy(w[pre + ".conv3.bias"])
        ref = torch.nn.functional.leaky_relu(torch.nn.functional.conv2d(a2.double(), W.double(), b.double()), 0.01).numpy()
        f32 = torch.nn.functional.leaky_relu(torch.nn.functional.conv2d(a2, W, b), 0.01).numpy().astype(np.float64)
        err_kernel, err_f32 = np.abs(a3 - ref).max(), np.abs(f32 - ref).max()
        mass = float(torch.nn.functional.conv2d(a2.double().abs(), W.double().abs()).max())
        P.MARGINS.check("accuracy", "conv3_fwd_units", err_kernel / (2.0 ** -24 * mass), "(%s: kernel %.3e, fp32 %.3e)" % (pre, err_kernel, err_f32))
        assert np.abs(a3 - ref).mean() <= 1.5 * np.abs(f32 - ref).mean() + 1e-12, (pre, np.abs(a3 - ref).mean(), np.abs(f32 - ref).mean())


KINDS = ["uniform", "pong_wide"]


def _inputs(n, seed, kind="uniform"):
    """(frames u8, actions, old_logps, advs, rets).  "uniform": dense random bytes, N(0,1) advantages (the worst case for time).
    "pong_wide": mostly flat Pong frames (utils.recipe.pong_frames) and advantages log-uniform over 1e-4 .. 10 with one sample
    x 1000 -- the realistic input and a dynamic range of eight decades across the samples."""
    rng = np.random.default_rng(seed)
    acts = rng.integers(0, 6, size=n).astype(np.float32)
    old = np.full(n, -1.79, dtype=np.float32)
    if kind == "uniform":
        frames = rng.integers(0, 256, size=(n, 4, 84, 84), dtype=np.uint8)
        adv, ret = rng.normal(size=n).astype(np.float32), rng.normal(size=n).astype(np.float32)
    else:
        from ddrl4nav_amd.utils.recipe import pong_frames
        frames = pong_frames(seed, n)
        adv = (rng.choice(np.array([-1.0, 1.0]), size=n) * 10.0 ** rng.uniform(-4.0, 1.0, size=n)).astype(np.float32)
        adv[n // 2] *= np.float32(1000.0)
        ret = (adv + rng.normal(size=n).astype(np.float32) * np.float32(0.1)).astype(np.float32)
    return frames, acts, old, adv, ret


def _bwd_setup(hp, n, seed, kind="uniform"):
    frames, acts, old, adv, ret = _inputs(n, seed, kind)
    w = make_weights(0)
    hp.set_params(flatten(w))
    hp.ppo_iter(dev(frames), dev(acts), dev(old), dev(adv), dev(ret))
    return w


def _leaky_mask(a, g):
    """the kernels' decision: slope 1 where the stored activation is positive, 0.01 elsewhere"""
    return torch.where(a > 0, g, g * 0.01)


@pytest.mark.parametrize("kind", ["uniform", "pong_wide"])
def test_dense_data_gradient_is_at_least_fp32_accurate(hp, kind):
    """fc_dgrad_planes_kernel (dh and the transposed weight planes as two scaled fp16 planes each, three plane products,
    leaky mask in the epilogue): given the kernel's own dh and a3, dz3 against the float64 product, in rounding units of
    sum_k |dh_k w_k| and beside torch's fp32 matmul."""
    n = 200
    w = _bwd_setup(hp, n, 41, kind)
    for enc, pre in ((0, "actor.pre"), (1, "critic.pre")):
        dh = hp.debug_buffer(7, (512,), n, enc).cpu()
        a3 = hp.debug_buffer(2, (3136,), n, enc).cpu()
        dz3 = hp.debug_buffer(6, (3136,), n, enc).cpu().numpy().astype(np.float64)
        W = torch.from_numpy(w[pre + ".linear.weight"])
        ref = _leaky_mask(a3.double(), dh.double() @ W.double()).numpy()
        f32 = _leaky_mask(a3, dh @ W).numpy().astype(np.float64)
        err_kernel, err_f32 = np.abs(dz3 - ref).max(), np.abs(f32 - ref).max()
        mass = float((dh.double().abs() @ W.double().abs()).max())
        P.MARGINS.check("accuracy", "dense_dgrad_units", err_kernel / (2.0 ** -24 * mass), "(%s: kernel %.3e, fp32 %.3e)" % (pre, err_kernel, err_f32))
        # mean error beside torch's fp32 operator (an f16x3 product keeps ~2^-22 of every term; torch's blocked fp32 sums ~2^-24)
        P.MARGINS.check("accuracy", "dense_dgrad_mean_vs_torch_fp32", np.abs(dz3 - ref).mean() / np.abs(f32 - ref).mean(), "(%s)" % pre)


@pytest.mark.parametrize("kind", ["uniform", "pong_wide"])
def test_dense_weight_gradient_is_at_least_fp32_accurate(hp, kind):
    """fc_wgrad_planes_kernel (dh and a3 split into two scaled fp16 planes while staged, fragments through the transposing LDS
    read, six plane products, split over the batch and summed in fixed order): given the kernel's own dh and a3,
    dW = dh^T a3 and db = column sums of dh against float64, beside torch's fp32 matmul.  n = 333 leaves a ragged last
    k-block and k-tile 24 is the half-empty one (3136 = 24.5 x 128)."""
    n = 333
    _bwd_setup(hp, n, 44, kind)
    got = _grad_views(hp)
    for enc, pre in ((0, "actor.pre"), (1, "critic.pre")):
        dh = hp.debug_buffer(7, (512,), n, enc).cpu()
        a3 = hp.debug_buffer(2, (3136,), n, enc).cpu()
        ref = (dh.double().T @ a3.double()).numpy()
        f32 = (dh.T @ a3).numpy().astype(np.float64)
        k = got[pre + ".linear.weight"].astype(np.float64)
        err_kernel, err_f32 = np.abs(k - ref).max(), np.abs(f32 - ref).max()
        mass = float((dh.double().abs().T @ a3.double().abs()).max())
        P.MARGINS.check("accuracy", "dense_wgrad_units", err_kernel / (2.0 ** -24 * mass), "(%s: kernel %.3e, fp32 %.3e)" % (pre, err_kernel, err_f32))
        P.MARGINS.check("accuracy", "dense_wgrad_mean_vs_torch_fp32", np.abs(k - ref).mean() / np.abs(f32 - ref).mean(), "(%s)" % pre)
        db = got[pre + ".linear.bias"].astype(np.float64)
        np.testing.assert_allclose(db, dh.double().sum(0).numpy(), rtol=0, atol=4 * 2.0 ** -24 * float(dh.double().abs().sum(0).max()))


@pytest.mark.parametrize("n", [101, 6])
@pytest.mark.parametrize("kind", ["uniform", "pong_wide"])
def test_conv3_weight_gradient_is_at_least_fp32_accurate(hp, n, kind):
    """conv_wgrad3_planes_kernel (dz3 and a2 staged channel-innermost as two scaled fp16 planes each, fragments through the
    transposing LDS read, 2 samples per stage, 128 sample splits summed in fixed order): given the kernel's own dz3 and
    a2, dW3 and db3 against float64, beside torch's fp32 weight gradient.  n = 101 leaves a one-sample last stage and
    most of the 128 splits empty; n = 6 leaves all but three empty."""
    _bwd_setup(hp, n, 45, kind)
    got = _grad_views(hp)
    g = torch.nn.grad
    for enc, pre in ((0, "actor.pre"), (1, "critic.pre")):
        dz3 = hp.debug_buffer(6, (64, 7, 7), n, enc).cpu()
        a2 = hp.debug_buffer(1, (64, 9, 9), n, enc).cpu()
        ref = g.conv2d_weight(a2.double(), (64, 64, 3, 3), dz3.double()).numpy()
        f32 = g.conv2d_weight(a2, (64, 64, 3, 3), dz3).numpy().astype(np.float64)
        k = got[pre + ".conv3.weight"].astype(np.float64)
        err_kernel, err_f32 = np.abs(k - ref).max(), np.abs(f32 - ref).max()
        mass = float(g.conv2d_weight(a2.double().abs(), (64, 64, 3, 3), dz3.double().abs()).max())
        P.MARGINS.check("accuracy", "conv3_wgrad_units", err_kernel / (2.0 ** -24 * mass), "(%s n=%d: kernel %.3e, fp32 %.3e)" % (pre, n, err_kernel, err_f32))
        db = got[pre + ".conv3.bias"].astype(np.float64)
        np.testing.assert_allclose(db, dz3.double().sum((0, 2, 3)).numpy(), rtol=0,
                                   atol=8 * 2.0 ** -24 * float(dz3.double().abs().sum((0, 2, 3)).max()))


@pytest.mark.parametrize("n", [77, 3])
@pytest.mark.parametrize("kind", ["uniform", "pong_wide"])
def test_conv2_weight_gradient_is_at_least_fp32_accurate(hp, n, kind):
    """conv_wgrad2_planes_kernel (dz2 and a1 staged channel-innermost as two scaled fp16 planes each, one sample per stage,
    fragments through the transposing LDS read): given the kernel's own dz2 and a1, dW2 and db2 against float64."""
    _bwd_setup(hp, n, 46, kind)
    got = _grad_views(hp)
    g = torch.nn.grad
    for enc, pre in ((0, "actor.pre"), (1, "critic.pre")):
        dz2 = hp.debug_buffer(5, (64, 9, 9), n, enc).cpu()
        a1 = hp.debug_buffer(0, (32, 20, 20), n, enc).cpu()
        ref = g.conv2d_weight(a1.double(), (64, 32, 4, 4), dz2.double(), stride=2).numpy()
        f32 = g.conv2d_weight(a1, (64, 32, 4, 4), dz2, stride=2).numpy().astype(np.float64)
        k = got[pre + ".conv2.weight"].astype(np.float64)
        err_kernel, err_f32 = np.abs(k - ref).max(), np.abs(f32 - ref).max()
        mass = float(g.conv2d_weight(a1.double().abs(), (64, 32, 4, 4), dz2.double().abs(), stride=2).max())
        P.MARGINS.check("accuracy", "conv2_wgrad_units", err_kernel / (2.0 ** -24 * mass), "(%s n=%d: kernel %.3e, fp32 %.3e)" % (pre, n, err_kernel, err_f32))
        db = got[pre + ".conv2.bias"].astype(np.float64)
        np.testing.assert_allclose(db, dz2.double().sum((0, 2, 3)).numpy(), rtol=0,
                                   atol=8 * 2.0 ** -24 * float(dz2.double().abs().sum((0, 2, 3)).max()))


@pytest.mark.parametrize("kind", ["uniform", "pong_wide"])
def test_conv3_data_gradient_is_at_least_fp32_accurate(hp, kind):
    """conv_dgrad3_planes_kernel: given the kernel's own dz3 (as [n,64,7,7]) and a2, dz2 = leaky'(a2) * conv_transpose(dz3, W3)
    against float64."""
    n = 101
    w = _bwd_setup(hp, n, 42, kind)
    for enc, pre in ((0, "actor.pre"), (1, "critic.pre")):
        dz3 = hp.debug_buffer(6, (64, 7, 7), n, enc).cpu()
        a2 = hp.debug_buffer(1, (64, 9, 9), n, enc).cpu()
        dz2 = hp.debug_buffer(5, (64, 9, 9), n, enc).cpu().numpy().astype(np.float64)
        W = torch.from_numpy(w[pre + ".conv3.weight"])
        ct = torch.nn.functional.conv_transpose2d
        ref = _leaky_mask(a2.double(), ct(dz3.double(), W.double())).numpy()
        f32 = _leaky_mask(a2, ct(dz3, W)).numpy().astype(np.float64)
        err_kernel, err_f32 = np.abs(dz2 - ref).max(), np.abs(f32 - ref).max()
        mass = float(ct(dz3.double().abs(), W.double().abs()).max())
        P.MARGINS.check("accuracy", "conv3_dgrad_units", err_kernel / (2.0 ** -24 * mass), "(%s: kernel %.3e, fp32 %.3e)" % (pre, err_kernel, err_f32))
        # mean error beside torch's fp32 operator (an f16x3 product keeps ~2^-22 of every term; torch's blocked fp32 sums ~2^-24)
        P.MARGINS.check("accuracy", "conv3_dgrad_mean_vs_torch_fp32", np.abs(dz2 - ref).mean() / np.abs(f32 - ref).mean(), "(%s)" % pre)


@pytest.mark.parametrize("kind", ["uniform", "pong_wide"])
def test_conv2_data_gradient_is_at_least_fp32_accurate(hp, kind):
    """conv_dgrad2_both_kernel: given the kernel's own dz2, the RAW gradient w.r.t. a1 (conv1's leaky mask is applied
    later, by the conv1 weight gradient) = conv_transpose(dz2, W2, stride 2) against float64."""
    n = 100
    w = _bwd_setup(hp, n, 43, kind)
    for enc, pre in ((0, "actor.pre"), (1, "critic.pre")):
        dz2 = hp.debug_buffer(5, (64, 9, 9), n, enc).cpu()
        da1 = hp.debug_buffer(4, (32, 20, 20), n, enc).cpu().numpy().astype(np.float64)
        W = torch.from_numpy(w[pre + ".conv2.weight"])
        ct = torch.nn.functional.conv_transpose2d
        ref = ct(dz2.double(), W.double(), stride=2).numpy()
        f32 = ct(dz2, W, stride=2).numpy().astype(np.float64)
        err_kernel, err_f32 = np.abs(da1 - ref).max(), np.abs(f32 - ref).max()
        mass = float(ct(dz2.double().abs(), W.double().abs(), stride=2).max())
        P.MARGINS.check("accuracy", "conv2_dgrad_units", err_kernel / (2.0 ** -24 * mass), "(%s: kernel %.3e, fp32 %.3e)" % (pre, err_kernel, err_f32))
        # mean error beside torch's fp32 operator (an f16x3 product keeps ~2^-22 of every term; torch's blocked fp32 sums ~2^-24)
        P.MARGINS.check("accuracy", "conv2_dgrad_mean_vs_torch_fp32", np.abs(da1 - ref).mean() / np.abs(f32 - ref).mean(), "(%s)" % pre)


def _adopt_kernel_decisions(h, net, n, x):
    """Leaky-ReLU decision boundaries.  A pre-activation within fp32 noise of zero can come out on either side
    depending on the summation order (the conv1 forward is a plane-product kernel whose output is CLOSER to float64 than
    an fp32 chain, but not the same bits), and ONE such element of conv2 changes hundreds of conv1 weight-gradient
    elements by a few 1e-3 of their size.  So the per-parameter gradient comparisons are made under identical
    decisions: this (1) runs the oracle forward, (2) checks that the kernel's activation signs differ from the
    oracle's pre-activation signs at a handful of elements only, all with |z| within 2e-5 of zero, and (3) makes
    the oracle's backward use the kernel's decisions (oracle Encoder.forced).  Call h.ppo_iter(...) first."""
    encs = [m for m in net.modules() if isinstance(m, O.Encoder)]
    with torch.no_grad():
        net(x)
    shapes = ((32, 20, 20), (64, 9, 9), (64, 7, 7))
    for e, enc in enumerate(encs):
        forced = []
        for k, shp in enumerate(shapes):
            act = h.debug_buffer(k, shp if k < 2 else (3136,), n, e).cpu().reshape((n,) + shp)
            pos = act > 0
            z = enc.last_z[k]
            differ = pos != (z > 0)
            assert int(differ.sum()) <= 8, (e, k, int(differ.sum()))
            if differ.any():
                assert float(z[differ].abs().max()) < 2e-5, (e, k, float(z[differ].abs().max()))
            forced.append(pos)
        enc.forced = forced


def _release_decisions(net):
    for m in net.modules():
        if isinstance(m, O.Encoder):
            m.forced = None


@pytest.mark.parametrize("n", [1, 33, 37])
def test_gradients_batch_fills_an_odd_capacity(onet, n):
    """n == max_batch, odd: the weight-gradient kernels pair samples and the dense-layer gradient reduces
    over 32-sample k-blocks, so their last pair / k-block is ragged exactly at the END of every workspace
    tensor and of the caller's frame buffer -- the tail paths must neither read past them nor let the
    missing samples contribute."""
    from ddrl4nav_amd.engine import HotPath
    h = HotPath(max_batch=n)
    try:
        h.set_params(flatten(make_weights(0)))
        rng = np.random.default_rng(4100 + n)
        frames = rng.integers(0, 256, size=(n, 4, 84, 84), dtype=np.uint8)
        acts = rng.integers(0, 6, size=n).astype(np.float32)
        old = (np.full(n, -1.79) + rng.normal(0, 0.3, n)).astype(np.float32)
        adv = rng.normal(size=n).astype(np.float32)
        ret = rng.normal(size=n).astype(np.float32)
        h.ppo_iter(dev(frames), dev(acts), dev(old), dev(adv), dev(ret))
        onet.load_weights(make_weights(0))
        onet.zero_grad()
        t = torch.from_numpy
        x = O.frames_to_f32(frames)
        _adopt_kernel_decisions(h, onet, n, x)
        _, al, vl, ent = O.ppo_losses(onet, x, t(acts), t(old), t(adv), t(ret))
        al.backward()
        vl.backward()
        got = _grad_views(h)
        for name, p in onet.named_parameters():
            want = p.grad.numpy()
            assert np.isfinite(got[name]).all(), name
            assert np.abs(got[name] - want).max() <= 1e-4 * np.abs(want).max() + 1e-12, (name, n)
    finally:
        _release_decisions(onet)
        h.close()


@pytest.mark.parametrize("n", [1, 37, 200, 511])
def test_gradients_ragged_batches_vs_oracle(hp, onet, n):
    """Odd sample counts exercise the zero-padded half of a sample pair in the weight-gradient
    kernels, partial column tiles and empty split-K slabs."""
    rng = np.random.default_rng(700 + n)
    frames = rng.integers(0, 256, size=(n, 4, 84, 84), dtype=np.uint8)
    acts = rng.integers(0, 6, size=n).astype(np.float32)
    old = (np.full(n, -1.79) + rng.normal(0, 0.3, n)).astype(np.float32)
    adv = rng.normal(size=n).astype(np.float32)
    ret = rng.normal(size=n).astype(np.float32)
    hp.set_params(flatten(make_weights(0)))
    hp.ppo_iter(dev(frames), dev(acts), dev(old), dev(adv), dev(ret))
    onet.load_weights(make_weights(0))
    onet.zero_grad()
    t = torch.from_numpy
    x = O.frames_to_f32(frames)
    _adopt_kernel_decisions(hp, onet, n, x)
    try:
        _, al, vl, ent = O.ppo_losses(onet, x, t(acts), t(old), t(adv), t(ret))
        al.backward()
        vl.backward()
    finally:
        _release_decisions(onet)
    tail = hp.grads[hp.n_params:hp.n_params + 3].cpu().numpy()
    np.testing.assert_allclose(tail, [al.item(), vl.item(), ent.item()], rtol=2e-5, atol=2e-6)
    got = _grad_views(hp)
    for name, p in onet.named_parameters():
        want = p.grad.numpy()
        scale = np.abs(want).max()
        # up to 200 x 400 products per element with heavy cancellation: both sides carry fp32
        # summation noise of ~sqrt(N) * 6e-8 of the partial sums, so the bound is looser than at B=64
        assert np.abs(got[name] - want).max() <= 1e-4 * scale + 1e-12, (name, n)
        g64, w64 = got[name].astype(np.float64).ravel(), want.astype(np.float64).ravel()
        assert g64 @ w64 / (np.linalg.norm(g64) * np.linalg.norm(w64) + 1e-300) > 1 - 1e-8, (name, n)


# ---- non-default learner modes: SHARE_CNN_NET=True and SMOOTH_L1_LOSS=True ---------------------
def _views(flat, shared):
    out, off = {}, 0
    for name, shape, _ in param_specs(shared=shared):
        n = int(np.prod(shape))
        out[name] = flat[off:off + n].reshape(shape)
        off += n
    return out


def _step4(h, args):
    h.ppo_iter(*args)
    h.clip_adam_step()
    s = h.stats()
    return [s["PpoTotalLoss"], s["ActorLoss"], s["VLoss"], s["EntLoss"]]


@pytest.fixture(scope="module")
def hp_shared():
    from ddrl4nav_amd.engine import HotPath
    h = HotPath(max_batch=512, share_cnn_net=1).keep_activations()
    assert h.n_params == 1684128 + 6 * 512 + 6 + 513 and h.n_actor == h.n_params
    h.set_params(flatten(make_weights(0, shared=True)))
    yield h
    h.close()


def test_shared_prenet_forward_loss_gradients_f10(hp_shared, golden):
    h = hp_shared
    g3, g = golden("f3_loss"), golden("f10_shared")
    frames = dev(g3["frames"])
    probs, value, _, logp = h.forward(frames, act=dev(g["actions"]))
    np.testing.assert_allclose(probs.cpu().numpy(), g["probs"], rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(value.cpu().numpy(), g["value"], rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(logp.cpu().numpy(), g["logp"], rtol=1e-5, atol=1e-6)
    ha, hc = h.last_features(64)
    assert torch.equal(ha, hc)  # one encoder feeds both heads
    np.testing.assert_allclose(ha.cpu().numpy()[:, :16], g["h"], rtol=1e-5, atol=2e-6)
    h.ppo_iter(frames, dev(g["actions"]), dev(g["old_logps"]), dev(g["advs"]), dev(g["rets"]))
    tail = h.grads[h.n_params:h.n_params + 3].cpu().numpy()
    np.testing.assert_allclose(tail, g["loss4"][1:], rtol=1e-5, atol=1e-6)
    # total_loss.backward(): value gradient scaled by V_LOSS_THETA, entropy term included
    net = O.OracleSharedPPO()
    net.load_weights(make_weights(0, shared=True))
    t = lambda k: torch.from_numpy(g[k])
    total, _, _, _ = O.ppo_losses(net, O.frames_to_f32(g3["frames"]), t("actions"), t("old_logps"), t("advs"), t("rets"))
    total.backward()
    got = _views(h.grads[:h.n_params].cpu().numpy(), True)
    for name, p in net.named_parameters():
        want = p.grad.numpy()
        scale = np.abs(want).max()
        assert np.abs(got[name] - want).max() <= 2e-5 * scale, (name, np.abs(got[name] - want).max(), scale)
        np.testing.assert_allclose(got[name].reshape(-1)[:64], g["ghead/" + name], rtol=0, atol=2e-5 * scale)
    gn = float(torch.linalg.vector_norm(h.grads[:h.n_params].double()).item())
    np.testing.assert_allclose(gn, g["gnorm"], rtol=1e-5)


def test_shared_prenet_entropy_gradient_is_present(hp_shared, golden):
    """Property: in the shared mode the entropy term has a gradient (ppo.py:108-112) -- changing
    ENTROPY_LOSS_THETA must change the actor-head gradient; in the default mode it must not."""
    from ddrl4nav_amd.engine import HotPath
    g3, g = golden("f3_loss"), golden("f10_shared")
    args = (dev(g3["frames"]), dev(g["actions"]), dev(g["old_logps"]), dev(g["advs"]), dev(g["rets"]))
    outs = []
    for shared in (1, 0):
        for theta in (0.05, 0.5):
            h = HotPath(max_batch=64, share_cnn_net=shared, ent_loss_theta=theta)
            h.set_params(flatten(make_weights(0, shared=bool(shared))))
            h.ppo_iter(*args)
            outs.append(_views(h.grads[:h.n_params].cpu().numpy(), bool(shared))["actor.actor_linear.weight"].copy())
            h.close()
    assert np.abs(outs[0] - outs[1]).max() > 1e-4
    assert np.array_equal(outs[2], outs[3])


def test_shared_prenet_learn_sequence_f10(hp_shared, golden):
    h = hp_shared
    g3, g = golden("f3_loss"), golden("f10_shared")
    h.set_params(flatten(make_weights(0, shared=True)))
    h.reset_optimizer()
    import parity_util as P
    ref = g["losses"]
    env = P.mode_loss_envelope("shared", ref, g["losses_f64"], g["losses_f32t8"])
    args = (dev(g3["frames"]), dev(g["actions"]), dev(g["old_logps"]), dev(g["advs"]), dev(g["rets"]))
    P.check_sequence("learn_f10_shared", "shared", lambda: _step4(h, args), lambda: h.params.cpu().numpy(), ref, env)


@pytest.mark.parametrize("n", [1, 37, 300])
def test_shared_prenet_ragged_gradients_vs_oracle(hp_shared, n):
    rng = np.random.default_rng(900 + n)
    frames = rng.integers(0, 256, size=(n, 4, 84, 84), dtype=np.uint8)
    acts = rng.integers(0, 6, size=n).astype(np.float32)
    old = (np.full(n, -1.79) + rng.normal(0, 0.3, n)).astype(np.float32)
    adv = rng.normal(size=n).astype(np.float32)
    ret = rng.normal(size=n).astype(np.float32)
    hp_shared.set_params(flatten(make_weights(0, shared=True)))
    hp_shared.ppo_iter(dev(frames), dev(acts), dev(old), dev(adv), dev(ret))
    net = O.OracleSharedPPO()
    net.load_weights(make_weights(0, shared=True))
    t = torch.from_numpy
    x = O.frames_to_f32(frames)
    _adopt_kernel_decisions(hp_shared, net, n, x)
    total, al, vl, ent = O.ppo_losses(net, x, t(acts), t(old), t(adv), t(ret))
    total.backward()
    tail = hp_shared.grads[hp_shared.n_params:hp_shared.n_params + 3].cpu().numpy()
    np.testing.assert_allclose(tail, [al.item(), vl.item(), ent.item()], rtol=2e-5, atol=2e-6)
    got = _views(hp_shared.grads[:hp_shared.n_params].cpu().numpy(), True)
    for name, p in net.named_parameters():
        want = p.grad.numpy()
        assert np.abs(got[name] - want).max() <= 1e-4 * np.abs(want).max() + 1e-12, (name, n)


def test_smooth_l1_value_loss_f11(golden):
    from ddrl4nav_amd.engine import HotPath
    g3, g = golden("f3_loss"), golden("f11_smooth_l1")
    h = HotPath(max_batch=64, smooth_l1_loss=1)
    h.set_params(flatten(make_weights(0)))
    args = (dev(g3["frames"]), dev(g3["actions"]), dev(g3["old_logps"]), dev(g3["advs"]), dev(g["rets"]))
    h.ppo_iter(*args)
    tail = h.grads[h.n_params:h.n_params + 3].cpu().numpy()
    np.testing.assert_allclose(tail, g["loss4"][1:], rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(h.debug_buffer(9, (), 64, 0).cpu().numpy(), g["dvalue"], rtol=1e-5, atol=1e-8)
    got = _views(h.grads[:h.n_params].cpu().numpy(), False)
    for name in got:
        scale = max(np.abs(g["ghead/" + name]).max(), g["gl2/" + name] / np.sqrt(got[name].size))
        np.testing.assert_allclose(got[name].reshape(-1)[:64], g["ghead/" + name], rtol=0, atol=5e-5 * scale)
        np.testing.assert_allclose(np.sqrt((got[name].astype(np.float64) ** 2).sum()), g["gl2/" + name], rtol=1e-4)
    h.reset_optimizer()
    ref = g["losses"]
    # what "the same computation" means for this sequence: the reference itself in float64, and the reference in
    # fp32 with the batch in three other orders (same mathematics -- full-batch means --, other summation order;
    # tests/golden/make_golden_reorder.py).  The permuted runs drift 3.5 x further by iteration 10 than the f64 one.
    import parity_util as P
    gb = golden("f11b_smooth_l1_reorder")
    env = P.mode_loss_envelope("smooth", ref, g["losses_f64"], gb["losses_perm"])
    # the smooth-L1 gradient is +-1/B for every |ret - v| > 1: many critic-side weight gradients sit at the fp32
    # noise floor, where Adam moves an element by O(lr) either way -- the reference's own fp32 runs sit up to 11 %
    # of the update norm away from its float64 run here (2.5 % in F4); the bound is relative to exactly that
    P.check_sequence("learn_f11_smooth", "smooth", lambda: _step4(h, args), lambda: h.params.cpu().numpy(), ref, env)
    h.close()


# ---- more than 8 actions: LDS-weight head kernels + head_wgrad (full Atari action set = 18) ----
def _views_a(flat, n_actions):
    out, off = {}, 0
    for name, shape, _ in param_specs(n_actions=n_actions):
        n = int(np.prod(shape))
        out[name] = flat[off:off + n].reshape(shape)
        off += n
    return out


def test_eighteen_actions_golden_f12(golden):
    from ddrl4nav_amd.engine import HotPath
    g3, g = golden("f3_loss"), golden("f12_actions18")
    h = HotPath(max_batch=64, n_actions=18)
    h.set_params(flatten(make_weights(0, n_actions=18), n_actions=18))
    frames = dev(g3["frames"])
    probs, value, _, logp = h.forward(frames, act=dev(g["actions"]))
    assert probs.shape == (64, 18)
    np.testing.assert_allclose(probs.cpu().numpy(), g["probs"], rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(value.cpu().numpy(), g["value"], rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(logp.cpu().numpy(), g["logp"], rtol=1e-5, atol=1e-6)
    _, _, ent = h.categorical_stats(probs)
    np.testing.assert_allclose(ent.cpu().numpy(), g["entropy"], rtol=1e-5, atol=1e-6)
    h.ppo_iter(frames, dev(g["actions"]), dev(g["old_logps"]), dev(g["advs"]), dev(g["rets"]))
    tail = h.grads[h.n_params:h.n_params + 3].cpu().numpy()
    np.testing.assert_allclose(tail, g["loss4"][1:], rtol=1e-5, atol=1e-6)
    got = _views_a(h.grads[:h.n_params].cpu().numpy(), 18)
    for key, name in (("grad_actor_linear_w", "actor.actor_linear.weight"), ("grad_actor_linear_b", "actor.actor_linear.bias")):
        scale = np.abs(g[key]).max()
        assert np.abs(got[name] - g[key]).max() <= 2e-5 * scale, name
    for name in got:
        np.testing.assert_allclose(np.sqrt((got[name].astype(np.float64) ** 2).sum()), g["gl2/" + name], rtol=2e-5)
        scale = np.abs(got[name]).max()
        np.testing.assert_allclose(got[name].reshape(-1)[:64], g["ghead/" + name], rtol=0, atol=2e-5 * scale)
    # 10 Adam steps follow the reference's loss trajectory within LOSS_LIMIT x the reference's own spread (tests/parity_util.py)
    # under a changed summation order (float64 / 8-thread runs stored in the fixture)
    h.reset_optimizer()
    ref = g["losses"]
    import parity_util as P
    env = P.mode_loss_envelope("actions18", ref, g["losses_f64"], g["losses_f32t8"])   # incl. the reference on torch's native backend
    args = (frames, dev(g["actions"]), dev(g["old_logps"]), dev(g["advs"]), dev(g["rets"]))
    for it in range(1, 11):
        got4 = np.array(_step4(h, args))
        excess = np.abs(got4 - ref[it - 1]) - (1e-5 * np.abs(ref[it - 1]) + 2e-6)
        P.MARGINS.check("learn_f12_actions18", "loss_env", max(0.0, float(np.max(excess / np.maximum(env[it - 1], 1e-12)))),
                        "(iteration %d)" % it)
    h.close()


@pytest.mark.parametrize("A", [2, 5, 6, 7, 8, 9, 13, 18])  # <= 6: reduced-together dot products, LDS weights; 7-8: register weights; > 8: LDS weights + head_wgrad
def test_action_counts_vs_oracle(A):
    from ddrl4nav_amd.engine import HotPath
    n = 70
    rng = np.random.default_rng(1200 + A)
    frames = rng.integers(0, 256, size=(n, 4, 84, 84), dtype=np.uint8)
    acts = rng.integers(0, A, size=n).astype(np.float32)
    old = (np.full(n, -np.log(A)) + rng.normal(0, 0.3, n)).astype(np.float32)
    adv = rng.normal(size=n).astype(np.float32)
    ret = rng.normal(size=n).astype(np.float32)
    w = make_weights(3, n_actions=A)
    h = HotPath(max_batch=n, n_actions=A)
    h.set_params(flatten(w, n_actions=A))
    net = O.OraclePPO(n_actions=A)
    net.load_weights(w)
    t = torch.from_numpy
    probs, value, _, logp = h.forward(dev(frames), act=dev(acts))
    with torch.no_grad():
        oprobs, op_hat, ologits, ov = net(O.frames_to_f32(frames))
    np.testing.assert_allclose(probs.cpu().numpy(), oprobs.numpy(), rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(logp.cpu().numpy(), O.categorical_log_prob(ologits, t(acts)).numpy(), rtol=1e-5, atol=1e-6)
    # sampler contract: inverse CDF over p_hat with the shared counter-based uniforms
    p2, _, action, lp2 = h.forward(dev(frames), act=None, seed=99, stream_id=5)
    p_hat, logits, _ = h.categorical_stats(p2)
    want = O.inverse_cdf_sample(p_hat.cpu().numpy(), sample_uniform(99, 5, n))
    assert np.array_equal(action.cpu().numpy(), want.astype(np.float32))
    h.ppo_iter(dev(frames), dev(acts), dev(old), dev(adv), dev(ret))
    _, al, vl, ent = O.ppo_losses(net, O.frames_to_f32(frames), t(acts), t(old), t(adv), t(ret))
    al.backward()
    vl.backward()
    tail = h.grads[h.n_params:h.n_params + 3].cpu().numpy()
    np.testing.assert_allclose(tail, [al.item(), vl.item(), ent.item()], rtol=2e-5, atol=2e-6)
    got = _views_a(h.grads[:h.n_params].cpu().numpy(), A)
    for name, p in net.named_parameters():
        want_g = p.grad.numpy()
        assert np.abs(got[name] - want_g).max() <= 5e-5 * np.abs(want_g).max() + 1e-12, (name, A)
    h.close()


@pytest.mark.parametrize("C", [1, 3])
def test_other_frame_stacks_golden_f20(golden, C):
    """AtariPreNet with 1 / 3 stacked frames (reference nn/atari_encoder.py:12-14 takes `num_inputs`; golden F20 =
    the reference's own forward and three PPO iterations, tests/golden/make_golden_channels.py): forward, losses and the whole
    gradient of the first iteration against the oracle (itself pinned to F20 in tests/test_oracle_golden.py), losses of
    iterations 2 and 3, parameter checksums after the third."""
    from ddrl4nav_amd.engine import HotPath
    g = golden("f20_channels")
    p = "c%d/" % C
    w = make_weights(seed=20 + C, num_inputs=C)
    h = HotPath(max_batch=16, in_channels=C)
    try:
        h.set_params(flatten(w, num_inputs=C))
        frames, acts = dev(g[p + "frames"]), dev(g[p + "actions"])
        probs, value, _, logp = h.forward(frames, act=acts)
        np.testing.assert_allclose(probs.cpu().numpy(), g[p + "probs"], rtol=1e-5, atol=1e-6)
        np.testing.assert_allclose(value.cpu().numpy(), g[p + "value"], rtol=1e-5, atol=1e-6)
        np.testing.assert_allclose(logp.cpu().numpy(), g[p + "logp"], rtol=1e-5, atol=1e-6)
        old, advs, rets = dev(g[p + "old_logps"]), dev(g[p + "advs"]), dev(g[p + "rets"])
        # gradient of iteration 1 against the oracle's autograd
        net = O.OraclePPO(num_inputs=C)
        net.load_weights(w)
        x = O.frames_to_f32(g[p + "frames"])
        t = lambda k: torch.from_numpy(g[p + k])
        _, al, vl, _ = O.ppo_losses(net, x, t("actions"), t("old_logps"), t("advs"), t("rets"))
        al.backward()
        vl.backward()
        h.ppo_iter(frames, acts, old, advs, rets)
        flat = h.grads[:h.n_params].cpu().numpy()
        off = 0
        for name, q in net.named_parameters():
            want = q.grad.numpy()
            got = flat[off:off + want.size].reshape(want.shape)
            off += want.size
            scale = np.abs(want).max()
            assert np.abs(got - want).max() <= 2e-5 * scale, (name, np.abs(got - want).max(), scale)
        assert off == h.n_params
        # three iterations: losses against the reference's (the same batch every iteration, as PPO.learn does)
        for it in range(3):
            if it:
                h.ppo_iter(frames, acts, old, advs, rets)
            h.clip_adam_step()
            s = h.stats()
            got = [s["PpoTotalLoss"], s["ActorLoss"], s["VLoss"], s["EntLoss"]]
            np.testing.assert_allclose(got, g[p + "losses"][it], rtol=2e-5 * (1 + 4 * it), atol=2e-6 * (1 + 4 * it))
        flat = h.params.cpu().numpy().astype(np.float64)
        off = 0
        for name, shape, _ in param_specs(C):
            n = int(np.prod(shape))
            a = flat[off:off + n]
            off += n
            np.testing.assert_allclose(np.sqrt((a ** 2).sum()), g[p + "it3/l2/" + name], rtol=2e-5)
    finally:
        h.close()


@pytest.mark.parametrize("n", [5, 37, 300])
def test_sign_masks_equal_the_signs_of_the_stored_activations(n):
    """The backward takes its leaky-ReLU decisions from sign masks the forward epilogues write: every bit of
    m1 / m2 / m3 must say `activation <= 0` of the STORED a1 / a2 / a3 -- the reference's `act > 0` -- in the layouts of
    csrc/common.h (m1: word per column, bit m1_bit(oc); m2 / m3: word per (sample, pixel, lane half), bit 16 i + 15 - r for
    channel 32 i + acc_row(r, half))."""
    from ddrl4nav_amd.engine import HotPath
    h = HotPath(max_batch=max(n, 8)).keep_activations()
    try:
        h.set_params(flatten(make_weights(0)))
        rng = np.random.default_rng(100 + n)
        frames = rng.integers(0, 256, size=(n, 4, 84, 84), dtype=np.uint8)
        frames[0] = 87   # flat samples: many activations on the leaky branch
        frames[n - 1, :, :, :40] = 0
        B = n
        f32 = lambda a: dev(np.asarray(a, np.float32))
        h.ppo_iter(dev(frames), f32(rng.integers(0, 6, size=B)), f32(np.full(B, -1.79)), f32(rng.normal(size=B)), f32(rng.normal(size=B)))
        words = lambda which, per, e: h.debug_buffer(which, (per,), n, e).view(torch.int32).cpu().numpy().astype(np.uint32)
        acc_row = lambda r, hi: (r & 3) + 8 * (r >> 2) + 4 * hi
        for e in range(2):
            a1 = h.debug_buffer(0, (32, 400), n, e).cpu().numpy()
            m1 = words(10, 400, e).reshape(n, 400)
            for oc in range(32):
                bit = ((oc >> 2) & 1) * 16 + 15 - ((oc & 3) + 4 * (oc >> 3))
                assert np.array_equal(((m1 >> bit) & 1).astype(bool), a1[:, oc, :] <= 0), ("m1", e, oc)
            for which, am, ch_pix in ((11, 1, 81), (12, 2, 49)):
                act = h.debug_buffer(am, (64, ch_pix), n, e).cpu().numpy()
                m = words(which, ch_pix * 2, e).reshape(n, ch_pix, 2)
                for i in range(2):
                    for half in range(2):
                        for r in range(16):
                            ch = 32 * i + acc_row(r, half)
                            got = ((m[:, :, half] >> (16 * i + 15 - r)) & 1).astype(bool)
                            assert np.array_equal(got, act[:, ch, :] <= 0), ("m%d" % (which - 9), e, ch)
    finally:
        h.close()
