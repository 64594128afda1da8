#!/usr/bin/env python3
"""Golden vectors for the image-only shared nav encoder, by IMPORTING THE REFERENCE:

  f25_navpre_shared.npz   NavPreNet(image_channel=1) shared + CategoricalActor(5): the SHARE_CNN_NET branch of create_net when the
                          env has no pedestrian map (runner/utils.py:104; encoder nn/nav_encoder.py:12-43) -- forward, one loss /
                          gradient evaluation and ten iterations of the reference's PPO.learn, as F13 / F14 (make_golden_nav.py)
  f25b_spread.npz         the reference's own fp32 spread around its float64 run on that batch (make_golden_nav_spread.py)
  f25c_backend_spread.npz ... and its runs on torch's native convolution backend (make_golden_backend_spread_nav.py)
  f26_navpre_unaligned.npz  the same net on a batch SELECTED so that no ReLU / max-pool decision of its forward lies near a tie:
                          the reference is run in float64 on N_CAND seeded candidates, every decision's margin is measured
                          (|pre-activation| for a ReLU whose window maximum it is; gap between the two largest entries of a
                          window whose maximum is positive; all relative to the layer's largest |pre-activation|), and the
                          B_SEL candidates with the largest smallest-margin are kept.  An fp32 implementation whose
                          pre-activations are accurate to well under that margin takes the SAME decisions as the reference, so its
                          full gradient can be checked against the reference's with NO transfer of decisions
                          (tests/test_generic_gpu.py::test_nav_gradient_unaligned).  Stored: the batch, the margins, per-layer
                          decision digests, the losses and the reference's fp32 gradient (whole tensors up to FULL_MAX elements,
                          a stride of the larger ones, L2 norm and sum of every tensor).

  f27_nav1d_unaligned.npz   the same construction for BASELINE config 4's own net -- NavPreNet1D x2 (actor's and critic's own encoders:
                          7x7 / 5x5 / 3x3 conv + ReLU + pool stack, the laser branch's Conv1d pair + fc_1d) + GaussionActor(2), the
                          non-shared branch of ppo.py:118-129 -- recipe seed 27: N_CAND1D candidates, every decision of BOTH encoders
                          measured, the B_SEL with the largest smallest-margin kept.

Inputs are seeded and stored in the fixtures; weights come from utils/recipe.py:hash_weights (seeds 25 / 27).
Usage: python tests/golden/make_golden_navpre.py [f25] [f25b] [f25c] [f26] [f27]     (default: all five)
"""
import os
import sys

import numpy as np
import torch
import torch.nn.functional as F

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)
import make_golden_nav as NV  # noqa: E402

NAME, SEED = "f25_navpre_shared", 25
N_CAND, B_SEL, FULL_MAX, STRIDE_N = 16384, 32, 80000, 4097
N_CAND1D = 12288


def build():
    from USTC_lab.nn import CategoricalActor, Critic, PPO
    from USTC_lab.nn.nav_encoder import NavPreNet
    cfg, cfg_nn = NV.cfgs({"discrete_action": True, "discrete_actions": list(range(5))})
    cfg_nn.SHARE_CNN_NET = True
    actor = CategoricalActor(action_output_dim=5, device="cpu", soft_max_grid=True, last_input_dim=512, nn_dtype=torch.float32)
    net = PPO(actor, Critic(device="cpu", last_input_dim=512), NavPreNet(image_channel=1, last_output_dim=512), None, cfg,
              cfg_nn).to("cpu")

    def reopt(n):
        n.optim = torch.optim.Adam(n.parameters(), cfg_nn.LEARNING_RATE)
    return net, reopt, cfg_nn


def maker():
    net, reopt, _ = build()
    return net, reopt, SEED


def sample_cat(dist):
    torch.manual_seed(251)
    return dist.sample().to(torch.float32)


def f25():
    net, reopt, cfg_nn = build()
    NV.load_recipe(net, SEED)
    rng = np.random.default_rng(SEED)
    B = 64
    img = (rng.random((B, 1, 48, 48)) < 0.3).astype(np.float32) * rng.uniform(0.25, 1.0, size=(B, 1, 48, 48)).astype(np.float32)
    vec = rng.normal(0, 1, size=(B, 9)).astype(np.float32)
    out = {"names": np.array([k for k, _ in net.named_parameters()])}
    NV.run_case(net, cfg_nn, [img, vec], sample_cat, B, rng, out, reopt)
    np.savez_compressed(os.path.join(HERE, NAME + ".npz"), **out)
    print("  %-26s %8d B" % (NAME + ".npz", os.path.getsize(os.path.join(HERE, NAME + ".npz"))))


def decision_margins(pre, img, vec, laser=None):
    """[n, sites] smallest decision margin per sample and site (conv1..3, fc0, fc1; with `laser`: NavPreNet1D, + fc_1d in front) of the
    reference encoder's forward, relative to the site's largest |pre-activation| over the batch; and the per-site decision digests
    (see the module docstring)."""
    out, digest = [], {}
    x = img
    with torch.no_grad():
        lz = None
        if laser is not None:      # nav_encoder.py:115-118: two un-activated Conv1d layers, then fc_1d = Linear + ReLU
            l = pre.conv1d2(pre.conv1d1(laser))
            lz = pre.fc_1d[0](l.view(l.shape[0], -1))
            out.append((lz.abs() / lz.abs().amax()).amin(1))
            digest["fc_1d_positive"] = (lz > 0).sum(1).numpy().astype(np.int64)
        for li, conv in enumerate((pre.conv1, pre.conv2, pre.conv3), 1):
            z = conv(x)
            s = z.abs().amax()
            n, c, h, w = z.shape
            win = z.view(n, c, h // 2, 2, w // 2, 2).permute(0, 1, 2, 4, 3, 5).reshape(n, c, h // 2, w // 2, 4)
            top2 = win.topk(2, dim=-1)
            mx, gap = top2.values[..., 0], top2.values[..., 0] - top2.values[..., 1]
            risk = torch.minimum(mx.abs(), torch.where(mx > 0, gap, torch.full_like(gap, 1e9)))
            out.append((risk / s).reshape(n, -1).amin(1))
            # digest: number of windows with a positive maximum, and the sum over those of the winner's index in PyTorch's scan order
            pos = mx > 0
            digest["conv%d_positive" % li] = pos.reshape(n, -1).sum(1).numpy().astype(np.int64)
            digest["conv%d_argsum" % li] = (top2.indices[..., 0] * pos).reshape(n, -1).sum(1).numpy().astype(np.int64)
            x = F.max_pool2d(F.relu(z), 2, 2)
        z0 = pre.fc0[0](x.view(x.size(0), -1))
        out.append((z0.abs() / z0.abs().amax()).amin(1))
        digest["fc0_positive"] = (z0 > 0).sum(1).numpy().astype(np.int64)
        z1 = pre.fc1[0](torch.cat((F.relu(z0), vec), 1) if lz is None else torch.cat((F.relu(lz), F.relu(z0), vec), 1))
        out.append((z1.abs() / z1.abs().amax()).amin(1))
        digest["fc1_positive"] = (z1 > 0).sum(1).numpy().astype(np.int64)
    return torch.stack(out, 1), digest


def f26():
    from USTC_lab.data import Experience
    net, _, cfg_nn = build()
    w, _ = NV.load_recipe(net, SEED)
    rng = np.random.default_rng(2600)
    img_all = rng.uniform(0, 1, size=(N_CAND, 1, 48, 48)).astype(np.float32)
    vec_all = rng.normal(0, 1, size=(N_CAND, 9)).astype(np.float32)
    torch.set_num_threads(8)
    net.double()
    best = []
    for lo in range(0, N_CAND, 512):
        m, _ = decision_margins(net.prenet, torch.from_numpy(img_all[lo:lo + 512]).double(), torch.from_numpy(vec_all[lo:lo + 512]).double())
        best.append(m.amin(1))
    best = torch.cat(best).numpy()
    pick = np.sort(np.argsort(-best)[:B_SEL])
    img, vec = img_all[pick].copy(), vec_all[pick].copy()
    # margins of the SELECTED batch (the per-site scale is now this batch's own maximum) and its decision digests, float64
    m64, digest = decision_margins(net.prenet, torch.from_numpy(img).double(), torch.from_numpy(vec).double())
    print("  f26: smallest margin of the %d selected of %d candidates: %.3e (per site %s); median candidate %.1e" % (
        B_SEL, N_CAND, float(m64.min()), " ".join("%.1e" % v for v in m64.amin(0).numpy()), float(np.median(best))))
    net.float()
    torch.set_num_threads(1)
    net.load_state_dict({k: torch.from_numpy(v.copy()) for k, v in w.items()}, strict=True)
    B = B_SEL
    st = [torch.from_numpy(img), torch.from_numpy(vec)]
    with torch.no_grad():
        (dist, _), values = net(st)
        actions = sample_cat(dist)
        old_logps = net.actor.log_prob_from_distribution(dist, actions)
        v0 = values[0][:, 0]
    old_logps = (old_logps + torch.from_numpy(rng.normal(0, 0.25, B).astype(np.float32))).contiguous()
    advs = torch.from_numpy(rng.normal(0, 1, B).astype(np.float32))
    rets = (v0 + advs).contiguous()
    exp = Experience(states=[img.copy(), vec.copy()], advs=advs.numpy(), actions=actions.numpy(), old_logps=old_logps.numpy(),
                     values=rets.numpy().reshape(1, B))
    exp.to_tensor(dtype=torch.float32, device="cpu")
    # the fp32 forward takes the float64 decisions (that is what the margin is for): checked here, asserted again by the tests
    m32, digest32 = decision_margins(net.prenet, st[0], st[1])
    for k in digest:
        assert np.array_equal(digest[k], digest32[k]), k
    out = {"names": np.array([k for k, _ in net.named_parameters()]), "state0": img, "state1": vec, "picked": pick,
           "actions": actions.numpy(), "old_logps": old_logps.numpy(), "advs": advs.numpy(), "rets": rets.numpy(),
           "margin_f64": m64.numpy(), "margin_candidates_median": np.float64(np.median(best))}
    for k, v in digest.items():
        out["digest/" + k] = v
    # the shared branch of ppo.py:110-117 differentiates total_loss
    net.zero_grad()
    pi, values = net(exp.states, exp.actions)
    dist, log_p = pi
    ratio = torch.exp(log_p - exp.old_logps)
    m = torch.min(ratio * exp.advs, torch.clamp(ratio, 1.0 - net.ppo_clip, 1.0 + net.ppo_clip) * exp.advs)
    actor_loss = -torch.mean(torch.where(exp.advs > 0, m, torch.max(m, net.duel_ppo_clip * exp.advs)))
    v_loss = net.vlossf(exp.values[0, :], values[0].squeeze())
    ent = torch.mean(dist.entropy())
    total = actor_loss + v_loss * net.v_loss_theta - ent * net.ent_loss_theta
    total.backward()
    out["loss4"] = np.array([total.item(), actor_loss.item(), v_loss.item(), ent.item()], np.float64)
    out["value"], out["logp"] = values[0].detach().numpy()[:, 0], log_p.detach().numpy()
    for k, p in net.named_parameters():
        g = p.grad.detach().numpy().reshape(-1)
        out["gl2/" + k] = np.float64(np.sqrt((g.astype(np.float64) ** 2).sum()))
        out["gsum/" + k] = np.float64(g.astype(np.float64).sum())
        out["gmax/" + k] = np.float64(np.abs(g).max())
        if g.size <= FULL_MAX:
            out["gfull/" + k] = g.copy()
        else:
            out["gstride/" + k] = g[::max(1, g.size // STRIDE_N)][:STRIDE_N].copy()
    fn = os.path.join(HERE, "f26_navpre_unaligned.npz")
    np.savez_compressed(fn, **out)
    print("  %-26s %8d B" % ("f26_navpre_unaligned.npz", os.path.getsize(fn)))


def build1d():
    from USTC_lab.nn import Critic, GaussionActor, PPO
    from USTC_lab.nn.nav_encoder import NavPreNet1D
    cfg, cfg_nn = NV.cfgs({"discrete_action": False, "act_dim": 2})
    actor = GaussionActor(action_output_dim=2, device="cpu", soft_max_grid=True, last_input_dim=512, nn_dtype=torch.float32,
                          pre=NavPreNet1D(image_channel=3, last_output_dim=512))
    critic = Critic(device="cpu", last_input_dim=512, pre=NavPreNet1D(image_channel=3, last_output_dim=512))
    return PPO(actor, critic, None, None, cfg, cfg_nn).to("cpu"), cfg_nn


def f27():
    from USTC_lab.data import Experience
    net, cfg_nn = build1d()
    w, _ = NV.load_recipe(net, 27)
    rng = np.random.default_rng(2700)
    laser_all = rng.uniform(0.05, 1.0, size=(N_CAND1D, 1, 960)).astype(np.float32)
    vec_all = rng.normal(0, 1, size=(N_CAND1D, 5)).astype(np.float32)
    ped_all = rng.uniform(0, 1, size=(N_CAND1D, 3, 48, 48)).astype(np.float32)
    torch.set_num_threads(8)
    net.double()
    encs = (("actor.pre", net.actor.pre), ("critic.pre", net.critic.pre))

    def both(lo, hi):
        d = lambda a: torch.from_numpy(a[lo:hi]).double()
        ms, digs = [], {}
        for name, pre in encs:
            m, dg = decision_margins(pre, d(ped_all), d(vec_all), d(laser_all))
            ms.append(m)
            digs.update({name + "/" + k: v for k, v in dg.items()})
        return torch.cat(ms, 1), digs

    best = []
    for lo in range(0, N_CAND1D, 256):
        best.append(both(lo, lo + 256)[0].amin(1))
        if lo % 2048 == 0:
            print("  f27: candidates %d / %d" % (lo, N_CAND1D), flush=True)
    best = torch.cat(best).numpy()
    pick = np.sort(np.argsort(-best)[:B_SEL])
    laser, vec, ped = laser_all[pick].copy(), vec_all[pick].copy(), ped_all[pick].copy()
    del laser_all, vec_all
    d = lambda a: torch.from_numpy(a).double()
    ms, digest = [], {}
    for name, pre in encs:
        m, dg = decision_margins(pre, d(ped), d(vec), d(laser))
        ms.append(m)
        digest.update({name + "/" + k: v for k, v in dg.items()})
    m64 = torch.cat(ms, 1)
    print("  f27: smallest margin of the %d selected of %d candidates: %.3e (per site %s); median candidate %.1e" % (
        B_SEL, N_CAND1D, float(m64.min()), " ".join("%.1e" % v for v in m64.amin(0).numpy()), float(np.median(best))))
    net.float()
    torch.set_num_threads(1)
    net.load_state_dict({k: torch.from_numpy(v.copy()) for k, v in w.items()}, strict=True)
    B = B_SEL
    st = [torch.from_numpy(laser), torch.from_numpy(vec), torch.from_numpy(ped)]
    with torch.no_grad():
        (dist, _), values = net(st)
        torch.manual_seed(271)
        actions = dist.sample().to(torch.float32)
        old_logps = net.actor.log_prob_from_distribution(dist, actions)
        v0 = values[0][:, 0]
    old_logps = (old_logps + torch.from_numpy(rng.normal(0, 0.25, B).astype(np.float32))).contiguous()
    advs = torch.from_numpy(rng.normal(0, 1, B).astype(np.float32))
    rets = (v0 + advs).contiguous()
    exp = Experience(states=[laser.copy(), vec.copy(), ped.copy()], advs=advs.numpy(), actions=actions.numpy(),
                     old_logps=old_logps.numpy(), values=rets.numpy().reshape(1, B))
    exp.to_tensor(dtype=torch.float32, device="cpu")
    for name, pre in encs:      # the fp32 forward takes the float64 decisions
        _, dg32 = decision_margins(pre, st[2], st[1], st[0])
        for k, v in dg32.items():
            assert np.array_equal(digest[name + "/" + k], v), (name, k)
    out = {"names": np.array([k for k, _ in net.named_parameters()]), "state0": laser, "state1": vec, "state2": ped, "picked": pick,
           "actions": actions.numpy(), "old_logps": old_logps.numpy(), "advs": advs.numpy(), "rets": rets.numpy(),
           "margin_f64": m64.numpy(), "margin_candidates_median": np.float64(np.median(best))}
    for k, v in digest.items():
        out["digest/" + k] = v
    # the non-shared branch of ppo.py:118-129: actor_loss.backward(); v_loss.backward() (no entropy gradient)
    net.zero_grad()
    pi, values = net(exp.states, exp.actions)
    dist, log_p = pi
    ratio = torch.exp(log_p - exp.old_logps)
    m = torch.min(ratio * exp.advs, torch.clamp(ratio, 1.0 - net.ppo_clip, 1.0 + net.ppo_clip) * exp.advs)
    actor_loss = -torch.mean(torch.where(exp.advs > 0, m, torch.max(m, net.duel_ppo_clip * exp.advs)))
    v_loss = net.vlossf(exp.values[0, :], values[0].squeeze())
    ent = torch.mean(dist.entropy())
    total = actor_loss + v_loss * net.v_loss_theta - ent * net.ent_loss_theta
    actor_loss.backward()
    v_loss.backward()
    out["loss4"] = np.array([total.item(), actor_loss.item(), v_loss.item(), ent.item()], np.float64)
    out["value"], out["logp"] = values[0].detach().numpy()[:, 0], log_p.detach().numpy()
    for k, p in net.named_parameters():
        g = p.grad.detach().numpy().reshape(-1)
        out["gl2/" + k] = np.float64(np.sqrt((g.astype(np.float64) ** 2).sum()))
        out["gsum/" + k] = np.float64(g.astype(np.float64).sum())
        out["gmax/" + k] = np.float64(np.abs(g).max())
        if g.size <= FULL_MAX:
            out["gfull/" + k] = g.copy()
        else:
            out["gstride/" + k] = g[::max(1, g.size // STRIDE_N)][:STRIDE_N].copy()
    fn = os.path.join(HERE, "f27_nav1d_unaligned.npz")
    np.savez_compressed(fn, **out)
    print("  %-26s %8d B" % ("f27_nav1d_unaligned.npz", os.path.getsize(fn)))


def main():
    from make_golden import _install_stubs
    _install_stubs()
    sys.path.insert(0, NV.REF)
    torch.set_num_threads(1)
    what = sys.argv[1:] or ["f25", "f25b", "f25c", "f26", "f27"]
    if "f25" in what:
        f25()
    if "f25b" in what:
        import make_golden_nav_spread as NS
        NS.spread_for({NAME: maker})
    if "f25c" in what:
        import make_golden_backend_spread_nav as BS
        out = {}
        BS.ppo_nets(out, {NAME: maker}, (NAME,))
        fn = os.path.join(HERE, "f25c_backend_spread.npz")
        np.savez_compressed(fn, **out)
        print("  %-26s %8d B" % ("f25c_backend_spread.npz", os.path.getsize(fn)))
    if "f26" in what:
        f26()
    if "f27" in what:
        f27()


if __name__ == "__main__":
    main()
