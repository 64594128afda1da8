#!/usr/bin/env python3
"""Measurement of the operator-composed robot_nav net (BASELINE config 4 shape: NavPreNet1D x2 +
GaussionActor(2) + Critic): PPO iterations on synthetic inputs, per-operator HIP-event times and
algorithmic TFLOP/s.  Not the headline bench (bench.py is); prints one JSON line.

Usage: python tools/bench_nav.py [B] [micro_batch] [iters] [T] [encoder] [loop_iters] [one-stream]
(T given: also one whole actor-learner loop at 512 envs x T steps with loop_iters PPO iterations, BASELINE config 4 shape:
T = 256, loop_iters = 10)"""
import json
import sys
import time
import types
from collections import defaultdict

import numpy as np
import torch

import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ddrl4nav_amd import ops  # noqa: E402
from ddrl4nav_amd.config import BaseConfig, ConfigNN  # noqa: E402
from ddrl4nav_amd.data import Experience  # noqa: E402
from ddrl4nav_amd.runner import create_net  # noqa: E402

events = []


def timed(name, flop_fn, fn):
    def wrap(self, *a, **k):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        out = fn(self, *a, **k)
        e1.record()
        events.append((name(self), flop_fn(self, *a, **k), e0, e1))
        return out
    return wrap


def conv_name(kind):
    return lambda c: "conv%dx%d_%d->%d_%s" % (c.kh, c.kw, c.cin, c.cout, kind)


def conv_flop(c, *a, n=None, **k):
    n = n if n is not None else a[0].shape[0]
    return 2.0 * n * c.oh * c.ow * c.cout * c.cin * c.kh * c.kw


def lin_name(kind):
    return lambda l: "linear_%d->%d_%s" % (l.K, l.N, kind)


def lin_flop(l, *a, **k):
    return 2.0 * a[-1] * l.K * l.N



def run(B=4096, CAP=1024, ITERS=3, T_loop=None, encoder="nav1d", loop_iters=None, encoder_streams=True):
    """One measurement; returns the record as a dict (bench.py's `nav` sub-record calls this).  The per-operator timing wraps
    ops.Conv / ops.Linear for the duration of the call only."""
    saved = {(cls, k): getattr(cls, k) for cls in (ops.Conv, ops.Linear) for k in ("forward", "dgrad", "wgrad")}
    saved.update({(ops.Conv, k): getattr(ops.Conv, k) for k in ("forward_pool", "dgrad_pooled", "wgrad_pooled")})
    events.clear()
    try:
        return _run(B, CAP, ITERS, T_loop, encoder, loop_iters, encoder_streams)
    finally:
        for (cls, k), fn in saved.items():
            setattr(cls, k, fn)


def _run(B, CAP, ITERS, T_loop, encoder, loop_iters=None, encoder_streams=True):
    ops.Conv.forward = timed(conv_name("fwd"), conv_flop, ops.Conv.forward)
    ops.Conv.dgrad = timed(conv_name("dgrad"), conv_flop, ops.Conv.dgrad)
    ops.Conv.wgrad = timed(conv_name("wgrad"), conv_flop, ops.Conv.wgrad)
    # the pooled forms (conv + ReLU + max-pool in one launch; gradients from d(pooled) + decision bytes) under the same names
    ops.Conv.forward_pool = timed(conv_name("fwd"), conv_flop, ops.Conv.forward_pool)
    ops.Conv.dgrad_pooled = timed(conv_name("dgrad"), conv_flop, ops.Conv.dgrad_pooled)
    ops.Conv.wgrad_pooled = timed(conv_name("wgrad"), conv_flop, ops.Conv.wgrad_pooled)
    ops.Linear.forward = timed(lin_name("fwd"), lin_flop, ops.Linear.forward)
    ops.Linear.dgrad = timed(lin_name("dgrad"), lin_flop, ops.Linear.dgrad)
    ops.Linear.wgrad = timed(lin_name("wgrad"), lin_flop, ops.Linear.wgrad)

    env = {"env_type": "gym", "env_name": "PongNoFrameskip-v4", "env_num": 8, "discrete_action": False, "act_dim": 2,
           "image_batch": 1, "ped_sim": {"total": 3}}
    cfg = BaseConfig(types.SimpleNamespace(task="bench", ip="127.0.0.1"), env)
    cfg.TASK_TYPE = "robot_nav"
    cfg_nn = ConfigNN(env)
    cfg_nn.TRAINING_ITER_TIME = ITERS
    # encoder = "navped": ONE shared NavPedPreNet(4) (runner/utils.py:98-102, SHARE_CNN_NET; the encoder of the GAIL nav configuration)
    cfg_nn.SHARE_CNN_NET = encoder == "navped"
    net = create_net({"config": cfg, "config_nn": cfg_nn, "config_env": env}, max_batch=CAP)
    net.encoder_streams = bool(encoder_streams)   # False: one stream, so that the per-operator events do not overlap
    net.deferred_stats = os.environ.get("NAV_SYNC_EVERY_ITER") != "1"   # one host sync per update, as bench.py's headline (nn/generic.py:learn)
    g = torch.Generator(device="cuda")
    g.manual_seed(4)
    if encoder == "navped":
        states = [torch.rand((B, 1, 48, 48), device="cuda", generator=g), torch.randn((B, 9), device="cuda", generator=g),
                  (torch.rand((B, 3, 48, 48), device="cuda", generator=g) < 0.15).float()]
    else:
        states = [torch.rand((B, 1, 960), device="cuda", generator=g), torch.randn((B, 5), device="cuda", generator=g),
                  (torch.rand((B, 3, 48, 48), device="cuda", generator=g) < 0.15).float()]
    (dist, _), values = net([s[:CAP] for s in states])
    exp = Experience(states=states, advs=torch.randn(B, device="cuda", generator=g),
                     actions=torch.randn((B, 2), device="cuda", generator=g), old_logps=torch.full((B,), -2.0, device="cuda"),
                     values=torch.randn((1, B), device="cuda", generator=g))
    for _ in net.learn(exp):  # warm-up (also sizes every lazily created buffer)
        pass
    events.clear()
    torch.cuda.synchronize()
    t0 = time.time()
    for _ in net.learn(exp):
        pass
    torch.cuda.synchronize()
    wall = (time.time() - t0) / ITERS
    agg = defaultdict(lambda: [0.0, 0.0, 0])
    for name, flop, e0, e1 in events:
        a = agg[name]
        a[0] += e0.elapsed_time(e1)
        a[1] += flop
        a[2] += 1
    rows = {k: {"ms_per_iter": round(v[0] / ITERS, 3), "tflops": round(v[1] / (v[0] * 1e-3) / 1e12, 2), "calls_per_iter": v[2] // ITERS}
            for k, v in sorted(agg.items(), key=lambda kv: -kv[1][0])}
    tot_ms = sum(v[0] for v in agg.values()) / ITERS
    tot_flop = sum(v[1] for v in agg.values()) / ITERS
    # ---- whole loop at the BASELINE config-4 shape: 512 envs, T steps of acting + bootstrap + GAE + one PPO update ----
    loop = None
    if T_loop is not None:
        from ddrl4nav_amd.agent import StateRollout
        N, T = 512, int(T_loop)
        ro = StateRollout(net, N, [(1, 960), (5,), (3, 48, 48)], horizon=T)
        ro.states[0].copy_(torch.rand(ro.states[0].shape, device="cuda", generator=g))
        ro.states[1].copy_(torch.randn(ro.states[1].shape, device="cuda", generator=g))
        ro.states[2].copy_((torch.rand(ro.states[2].shape, device="cuda", generator=g) < 0.15).float())
        u = torch.rand((T, N), device="cuda", generator=g)
        ro.rewards.copy_(torch.where(u < 0.01, -1.0, torch.where(u > 0.99, 1.0, 0.0)))
        ro.dones.copy_((torch.rand((T, N), device="cuda", generator=g) < 1.0 / 800).to(torch.uint8))

        def one_loop(t_act, iters):
            e0, e1, e2 = (torch.cuda.Event(enable_timing=True) for _ in range(3))
            net.training_iter_time = iters
            e0.record()
            for t in range(t_act):
                ro.act(t)
            ro.bootstrap()
            ro.finish()
            e1.record()
            for _ in net.learn(ro.batch()):
                pass
            e2.record()
            torch.cuda.synchronize()
            return e0.elapsed_time(e1), e1.elapsed_time(e2)

        L_IT = int(loop_iters) if loop_iters else ITERS
        one_loop(T, 1)              # warm-up: the whole rollout, one PPO iteration over the whole batch (every buffer sized)
        events.clear()
        act_ms, upd_ms = one_loop(T, L_IT)
        net.training_iter_time = ITERS
        loop = {"envs": N, "horizon": T, "ppo_iters": L_IT, "samples": N * T, "micro_batch": CAP,
                "acting_plus_gae_ms": round(act_ms, 1), "update_ms": round(upd_ms, 1),
                "ppo_iter_ms": round(upd_ms / L_IT, 2), "ppo_iter_ms_per_4096_samples": round(upd_ms / L_IT * 4096 / (N * T), 3),
                "env_steps_per_s": round(N * T / ((act_ms + upd_ms) * 1e-3), 1),
                "acting_env_steps_per_s": round(N * (T + 1) / (act_ms * 1e-3), 1)}
    return ({"workload": ("robot_nav: shared NavPedPreNet(4) + GaussionActor(2), PPO iteration" if encoder == "navped" else
                          "robot_nav: NavPreNet1D x2 + GaussionActor(2), PPO iteration"), "B": B, "micro_batch": CAP,
                      "whole_loop": loop,
                      "ms_per_ppo_iter_wall": round(wall * 1e3, 2), "gemm_ops_ms_per_iter": round(tot_ms, 2),
                      "algorithmic_tflops_over_gemm_ops": round(tot_flop / (tot_ms * 1e-3) / 1e12, 2),
                      "samples_per_s": round(B / wall, 1), "ops": rows})


if __name__ == "__main__":
    print(json.dumps(run(int(sys.argv[1]) if len(sys.argv) > 1 else 4096, int(sys.argv[2]) if len(sys.argv) > 2 else 1024,
                         int(sys.argv[3]) if len(sys.argv) > 3 else 3, int(sys.argv[4]) if len(sys.argv) > 4 and sys.argv[4] != "-" else None,
                         sys.argv[5] if len(sys.argv) > 5 else "nav1d", int(sys.argv[6]) if len(sys.argv) > 6 and sys.argv[6] != "-" else None,
                         encoder_streams=not (len(sys.argv) > 7 and sys.argv[7] == "one-stream"))))
