"""One GAIL.learn of the nav configuration (BASELINE config 5: shared NavPedPreNet(4) generator + discriminator over a copy of the
encoder + the GAIL critic; reference nn/GAIL.py:19-66, runner/utils.py:161-168) on synthetic data: the discriminator's WGAN step
(policy batch + expert batches) and the PPO iterations with the extra value head, timed with HIP events.
    python tools/bench_gail_nav.py [B=4096] [CAP=4096] [ITERS=3] [EXPERT_BATCHES=4 of 128]"""
import json
import os
import sys
import time
import types

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ddrl4nav_amd.config import BaseConfig, ConfigNN  # noqa: E402
from ddrl4nav_amd.data import Experience  # noqa: E402
from ddrl4nav_amd.runner import create_net  # noqa: E402


def run(B=4096, CAP=4096, ITERS=3, NEXP=4):
    env = {"env_type": "robot_nav", "env_name": "robot_nav", "env_num": 8, "discrete_action": True, "discrete_actions": list(range(5)),
           "image_batch": 1, "ped_sim": {"total": 3}}
    cfg = BaseConfig(types.SimpleNamespace(task="gail", ip="127.0.0.1"), env)
    cfg.TASK_TYPE = "robot_nav"
    cfg_nn = ConfigNN(env)
    cfg_nn.NETWORK_TYPE, cfg_nn.SHARE_CNN_NET = "gail", True
    cfg_nn.TRAINING_ITER_TIME = ITERS
    g = torch.Generator(device="cuda")
    g.manual_seed(5)

    def batch(n):
        return [torch.rand((n, 1, 48, 48), device="cuda", generator=g), torch.randn((n, 9), device="cuda", generator=g),
                (torch.rand((n, 3, 48, 48), device="cuda", generator=g) < 0.15).float()]

    nb = cfg_nn.GAN_D_BATCH_SIZE
    expert = [(batch(nb), torch.randint(0, 5, (nb, 1), device="cuda", generator=g).float()) for _ in range(NEXP)]
    net = create_net({"config": cfg, "config_nn": cfg_nn, "config_env": env}, max_batch=CAP, expert_data=expert)
    states = batch(B)
    exp = Experience(states=states, advs=torch.randn(B, device="cuda", generator=g),
                     actions=torch.randint(0, 5, (B,), device="cuda", generator=g).float(), old_logps=torch.full((B,), -1.6, device="cuda"),
                     values=torch.randn((2, B), device="cuda", generator=g))
    for _ in net.learn(exp):   # warm-up (sizes every lazily created buffer)
        pass
    torch.cuda.synchronize()
    t0 = time.time()
    for _ in net.learn(exp):
        pass
    torch.cuda.synchronize()
    whole = time.time() - t0
    # the discriminator's step alone
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in net.discriminator.learn(exp):   # one WGAN step: the policy batch and ONE expert batch (GAIL.py:75-92 breaks after the first)
        pass
    e1.record()
    torch.cuda.synchronize()
    d_ms = e0.elapsed_time(e1)
    return {"workload": "GAIL over robot_nav: shared NavPedPreNet(4) + CategoricalActor(5) + 2 critics, discriminator over a copy of the encoder",
            "B": B, "micro_batch": CAP, "ppo_iters": ITERS, "expert_batches": "%d x %d" % (NEXP, nb),
            "gail_learn_ms": round(whole * 1e3, 2), "discriminator_step_ms": round(d_ms, 2),
            "ppo_iter_ms": round((whole * 1e3 - d_ms) / ITERS, 2), "samples_per_s": round(B / whole, 1)}


if __name__ == "__main__":
    a = [int(x) for x in sys.argv[1:]]
    print(json.dumps(run(*a)))
