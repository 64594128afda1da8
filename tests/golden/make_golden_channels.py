#!/usr/bin/env python3
"""Golden vectors for AtariPreNet with OTHER frame stacks than four (reference nn/atari_encoder.py:12-14 takes `num_inputs`),
by IMPORTING THE REFERENCE (build container only; outputs are stored, nothing is copied):

  f20_channels.npz   for C in (1, 3): frames u8 [16, C, 84, 84] (seeded), actions / old_logps / advs / rets; the reference's
                     forward (probs, p_hat, logits, value, logp) and its PPO.learn for three iterations on the two-encoder net
                     (four losses per iteration, float64 checksums of every parameter after iteration 3)

Weights come from the recipe (ddrl4nav_amd/utils/recipe.py: make_weights(seed=20 + C, num_inputs=C)).
Usage:  python tests/golden/make_golden_channels.py
"""
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)
REF = os.environ.get("DDRL_REFERENCE", "/root/reference")


def build(C, weights):
    from USTC_lab.config.config_nn import ConfigNN
    from USTC_lab.nn import AtariPreNet, CategoricalActor, Critic, PPO
    cfg = types.SimpleNamespace(MIDDLE_REDIS_HOST="127.0.0.1", MIDDLE_REDIS_PORT=0, TASK_NAME="golden", MODULE_KEY="MODEL", DEVICE="cpu")
    cfg_nn = ConfigNN({"discrete_action": True, "discrete_actions": list(range(6))})
    cfg_nn.DEVICE = "cpu"
    actor = CategoricalActor(action_output_dim=6, device="cpu", soft_max_grid=True, last_input_dim=512,
                             pre=AtariPreNet(C, last_output_dim=512, device="cpu"), nn_dtype=torch.float32)
    critic = Critic(device="cpu", last_input_dim=512, pre=AtariPreNet(C, last_output_dim=512, device="cpu"))
    net = PPO(actor, critic, None, None, cfg, cfg_nn).to("cpu")
    net.load_state_dict({k: torch.from_numpy(v.copy()) for k, v in weights.items()}, strict=True)
    net.actor_optim = torch.optim.Adam(net.actor.parameters(), cfg_nn.ACTOR_LEARNING_RATE)
    net.critic_optim = torch.optim.Adam(net.critic.parameters(), cfg_nn.CRITIC_LEARNING_RATE)
    net.update_time = 0
    return net, cfg_nn


def main():
    from make_golden import _install_stubs
    _install_stubs()
    sys.path.insert(0, REF)
    from ddrl4nav_amd.utils.recipe import make_weights, param_specs
    from USTC_lab.data import Experience
    torch.set_num_threads(1)
    out = {}
    B = 16
    for C in (1, 3):
        rng = np.random.default_rng(2000 + C)
        frames = rng.integers(0, 256, size=(B, C, 84, 84), dtype=np.uint8)
        frames[B - 1] = 87                      # a flat, Pong-like sample: the leaky branch is exercised
        frames[B - 1, :, 10:26, 4:8] = 147
        weights = make_weights(seed=20 + C, num_inputs=C)
        net, cfg_nn = build(C, weights)
        assert [k for k, _ in net.named_parameters()] == [n for n, _, _ in param_specs(C)]
        x = torch.tensor(frames / 255.0, dtype=torch.float32)   # f64 divide -> f32, as forward.py:102-104
        acts = rng.integers(0, 6, size=B).astype(np.float32)
        with torch.no_grad():
            (dist, logp), values = net([x], torch.from_numpy(acts))
            (probs, _), _ = net([x], play_mode=True)
        old = (logp.numpy() + 0.1 * rng.normal(size=B)).astype(np.float32)
        advs = rng.normal(size=B).astype(np.float32)
        rets = (values[0].numpy()[:, 0] + rng.normal(size=B)).astype(np.float32)
        p = "c%d/" % C
        out.update({p + "frames": frames, p + "actions": acts, p + "old_logps": old, p + "advs": advs, p + "rets": rets,
                    p + "probs": probs.numpy(), p + "p_hat": dist.probs.numpy(), p + "logits": dist.logits.numpy(),
                    p + "value": values[0].numpy()[:, 0], p + "logp": logp.numpy()})
        cfg_nn.TRAINING_ITER_TIME = 3
        net.training_iter_time = 3
        exp = Experience(states=[x.numpy()], advs=advs, actions=acts, old_logps=old, values=rets.reshape(1, B))
        exp.to_tensor(dtype=torch.float32, device="cpu")
        losses = []
        for ld, _, _ in net.learn(exp):
            losses.append([ld["PpoTotalLoss"], ld["ActorLoss"], ld["VLoss"], ld["EntLoss"]])
        out[p + "losses"] = np.asarray(losses, np.float64)
        for name, q in net.named_parameters():
            a = q.detach().double().numpy()
            out[p + "it3/sum/" + name] = np.float64(a.sum())
            out[p + "it3/l2/" + name] = np.float64(np.sqrt((a ** 2).sum()))
        print("C=%d losses" % C, out[p + "losses"][-1])
    np.savez_compressed(os.path.join(HERE, "f20_channels.npz"), **out)
    print("f20_channels.npz", os.path.getsize(os.path.join(HERE, "f20_channels.npz")), "B")


if __name__ == "__main__":
    main()
