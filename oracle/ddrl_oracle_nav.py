"""CPU ORACLE for the non-Atari nets.  TEST INFRASTRUCTURE ONLY.

Plain PyTorch-CPU fp32 restatement of the reference's nav / MLP encoders, Gaussian actor and the
PPO loss / optimise block over them; pinned against golden vectors produced by importing the
reference (tests/golden/make_golden_nav.py -> f13..f15, tests/test_oracle_golden.py).  Only
tests/, __graft_entry__.smoke() and bench.py's cpu_baseline may import this file.

Reference sites restated (under /root/reference/USTC_lab):
  * nn/nav_encoder.py:12-44    NavPreNet      * nn/nav_encoder.py:47-83   NavPedPreNet
  * nn/nav_encoder.py:86-128   NavPreNet1D    * nn/mlp_encoder.py:12-29   MLPPreNet
  * nn/utils.py:10-20          mlp            * nn/actor.py:43-70         GaussionActor
  * nn/actor.py:73-101         CategoricalActor   * nn/critic.py:8-21     Critic
  * nn/ppo.py:72-129           PPO.forward / learn (shared and non-shared branch)
"""
import math
import time

import numpy as np
import torch
import torch.nn.functional as F
from torch import nn

from oracle.ddrl_oracle import (ACTOR_LR, CLIP_GRAD_NUM, CRITIC_LR, DUEL_PPO_CLIP, ENT_LOSS_THETA, PPO_CLIP, SHARED_LR,
                                TRAINING_ITER_TIME, V_LOSS_THETA, categorical_entropy, categorical_log_prob,
                                categorical_logits)


def mlp(spec):
    layers = []
    for i, o, af in spec:
        layers.append(nn.Linear(i, o))
        if af == "relu":
            layers.append(nn.ReLU())
    return nn.Sequential(*layers)


def _act(net, key, z):
    """relu(z).  Test hook (not part of the reference): with ``net.sub = {key: a_k}`` -- the ReLU OUTPUT another implementation
    computed at this site for the same batch -- the forward value becomes a_k and the backward mask (a_k > 0), while the gradient
    still flows into z.  Downstream max-pools then route by a_k, i.e. every ReLU / max-pool DECISION of the backward is the other
    implementation's (an activation within fp32 noise of a decision boundary comes out on either side depending on the summation
    order; one such flip moves every conv-weight gradient upstream of it by ~1e-3 of its size), the arithmetic stays the oracle's."""
    rec = getattr(net, "record", None)
    if rec is not None:          # test hook: the pre-activations of this forward, for the "decisions differ only within fp32 noise" checks
        rec[key] = z.detach()
    sub = getattr(net, "sub", None)
    if sub is None or key not in sub:
        return F.relu(z)
    a_k = sub[key].to(z.dtype).reshape(z.shape)
    z_sub = torch.where(a_k > 0, a_k, -torch.ones_like(a_k))
    return F.relu(z + (z_sub - z).detach())


def _pool3(net, x):
    x = F.max_pool2d(_act(net, "conv1", net.conv1(x)), 2, stride=2)
    x = F.max_pool2d(_act(net, "conv2", net.conv2(x)), 2, stride=2)
    x = F.max_pool2d(_act(net, "conv3", net.conv3(x)), 2, stride=2)
    return x.view(x.size(0), -1)


class MLPPreNet(nn.Module):
    def __init__(self, input_dim=4, last_output_dim=512):
        super().__init__()
        self.fc0 = mlp([(input_dim, last_output_dim, "relu")])

    def forward(self, state):
        return _act(self, "fc0", self.fc0[0](state[0]))      # fc0 = Linear + ReLU (mlp_encoder.py:18)


class NavPreNet(nn.Module):
    def __init__(self, image_channel=1):
        super().__init__()
        self.conv1 = nn.Conv2d(image_channel, 64, 3, stride=1, padding=(1, 1))
        self.conv2 = nn.Conv2d(64, 128, 3, stride=1, padding=(1, 1))
        self.conv3 = nn.Conv2d(128, 256, 3, stride=1, padding=(1, 1))
        self.fc0 = mlp([(256 * 6 * 6, 512, "relu")])
        self.fc1 = mlp([(512 + 9, 512, "relu")])
        self.fc2 = nn.Linear(512, 512)

    def image(self, state):
        return state[0]

    def forward(self, state):
        x = _act(self, "fc0", self.fc0[0](_pool3(self, self.image(state))))
        return self.fc2(_act(self, "fc1", self.fc1[0](torch.cat((x, state[1]), dim=1))))


class NavPedPreNet(NavPreNet):
    def __init__(self, image_channel=4):
        super().__init__(image_channel)

    def image(self, state):
        return torch.cat([state[0], state[2]], dim=1)


class NavPreNet1D(nn.Module):
    def __init__(self, image_channel=3):
        super().__init__()
        self.conv1 = nn.Conv2d(image_channel, 64, 7, stride=1, padding=(1, 1))
        self.conv2 = nn.Conv2d(64, 128, 5, stride=1, padding=(1, 1))
        self.conv3 = nn.Conv2d(128, 256, 3, stride=1, padding=(1, 1))
        self.conv1d1 = nn.Conv1d(1, 32, 5, 2, "valid")
        self.conv1d2 = nn.Conv1d(32, 32, 3, 2, "valid")
        self.fc_1d = mlp([(7616, 256, "relu")])
        self.fc0 = mlp([(6400, 512, "relu")])
        self.fc1 = mlp([(256 + 512 + 5, 512, "relu")])
        self.fc2 = nn.Linear(512, 512)

    def forward(self, state):
        l = self.conv1d2(self.conv1d1(state[0]))
        l = _act(self, "fc_1d", self.fc_1d[0](l.view(l.shape[0], -1)))
        x = _act(self, "fc0", self.fc0[0](_pool3(self, state[2])))
        return self.fc2(_act(self, "fc1", self.fc1[0](torch.cat((l, x, state[1]), dim=1))))


class _Actor(nn.Module):
    def __init__(self, n_out, pre, gaussian):
        super().__init__()
        self.pre = pre
        self.actor_linear = nn.Linear(512, n_out)
        if gaussian:
            self.log_std = nn.Parameter(-0.5 * torch.ones(n_out))


class _Critic(nn.Module):
    def __init__(self, pre):
        super().__init__()
        self.critic_linear = nn.Linear(512, 1)
        self.pre = pre


class OracleNet(nn.Module):
    """PPO module tree (ppo.py:26-28): prenet (shared) | actor(pre) | critic(pre)."""

    def __init__(self, make_pre, n_out, gaussian, shared):
        super().__init__()
        self.prenet = make_pre() if shared else None
        self.actor = _Actor(n_out, None if shared else make_pre(), gaussian)
        self.critic = _Critic(None if shared else make_pre())
        self.gaussian, self.shared = gaussian, shared
        self.update_time = 0

    def load_weights(self, weights):
        self.load_state_dict({k: torch.from_numpy(np.ascontiguousarray(v)) for k, v in weights.items()}, strict=True)

    def features(self, states):
        if self.shared:
            h = self.prenet(states)
            return h, h
        return self.actor.pre(states), self.critic.pre(states)

    def forward(self, states, actions):
        """-> (dist_out, logp [n], entropy (per element), value [n,1])"""
        ha, hc = self.features(states)
        out = self.actor.actor_linear(ha)
        v = self.critic.critic_linear(hc)
        if self.gaussian:
            pi = torch.distributions.Normal(out, torch.exp(self.actor.log_std))
            return out, pi.log_prob(actions).sum(axis=-1), pi.entropy(), v
        probs = F.softmax(out, dim=-1)
        p_hat, logits = categorical_logits(probs)
        return probs, categorical_log_prob(logits, actions), categorical_entropy(p_hat, logits), v

    def make_optims(self):
        if self.shared:
            return (torch.optim.Adam(self.parameters(), SHARED_LR),)
        return (torch.optim.Adam(self.actor.parameters(), ACTOR_LR), torch.optim.Adam(self.critic.parameters(), CRITIC_LR))


def losses(net, states, actions, old_logps, advs, rets):
    _, log_p, ent_el, v = net(states, actions)
    ratio = torch.exp(log_p - old_logps)
    m = torch.min(ratio * advs, torch.clamp(ratio, 1.0 - PPO_CLIP, 1.0 + PPO_CLIP) * advs)
    actor_loss = -torch.mean(torch.where(advs > 0, m, torch.max(m, DUEL_PPO_CLIP * advs)))
    v_loss = torch.mean((rets - v.squeeze()) ** 2) / 2
    ent = torch.mean(ent_el)
    return actor_loss + v_loss * V_LOSS_THETA - ent * ENT_LOSS_THETA, actor_loss, v_loss, ent


def learn(net, optims, states, actions, old_logps, advs, rets, iters=TRAINING_ITER_TIME):
    for _ in range(iters):
        t0 = time.time()
        total, actor_loss, v_loss, ent = losses(net, states, actions, old_logps, advs, rets)
        for o in optims:
            o.zero_grad()
        if net.shared:
            total.backward()
        else:
            actor_loss.backward()
            v_loss.backward()
        gnorm = torch.nn.utils.clip_grad_norm_(net.parameters(), CLIP_GRAD_NUM)
        for o in optims:
            o.step()
        net.update_time += 1
        yield ({"PpoTotalLoss": total.item(), "ActorLoss": actor_loss.item(), "VLoss": v_loss.item(), "EntLoss": ent.item(),
                "PpoBackUpTime": time.time() - t0, "GradNorm": float(gnorm)}, net.update_time, True)


def box_muller(u1, u2):
    """Sampler contract of the Gaussian head: z = sqrt(-2 ln(1 - u1)) cos(2 pi u2) on the
    counter-based uniform stream (indices 2i, 2i+1), action = mu + std * z."""
    u1 = np.float32(1.0) - np.asarray(u1, np.float32)
    return (np.sqrt(np.float32(-2.0) * np.log(u1)) * np.cos(np.float32(2.0 * math.pi) * np.asarray(u2, np.float32))).astype(np.float32)
