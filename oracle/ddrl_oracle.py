"""CPU ORACLE for the DDRL4NAV actor-learner hot path.  TEST INFRASTRUCTURE ONLY.

This file restates, in plain PyTorch-CPU fp32 + NumPy, the arithmetic that the reference
performs on its Forward / Env / Backward path.  It is the *checker* used by ``tests/``,
``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of ``bench.py``; nothing under
``ddrl4nav_amd/`` may import it and the product path never falls back to it.

Pinned: every function here is checked against golden vectors produced by importing the
reference itself (``tests/golden/make_golden.py`` -> ``tests/golden/*.npz``), see
``tests/test_oracle_golden.py``.

Reference sites restated (all under /root/reference/USTC_lab):
  * nn/atari_encoder.py:12-32      AtariPreNet   -> ``Encoder``
  * nn/actor.py:73-101             CategoricalActor -> ``ActorHead`` / ``categorical_*``
  * nn/critic.py:8-21              Critic        -> ``CriticHead``
  * nn/ppo.py:72-75                PPO.forward   -> ``OraclePPO.forward``
  * nn/ppo.py:77-146               PPO.learn     -> ``learn`` (both the shared and the non-shared branch)
  * runner/utils.py:136-143        SHARE_CNN_NET=True net -> ``OracleSharedPPO``
  * agent/agent.py:124-140         Agents._accumulate_rewards -> ``gae``
  * agent/statistics.py:118-123    Status.update_reward_status -> ``episode_returns``
  * env/gym_env/wrapper/warputils.py:300 + server/forward.py:102-104  u8/255.0 -> f32 -> ``u8_lut``
"""
import time
from collections import OrderedDict

import numpy as np
import torch
import torch.nn.functional as F
from torch import nn

# ConfigNN defaults (reference config/config_nn.py:27-57)
GAMMA = 0.99
LANDA = 0.95
ACTOR_LR = 5e-5
CRITIC_LR = 1e-3
SHARED_LR = 2e-4
V_LOSS_THETA = 1.0
ENT_LOSS_THETA = 0.05
PPO_CLIP = 0.2
DUEL_PPO_CLIP = 3
TRAINING_ITER_TIME = 10
CLIP_GRAD_NUM = 0.5


def u8_lut():
    """float32(uint8 / 255.0): the env wrapper divides in float64 (warputils.py:300) and the
    forward/backward servers cast to float32 (forward.py:102-104, experience.py:56-62)."""
    return (np.arange(256, dtype=np.uint8) / 255.0).astype(np.float32)


def frames_to_f32(frames_u8):
    return torch.from_numpy(u8_lut()[np.asarray(frames_u8)])


class _LeakyForced(torch.autograd.Function):
    """leaky_relu whose BACKWARD slope follows a given decision tensor instead of sign(z) (tests only: lets a
    gradient comparison be made under identical ReLU decisions when a pre-activation sits within fp32 noise of 0)."""

    @staticmethod
    def forward(ctx, z, positive):
        ctx.save_for_backward(positive)
        return F.leaky_relu(z)

    @staticmethod
    def backward(ctx, g):
        (positive,) = ctx.saved_tensors
        return g * torch.where(positive, torch.ones_like(g), torch.full_like(g, 0.01)), None


class Encoder(nn.Module):
    """AtariPreNet (atari_encoder.py:12-32): 3 x conv + leaky_relu(0.01), flatten, linear.

    Test hooks (not part of the reference): ``last_z`` keeps the three pre-activations of the latest forward;
    ``forced`` = [pos1, pos2, pos3] (bool tensors) makes the backward use those decisions (_LeakyForced); ``forced_seq`` = a list
    of such triples, one per successive forward (the discriminator step runs the encoder on two batches);
    ``tap`` = {} collects, at the next backward, d loss / d (conv1 pre-activation) as "z1", d loss / d (conv2 / conv3
    pre-activation) as "z2" / "z3" and d loss / d (encoder output) as "h" (per-sample gradient tensors)."""

    def __init__(self, num_inputs=4):
        super().__init__()
        self.conv1 = nn.Conv2d(num_inputs, 32, 8, stride=4)
        self.conv2 = nn.Conv2d(32, 64, 4, stride=2)
        self.conv3 = nn.Conv2d(64, 64, 3, stride=1)
        self.linear = nn.Linear(3136, 512)
        self.forced = None
        self.forced_seq = None   # [[pos1, pos2, pos3], ...]: decisions of the successive forwards of one step (GAIL's D runs two), cyclic
        self._forward_calls = 0
        self.last_z = None
        self.tap = None

    def _tap(self, t, key):
        if self.tap is not None and t.requires_grad:
            t.register_hook(lambda g, d=self.tap, k=key: d.__setitem__(k, g.detach().clone()))

    def _act(self, z, k):
        self.last_z.append(z.detach())
        self._tap(z, "z%d" % (k + 1))
        return F.leaky_relu(z) if self.forced is None else _LeakyForced.apply(z, self.forced[k])

    def forward(self, x):
        self.last_z = []
        if self.forced_seq is not None:
            self.forced = self.forced_seq[self._forward_calls % len(self.forced_seq)]
            self._forward_calls += 1
        x = self._act(self.conv1(x), 0)
        x = self._act(self.conv2(x), 1)
        x = self._act(self.conv3(x), 2)
        h = self.linear(x.view(x.size(0), -1))
        self._tap(h, "h")
        return h


class ActorHead(nn.Module):
    """CategoricalActor with its own encoder (actor.py:73-101); attribute order = ``pre`` then
    ``actor_linear`` so that named_parameters() matches the reference."""

    def __init__(self, n_actions=6, num_inputs=4):
        super().__init__()
        self.pre = Encoder(num_inputs)
        self.actor_linear = nn.Linear(512, n_actions)

    def forward(self, x):
        return F.softmax(self.actor_linear(self.pre(x)), dim=-1)


class CriticHead(nn.Module):
    """Critic (critic.py:8-21); ``critic_linear`` is registered before ``pre`` in the reference."""

    def __init__(self, num_inputs=4):
        super().__init__()
        self.critic_linear = nn.Linear(512, 1)
        self.pre = Encoder(num_inputs)

    def forward(self, x):
        return self.critic_linear(self.pre(x))


EPS = float(torch.finfo(torch.float32).eps)


def categorical_logits(probs):
    """torch.distributions.Categorical(probs=p): p_hat = p / sum(p); logits = log(clamp(p_hat, eps, 1-eps))."""
    p_hat = probs / probs.sum(-1, keepdim=True)
    eps = torch.finfo(probs.dtype).eps  # torch.distributions.utils.clamp_probs: eps of the probs' dtype
    return p_hat, torch.log(torch.clamp(p_hat, eps, 1.0 - eps))


def categorical_log_prob(logits, act):
    return logits.gather(-1, act.long().unsqueeze(-1)).squeeze(-1)


def categorical_entropy(p_hat, logits):
    min_real = torch.finfo(logits.dtype).min
    return -(torch.clamp(logits, min=min_real) * p_hat).sum(-1)


class OraclePPO(nn.Module):
    """Default (SHARE_CNN_NET=False) Pong net of runner/utils.py:122-134 + ppo.py:18-59."""

    def __init__(self, n_actions=6, num_inputs=4):
        super().__init__()
        self.actor = ActorHead(n_actions, num_inputs)
        self.critic = CriticHead(num_inputs)
        self.update_time = 0

    def load_weights(self, weights):
        sd = OrderedDict((k, torch.from_numpy(np.ascontiguousarray(v))) for k, v in weights.items())
        self.load_state_dict(sd, strict=True)

    def forward(self, x):
        """x: float32 [n,4,84,84] -> (probs [n,A], p_hat, logits, value [n,1])."""
        probs = self.actor(x)
        p_hat, logits = categorical_logits(probs)
        return probs, p_hat, logits, self.critic(x)

    def make_optims(self):
        return (torch.optim.Adam(self.actor.parameters(), ACTOR_LR),
                torch.optim.Adam(self.critic.parameters(), CRITIC_LR))


class OracleSharedPPO(nn.Module):
    """SHARE_CNN_NET=True net of runner/utils.py:136-143: one ``prenet`` feeding a pre-less
    CategoricalActor and Critic (ppo.py:72-75); one Adam over everything (ppo.py:39)."""

    def __init__(self, n_actions=6, num_inputs=4):
        super().__init__()
        self.prenet = Encoder(num_inputs)
        self.actor = nn.Module()
        self.actor.actor_linear = nn.Linear(512, n_actions)
        self.critic = nn.Module()
        self.critic.critic_linear = nn.Linear(512, 1)
        self.update_time = 0

    load_weights = OraclePPO.load_weights

    def forward(self, x):
        h = self.prenet(x)
        probs = F.softmax(self.actor.actor_linear(h), dim=-1)
        p_hat, logits = categorical_logits(probs)
        return probs, p_hat, logits, self.critic.critic_linear(h)

    def make_optims(self):
        return (torch.optim.Adam(self.parameters(), SHARED_LR),)


def ppo_losses(net, x, actions, old_logps, advs, rets, smooth_l1=False):
    """Loss block of ppo.py:82-108.  Returns (total, actor_loss, v_loss, entropy) tensors."""
    _, p_hat, logits, v = net(x)
    log_p = categorical_log_prob(logits, actions)
    ratio = torch.exp(log_p - old_logps)
    m = torch.min(ratio * advs, torch.clamp(ratio, 1.0 - PPO_CLIP, 1.0 + PPO_CLIP) * advs)
    actor_loss = -torch.mean(torch.where(advs > 0, m, torch.max(m, DUEL_PPO_CLIP * advs)))
    if smooth_l1:  # SMOOTH_L1_LOSS (ppo.py:53-54): vlossf(data.values[0], values[0].squeeze())
        v_loss = F.smooth_l1_loss(rets, v.squeeze())
    else:
        v_loss = torch.mean((rets - v.squeeze()) ** 2) / 2
    ent = torch.mean(categorical_entropy(p_hat, logits))
    total = actor_loss + v_loss * V_LOSS_THETA - ent * ENT_LOSS_THETA
    return total, actor_loss, v_loss, ent


def learn(net, optims, x, actions, old_logps, advs, rets, iters=TRAINING_ITER_TIME, hook=None, smooth_l1=False):
    """Optimise block of ppo.py:110-129 as a generator like ppo.py:142.  Two optimisers = the
    default non-shared branch (actor_loss.backward(); v_loss.backward()); one optimiser = the
    shared branch (total_loss.backward())."""
    for _ in range(iters):
        t0 = time.time()
        total, actor_loss, v_loss, ent = ppo_losses(net, x, actions, old_logps, advs, rets, smooth_l1)
        for o in optims:
            o.zero_grad()
        if len(optims) == 1:
            total.backward()
        else:
            actor_loss.backward()
            v_loss.backward()
        gnorm = torch.nn.utils.clip_grad_norm_(net.parameters(), CLIP_GRAD_NUM)
        if hook is not None:
            hook(net, float(gnorm))
        for o in optims:
            o.step()
        net.update_time += 1
        yield ({"PpoTotalLoss": total.item(), "ActorLoss": actor_loss.item(), "VLoss": v_loss.item(),
                "EntLoss": ent.item(), "PpoBackUpTime": time.time() - t0, "GradNorm": float(gnorm)},
               net.update_time, True)


def gae(values, rewards, dones, gamma=GAMMA, landa=LANDA):
    """Agents._accumulate_rewards (agent.py:124-140) for one value head.

    values  float32 [T+1, N]  (row T = bootstrap value of the (T+1)-th stored step)
    rewards float32 [>=T, N]  (rewards_step[t][0])
    dones   uint8   [>=T, N]  (experiences[t].dones[0])
    returns (adv [T,N], ret [T,N]) float32.  Operation order follows the reference exactly:
    ``g *= (1-d)``; ``g = (gamma*landa)*g + ((gamma*V_next)*(1-d) - V_t + r_t)``.
    """
    values = np.asarray(values, dtype=np.float32)
    T = values.shape[0] - 1
    discounts = np.array([gamma], dtype=np.float32).reshape(1, 1)
    g = np.zeros_like(values[0:1])
    next_v = values[T:T + 1]
    adv = np.empty((T, values.shape[1]), np.float32)
    ret = np.empty((T, values.shape[1]), np.float32)
    for t in reversed(range(T)):
        d = dones[t:t + 1]
        v = values[t:t + 1]
        g = g * (1 - d)
        g = discounts * landa * g + (discounts * next_v * (1 - d) - v + rewards[t:t + 1])
        next_v = v
        ret[t] = (v + g)[0]
        adv[t] = g[0] * 1.0
    return adv, ret


def episode_returns(rewards, dones):
    """Status.update_reward_status (statistics.py:118-123) run over a [T,N] stream; returns the
    per-step 'latest finished episode reward' trace and the final running sums."""
    rewards = np.asarray(rewards, np.float32)
    N = rewards.shape[1]
    rsum = np.zeros(N, np.float32)
    rep = np.zeros(N, np.float32)
    trace = np.empty_like(rewards)
    for t in range(rewards.shape[0]):
        d = dones[t]
        rsum = rsum + rewards[t]
        rep = rep * (1 - d) + rsum * d
        rsum = rsum * (1 - d)
        trace[t] = rep
    return trace, rsum


def inverse_cdf_sample(p_hat, u):
    """Sampler contract of the build (the reference's torch.multinomial stream is RNG-specific,
    forward.py:137): action = first index whose running float32 sum of p_hat exceeds u."""
    p = np.asarray(p_hat, np.float32)
    c = np.zeros(p.shape[0], np.float32)
    act = np.full(p.shape[0], p.shape[1] - 1, np.int64)
    done = np.zeros(p.shape[0], bool)
    for j in range(p.shape[1]):
        c = (c + p[:, j]).astype(np.float32)
        hit = (~done) & (u < c)
        act[hit] = j
        done |= hit
    return act
