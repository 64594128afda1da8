"""CPU ORACLE for the GAIL path.  TEST INFRASTRUCTURE ONLY.

Plain PyTorch-CPU restatement of the reference's discriminator, its WGAN-style update and the PPO update with the
extra GAIL critic; pinned against golden vectors produced by importing the reference itself
(tests/golden/make_golden_gail.py -> f16_gail_classical.npz, f17_gail_atari.npz, f18_gae_two_rows.npz;
tests/golden/make_golden_gail_nav.py -> f22_gail_navped.npz: GAIL over a shared NavPedPreNet; checked in tests/test_oracle_golden.py).  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline may import this file.

Reference sites restated (under /root/reference/USTC_lab):
  * nn/GAIL.py:19-71     Discriminator: mlp(GAN_D_MLP_LIST) on cat(pre(state), action)
  * nn/GAIL.py:73-94     Discriminator.learn: mean D(generator batch) - mean D(expert batch), clip_grad_norm_(WGAN_CLIP_GRAD_NUM),
                         RMSprop(lr GAN_D_LEARNING_RATE, alpha 0.9), StepLR(250, 0.95)
  * nn/GAIL.py:103-158   GAIL: module tree, forward routing, learn = D update then generator update
  * nn/ppo.py:61-62,72-75,95-107   add_critic, two-critic forward, v_loss = ppov_loss + gailv_loss on data.values[-1]
  * runner/utils.py:161-168        gail_critic = deepcopy(critic), D_prenet = deepcopy(prenet)
  * agent/agent.py:97-101,124-140  _accumulate_rewards with one discount per value row
The GAIL critic is registered on the GAIL module only and is in no optimiser (ppo.py:39 builds Adam before add_critic
appends to a plain list): it is never trained.  The oracle keeps that behaviour.
"""
import time

import numpy as np
import torch
import torch.nn.functional as F
from torch import nn

from oracle import ddrl_oracle as O
from oracle import ddrl_oracle_nav as N

GAN_D_LEARNING_RATE = 5e-5   # config_nn.py:93
WGAN_CLIP_GRAD_NUM = 0.01    # config_nn.py:99
GAN_DISCOUNT = 0.99          # config_nn.py:91


class AtariPre(O.Encoder):
    """AtariPreNet called the way PPO.forward calls a shared prenet: on the state LIST (atari_encoder.py:26)."""

    def forward(self, state):
        return super().forward(state[0])


class OracleDiscriminator(nn.Module):
    def __init__(self, pre, mlp_list):
        super().__init__()
        self.mlp_layer = N.mlp(mlp_list)   # registered before `pre` (GAIL.py:26-27)
        self.pre = pre

    def forward(self, x):
        state, action = x
        # Test hook (not part of the reference): ``sub_seq`` = one {site: ReLU output} dict per coming forward call of a NAV encoder
        # (ddrl_oracle_nav._act): the discriminator step runs `pre` twice, on the policy batch and on the expert batch.
        seq = getattr(self, "sub_seq", None)
        if seq:
            self.pre.sub = seq.pop(0)
        try:
            h = self.pre(state)
        finally:
            if seq is not None:
                self.pre.sub = None
        return self.mlp_layer(torch.cat((h, action), dim=-1))


class OracleGAIL(nn.Module):
    """GAIL(generator=PPO(actor, critic, prenet), discriminator, gail_critic) with SHARE_CNN_NET=True."""

    def __init__(self, make_pre, n_out, gaussian, d_mlp_list, action_dim=1):
        super().__init__()
        self.generator = N.OracleNet(make_pre, n_out, gaussian, shared=True)
        self.discriminator = OracleDiscriminator(make_pre(), d_mlp_list)
        self.gail_critic = N._Critic(None)
        self.action_dim = action_dim
        self.g_updates = self.d_updates = 0

    def load_weights(self, weights):
        self.load_state_dict({k: torch.from_numpy(np.ascontiguousarray(v)) for k, v in weights.items()}, strict=True)

    def make_optims(self):
        """(Adam over the GENERATOR's parameters only, RMSprop over the discriminator's, its StepLR)."""
        g = torch.optim.Adam(self.generator.parameters(), O.SHARED_LR)
        d = torch.optim.RMSprop(self.discriminator.parameters(), lr=GAN_D_LEARNING_RATE, alpha=0.9)
        return g, d, torch.optim.lr_scheduler.StepLR(d, step_size=250, gamma=0.95)

    def forward(self, states, actions):
        """-> (dist_out, logp, entropy per element, [value0 [n,1], value_gail [n,1]])"""
        g = self.generator
        h = g.prenet(states)
        out = g.actor.actor_linear(h)
        values = [g.critic.critic_linear(h), self.gail_critic.critic_linear(h)]
        if g.gaussian:
            pi = torch.distributions.Normal(out, torch.exp(g.actor.log_std))
            return out, pi.log_prob(actions).sum(axis=-1), pi.entropy(), values
        probs = F.softmax(out, dim=-1)
        p_hat, logits = O.categorical_logits(probs)
        return probs, O.categorical_log_prob(logits, actions), O.categorical_entropy(p_hat, logits), values

    def d_reward(self, states, actions):
        return self.discriminator((states, actions.reshape(actions.shape[0], self.action_dim)))


def g_losses(net, states, actions, old_logps, advs, rets2):
    """Loss block of ppo.py:82-108 with gail_critic=True: rets2 [2, B] = data.values."""
    _, log_p, ent_el, values = net(states, actions)
    ratio = torch.exp(log_p - old_logps)
    m = torch.min(ratio * advs, torch.clamp(ratio, 1.0 - O.PPO_CLIP, 1.0 + O.PPO_CLIP) * advs)
    actor_loss = -torch.mean(torch.where(advs > 0, m, torch.max(m, O.DUEL_PPO_CLIP * advs)))
    ppov = torch.mean((rets2[0] - values[0].squeeze()) ** 2) / 2
    gailv = torch.mean((rets2[-1] - values[-1].squeeze()) ** 2) / 2
    v_loss = ppov + 0 + gailv                      # ppov_loss + rndv_loss + gailv_loss (ppo.py:107)
    ent = torch.mean(ent_el)
    return actor_loss + v_loss * O.V_LOSS_THETA - ent * O.ENT_LOSS_THETA, actor_loss, v_loss, ent


def d_step(net, d_optim, d_sched, states, actions, expert_states, expert_actions):
    """One pass of Discriminator.learn's inner loop (GAIL.py:76-91)."""
    t0 = time.time()
    D = net.discriminator
    g_loss = torch.mean(D((states, actions.reshape(actions.shape[0], net.action_dim))))
    expert_loss = -torch.mean(D((expert_states, expert_actions)))
    total = g_loss + expert_loss
    d_optim.zero_grad()
    total.backward()
    gnorm = torch.nn.utils.clip_grad_norm_(D.parameters(), WGAN_CLIP_GRAD_NUM)
    d_optim.step()
    d_sched.step()
    net.d_updates += 1
    return {"Gail[D]BackUpTime": time.time() - t0, "Gail[D]Loss": total.item(), "GradNorm": float(gnorm)}, net.d_updates, True


def learn(net, optims, states, actions, old_logps, advs, rets2, expert_states, expert_actions, iters=O.TRAINING_ITER_TIME,
          d_epochs=1):
    """GAIL.learn (GAIL.py:149-158): discriminator epochs (yielded with last=False), then the generator's PPO
    iterations (last=True)."""
    g_optim, d_optim, d_sched = optims
    for _ in range(d_epochs):
        item, ut, _ = d_step(net, d_optim, d_sched, states, actions, expert_states, expert_actions)
        yield item, ut, False
    for _ in range(iters):
        yield g_step(net, g_optim, states, actions, old_logps, advs, rets2)


def g_step(net, g_optim, states, actions, old_logps, advs, rets2):
    """One generator iteration of GAIL.learn (the PPO iteration of ppo.py:82-129 with the GAIL critic's value loss)."""
    t0 = time.time()
    total, actor_loss, v_loss, ent = g_losses(net, states, actions, old_logps, advs, rets2)
    g_optim.zero_grad()
    total.backward()   # SHARE_CNN_NET branch (ppo.py:110-117); the GAIL critic's .grad accumulates, nobody reads it
    gnorm = torch.nn.utils.clip_grad_norm_(net.generator.parameters(), O.CLIP_GRAD_NUM)
    g_optim.step()
    net.g_updates += 1
    return ({"PpoTotalLoss": total.item(), "ActorLoss": actor_loss.item(), "VLoss": v_loss.item(), "EntLoss": ent.item(),
             "PpoBackUpTime": time.time() - t0, "GradNorm": float(gnorm)}, net.g_updates, True)


def gae_rows(values, rewards, dones, discounts, landa=O.LANDA):
    """Agents._accumulate_rewards (agent.py:124-140) with K value rows: values [T+1, K, N], rewards [>=T, K, N],
    dones [>=T, K, N] uint8, discounts [K].  Returns (adv [T, N] = row 0 of the accumulated sum, ret [T, K, N])."""
    values = np.asarray(values, np.float32)
    T = values.shape[0] - 1
    disc = np.asarray(discounts, np.float32).reshape(-1, 1)
    g = np.zeros_like(rewards[0], dtype=np.float32)
    next_v = values[T]
    adv = np.empty((T, values.shape[2]), np.float32)
    ret = np.empty((T,) + values.shape[1:], np.float32)
    for t in reversed(range(T)):
        g = g * (1 - dones[t])
        g = disc * landa * g + (disc * next_v * (1 - dones[t]) - values[t] + rewards[t])
        next_v = values[t]
        ret[t] = values[t] + g
        adv[t] = g[0] * 1.0
    return adv, ret
