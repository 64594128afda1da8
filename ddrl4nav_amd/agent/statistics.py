"""EpisodeReturns: the reward bookkeeping of the reference's ``Status`` (USTC_lab/agent/statistics.py:118-123,
``update_reward_status``) for a device-resident rollout.

The reference updates two per-env float32 vectors after every env step on the host:
    rewards_sum += rewards
    rewards_episode = rewards_episode * (1 - dones) + rewards_sum * dones     # the latest finished episode's return
    rewards_sum *= (1 - dones)
Here the rewards and dones of a whole rollout already sit in the experience pool ([T, N], agent/rollout.py), so the same
recurrence runs once per rollout in one small kernel (ddrl_episode_returns: one lane per env, the reference's operation
order, fp32) and the two vectors stay on the GPU between rollouts.  Pinned by golden F6 (the reference's own Status run)."""
from ctypes import c_void_p

import torch

from ddrl4nav_amd import _lib
from ddrl4nav_amd._lib import check


class EpisodeReturns:
    def __init__(self, n_envs, device):
        self.lib = _lib.load()
        self.N = int(n_envs)
        dev = torch.device(device)
        self.rewards_sum = torch.zeros(self.N, dtype=torch.float32, device=dev)      # Status.rewards_sum
        self.rewards_episode = torch.zeros(self.N, dtype=torch.float32, device=dev)  # Status.rewards_episode
        self.episodes_finished = torch.zeros(self.N, dtype=torch.int32, device=dev)

    def update(self, rewards, dones, trace=None):
        """rewards f32 [T, N], dones u8 [T, N] (device, contiguous): T calls of update_reward_status in one launch.
        `trace` f32 [T, N] (optional) receives rewards_episode after every step."""
        T, N = rewards.shape
        if N != self.N or tuple(dones.shape) != (T, N):
            raise ValueError("rewards %s / dones %s do not match %d envs" % (tuple(rewards.shape), tuple(dones.shape), self.N))
        if rewards.dtype != torch.float32 or dones.dtype != torch.uint8 or not rewards.is_cuda:
            raise TypeError("rewards must be float32 and dones uint8 device tensors")
        if not (rewards.is_contiguous() and dones.is_contiguous()):
            raise ValueError("rewards / dones must be contiguous")
        if trace is not None and (tuple(trace.shape) != (T, N) or trace.dtype != torch.float32 or not trace.is_contiguous()):
            raise ValueError("trace must be a contiguous float32 [T, N] tensor")
        p = lambda t: c_void_p(0) if t is None else c_void_p(t.data_ptr())
        check(self.lib.ddrl_episode_returns(p(rewards), p(dones), T, N, p(self.rewards_sum), p(self.rewards_episode), p(trace),
                                            p(self.episodes_finished), c_void_p(torch.cuda.current_stream().cuda_stream)))
        return self.rewards_episode

    def mean_return(self):
        """Mean of the latest finished episode's return over the envs that have finished one (host float; synchronises)."""
        done = self.episodes_finished > 0
        k = int(done.sum().item())
        return float(self.rewards_episode[done].mean().item()) if k else float("nan")
