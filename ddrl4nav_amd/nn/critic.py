"""Critic head (mirror of USTC_lab/nn/critic.py:6-21); note critic_linear is registered before
pre, which fixes the parameter order of the flat arena."""
from torch import nn


class Critic(nn.Module):
    def __init__(self, device='cpu', last_input_dim=512, pre=None):
        super().__init__()
        self.device = device
        self.critic_linear = nn.Linear(last_input_dim, 1)
        self.pre = pre

    def forward(self, x):
        raise RuntimeError("the critic runs inside ddrl4nav_amd.nn.PPO (HIP kernels)")
