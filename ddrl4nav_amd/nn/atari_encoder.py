"""AtariPreNet: parameter holder of the 3 x conv + FC encoder (mirror of
USTC_lab/nn/atari_encoder.py:11-23).  Its arithmetic runs in the HIP kernels of the owning PPO
(csrc/conv2.hip, wgrad2.hip, fc2.hip); calling the module on its own is not a product path."""
from torch import nn

from ddrl4nav_amd.nn.base import PreNet


class AtariPreNet(PreNet):
    def __init__(self, num_inputs=1, last_output_dim=512, device='cpu'):
        super().__init__()
        self.device = device
        self.conv1 = nn.Conv2d(num_inputs, 32, 8, stride=4)
        self.conv2 = nn.Conv2d(32, 64, 4, stride=2)
        self.conv3 = nn.Conv2d(64, 64, 3, stride=1)
        self.linear = nn.Linear(3136, 512)
        assert self.linear.out_features == last_output_dim

    def forward(self, x):
        raise RuntimeError("AtariPreNet runs inside ddrl4nav_amd.nn.PPO (HIP kernels); wrap it in a PPO net")
