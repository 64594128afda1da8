"""AtariPreNet: parameter holder of the Pong / Atari image encoder.

Shape contract (reference USTC_lab/nn/atari_encoder.py:11-32, which computes it with three
``F.leaky_relu(conv)`` calls and a linear layer):

    frames [n, C, 84, 84] --conv1 8x8 /4--> [n, 32, 20, 20] --conv2 4x4 /2--> [n, 64, 9, 9]
           --conv3 3x3 /1--> [n, 64, 7, 7] --flatten--> [n, 3136] --linear--> [n, 512]

with leaky-ReLU (slope 0.01) after every convolution and no activation after the linear layer.
The attribute names (``conv1`` ``conv2`` ``conv3`` ``linear``) are the reference's, so checkpoints
and the Redis weight blob interchange.  The computation itself lives in the fused HIP kernels of
the owning PPO net: csrc/conv2.hip (forward, data gradients), csrc/wgrad2.hip (weight gradients),
csrc/fc2.hip (the 3136 -> 512 layer).
"""
from torch import nn

from ddrl4nav_amd.nn.base import PreNet

_GEOMETRY = (("conv1", 32, 8, 4), ("conv2", 64, 4, 2), ("conv3", 64, 3, 1))  # name, out channels, kernel, stride
_FLAT = 64 * 7 * 7


class AtariPreNet(PreNet):
    def __init__(self, num_inputs=1, last_output_dim=512, device="cpu"):
        PreNet.__init__(self)
        channels = int(num_inputs)
        for name, out_ch, kernel, stride in _GEOMETRY:
            self.add_module(name, nn.Conv2d(channels, out_ch, kernel, stride=stride))
            channels = out_ch
        self.add_module("linear", nn.Linear(_FLAT, 512))
        if int(last_output_dim) != 512:
            raise ValueError("the encoder's feature width is fixed at 512 (AC_INPUT_DIM), got %r" % (last_output_dim,))
        self.device = device

    def forward(self, x):
        raise RuntimeError("AtariPreNet is evaluated by the fused HIP kernels of ddrl4nav_amd.nn.PPO; wrap it in a PPO net")
