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

Two ways to run it:
  * inside ``nn.PPO`` (the Atari fast path): the whole iteration is one C call (ddrl_ppo_iter);
  * as an operator-composed encoder (``build`` / ``forward_dev`` / ``backward_dev`` / ``pack``, the protocol of
    nn/generic.py) through the encoder-only entry points ddrl_encoder_forward / ddrl_encoder_backward -- used where
    the reference composes the encoder with other modules: the GAIL discriminator's ``pre`` (GAIL.py:27,65-66) and a
    generator whose heads carry the GAIL critic.
"""
from ctypes import byref, c_int64, c_void_p

import torch
from torch import nn

from ddrl4nav_amd.nn.base import PreNet
from ddrl4nav_amd.utils.staging import to_device

_GEOMETRY = (("conv1", 32, 8, 4), ("conv2", 64, 4, 2), ("conv3", 64, 3, 1))  # name, out channels, kernel, stride
_FLAT = 64 * 7 * 7


def frames_u8(states, device):
    """states[0] as a contiguous uint8 device tensor.  Float inputs are the reference's
    float32(uint8/255.0) frames (forward.py:102-104); x*255 rounds back to the byte exactly."""
    x = states[0] if isinstance(states, (list, tuple)) else states
    x = torch.as_tensor(x)
    if x.dtype != torch.uint8:
        x = torch.round(x.to(torch.float32) * 255.0).clamp_(0, 255).to(torch.uint8)
    return to_device(x, device).contiguous()


class AtariPreNet(PreNet):
    n_inputs = 1
    raw_u8 = True  # forward_dev takes the uint8 frames themselves (the kernels apply float32(u8 / 255.0))

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
        self._ctx = None

    def forward(self, x):
        raise RuntimeError("AtariPreNet is evaluated by the fused HIP kernels of ddrl4nav_amd.nn.PPO; wrap it in a PPO net")

    # ---- operator-composed use (protocol of nn/generic.py:GenericPreNet) -------------------------------------------
    def build(self, cap, device):
        """Bind an encoder-only libddrl_hip context to this module's parameters, which by now are views into the owner's
        flat arena (``p.data``) with gradient views beside them (``p.grad_view``)."""
        from ddrl4nav_amd import _lib
        from ddrl4nav_amd._lib import check
        lib = _lib.load()
        ps = list(self.parameters())
        base_p, base_g, off = ps[0].data_ptr(), ps[0].grad_view.data_ptr(), 0
        for p in ps:  # the kernels address the encoder as ONE run: conv1.weight ... linear.bias
            if p.data_ptr() != base_p + 4 * off or p.grad_view.data_ptr() != base_g + 4 * off:
                raise ValueError("AtariPreNet parameters must be contiguous views of one flat arena")
            off += p.numel()
        if base_p % 16 or base_g % 16:
            raise ValueError("the encoder's slice of the parameter / gradient arena must be 16-byte aligned")
        self._lib, self._device, self.cap = lib, torch.device(device), int(cap)
        self._cfg = _lib.default_config(max_batch=self.cap, n_actions=6, in_channels=self.conv1.in_channels, share_cnn_net=1)
        wb = c_int64()
        check(lib.ddrl_workspace_bytes(byref(self._cfg), byref(wb)))
        self._workspace = torch.empty(wb.value, dtype=torch.uint8, device=self._device)
        ctx = c_void_p()
        check(lib.ddrl_ctx_create(byref(self._cfg), c_void_p(base_p), c_void_p(base_g), c_void_p(0), c_void_p(0),
                                  c_void_p(self._workspace.data_ptr()), wb.value, byref(ctx)))
        self._ctx = ctx
        h, dh = c_void_p(), c_void_p()
        check(lib.ddrl_encoder_buffers(ctx, byref(h), byref(dh)))
        f32 = self._workspace.view(torch.float32)
        view = lambda ptr: f32[(ptr.value - self._workspace.data_ptr()) // 4:][:self.cap * 512].view(self.cap, 512)
        self.h, self._dh = view(h), view(dh)
        self._frames = None

    def pack(self):
        from ddrl4nav_amd._lib import check
        check(self._lib.ddrl_params_changed(self._ctx))

    def forward_dev(self, states, n):
        from ddrl4nav_amd._lib import check
        self._frames = frames_u8(states, self._device)
        assert self._frames.shape[0] == n
        check(self._lib.ddrl_encoder_forward(self._ctx, c_void_p(self._frames.data_ptr()), n,
                                             c_void_p(torch.cuda.current_stream().cuda_stream)))
        return self.h

    def backward_dev(self, dh, n):
        from ddrl4nav_amd._lib import check
        if dh.data_ptr() != self._dh.data_ptr():
            self._dh[:n].copy_(dh[:n])
        check(self._lib.ddrl_encoder_backward(self._ctx, c_void_p(self._frames.data_ptr()), n,
                                              c_void_p(torch.cuda.current_stream().cuda_stream)))

    def __del__(self):
        try:
            if self._ctx is not None:
                self._lib.ddrl_ctx_destroy(self._ctx)
                self._ctx = None
        except Exception:
            pass
