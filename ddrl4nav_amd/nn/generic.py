"""Encoders and PPO for the reference's non-Atari nets, composed from the operator-level C ABI.

The reference builds these from torch modules (USTC_lab/nn/nav_encoder.py:12-128,
mlp_encoder.py:12-29, actor.py:43-70, critic.py:8-21, ppo.py:17-146) and lets autograd derive the
backward.  Here every module keeps the reference's parameter names and shapes (so ``state_dict``
and the Redis weight blob interchange), but its forward and backward are explicit sequences of
ddrl_op_* launches (csrc/gconv.hip, glinear.hip, gheads.hip, heads.hip, optim.hip) on buffers it
owns.  torch supplies memory, streams and input staging (dtype casts, channel concat) only.

The Atari encoder does not go through this file: it has its own fused kernels (nn/ppo.py).
"""
import time
from ctypes import byref, c_int64, c_void_p

import torch
from torch import nn

from ddrl4nav_amd import _lib
from ddrl4nav_amd._lib import STATS_FLOATS, HeadsDesc, check
from ddrl4nav_amd.data import Experience
from ddrl4nav_amd.nn.base import Basenn, PreNet
from ddrl4nav_amd.ops import Conv, Linear, maxpool2_idx, maxpool2_backward_idx, sample_amax, _p, _st
from ddrl4nav_amd.utils.staging import to_device

FEAT = 512
# per-sample magnitudes travel from a tensor's producer to its consumer (include/ddrl.h); False: a pre-pass in front of every consumer.
# Default for encoders BUILT from here on: each encoder keeps its own `producer_amax` (like GenericPPO.encoder_streams), so flipping
# this mid-training does not reach live nets.
PRODUCER_AMAX = True


def _pad4(k):
    return (k + 3) // 4 * 4


def mlp(input_mlp):
    """Parameter holder with the module tree of the reference's ``mlp`` helper
    (USTC_lab/nn/utils.py:10-20): Linear at index 0, activation at index 1."""
    layers = []
    for in_dim, out_dim, af in input_mlp:
        layers.append(nn.Linear(in_dim, out_dim, bias=True))
        if af == "relu":
            layers.append(nn.ReLU())
        elif af == "sigmoid":
            raise NotImplementedError("sigmoid activations are not used by any encoder of the reference")
    return nn.Sequential(*layers)


class _Dense:
    """nn.Linear (+ReLU) bound to arena views: forward / backward through ddrl_op_linear_*."""

    def __init__(self, module, relu, cap, device):
        self.m, self.relu = module, relu
        self.K, self.N = module.in_features, module.out_features
        self.op = Linear(self.K, self.N, max_n=cap, device=device)
        self._scale_buffers(cap, device)

    def _scale_buffers(self, cap, device):
        # per-row magnitudes of the layer's input and of d(output) (include/ddrl.h "per-sample magnitudes"): found once per pass and
        # shared by the operators that read the same tensor (forward + weight gradient; data + weight gradient).  A caller that knows
        # the tensor's producer hands the producer's `out_amax` in instead (in_amax / dout_amax below) and the pre-pass is skipped.
        share = self.op.uses_planes(cap)
        self.in_sc = torch.empty((cap,), dtype=torch.float32, device=device) if share else None
        self.dout_sc = torch.empty((cap,), dtype=torch.float32, device=device) if share else None

    def _in_amax(self, x, ld_in, n, given=None):
        if self.in_sc is None or not self.op.uses_planes(n):
            return None
        return given if given is not None else self.op.row_amax(x, ld_in, self.K, n, self.in_sc)

    def _dout_amax(self, dout, ld_dout, n, given=None):
        if self.dout_sc is None or not self.op.uses_planes(n):
            return None
        return given if given is not None else self.op.row_amax(dout, ld_dout, self.N, n, self.dout_sc)

    def pack(self):
        self.op.pack(self.m.weight.data)

    def forward(self, x, ld_in, out, ld_out, n, in_amax=None):
        self._fwd_sc = self._in_amax(x, ld_in, n, in_amax)     # kept for the weight gradient of the same pass
        return self.op.forward(x, ld_in, self.m.bias.data, self.relu, out, ld_out, n, in_amax=self._fwd_sc)

    def backward(self, x, ld_in, dout, ld_dout, n, din=None, ld_din=0, mask_src=None, ld_mask=0, dout_amax=None, din_amax=None,
                 amax_cols=None):
        """dout = gradient w.r.t. this layer's PRE-activation output (the consumer applied the ReLU mask).  din_amax (zeroed by the
        caller): raised to every row's largest |din| over the columns amax_cols -- what the layer below will ask for."""
        ds = self._dout_amax(dout, ld_dout, n, dout_amax)
        self.op.wgrad(x, ld_in, dout, ld_dout, self.m.weight.grad_view, self.m.bias.grad_view, n, in_amax=getattr(self, "_fwd_sc", None),
                      dout_amax=ds)
        if din is not None:
            self.op.dgrad(dout, ld_dout, mask_src, ld_mask, din, ld_din, n, dout_amax=ds, din_amax=din_amax, amax_cols=amax_cols)


class _PaddedDense(_Dense):
    """nn.Linear whose out_features is not a multiple of 4 (the discriminator's 1-wide score layer, GAIL.py:26):
    the operator runs on a zero-padded copy [Np][K] of the weights; forward writes [n][Np] (columns >= N are the
    padded zeros + zero bias), the weight gradient is computed at the padded width and its first N rows are copied
    to the parameter's gradient view."""

    def __init__(self, module, relu, cap, device):
        self.m, self.relu = module, relu
        self.K, self.N_real = module.in_features, module.out_features
        self.N = _pad4(self.N_real)
        self.op = Linear(self.K, self.N, max_n=cap, device=device)
        self._scale_buffers(cap, device)
        f = dict(dtype=torch.float32, device=device)
        self.wpad, self.bpad = torch.zeros((self.N, self.K), **f), torch.zeros(self.N, **f)
        self.dwpad, self.dbpad = torch.zeros((self.N, self.K), **f), torch.zeros(self.N, **f)

    def pack(self):
        self.wpad[:self.N_real].copy_(self.m.weight.data)
        self.bpad[:self.N_real].copy_(self.m.bias.data)
        self.op.pack(self.wpad)

    def forward(self, x, ld_in, out, ld_out, n, in_amax=None):
        self._fwd_sc = self._in_amax(x, ld_in, n, in_amax)
        return self.op.forward(x, ld_in, self.bpad, self.relu, out, ld_out, n, in_amax=self._fwd_sc)

    def backward(self, x, ld_in, dout, ld_dout, n, din=None, ld_din=0, mask_src=None, ld_mask=0, dout_amax=None, din_amax=None,
                 amax_cols=None):
        ds = self._dout_amax(dout, ld_dout, n, dout_amax)
        self.op.wgrad(x, ld_in, dout, ld_dout, self.dwpad, self.dbpad, n, in_amax=getattr(self, "_fwd_sc", None), dout_amax=ds)
        self.m.weight.grad_view.copy_(self.dwpad[:self.N_real])
        # bias gradient = column sums of dout, correctly rounded (see csrc/gail.hip:colsum_kernel)
        check(_lib.load().ddrl_op_colsum(_p(dout), ld_dout, n, self.N_real, _p(self.m.bias.grad_view), _st()))
        if din is not None:
            self.op.dgrad(dout, ld_dout, mask_src, ld_mask, din, ld_din, n, dout_amax=ds, din_amax=din_amax, amax_cols=amax_cols)


def dense_layer(module, relu, cap, device):
    return (_Dense if module.out_features % 4 == 0 else _PaddedDense)(module, relu, cap, device)


class _ConvPool:
    """Conv2d + ReLU + max_pool2d(2) (nav_encoder.py:27-31).  Layers whose kernels pool in their epilogue (`fused`: the conv layers of the
    three nav encoders) keep the pooled map and one decision byte per window and run their backward from d(pooled); every other layer keeps
    the full-resolution ReLU output and d(pre-activation) around the stand-alone pool operators."""

    def __init__(self, module, h, w, cap, device, pool=True, relu=True):
        self.m, self.pool, self.relu = module, pool, relu
        kh, kw = module.kernel_size
        pad = module.padding if isinstance(module.padding, tuple) else (module.padding, module.padding)
        if len(pad) == 1:
            pad = (0, pad[0])
        self.op = Conv(module.in_channels, h, w, module.out_channels, kh, kw, stride=module.stride[0], pad=pad, max_n=cap,
                       device=device)
        oh, ow = self.op.oh, self.op.ow
        self.oh, self.ow = oh, ow
        f = dict(dtype=torch.float32, device=device)
        # layers whose kernels pool in their epilogue never write the full-resolution activations (ddrl_op_conv_forward_pool)
        self.fused = bool(pool and relu and self.op.has_forward_pool())
        self._a = None if self.fused else torch.empty((cap, module.out_channels, oh, ow), **f)   # relu(conv)
        # ... and their backward reads d(pooled) + the decision bytes (ddrl_op_conv_*_pooled): no full-resolution gradient either
        self.fused_bwd = self.fused
        # per-sample magnitudes of the layer's input and of d(pooled) (include/ddrl.h): taken from the tensors' producers where the
        # encoder hands them in (in_amax / dp_amax below), else found by a pre-pass into these buffers; one array per tensor serves the
        # operators that read it (forward + weight gradient; data + weight gradient)
        self.in_sc = self.dp_sc = None
        if self.fused and self.op.pooled_uses_scales():
            self.in_sc = torch.empty((cap,), **f)
            self.dp_sc = torch.empty((cap,), **f)
        self.dz = None if self.fused_bwd else torch.empty((cap, module.out_channels, oh, ow), **f)   # d(loss)/d(pre-activation)
        if pool:
            self.p = torch.empty((cap, module.out_channels, oh // 2, ow // 2), **f)
            self.dp = torch.empty_like(self.p)
            # one decision byte per window (first maximum + ReLU sign): the backward reads it instead of the full-resolution `a`
            self.code = torch.empty((cap, module.out_channels, oh // 2, ow // 2), dtype=torch.uint8, device=device)
        self.out_shape = (module.out_channels, oh // 2, ow // 2) if pool else (module.out_channels, oh, ow)

    def pack(self):
        w = self.m.weight.data
        self.op.pack(w if w.dim() == 4 else w.unsqueeze(2))

    @property
    def a(self):
        """relu(conv) at full resolution.  A fused layer keeps only the pooled map and the decisions: what comes back then is zero
        except at each window's first maximum, which holds the pooled value -- the same pooled map, the same max-pool routing and
        the same ReLU mask under it as the activations the kernel had in its registers (what tests/parity_util.py substitutes into
        the float64 yardstick)."""
        return self.relu_output(self.p.shape[0] if self.fused else self._a.shape[0])

    def relu_output(self, n):
        """`a` of the first n samples (the surrogate of a fused layer is built for those n only)."""
        if not self.fused:
            return self._a[:n]
        p, am = self.p[:n], self.code[:n] & 3
        a = torch.zeros((n, p.shape[1], self.oh, self.ow), dtype=torch.float32, device=p.device)
        for k in range(4):
            a[:, :, (k >> 1)::2, (k & 1)::2] = torch.where(am == k, p, torch.zeros_like(p))
        return a

    def forward(self, x, n, in_amax=None, out_amax=None):
        """in_amax: the samples' largest |x| as x's producer left them (None: pre-pass); out_amax (zeroed by the caller): raised to the
        samples' largest |output| for the layer that reads this one's output."""
        if self.fused:
            self._in_amax = None
            if self.in_sc is not None:
                self._in_amax = in_amax if in_amax is not None else sample_amax(x, n, self.in_sc)
            self.op.forward_pool(x, self.m.bias.data, self.p, self.code, n=n, in_amax=self._in_amax, out_amax=out_amax)
            return self.p
        # a pooled block's consumer reads the pooled map: every pooled value is one of `a`'s, so max|a| bounds it (equal under ReLU on
        # even maps); the row must be raised here too, the next block trusts it (conv1 with a non-specialised cin runs this path)
        self.op.forward(x, self.m.bias.data, self.relu, out=self._a, n=n, out_amax=out_amax)
        if not self.pool:
            return self._a
        maxpool2_idx(self._a[:n], out=self.p, code=self.code)
        return self.p

    def out_grad_buffer(self):
        """Where the consumer writes d(loss)/d(output of this block)."""
        return self.dp if self.pool else self.dz

    def backward(self, x, n, din=None, dp_amax=None, din_amax=None):
        """dp_amax: the samples' largest |d(pooled)| as its producer left them (None: pre-pass); din_amax (zeroed by the caller): raised
        to the samples' largest |din|."""
        if self.fused_bwd:
            dpm = None
            if self.dp_sc is not None:
                dpm = dp_amax if dp_amax is not None else sample_amax(self.dp, n, self.dp_sc)
            self.op.wgrad_pooled(x, self.dp, self.code, self.m.weight.grad_view, self.m.bias.grad_view, n=n,
                                 in_amax=getattr(self, "_in_amax", None), dpool_amax=dpm)
            if din is not None:
                self.op.dgrad_pooled(self.dp, self.code, din=din, n=n, dpool_amax=dpm, din_amax=din_amax)
            return
        if self.pool:
            maxpool2_backward_idx(self.dp[:n], self.code, self.oh, self.ow, dz=self.dz)
        self.op.wgrad(x, self.dz, self.m.weight.grad_view, self.m.bias.grad_view, n=n)
        if din is not None:
            self.op.dgrad(self.dz, din=din, n=n)


class GenericPreNet(PreNet):
    """Base of the operator-composed encoders: subclasses define the torch parameter tree in
    __init__ (reference names) and implement build / forward_dev / backward_dev."""
    n_inputs = 1

    def forward(self, x):
        raise RuntimeError("%s runs inside ddrl4nav_amd.nn.PPO (HIP operators); wrap it in a PPO net" % type(self).__name__)

    def _f(self, cap, *shape):
        return torch.empty((cap,) + shape, dtype=torch.float32, device=self._device)

    def pack(self):
        for b in self._blocks:
            b.pack()


class MLPPreNet(GenericPreNet):
    """mlp_encoder.py:12-29: fc0 = Linear(input_dim, last_output_dim) + ReLU on state[0]."""

    def __init__(self, input_dim=4, last_output_dim=128):
        super().__init__()
        self.fc0 = mlp([(input_dim, last_output_dim, "relu")])
        self.input_dim, self.out_dim = input_dim, last_output_dim

    def build(self, cap, device):
        self._device = device
        self.ld = _pad4(self.input_dim)
        self.x = torch.zeros((cap, self.ld), dtype=torch.float32, device=device)
        self.d0 = _Dense(self.fc0[0], True, cap, device)
        self.h = self._f(cap, self.out_dim)
        self._blocks = [self.d0]

    def forward_dev(self, states, n):
        self.x[:n, :self.input_dim].copy_(states[0].reshape(n, -1))
        self.d0.forward(self.x, self.ld, self.h, self.out_dim, n)
        return self.h

    def backward_dev(self, dh, n):
        # dh arrives w.r.t. the ReLU output of fc0 (the encoder's last layer): mask it first
        check(_lib.load().ddrl_op_relu_mask(_p(dh), self.out_dim, _p(self.h), self.out_dim, n, self.out_dim, _st()))
        self.d0.backward(self.x, self.ld, dh, self.out_dim, n)


class _NavBase(GenericPreNet):
    """conv stack + fc0 (+ReLU), cat with the vector state, fc1 (+ReLU), fc2 (nav_encoder.py:34-44)."""

    def _build_tail(self, cap, device, flat_dim, vec_dim, extra=0):
        self.vec_dim, self.extra = vec_dim, extra
        self.cat_k = extra + 512 + vec_dim
        self.ld_cat = _pad4(self.cat_k)
        self.cat = torch.zeros((cap, self.ld_cat), dtype=torch.float32, device=device)
        self.dcat = torch.zeros((cap, self.ld_cat), dtype=torch.float32, device=device)
        self.d0 = _Dense(self.fc0[0], True, cap, device)
        self.d1 = _Dense(self.fc1[0], True, cap, device)
        self.d2 = _Dense(self.fc2, False, cap, device)
        self.f1 = self._f(cap, 512)
        self.df1 = self._f(cap, 512)
        self.h = self._f(cap, 512)
        self.flat_dim = flat_dim

    def _links(self, *rows):
        """The arena rows handed from producers to consumers; producer_amax = False (tools/ab_nav_amax.py, same-box A/B) hands out
        None instead, i.e. every consumer runs its own pre-pass as in round 4."""
        return rows if self.producer_amax else (None,) * len(rows)

    def _amax_arena(self, cap, device, n_fwd, n_bwd):
        """Per-sample magnitudes that travel from a tensor's producer to its consumer (include/ddrl.h): one row per tensor, zeroed at
        the start of a pass (one fill per pass and encoder), raised by the producers' epilogues."""
        self._amax_f = torch.zeros((n_fwd, cap), dtype=torch.float32, device=device)
        self._amax_b = torch.zeros((n_bwd, cap), dtype=torch.float32, device=device)
        self.producer_amax = bool(PRODUCER_AMAX)

    def _tail_forward(self, flat, vec, n, flat_amax=None):
        # torch.cat((x, state[1]), dim=1): fc0 writes its slice of the cat buffer directly
        self.d0.forward(flat, self.flat_dim, self.cat[:, self.extra:], self.ld_cat, n, in_amax=flat_amax)
        self.cat[:n, self.extra + 512:self.cat_k].copy_(vec.reshape(n, -1))
        self.d1.forward(self.cat, self.ld_cat, self.f1, 512, n)
        self.d2.forward(self.f1, 512, self.h, 512, n)
        return self.h

    def _tail_backward(self, dh, flat, dflat, n, dflat_amax=None):
        self.d2.backward(self.f1, 512, dh, 512, n, din=self.df1, ld_din=512, mask_src=self.f1, ld_mask=512)
        # the first extra+512 columns of cat are ReLU outputs: mask with the cat values themselves
        self.d1.backward(self.cat, self.ld_cat, self.df1, 512, n, din=self.dcat, ld_din=self.ld_cat, mask_src=self.cat,
                         ld_mask=self.ld_cat)
        # d(flat) = d(pooled) of the last conv block: its data gradient leaves the samples' magnitudes for that block's backward
        self.d0.backward(flat, self.flat_dim, self.dcat[:, self.extra:], self.ld_cat, n, din=dflat, ld_din=self.flat_dim,
                         din_amax=dflat_amax)


class NavPreNet(_NavBase):
    """nav_encoder.py:12-44: state = [image [n,C,48,48], vector [n,9]]."""
    n_inputs = 2
    vec_dim = 9

    def __init__(self, image_channel=1, last_output_dim=512):
        super().__init__()
        self.conv1 = nn.Conv2d(image_channel, 64, 3, stride=1, padding=(1, 1))
        self.conv2 = nn.Conv2d(64, 128, 3, stride=1, padding=(1, 1))
        self.conv3 = nn.Conv2d(128, 256, 3, stride=1, padding=(1, 1))
        self.fc0 = mlp([(256 * 6 * 6, 512, "relu")])
        self.fc1 = mlp([(512 + 9, 512, "relu")])
        self.fc2 = nn.Linear(512, 512)
        self.image_channel = image_channel

    def build(self, cap, device):
        self._device = device
        self.c1 = _ConvPool(self.conv1, 48, 48, cap, device)
        self.c2 = _ConvPool(self.conv2, 24, 24, cap, device)
        self.c3 = _ConvPool(self.conv3, 12, 12, cap, device)
        self._build_tail(cap, device, 256 * 6 * 6, 9)
        self._amax_arena(cap, device, 3, 2)
        self._blocks = [self.c1, self.c2, self.c3, self.d0, self.d1, self.d2]

    def _image(self, states, n):
        return states[0].reshape(n, self.image_channel, 48, 48).contiguous()

    def forward_dev(self, states, n):
        self.img = self._image(states, n)
        self._amax_f.zero_()
        A = self._links(*self._amax_f)
        p1 = self.c1.forward(self.img, n, out_amax=A[0])                 # every block leaves the magnitudes the next one scales by
        p2 = self.c2.forward(p1, n, in_amax=A[0], out_amax=A[1])
        p3 = self.c3.forward(p2, n, in_amax=A[1], out_amax=A[2])
        return self._tail_forward(p3, states[1], n, flat_amax=A[2])

    def backward_dev(self, dh, n):
        self._amax_b.zero_()
        G = self._links(*self._amax_b)
        self._tail_backward(dh, self.c3.p, self.c3.dp, n, dflat_amax=G[0])
        self.c3.backward(self.c2.p, n, din=self.c2.dp, dp_amax=G[0], din_amax=G[1])
        self.c2.backward(self.c1.p, n, din=self.c1.dp, dp_amax=G[1])
        self.c1.backward(self.img, n)


class NavPedPreNet(NavPreNet):
    """nav_encoder.py:47-83: the image is torch.cat([state[0], state[2]], axis=1)."""
    n_inputs = 3

    def __init__(self, image_channel=4, last_output_dim=512):
        super().__init__(image_channel, last_output_dim)

    def _image(self, states, n):
        return torch.cat([states[0].reshape(n, -1, 48, 48), states[2].reshape(n, -1, 48, 48)], dim=1).contiguous()


class NavPreNet1D(_NavBase):
    """nav_encoder.py:86-128: state = [laser [n,1,960], vector [n,5], pedestrian image [n,3,48,48]];
    two un-activated Conv1d layers + fc_1d on the laser scan, a 7x7 / 5x5 / 3x3 conv stack on the image."""
    n_inputs = 3

    def __init__(self, image_channel=1, last_output_dim=512):
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
        self.image_channel = image_channel

    def build(self, cap, device):
        self._device = device
        self.c1 = _ConvPool(self.conv1, 48, 48, cap, device)    # -> 64 x 44 x 44 -> 22 x 22
        self.c2 = _ConvPool(self.conv2, 22, 22, cap, device)    # -> 128 x 20 x 20 -> 10 x 10
        self.c3 = _ConvPool(self.conv3, 10, 10, cap, device)    # -> 256 x 10 x 10 -> 5 x 5
        self.l1 = _ConvPool(_as2d(self.conv1d1), 1, 960, cap, device, pool=False, relu=False)  # -> 32 x 478
        self.l2 = _ConvPool(_as2d(self.conv1d2), 1, 478, cap, device, pool=False, relu=False)  # -> 32 x 238
        self.l1.m, self.l2.m = self.conv1d1, self.conv1d2
        self.d1d = _Dense(self.fc_1d[0], True, cap, device)
        self._build_tail(cap, device, 6400, 5, extra=256)
        self._amax_arena(cap, device, 4, 2)
        self.dl2 = self._f(cap, 7616)
        self._blocks = [self.c1, self.c2, self.c3, self.l1, self.l2, self.d1d, self.d0, self.d1, self.d2]

    def forward_dev(self, states, n):
        self.laser = states[0].reshape(n, 1, 1, 960).contiguous()
        self.img = states[2].reshape(n, self.image_channel, 48, 48).contiguous()
        self._amax_f.zero_()
        A = self._links(*self._amax_f)
        a1 = self.l1.forward(self.laser, n)
        a2 = self.l2.forward(a1, n, out_amax=A[3])
        self.d1d.forward(a2, 7616, self.cat, self.ld_cat, n, in_amax=A[3])  # encoded laser -> cat[:, 0:256]
        p1 = self.c1.forward(self.img, n, out_amax=A[0])                 # every block leaves the magnitudes the next one scales by
        p2 = self.c2.forward(p1, n, in_amax=A[0], out_amax=A[1])
        p3 = self.c3.forward(p2, n, in_amax=A[1], out_amax=A[2])
        return self._tail_forward(p3, states[1], n, flat_amax=A[2])

    def backward_dev(self, dh, n):
        self._amax_b.zero_()
        G = self._links(*self._amax_b)
        self._tail_backward(dh, self.c3.p, self.c3.dp, n, dflat_amax=G[0])
        self.c3.backward(self.c2.p, n, din=self.c2.dp, dp_amax=G[0], din_amax=G[1])
        self.c2.backward(self.c1.p, n, din=self.c1.dp, dp_amax=G[1])
        self.c1.backward(self.img, n)
        # laser branch: dcat[:, 0:256] already carries the ReLU mask of fc_1d (applied by fc1's dgrad)
        self.d1d.backward(self.l2.a, 7616, self.dcat, self.ld_cat, n, din=self.l2.dz, ld_din=7616)
        self.l2.backward(self.l1.a, n, din=self.l1.dz)
        self.l1.backward(self.laser, n)


class _Conv1dAs2d:
    """View of an nn.Conv1d with the attributes _ConvPool reads from a Conv2d (h = kh = 1)."""

    def __init__(self, m):
        self.in_channels, self.out_channels = m.in_channels, m.out_channels
        self.kernel_size = (1, m.kernel_size[0])
        self.stride = (m.stride[0], m.stride[0])
        self.padding = (0, 0)  # "valid"
        self.weight, self.bias = m.weight, m.bias


def _as2d(m):
    return _Conv1dAs2d(m)


# ------------------------------------------------------------------------------------------------
class HipNormal:
    """What PPO.forward returns where the reference returns torch.distributions.Normal
    (actor.py:62-66): .mean .stddev .sample() .log_prob(a) (per dim) .entropy()."""

    def __init__(self, net, mu, log_std, action, logp, version):
        self._net, self.mean, self._log_std = net, mu, log_std
        self.stddev = torch.exp(log_std).expand_as(mu)
        self._action, self._logp, self._version = action, logp, version
        self._draws = 0

    loc = property(lambda self: self.mean)
    scale = property(lambda self: self.stddev)

    def sample(self):
        """First call: the draw fused into the forward; later calls draw again on the retained
        features with a fresh stream id (draw index in bits 48.. of the id)."""
        if self._draws == 0 and self._action is not None:
            self._draws = 1
            return self._action
        self._draws += 1
        self._action, self._logp = self._net._resample(self._version, self._draws)
        return self._action

    def summed_log_prob(self, value):
        if value is self._action and self._logp is not None:
            return self._logp
        return self._net._eval_logp(value, self._version)

    def log_prob(self, value):
        # per-dimension Normal log-density (torch semantics); staging-level torch arithmetic on [n, D]
        var = self.stddev ** 2
        return -((value - self.mean) ** 2) / (2 * var) - self._log_std - 0.9189385332046727

    def entropy(self):
        return (0.5 + 0.9189385332046727 + self._log_std).expand_as(self.mean)


class GenericPPO(Basenn):
    """PPO over operator-composed encoders (any PreNet above) with a Categorical or Gaussian actor.
    Same constructor and protocol as the reference's PPO (ppo.py:17-146); see nn/ppo.py for the
    Atari fast path."""

    def __init__(self, actor, critic, prenet=None, rnd=None, config=None, config_nn=None, max_batch=None,
                 process_group=None):
        super().__init__(config, config_nn)
        if rnd is not None:
            raise NotImplementedError("RND is disabled in the reference defaults (USE_RND=False) and out of scope")
        if bool(config_nn.SHARE_CNN_NET) != (prenet is not None):
            raise ValueError("SHARE_CNN_NET=True needs a shared prenet; SHARE_CNN_NET=False needs prenet=None")
        if not torch.cuda.is_available():
            raise _lib.DdrlError("ddrl4nav_amd needs a ROCm GPU; there is no CPU fallback")
        self.lib = _lib.load()
        self.device = torch.device(config.DEVICE if str(config.DEVICE) != "cuda" else "cuda:%d" % torch.cuda.current_device())
        self.prenet, self.actor, self.critic = prenet, actor, critic
        self._critics = [self.critic]
        self.rnd, self.gail_critic = rnd, False
        self.share_cnn_net = bool(config_nn.SHARE_CNN_NET)
        self.training_iter_time = config_nn.TRAINING_ITER_TIME
        self.update_time = 0
        self._cfg_nn, self._process_group = config_nn, process_group
        self._seed = int(torch.initial_seed()) & (2 ** 63 - 1)
        self._calls = 0
        self.continuous = hasattr(actor, "log_std")
        self.n_actions = actor.action_output_dim
        # micro-batch: every layer keeps activations + gradients for this many samples (NavPreNet1D x2:
        # ~4.4 MB per sample); larger micro-batches fill the GPU better (1024 -> 4096: +11 % samples/s)
        self.cap = int(max_batch if max_batch is not None else 4096)
        self._encs = [prenet] if self.share_cnn_net else [actor.pre, critic.pre]
        for e in self._encs:
            if not all(hasattr(e, k) for k in ("build", "forward_dev", "backward_dev", "pack")):
                raise TypeError("GenericPPO needs operator-composed encoders (MLPPreNet, NavPreNet, AtariPreNet, ...), "
                                "got %r" % type(e))
        self._raw_u8 = bool(getattr(self._encs[0], "raw_u8", False))  # AtariPreNet: uint8 frames go to the kernels as they are
        self._extra = []   # [(critic module, value-loss scratch)]: PPO.add_critic (ppo.py:61-62)
        self._bind_arena()
        for e in self._encs:
            e.build(self.cap, self.device)
            if e.h.shape[1] != FEAT:
                raise NotImplementedError("the head kernels take %d-wide features (AC_INPUT_DIM, config_nn.py:23); "
                                          "this encoder emits %d" % (FEAT, e.h.shape[1]))
        self._build_heads()
        self._dirty = True
        self._step = 0
        self.encoder_streams = True     # actor and critic encoders on two HIP streams (_side)

    # ---- flat arenas (reference named_parameters() order) ------------------------------------------
    def _bind_arena(self):
        params = list(self.named_parameters())
        total = sum(p.numel() for _, p in params)
        f = dict(dtype=torch.float32, device=self.device)
        self.n_params = total
        self.params = torch.zeros(total, **f)
        self.grads = torch.zeros(total + STATS_FLOATS, **f)
        self.gtmp = torch.zeros(total + STATS_FLOATS, **f)
        self.adam_m = torch.zeros(total, **f)
        self.adam_v = torch.zeros(total, **f)
        self._offsets, off = {}, 0
        with torch.no_grad():
            for name, p in params:
                n = p.numel()
                view = self.params[off:off + n].view(p.shape)
                view.copy_(p.detach().to(self.device, torch.float32))
                p.data = view
                p.requires_grad_(False)
                p.grad_view = self.gtmp[off:off + n].view(p.shape)  # operators write gradients here
                self._offsets[name] = off
                off += n
        # actor group = the leading run of actor.* parameters (ppo.py:40-42); everything when shared
        self.n_actor = total if self.share_cnn_net else sum(p.numel() for k, p in params if k.startswith("actor."))
        if not self.share_cnn_net:
            assert all(k.startswith("actor.") for k, _ in params[:sum(1 for k, _ in params if k.startswith("actor."))])

    def _build_heads(self):
        o = self._offsets
        d = HeadsDesc()
        d.continuous = 1 if self.continuous else 0
        d.n_actions, d.shared = self.n_actions, 1 if self.share_cnn_net else 0
        d.actor_w, d.actor_b = o["actor.actor_linear.weight"], o["actor.actor_linear.bias"]
        d.log_std = o.get("actor.log_std", 0)
        d.critic_w, d.critic_b = o["critic.critic_linear.weight"], o["critic.critic_linear.bias"]
        d.n_params = self.n_params
        self._hd = d
        wf, cb = c_int64(), c_int64()
        check(self.lib.ddrl_op_heads_ws_floats(byref(d), self.cap, byref(wf)))
        check(self.lib.ddrl_op_clip_adam_ws_bytes(byref(cb)))
        f = dict(dtype=torch.float32, device=self.device)
        self._heads_ws = torch.empty(wf.value, **f)
        self._adam_ws = torch.empty(cb.value, dtype=torch.uint8, device=self.device)
        # an encoder that owns its d(loss)/d(h) buffer (AtariPreNet: inside its kernel workspace) gets the gradient written there
        self._dh = [e._dh if getattr(e, "_dh", None) is not None else torch.empty((self.cap, FEAT), **f) for e in self._encs]
        c = self._cfg_nn
        self._cfg = _lib.default_config(
            max_batch=self.cap, n_actions=max(2, min(self.n_actions, 18)), share_cnn_net=1 if self.share_cnn_net else 0,
            learning_rate=float(c.LEARNING_RATE), smooth_l1_loss=1 if c.SMOOTH_L1_LOSS else 0,
            clip_grad=1 if c.CLIP_GRID else 0, clip_grad_norm=float(c.CLIP_GRID_NUM), actor_lr=float(c.ACTOR_LEARNING_RATE),
            critic_lr=float(c.CRITIC_LEARNING_RATE), ppo_clip=float(c.PPO_CLIP), dual_clip=float(c.DUEL_PPO_CLIP),
            v_loss_theta=float(c.V_LOSS_THETA), ent_loss_theta=float(c.ENTROPY_LOSS_THETA))

    def _ensure_packed(self):
        if self._dirty:
            for e in self._encs:
                e.pack()
            self._dirty = False

    def params_changed(self):
        self._dirty = True

    def load_state_dict(self, state_dict, strict=True, **kw):
        out = super().load_state_dict(state_dict, strict=strict, **kw)
        self._dirty = True
        return out

    def to(self, *args, **kwargs):
        return self

    def reset_optimizer(self):
        self.adam_m.zero_()
        self.adam_v.zero_()
        self._step = 0

    # ---- forward -----------------------------------------------------------------------------------
    def _stage(self, states, lo, hi):
        if self._raw_u8:
            return [torch.as_tensor(s)[lo:hi] for s in states]
        return [to_device(torch.as_tensor(s)[lo:hi], self.device, torch.float32) for s in states]

    def _side(self):
        """The second encoder's stream: actor and critic encoders are independent networks with their own buffers and their own slices
        of the gradient arena, so the critic's runs beside the actor's (its small, latency-bound launches -- dense layers, the laser
        branch, slab reductions -- fill what the other's leave idle).  `net.encoder_streams = False`: one stream (per-operator timing)."""
        if self.share_cnn_net or not self.encoder_streams:
            return None
        if getattr(self, "_side_stream", None) is None:
            self._side_stream = torch.cuda.Stream(device=self.device)
        return self._side_stream

    def _features(self, st, n):
        side = self._side()
        if side is None:
            hs = [e.forward_dev(st, n) for e in self._encs]
            return (hs[0], hs[0]) if self.share_cnn_net else (hs[0], hs[1])
        cur = torch.cuda.current_stream(self.device)
        side.wait_stream(cur)
        with torch.cuda.stream(side):
            hc = self._encs[1].forward_dev(st, n)
        ha = self._encs[0].forward_dev(st, n)
        cur.wait_stream(side)
        return ha, hc

    def _backward_both(self, dha, dhc, n):
        side = self._side()
        if side is None:
            self._encs[0].backward_dev(dha, n)
            if not self.share_cnn_net:
                self._encs[1].backward_dev(dhc, n)
            return
        cur = torch.cuda.current_stream(self.device)
        side.wait_stream(cur)
        with torch.cuda.stream(side):
            self._encs[1].backward_dev(dhc, n)
        self._encs[0].backward_dev(dha, n)
        cur.wait_stream(side)

    def forward(self, states, act=None, play_mode=False):
        states = states if isinstance(states, (list, tuple)) else [states]
        n = int(torch.as_tensor(states[0]).shape[0])
        self._ensure_packed()
        self._calls += 1
        A = self.n_actions
        f = dict(dtype=torch.float32, device=self.device)
        dist_out = torch.empty((n, A), **f)
        value, logp = torch.empty(n, **f), torch.empty(n, **f)
        a_in = None if act is None else torch.as_tensor(act, **f).contiguous()
        action = a_in if a_in is not None else torch.empty((n, A) if self.continuous else (n,), **f)
        extra_values = [torch.empty(n, **f) for _ in self._extra]
        for lo in range(0, n, self.cap):
            hi = min(n, lo + self.cap)
            ha, hc = self._features(self._stage(states, lo, hi), hi - lo)
            for (crit, _), xv in zip(self._extra, extra_values):   # [critic(states) for critic in self._critics] (ppo.py:75)
                check(self.lib.ddrl_op_value_head_forward(_p(crit.critic_linear.weight.data), _p(crit.critic_linear.bias.data),
                                                          _p(hc), FEAT, hi - lo, _p(xv[lo:hi]), _st()))
            check(self.lib.ddrl_op_heads_act(
                byref(self._hd), _p(self.params), _p(ha), _p(hc), hi - lo, _p(None if a_in is None else a_in[lo:hi]),
                self._seed, self._calls * 4096 + lo // self.cap, _p(dist_out[lo:hi]), _p(value[lo:hi]),
                _p(None) if a_in is not None else _p(action[lo:hi]), _p(logp[lo:hi]), _st()))
        self._last = (n, self._calls)
        values = [value.view(n, 1)] + [x.view(n, 1) for x in extra_values]
        if play_mode:
            return (dist_out, logp if act is not None else None), values
        if self.continuous:
            dist = HipNormal(self, dist_out, self.actor.log_std.data, action, logp, self._calls)
        else:
            from ddrl4nav_amd.nn.distribution import HipCategorical
            dist = HipCategorical(_CatOps(self), dist_out, self._seed, self._calls, action, logp)
        return (dist, logp if act is not None else None), values

    def _eval_logp(self, value, version):
        n, v = self._last
        if v != version or n > self.cap:
            raise RuntimeError("log_prob of new actions needs the features of the forward that made this distribution "
                               "(call it before the next forward, batch <= max_batch)")
        ha = self._encs[0].h
        hc = ha if self.share_cnn_net else self._encs[1].h
        f = dict(dtype=torch.float32, device=self.device)
        logp, val = torch.empty(n, **f), torch.empty(n, **f)
        a = torch.as_tensor(value, **f).contiguous()
        check(self.lib.ddrl_op_heads_act(byref(self._hd), _p(self.params), _p(ha), _p(hc), n, _p(a), 0, 0, _p(None), _p(val),
                                         _p(None), _p(logp), _st()))
        return logp

    def _resample(self, version, draw):
        """A fresh Gaussian draw (and its summed log-prob) on the features of forward `version`."""
        n, v = self._last
        if v != version or n > self.cap:
            raise RuntimeError("sample() after another forward (or on a batch > max_batch) has no features to draw from")
        ha = self._encs[0].h
        hc = ha if self.share_cnn_net else self._encs[1].h
        f = dict(dtype=torch.float32, device=self.device)
        mu, val, logp = torch.empty((n, self.n_actions), **f), torch.empty(n, **f), torch.empty(n, **f)
        action = torch.empty((n, self.n_actions), **f)
        check(self.lib.ddrl_op_heads_act(byref(self._hd), _p(self.params), _p(ha), _p(hc), n, _p(None), self._seed,
                                         (version * 4096) ^ (int(draw) << 48), _p(mu), _p(val), _p(action), _p(logp), _st()))
        return action, logp

    def add_critic(self, critic):
        """PPO.add_critic (ppo.py:61-62): one more value head on the features the critic reads.  As in the reference the
        head is appended to a plain list: it is NOT a parameter of this module, so neither the grad-norm clip nor the
        Adam of this net sees it (ppo.py:39 builds the optimiser before GAIL.__init__ calls add_critic, GAIL.py:116-117)
        -- it is never trained, but its value loss is part of VLoss / PpoTotalLoss and, with a shared prenet, its gradient
        flows into the encoder."""
        if not self.share_cnn_net or getattr(critic, "pre", None) is not None:
            raise NotImplementedError("an extra critic with its own encoder (SHARE_CNN_NET=False: deepcopy(critic) carries a "
                                      "third encoder, runner/utils.py:162) is not built; GAIL itself needs the shared prenet "
                                      "(D_prenet = deepcopy(prenet) if prenet else None, utils.py:164)")
        critic.to(self.device)
        for p in critic.parameters():
            p.requires_grad_(False)
        wf = c_int64()
        check(self.lib.ddrl_op_value_head_ws_floats(byref(wf)))
        self._critics.append(critic)
        self._extra.append((critic, torch.empty(wf.value, dtype=torch.float32, device=self.device)))

    def get_rnd(self, states):
        raise NotImplementedError("RND is out of scope on this path")

    def states_normalization(self, states):
        return states / 255

    # ---- learn (ppo.py:77-146) -----------------------------------------------------------------------
    def _iter_chunk(self, st, n, actions, old_logps, advs, rets, b_global, extra_rets=()):
        """forward + loss + backward of one micro-batch; gradients land in self.gtmp (overwritten)."""
        ha, hc = self._features(st, n)
        dha = self._dh[0]
        dhc = dha if self.share_cnn_net else self._dh[1]
        check(self.lib.ddrl_op_heads_loss(byref(self._hd), byref(self._cfg), _p(self.params), _p(ha), _p(hc), n, _p(actions),
                                          _p(old_logps), _p(advs), _p(rets), b_global, _p(dha), _p(dhc), _p(self.gtmp),
                                          _p(self._heads_ws), _st()))
        # v_loss = ppov_loss + rndv_loss + gailv_loss (ppo.py:95-107): each extra head adds its loss share to the VLoss
        # slot of the statistics tail and its d(loss)/d(h) to the shared encoder's gradient
        for (crit, ws), xr in zip(self._extra, extra_rets):
            check(self.lib.ddrl_op_value_head_loss(byref(self._cfg), 1, _p(crit.critic_linear.weight.data),
                                                   _p(crit.critic_linear.bias.data), _p(hc), FEAT, n, _p(xr), b_global, _p(dha),
                                                   FEAT, _p(None), _p(None), _p(self.gtmp[self.n_params + 1:]), _p(ws), _st()))
        self._backward_both(dha, dhc, n)

    def learn(self, data: Experience):
        states = data.states if isinstance(data.states, (list, tuple)) else [data.states]
        B = int(torch.as_tensor(states[0]).shape[0])
        f32 = lambda t: torch.as_tensor(t, dtype=torch.float32, device=self.device).contiguous()
        actions, old_logps, advs = f32(data.actions), f32(data.old_logps), f32(data.advs)
        vals = f32(data.values)
        rets = vals[0].contiguous()
        assert rets.shape == (B,)
        # ppo.py:97-104: the GAIL critic's targets are the LAST row of data.values (with RND unbuilt there is one extra head)
        assert len(self._extra) <= 1 and (not self._extra or vals.shape[0] >= 2), "data.values needs one row per critic"
        extra_rets = [vals[-1].contiguous()] if self._extra else []
        import torch.distributed as dist
        from ddrl4nav_amd.dist import global_batch
        world = dist.get_world_size(self._process_group) if dist.is_available() and dist.is_initialized() else 1
        b_global = global_batch(B, self._process_group)  # shards may be uneven
        total = self.n_params + STATS_FLOATS
        # the batch is read by every one of the TRAINING_ITER_TIME iterations: stage it on the device once
        if self._raw_u8:
            from ddrl4nav_amd.nn.atari_encoder import frames_u8
            dstates = [frames_u8(states, self.device)]
        else:
            dstates = [to_device(s, self.device, torch.float32) for s in states]
        for _ in range(self.training_iter_time):
            t0 = time.time()
            self._ensure_packed()
            for ci, lo in enumerate(range(0, B, self.cap)):
                hi = min(B, lo + self.cap)
                self._iter_chunk([s[lo:hi] for s in dstates], hi - lo, actions[lo:hi], old_logps[lo:hi], advs[lo:hi],
                                 rets[lo:hi], b_global, [x[lo:hi] for x in extra_rets])
                if ci == 0:
                    self.grads.copy_(self.gtmp)
                else:
                    check(self.lib.ddrl_op_accumulate(_p(self.grads), _p(self.gtmp), total, _st()))
            if world > 1:
                from ddrl4nav_amd.dist import allreduce_flat
                allreduce_flat(self.grads, self._process_group)
            self._step += 1
            check(self.lib.ddrl_op_clip_adam(byref(self._cfg), _p(self.params), _p(self.grads), _p(self.adam_m),
                                             _p(self.adam_v), self.n_params, self.n_actor, 1 if self.share_cnn_net else 0,
                                             self._step, _p(self._adam_ws), _st()))
            self._dirty = True
            self.update_time += 1
            s = self.grads[self.n_params:self.n_params + 6].cpu().numpy()
            yield ({"PpoTotalLoss": float(s[3]), "ActorLoss": float(s[0]), "VLoss": float(s[1]), "EntLoss": float(s[2]),
                    "PpoBackUpTime": time.time() - t0}, self.update_time, True)

    def stats(self):
        s = self.grads[self.n_params:self.n_params + 6].cpu().numpy()
        return {"ActorLoss": float(s[0]), "VLoss": float(s[1]), "EntLoss": float(s[2]), "PpoTotalLoss": float(s[3]),
                "GradNorm": float(s[4]), "ClipCoef": float(s[5])}


class _CatOps:
    """categorical_stats / categorical_sample provider for HipCategorical (same C entry points as HotPath)."""

    def __init__(self, net):
        self.lib = net.lib

    def categorical_stats(self, probs):
        n, A = probs.shape
        p_hat, logits = torch.empty_like(probs), torch.empty_like(probs)
        ent = torch.empty(n, dtype=torch.float32, device=probs.device)
        check(self.lib.ddrl_categorical_stats(_p(probs), n, A, _p(p_hat), _p(logits), _p(ent), _st()))
        return p_hat, logits, ent

    def categorical_sample(self, probs, seed, stream_id):
        n, A = probs.shape
        action = torch.empty(n, dtype=torch.float32, device=probs.device)
        logp = torch.empty(n, dtype=torch.float32, device=probs.device)
        check(self.lib.ddrl_categorical_sample(_p(probs), n, A, int(seed) & (2 ** 64 - 1), int(stream_id) & (2 ** 64 - 1),
                                               _p(action), _p(logp), _st()))
        return action, logp
