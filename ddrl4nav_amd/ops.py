"""Thin ctypes wrappers of the operator-level C ABI (include/ddrl.h, ddrl_op_*): generic
convolution, 2x2 max-pool and dense layers on torch-owned device buffers.  torch supplies memory
and streams only; the arithmetic runs in csrc/gconv.hip and csrc/glinear.hip.  Used by
ddrl4nav_amd.nn.generic to compose the reference's non-Atari encoders
(USTC_lab/nn/nav_encoder.py, mlp_encoder.py)."""
from ctypes import byref, c_int32, c_int64, c_void_p

import torch

from . import _lib
from ._lib import ConvDesc, check


def _p(t):
    return c_void_p(0) if t is None else c_void_p(t.data_ptr())


def _st():
    return c_void_p(torch.cuda.current_stream().cuda_stream)


def _f32(t):
    assert t.dtype == torch.float32 and t.is_cuda and t.is_contiguous(), "expected a contiguous fp32 device tensor"
    return t


class Conv:
    """One Conv2d / Conv1d layer (torch weight layout [cout][cin][kh][kw]; Conv1d: h = kh = 1)."""

    def __init__(self, cin, h, w, cout, kh, kw, stride=1, pad=(0, 0), max_n=1, device="cuda"):
        self.lib = _lib.load()
        self.cin, self.h, self.w, self.cout, self.kh, self.kw = cin, h, w, cout, kh, kw
        self.stride, self.pad = stride, tuple(pad)
        self.device = torch.device(device)
        d = self.desc(max_n)
        oh, ow, pf, wf, sf = c_int32(), c_int32(), c_int64(), c_int64(), c_int64()
        check(self.lib.ddrl_op_conv_out_shape(byref(d), byref(oh), byref(ow)))
        check(self.lib.ddrl_op_conv_pack_floats(byref(d), byref(pf)))
        check(self.lib.ddrl_op_conv_ws_floats(byref(d), byref(wf)))
        check(self.lib.ddrl_op_conv_scratch_floats(byref(d), byref(sf)))
        self.oh, self.ow, self.max_n = oh.value, ow.value, max_n
        self.packed = torch.zeros(pf.value, dtype=torch.float32, device=self.device)   # read-only after pack()
        self.ws = torch.empty(wf.value, dtype=torch.float32, device=self.device)
        # per-sample plane scales of a launch that is not handed its scales (fp16-plane layers; max_n floats): one launch per layer
        # object at a time, like `ws`
        self.scratch = torch.empty(sf.value, dtype=torch.float32, device=self.device) if sf.value else None

    def desc(self, n, in_sn=0, out_sn=0):
        return ConvDesc(n, self.cin, self.h, self.w, self.cout, self.kh, self.kw, self.stride, self.pad[0], self.pad[1],
                        in_sn, out_sn)

    def pack(self, weight):
        check(self.lib.ddrl_op_conv_pack(byref(self.desc(1)), _p(_f32(weight)), _p(self.packed), _st()))

    def forward(self, x, bias, relu, out=None, n=None, out_amax=None):
        """out_amax (optional, [n] floats zeroed by the caller): raised to every sample's largest |output| (include/ddrl.h,
        "per-sample magnitudes")."""
        n = x.shape[0] if n is None else n
        assert n <= self.max_n, "batch larger than the layer's scratch was sized for"
        if out is None:
            out = torch.empty((n, self.cout, self.oh, self.ow), dtype=torch.float32, device=x.device)
        check(self.lib.ddrl_op_conv_forward(byref(self.desc(n)), _p(_f32(x)), _p(self.packed), _p(_f32(bias)),
                                            1 if relu else 0, _p(out), _p(self.scratch), _p(out_amax), _st()))
        return out

    def has_forward_pool(self):
        """True when the layer's kernels take ReLU + max_pool2d(2) into their epilogue (include/ddrl.h ddrl_op_conv_forward_pool)."""
        return bool(self.lib.ddrl_op_conv_has_forward_pool(byref(self.desc(1))))

    def pooled_uses_scales(self):
        """True when the pooled operators of this layer read per-sample magnitudes (in_amax / dpool_amax; sample_amax below)."""
        return bool(self.lib.ddrl_op_conv_pooled_uses_scales(byref(self.desc(1))))

    def forward_pool(self, x, bias, pooled, code, n=None, in_amax=None, out_amax=None):
        """max_pool2d(relu(conv(x)), 2) in one launch: pooled [n][cout][oh/2][ow/2] + one decision byte per window.
        in_amax: the samples' largest |x| (from x's producer or sample_amax; None = own pre-pass); out_amax: raised to the samples'
        largest pooled value (zeroed by the caller)."""
        n = x.shape[0] if n is None else n
        assert n <= self.max_n, "batch larger than the layer's scratch was sized for"
        check(self.lib.ddrl_op_conv_forward_pool(byref(self.desc(n)), _p(_f32(x)), _p(self.packed), _p(_f32(bias)), _p(pooled),
                                                 _p(code), _p(in_amax), _p(self.scratch), _p(out_amax), _st()))
        return pooled

    def dgrad_pooled(self, dpool, code, din=None, n=None, dpool_amax=None, din_amax=None):
        """Data gradient of a forward_pool layer from d(pooled) + decision bytes (no full-resolution gradient in between).
        din_amax: raised to the samples' largest |din| (zeroed by the caller)."""
        n = dpool.shape[0] if n is None else n
        assert n <= self.max_n, "batch larger than the layer's scratch was sized for"
        if din is None:
            din = torch.empty((n, self.cin, self.h, self.w), dtype=torch.float32, device=dpool.device)
        check(self.lib.ddrl_op_conv_dgrad_pooled(byref(self.desc(n)), _p(_f32(dpool)), _p(code), _p(self.packed), _p(din),
                                                 _p(dpool_amax), _p(self.scratch), _p(din_amax), _st()))
        return din

    def wgrad_pooled(self, x, dpool, code, dw, db, n=None, in_amax=None, dpool_amax=None):
        n = x.shape[0] if n is None else n
        assert n <= self.max_n, "batch larger than the split-K scratch was sized for"
        check(self.lib.ddrl_op_conv_wgrad_pooled(byref(self.desc(n)), _p(_f32(x)), _p(_f32(dpool)), _p(code), _p(self.packed),
                                                 _p(self.ws), _p(dw), _p(db), _p(in_amax), _p(dpool_amax), _st()))

    def dgrad(self, dz, din=None, n=None):
        n = dz.shape[0] if n is None else n
        assert n <= self.max_n, "batch larger than the layer's scratch was sized for"
        if din is None:
            din = torch.empty((n, self.cin, self.h, self.w), dtype=torch.float32, device=dz.device)
        check(self.lib.ddrl_op_conv_dgrad(byref(self.desc(n)), _p(_f32(dz)), _p(self.packed), _p(din), _p(self.scratch), _st()))
        return din

    def wgrad(self, x, dz, dw, db, n=None):
        n = x.shape[0] if n is None else n
        assert n <= self.max_n, "batch larger than the split-K scratch was sized for"
        check(self.lib.ddrl_op_conv_wgrad(byref(self.desc(n)), _p(_f32(x)), _p(_f32(dz)), _p(self.packed), _p(self.ws),
                                          _p(dw), _p(db), _st()))


def sample_amax(x, n, out):
    """Largest magnitude of every sample of x[:n] (dense samples): the pre-pass for tensors whose producer leaves none (include/ddrl.h)."""
    elems = x[0].numel()
    check(_lib.load().ddrl_op_sample_amax(_p(_f32(x)), elems, elems, n, _p(out), _st()))
    return out


def maxpool2(x, out=None):
    n, c, h, w = x.shape
    if out is None:
        out = torch.empty((n, c, h // 2, w // 2), dtype=torch.float32, device=x.device)
    check(_lib.load().ddrl_op_maxpool2_forward(_p(_f32(x)), n * c, h, w, _p(out), _st()))
    return out


def maxpool2_relu_backward(a, dpool, dz=None):
    n, c, h, w = a.shape
    if dz is None:
        dz = torch.empty_like(a)
    check(_lib.load().ddrl_op_maxpool2_relu_backward(_p(_f32(a)), _p(_f32(dpool)), n * c, h, w, _p(dz), _st()))
    return dz


def maxpool2_idx(x, out=None, code=None):
    """max_pool2d(x, 2) that also leaves one decision byte per window for maxpool2_backward_idx (include/ddrl.h)."""
    n, c, h, w = x.shape
    if out is None:
        out = torch.empty((n, c, h // 2, w // 2), dtype=torch.float32, device=x.device)
    if code is None:
        code = torch.empty((n, c, h // 2, w // 2), dtype=torch.uint8, device=x.device)
    check(_lib.load().ddrl_op_maxpool2_forward_idx(_p(_f32(x)), n * c, h, w, _p(out), _p(code), _st()))
    return out, code


def maxpool2_backward_idx(dpool, code, h, w, dz=None):
    n, c = dpool.shape[:2]
    if dz is None:
        dz = torch.empty((n, c, h, w), dtype=torch.float32, device=dpool.device)
    check(_lib.load().ddrl_op_maxpool2_backward_idx(_p(_f32(dpool)), _p(code), n * c, h, w, _p(dz), _st()))
    return dz


class Linear:
    """One nn.Linear(K, N) (+ReLU) layer; weight [N][K]."""

    def __init__(self, K, N, max_n=1, device="cuda"):
        self.lib = _lib.load()
        self.K, self.N, self.max_n = K, N, max_n
        self.device = torch.device(device)
        a, b, wf = c_int64(), c_int64(), c_int64()
        check(self.lib.ddrl_op_linear_pack_floats(K, N, byref(a), byref(b)))
        check(self.lib.ddrl_op_linear_ws_floats(max_n, K, N, byref(wf)))
        self.wt = torch.zeros(a.value, dtype=torch.float32, device=self.device)
        self.wn = torch.zeros(b.value, dtype=torch.float32, device=self.device)
        self.ws = torch.empty(wf.value, dtype=torch.float32, device=self.device)

    def pack(self, weight):
        check(self.lib.ddrl_op_linear_pack(_p(_f32(weight)), self.K, self.N, _p(self.wt), _p(self.wn), _st()))

    def uses_planes(self, n):
        """True when a launch of n rows runs on the fp16 plane kernels (and therefore reads per-row scales)."""
        return bool(self.lib.ddrl_op_linear_uses_planes(n, self.K, self.N))

    def row_amax(self, x, ld, width, n, out, accumulate=False):
        """Largest magnitude of every row of x[:n] (one pass; hand it to the operators that read the same tensor)."""
        check(self.lib.ddrl_op_row_amax(_p(x), ld, width, n, _p(out), 1 if accumulate else 0, _st()))
        return out

    def forward(self, x, ld_in, bias, relu, out, ld_out, n, in_amax=None):
        assert n <= self.max_n
        check(self.lib.ddrl_op_linear_forward(_p(x), ld_in, _p(self.wt), _p(_f32(bias)), 1 if relu else 0, _p(out), ld_out,
                                              n, self.K, self.N, _p(self.ws), _p(in_amax), _st()))
        return out

    def dgrad(self, dout, ld_dout, mask_src, ld_mask, din, ld_din, n, dout_amax=None, din_amax=None, amax_cols=None):
        """din_amax ([n] floats zeroed by the caller): raised to every row's largest |din| over the columns amax_cols = (lo, hi)
        (default: all K)."""
        assert n <= self.max_n
        lo, hi = amax_cols if amax_cols is not None else (0, 0)
        check(self.lib.ddrl_op_linear_dgrad(_p(dout), ld_dout, _p(self.wn), _p(mask_src), ld_mask, _p(din), ld_din, n,
                                            self.K, self.N, _p(self.ws), _p(dout_amax), _p(din_amax), lo, hi, _st()))
        return din

    def wgrad(self, x, ld_in, dout, ld_dout, dw, db, n, in_amax=None, dout_amax=None):
        assert n <= self.max_n
        check(self.lib.ddrl_op_linear_wgrad(_p(x), ld_in, _p(dout), ld_dout, _p(self.ws), _p(dw), _p(db), n, self.K,
                                            self.N, _p(in_amax), _p(dout_amax), _st()))
