"""GPU tests of the operator-level C ABI (ddrl_op_*): every generic HIP operator against the
plain PyTorch fp32 CPU operator it replaces (the operators of the reference's nav / MLP encoders,
USTC_lab/nn/nav_encoder.py, mlp_encoder.py).  Tolerance: |d| <= 2e-5 * max|want| (+1e-6) --
same products, different fp32 summation order."""
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


def close(got, want, tol=2e-5):
    got, want = got.detach().cpu().double(), want.detach().cpu().double()
    scale = max(want.abs().max().item(), 1e-30)
    err = (got - want).abs().max().item()
    assert err <= tol * scale + 1e-7, (err, scale)


# (n, cin, h, w, cout, kh, kw, stride, pad) -- the nav encoder layers plus ragged shapes
CONVS = [
    (5, 3, 48, 48, 64, 7, 7, 1, (1, 1)),     # NavPreNet1D.conv1
    (300, 3, 48, 48, 64, 7, 7, 1, (1, 1)),   # the same with more samples than persistent workgroups (csrc/fconv.hip walks them)
    (3, 64, 22, 22, 128, 5, 5, 1, (1, 1)),   # NavPreNet1D.conv2
    (7, 128, 10, 10, 256, 3, 3, 1, (1, 1)),  # NavPreNet1D.conv3
    (4, 1, 48, 48, 64, 3, 3, 1, (1, 1)),     # NavPreNet.conv1
    (261, 4, 48, 48, 64, 3, 3, 1, (1, 1)),   # NavPedPreNet.conv1 (image + 3 pedestrian maps), more samples than persistent workgroups
    (2, 64, 24, 24, 128, 3, 3, 1, (1, 1)),   # NavPreNet.conv2
    (3, 128, 12, 12, 256, 3, 3, 1, (1, 1)),  # NavPreNet.conv3
    (7, 64, 9, 9, 64, 3, 3, 1, (0, 0)),      # AtariPreNet.conv3 as an operator (no padding)
    (6, 1, 1, 960, 32, 1, 5, 2, (0, 0)),     # conv1d1
    (6, 32, 1, 478, 32, 1, 3, 2, (0, 0)),    # conv1d2
    (133, 32, 1, 478, 32, 1, 3, 2, (0, 0)),  # conv1d2, enough samples for several slabs of its weight gradient (csrc/c1d.hip)
    (5, 32, 1, 101, 32, 1, 3, 2, (0, 0)),    # the same kernels on an odd width
    (1, 5, 9, 11, 70, 3, 2, 1, (2, 0)),      # odd everything, one sample
    (9, 4, 20, 20, 32, 4, 4, 2, (0, 0)),     # stride 2 square
    (2, 3, 16, 16, 8, 8, 8, 4, (0, 0)),      # stride 4
]


@pytest.mark.parametrize("shape", CONVS)
def test_conv_forward_backward_vs_torch(shape):
    from ddrl4nav_amd.ops import Conv
    n, cin, h, w, cout, kh, kw, s, pad = shape
    g = torch.Generator().manual_seed(hash(shape) % 1000)
    x = torch.randn(n, cin, h, w, generator=g)
    wt = torch.randn(cout, cin, kh, kw, generator=g) / (cin * kh * kw) ** 0.5
    b = torch.randn(cout, generator=g)
    if n > 4:
        x[3] *= 1e-5   # a faint and an empty input sample: per-sample input scales of the plane kernels
        x[4] = 0.0
    x.requires_grad_(True)
    wt.requires_grad_(True)
    b.requires_grad_(True)
    z = F.conv2d(x, wt, b, stride=s, padding=pad)
    out = F.relu(z)
    dz = torch.randn(z.shape, generator=g)
    if n > 2:
        # one sample without gradient (an advantage of exactly 0) and one 1e-4 below the rest: the per-sample / batch plane scales of
        # the fp16-plane kernels (csrc/pconv.hip) must not be pinned by the empty sample, nor lose the small one
        dz[1] = 0.0
        dz[2] *= 1e-4
    z.backward(dz)
    conv = Conv(cin, h, w, cout, kh, kw, stride=s, pad=pad, max_n=n)
    assert (conv.oh, conv.ow) == tuple(z.shape[2:])
    conv.pack(wt.detach().cuda())
    xd, bd, dzd = x.detach().cuda(), b.detach().cuda(), dz.cuda()
    close(conv.forward(xd, bd, relu=True), out)
    got = conv.forward(xd, bd, relu=False)
    close(got, z)
    if n > 4:   # the faint sample: bias + 1e-5-sized terms, right to the rounding of the fp32 sum (a batch-wide scale would lose the terms)
        assert float((got[3].cpu() - z[3].detach()).abs().max()) <= 1e-6
    din = conv.dgrad(dzd)
    close(din, x.grad)
    if n > 2:   # the small sample on its own scale: per-sample relative accuracy of the data gradient
        close(din[2], x.grad[2])
        assert float(din[1].abs().max()) == 0.0
    if conv.oh * conv.ow >= 32:
        dw = torch.full_like(wt.detach(), 7.0).cuda()
        db = torch.full((cout,), 7.0).cuda()
        conv.wgrad(xd, dzd, dw, db)
        close(dw, wt.grad)
        close(db, b.grad)


@pytest.mark.parametrize("shape", [(5, 3, 48, 64, 7, 1), (259, 3, 48, 64, 7, 1), (9, 64, 22, 128, 5, 1), (11, 128, 10, 256, 3, 1),
                                   (4, 64, 24, 128, 3, 1), (6, 128, 12, 256, 3, 1), (3, 64, 9, 64, 3, 0), (5, 1, 48, 64, 3, 1),
                                   (7, 4, 48, 64, 3, 1)])
def test_conv_relu_pool_in_one_launch_vs_torch(shape):
    """ddrl_op_conv_forward_pool: max_pool2d(relu(conv(x)), 2) from the convolution's epilogue (csrc/fconv.hip, csrc/pconv.hip) --
    pooled values to fp32 rounding, and the decision bytes route d(pooled) exactly as torch's autograd does wherever the window's
    maximum is clear of fp32 noise."""
    from ddrl4nav_amd.ops import Conv, maxpool2_backward_idx
    n, cin, h, cout, ks, pad = shape
    g = torch.Generator().manual_seed(sum(shape))
    x = torch.randn(n, cin, h, h, generator=g)
    x[1] *= 1e-3
    wt = torch.randn(cout, cin, ks, ks, generator=g) / (cin * ks * ks) ** 0.5
    b = torch.randn(cout, generator=g) * 0.1
    conv = Conv(cin, h, h, cout, ks, ks, pad=(pad, pad), max_n=n)
    conv.pack(wt.cuda())
    if (cin, h) == (64, 9):           # AtariPreNet.conv3 as an operator: no pooling epilogue -> the caller composes the two operators
        assert not conv.has_forward_pool()
        return
    assert conv.has_forward_pool()
    z = F.conv2d(x.double(), wt.double(), b.double(), padding=pad).requires_grad_(True)
    a = F.relu(z)
    want = F.max_pool2d(a, 2, stride=2)
    oh = conv.oh
    pooled = torch.full((n, cout, oh // 2, oh // 2), -7.0).cuda()
    code = torch.full((n, cout, oh // 2, oh // 2), 255, dtype=torch.uint8).cuda()
    conv.forward_pool(x.cuda(), b.cuda(), pooled, code, n=n)
    if conv.pooled_uses_scales():   # the caller's per-sample scales (shared by forward and weight gradient) = the operator's own pre-pass
        from ddrl4nav_amd.ops import sample_amax
        sc = sample_amax(x.cuda(), n, torch.empty(n).cuda())
        p2, c2 = torch.empty_like(pooled), torch.empty_like(code)
        conv.forward_pool(x.cuda(), b.cuda(), p2, c2, n=n, in_amax=sc)
        assert torch.equal(p2, pooled) and torch.equal(c2, code)
    else:
        assert cin <= 4       # the few-channel first layers find their scales inside their kernels (csrc/fconv.hip)
    close(pooled, want.float())
    close(pooled[1], want[1].float())                      # the faint sample on its own scale
    assert int(code.max()) < 8
    dpool = torch.randn(want.shape, generator=g)
    want.backward(dpool.double())
    dz = maxpool2_backward_idx(dpool.cuda(), code, oh, oh).cpu()
    # windows whose two largest activations are closer than fp32 noise may route either way: compare where the decision is clear
    top2 = F.unfold(a.detach().reshape(n * cout, 1, oh, oh), 2, stride=2).topk(2, dim=1).values
    clear = ((top2[:, 0] - top2[:, 1]).abs() > 1e-5 * (1 + top2[:, 0].abs())) | (top2[:, 0] == 0)
    clear &= (top2[:, 0] == 0) | (top2[:, 0] > 1e-5)
    clear = clear.reshape(n, cout, oh // 2, oh // 2)
    mask = clear.repeat_interleave(2, 2).repeat_interleave(2, 3)
    assert float(clear.double().mean()) > 0.99
    assert torch.equal(dz[mask], z.grad.float()[mask])
    # the backward straight from d(pooled) + decision bytes equals the backward from the unpooled gradient (same kernels, the
    # gradient formed while it is staged): weight / bias gradient, and the data gradient where the layer has one
    dzd = dz.cuda()
    dw_a, db_a = torch.empty_like(wt).cuda(), torch.empty(cout).cuda()
    dw_b, db_b = torch.empty_like(wt).cuda(), torch.empty(cout).cuda()
    conv.wgrad(x.cuda(), dzd, dw_a, db_a)
    conv.wgrad_pooled(x.cuda(), dpool.cuda(), code, dw_b, db_b)
    close(dw_b, dw_a, tol=2e-6)
    close(db_b, db_a, tol=2e-6)
    want_dw = torch.nn.grad.conv2d_weight(x.double(), wt.shape, dz.double(), padding=pad)
    close(dw_b, want_dw.float())
    if cin > 4:
        din_a = conv.dgrad(dzd)
        din_b = conv.dgrad_pooled(dpool.cuda(), code)
        close(din_b, din_a, tol=2e-6)
        close(din_b, torch.nn.grad.conv2d_input(x.shape, wt.double(), dz.double(), padding=pad).float())
        close(din_b[1], torch.nn.grad.conv2d_input(x.shape, wt.double(), dz.double(), padding=pad)[1].float())


def test_conv_strided_sample_layout():
    """in_sn / out_sn: the layer reads and writes slices of wider per-sample records."""
    from ddrl4nav_amd import _lib
    from ddrl4nav_amd.ops import Conv, _p, _st
    from ctypes import byref
    n, cin, h, w, cout = 3, 2, 8, 8, 5
    g = torch.Generator().manual_seed(1)
    x = torch.randn(n, cin * h * w + 13, generator=g)
    wt = torch.randn(cout, cin, 3, 3, generator=g)
    b = torch.randn(cout, generator=g)
    conv = Conv(cin, h, w, cout, 3, 3, pad=(1, 1), max_n=n)
    conv.pack(wt.cuda())
    out = torch.zeros(n, cout * h * w + 6).cuda()
    d = conv.desc(n, in_sn=x.shape[1], out_sn=out.shape[1])
    xd, bd = x.cuda(), b.cuda()  # keep the device copies alive across the asynchronous launch
    _lib.check(_lib.load().ddrl_op_conv_forward(byref(d), _p(xd), _p(conv.packed), _p(bd), 0, _p(out), _p(conv.scratch), _p(None), _st()))
    want = F.conv2d(x[:, :cin * h * w].reshape(n, cin, h, w), wt, b, padding=1)
    close(out[:, :cout * h * w].reshape(n, cout, h, w), want)
    assert float(out[:, cout * h * w:].abs().max()) == 0.0


@pytest.mark.parametrize("planes_hw", [((3, 5), 44, 44), ((2, 7), 20, 20), ((1, 1), 2, 2), ((4, 3), 10, 6)])
def test_maxpool_relu_pair_vs_torch(planes_hw):
    from ddrl4nav_amd.ops import maxpool2, maxpool2_relu_backward, maxpool2_idx, maxpool2_backward_idx
    (n, c), h, w = planes_hw
    g = torch.Generator().manual_seed(h * 100 + w)
    z = torch.randn(n, c, h, w, generator=g)
    z[0, 0, :2, :2] = -1.0                      # a window that is entirely clipped by the ReLU
    z[0, 0, 0, 0] = z[0, 0, 0, 1] = 0.75 if h > 2 else -1.0  # an exact tie inside a window
    z.requires_grad_(True)
    a = F.relu(z)
    pooled = F.max_pool2d(a, 2, stride=2)
    dpool = torch.randn(pooled.shape, generator=g)
    pooled.backward(dpool)
    ad = a.detach().cuda()
    assert torch.equal(maxpool2(ad).cpu(), pooled.detach())
    assert torch.equal(maxpool2_relu_backward(ad, dpool.cuda()).cpu(), z.grad)
    # the pair that keeps one decision byte per window instead of re-reading the activations
    out, code = maxpool2_idx(ad)
    assert torch.equal(out.cpu(), pooled.detach())
    assert int(code.max()) < 8
    assert torch.equal(maxpool2_backward_idx(dpool.cuda(), code, h, w).cpu(), z.grad)


LINEARS = [(300, 7616, 256), (257, 6400, 512), (130, 773, 512), (64, 512, 512), (1, 512, 512), (33, 4, 128), (5, 37, 12),
           (1100, 1024, 192), (128, 128, 64)]  # several row tiles + split K; the smallest layer / launch of the 16-bit plane kernels


@pytest.mark.parametrize("shape", LINEARS)
def test_linear_forward_backward_vs_torch(shape):
    from ddrl4nav_amd.ops import Linear
    n, K, N = shape
    g = torch.Generator().manual_seed(n + K + N)
    ld_in = (K + 3) // 4 * 4 + 4  # padded leading dimension (as in the concat buffer)
    xfull = torch.zeros(n, ld_in)
    xfull[:, :K] = torch.relu(torch.randn(n, K, generator=g))
    x = xfull[:, :K].clone().requires_grad_(True)
    wt = (torch.randn(N, K, generator=g) / K ** 0.5).requires_grad_(True)
    b = torch.randn(N, generator=g).requires_grad_(True)
    z = F.linear(x, wt, b)
    dz = torch.randn(n, N, generator=g)
    z.backward(dz)
    lin = Linear(K, N, max_n=n)
    lin.pack(wt.detach().cuda())
    xd, bd, dzd = xfull.cuda(), b.detach().cuda(), dz.cuda()
    out = torch.full((n, N + 8), 3.0).cuda()
    lin.forward(xd, ld_in, bd, True, out, N + 8, n)
    close(out[:, :N], F.relu(z))
    assert float((out[:, N:] - 3.0).abs().max()) == 0.0
    lin.forward(xd, ld_in, bd, False, out, N + 8, n)
    close(out[:, :N], z)
    din = torch.full((n, ld_in), 5.0).cuda()
    lin.dgrad(dzd, N, None, 0, din, ld_in, n)
    close(din[:, :K], x.grad)
    # with the ReLU mask of the producing layer (its output = this layer's input)
    lin.dgrad(dzd, N, xd, ld_in, din, ld_in, n)
    close(din[:, :K], x.grad * (x.detach() > 0))
    dw = torch.full((N, K), 9.0).cuda()
    db = torch.full((N,), 9.0).cuda()
    lin.wgrad(xd, ld_in, dzd, N, dw, db, n)
    close(dw, wt.grad)
    close(db, b.grad)


def test_linear_rows_of_very_different_magnitude_keep_their_precision():
    """The plane kernels of the dense layers (csrc/plin.hip) scale every ROW of the activations / gradients by its own power of two:
    a sample 1e-6 below the batch's largest keeps fp32 accuracy relative to ITSELF, an all-zero sample does not pin the batch scale
    of the weight gradient."""
    from ddrl4nav_amd.ops import Linear
    n, K, N = 384, 1536, 256
    g = torch.Generator().manual_seed(5)
    mag = 10.0 ** (torch.rand(n, 1, generator=g, dtype=torch.float64) * 8.0 - 6.0)
    mag[7] = 0.0
    x = (torch.relu(torch.randn(n, K, generator=g, dtype=torch.float64)) * mag).float()
    dz = (torch.randn(n, N, generator=g, dtype=torch.float64) * mag.flip(0)).float()
    wt = torch.randn(N, K, generator=g) / K ** 0.5
    b = torch.zeros(N)
    lin = Linear(K, N, max_n=n)
    lin.pack(wt.cuda())
    out = torch.empty(n, N).cuda()
    lin.forward(x.cuda(), K, b.cuda(), False, out, N, n)
    want = x.double() @ wt.double().t()
    row = want.abs().amax(1, keepdim=True).clamp_min(1e-300)
    assert float(((out.cpu().double() - want).abs() / row).max()) < 2e-6      # per row, not per batch
    assert float(out[7].abs().max()) == 0.0
    din = torch.empty(n, K).cuda()
    lin.dgrad(dz.cuda(), N, None, 0, din, K, n)
    want = dz.double() @ wt.double()
    row = want.abs().amax(1, keepdim=True).clamp_min(1e-300)
    assert float(((din.cpu().double() - want).abs() / row).max()) < 2e-6
    dw, db = torch.empty(N, K).cuda(), torch.empty(N).cuda()
    lin.wgrad(x.cuda(), K, dz.cuda(), N, dw, db, n)
    want = dz.double().t() @ x.double()
    ref32 = dz.t() @ x                                                       # torch fp32 on the CPU
    err, err32 = (dw.cpu().double() - want).abs().max().item(), (ref32.double() - want).abs().max().item()
    assert err <= 2.0 * err32 + 1e-30, (err, err32)
    close(db, dz.double().sum(0).float())


def test_ops_reject_bad_arguments():
    from ddrl4nav_amd import _lib
    from ddrl4nav_amd.ops import Conv, Linear
    with pytest.raises(_lib.DdrlError):
        Conv(3, 8, 8, 4, 3, 3, stride=3)          # stride must be 1, 2 or 4
    with pytest.raises(_lib.DdrlError):
        Linear(16, 6)                              # N must be a multiple of 4
    lin = Linear(16, 8, max_n=4)
    x = torch.zeros(4, 16).cuda()
    with pytest.raises(_lib.DdrlError):
        lin.forward(x, 15, torch.zeros(8).cuda(), False, torch.zeros(4, 8).cuda(), 8, 4)  # ld not a multiple of 4


@pytest.mark.parametrize("shape", [(512, 3, 48, 48, 64, 7, 7, 1, (1, 1)), (512, 64, 22, 22, 128, 5, 5, 1, (1, 1)),
                                   (512, 128, 10, 10, 256, 3, 3, 1, (1, 1)), (512, 32, 1, 478, 32, 1, 3, 2, (0, 0))])
def test_conv_full_batch_size_vs_torch_gpu(shape):
    """BASELINE config 4 batch (512 envs): the nav layers at n = 512 against torch's own GPU
    convolution (MIOpen) -- a second, independent implementation at a size the CPU oracle would
    take minutes for.  Both sides are fp32 with different summation orders."""
    from ddrl4nav_amd.ops import Conv
    n, cin, h, w, cout, kh, kw, s, pad = shape
    g = torch.Generator(device="cuda").manual_seed(n + cin)
    x = torch.randn(n, cin, h, w, device="cuda", generator=g).requires_grad_(True)
    wt = (torch.randn(cout, cin, kh, kw, device="cuda", generator=g) / (cin * kh * kw) ** 0.5).requires_grad_(True)
    b = torch.randn(cout, device="cuda", generator=g).requires_grad_(True)
    z = F.conv2d(x, wt, b, stride=s, padding=pad)
    dz = torch.randn(z.shape, device="cuda", generator=g)
    z.backward(dz)
    conv = Conv(cin, h, w, cout, kh, kw, stride=s, pad=pad, max_n=n)
    conv.pack(wt.detach())
    close(conv.forward(x.detach(), b.detach(), relu=False), z, tol=5e-5)
    close(conv.dgrad(dz), x.grad, tol=5e-5)
    dw, db = torch.empty_like(wt.detach()), torch.empty_like(b.detach())
    conv.wgrad(x.detach(), dz, dw, db)
    close(dw, wt.grad, tol=2e-4)   # sums over 512 x oh x ow signed terms
    close(db, b.grad, tol=2e-4)


@pytest.mark.parametrize("shape", [(512, 3, 48, 64, 7), (512, 64, 22, 128, 5), (512, 128, 10, 256, 3), (512, 4, 48, 64, 3), (512, 64, 24, 128, 3)])
def test_pooled_conv_block_full_batch_size_vs_torch_gpu(shape):
    """The same at n = 512 for the blocks that pool in their epilogue: conv + ReLU + max-pool in one launch and the backward from
    d(pooled), against torch's GPU autograd through F.max_pool2d(F.relu(conv)) (fp32, MIOpen).  Pooled values agree to fp32
    rounding; the per-window routing agrees except where the window's two largest activations lie within fp32 noise of each other
    (either implementation may take either: a few windows in 10^7, each of which moves a weight-gradient element by ~1e-3 of its
    size -- so the gradients are compared under torch's OWN routing, which separates the arithmetic from those coin flips)."""
    from ddrl4nav_amd.ops import Conv, sample_amax, maxpool2_backward_idx
    n, cin, h, cout, ks = shape
    g = torch.Generator(device="cuda").manual_seed(n + cin + h)
    x = torch.randn(n, cin, h, h, device="cuda", generator=g).requires_grad_(True)
    wt = (torch.randn(cout, cin, ks, ks, device="cuda", generator=g) / (cin * ks * ks) ** 0.5).requires_grad_(True)
    b = (0.1 * torch.randn(cout, device="cuda", generator=g)).requires_grad_(True)
    z = F.conv2d(x, wt, b, padding=1)
    z.retain_grad()
    pooled_ref = F.max_pool2d(F.relu(z), 2, stride=2)
    dpool = torch.randn(pooled_ref.shape, device="cuda", generator=g)
    pooled_ref.backward(dpool)
    conv = Conv(cin, h, h, cout, ks, ks, pad=(1, 1), max_n=n)
    conv.pack(wt.detach())
    assert conv.has_forward_pool()
    pooled, code = torch.empty_like(pooled_ref), torch.empty(pooled_ref.shape, dtype=torch.uint8, device="cuda")
    sc = sample_amax(x.detach(), n, torch.empty(n, device="cuda")) if conv.pooled_uses_scales() else None
    conv.forward_pool(x.detach(), b.detach(), pooled, code, n=n, in_amax=sc)
    close(pooled, pooled_ref, tol=5e-5)
    oh = conv.oh
    dz_k = maxpool2_backward_idx(dpool, code, oh, oh)
    differ = (dz_k != z.grad).reshape(n, cout, oh // 2, 2, oh // 2, 2).any(5).any(3)          # windows routed differently
    assert float(differ.float().mean()) < 2e-5, float(differ.float().mean())
    if differ.any():   # ... only where the decision was a coin flip: the two candidates (or the maximum and zero) within fp32 noise
        a = F.relu(z.detach()).reshape(n, cout, oh // 2, 2, oh // 2, 2).permute(0, 1, 2, 4, 3, 5).reshape(n, cout, oh // 2, oh // 2, 4)
        top2 = a[differ].topk(2, dim=1).values
        assert float(((top2[:, 0] - top2[:, 1]).abs().minimum(top2[:, 0])).max()) < 2e-5
    # gradients under torch's routing: overwrite the decision bytes of the few coin-flip windows with torch's choice
    zg = z.grad.reshape(n, cout, oh // 2, 2, oh // 2, 2).permute(0, 1, 2, 4, 3, 5).reshape(n, cout, oh // 2, oh // 2, 4)
    t_am = (zg != 0).float().argmax(4).to(torch.uint8)
    t_pos = (zg != 0).any(4) | (dpool == 0)
    code_t = torch.where(differ, t_am | (t_pos.to(torch.uint8) * 4), code)
    rel = lambda got, want: float((got.double() - want.double()).norm() / want.double().norm())
    dw, db = torch.empty_like(wt.detach()), torch.empty_like(b.detach())
    conv.wgrad_pooled(x.detach(), dpool, code_t, dw, db, n=n, in_amax=sc)
    assert rel(db, b.grad) < 2e-5, rel(db, b.grad)
    assert rel(dw, wt.grad) < 2e-5, rel(dw, wt.grad)
    if cin > 4:
        din = conv.dgrad_pooled(dpool, code_t, n=n)
        assert rel(din, x.grad) < 2e-5, rel(din, x.grad)


@pytest.mark.parametrize("order", ["critic_below_actor", "critic_above_actor"])
def test_heads_take_the_two_feature_buffers_at_any_relative_address(order):
    """ddrl_op_heads_act / ddrl_op_heads_loss receive the actor's and the critic's feature (and gradient) buffers as two
    independent pointers; the kernels address the critic's as actor + stride, and that stride may be NEGATIVE.  (A "stride < 0
    means unset" test once sent every net whose critic buffer lay below its actor buffer to actor + n * 512: wrong values in a
    fresh process, right ones whenever the allocator happened to order the two the other way.)"""
    from ctypes import byref, c_void_p
    from ddrl4nav_amd import _lib
    from ddrl4nav_amd._lib import HeadsDesc, check, default_config
    lib = _lib.load()
    n, A = 37, 5
    g = torch.Generator().manual_seed(5)
    Wa, ba = torch.randn(A, 512, generator=g) * 0.05, torch.randn(A, generator=g) * 0.1
    wc, bc = torch.randn(512, generator=g) * 0.05, torch.randn(1, generator=g)
    params = torch.cat([Wa.reshape(-1), ba, wc, bc]).cuda()
    d = HeadsDesc()
    d.continuous, d.n_actions, d.shared = 0, A, 0
    d.actor_w, d.actor_b, d.critic_w, d.critic_b, d.n_params = 0, A * 512, A * 512 + A, A * 512 + A + 512, params.numel()
    feats = torch.randn(2, n, 512, generator=g).cuda()
    ia, ic = (1, 0) if order == "critic_below_actor" else (0, 1)
    ha, hc = feats[ia], feats[ic]
    acts = torch.randint(0, A, (n,), generator=g).float().cuda()
    f = dict(dtype=torch.float32, device="cuda")
    probs, value, logp = torch.empty((n, A), **f), torch.empty(n, **f), torch.empty(n, **f)
    st = c_void_p(torch.cuda.current_stream().cuda_stream)
    p = lambda t: c_void_p(t.data_ptr())
    check(lib.ddrl_op_heads_act(byref(d), p(params), p(ha), p(hc), n, p(acts), 0, 0, p(probs), p(value), c_void_p(0), p(logp), st))
    want_v = hc.cpu() @ wc + bc
    want_p = torch.softmax(ha.cpu() @ Wa.T + ba, dim=-1)
    close(value, want_v)
    close(probs, want_p)
    # loss + head backward: d loss / d h of BOTH encoders land in their own buffers
    from ctypes import c_int64
    wf = c_int64()
    check(lib.ddrl_op_heads_ws_floats(byref(d), n, byref(wf)))
    ws = torch.empty(wf.value, **f)
    dh = torch.zeros(2, n, 512, **f)
    grads = torch.zeros(params.numel() + 8, **f)
    cfg = default_config(max_batch=n, n_actions=A)
    old = (torch.log(want_p.gather(1, acts.cpu().long()[:, None])[:, 0]) + 0.1 * torch.randn(n, generator=g)).cuda()
    adv, ret = torch.randn(n, generator=g).cuda(), torch.randn(n, generator=g).cuda()
    check(lib.ddrl_op_heads_loss(byref(d), byref(cfg), p(params), p(ha), p(hc), n, p(acts), p(old), p(adv), p(ret), n, p(dh[ia]),
                                 p(dh[ic]), p(grads), p(ws), st))
    want_dv = (want_v - ret.cpu()) / n                      # d (mean((ret - v)^2) / 2) / d v
    close(dh[ic], want_dv[:, None] * wc[None, :])
    assert float(dh[ia].abs().max()) > 0


def test_linear_launch_smaller_than_capacity_takes_more_splits():
    """A dense layer built for max_n samples and launched with fewer: the split-K count of the forward fills the chip, so 40,000
    samples take two splits where 65,536 take one -- the partial-sum workspace must hold the LARGEST splits(n') n' N over n' <= max_n
    (sizing by max_n alone overflowed it: a GPU memory fault, found by the full-size GAIL property test)."""
    from ddrl4nav_amd.ops import Linear
    K, N, cap = 516, 64, 65536
    g = torch.Generator().manual_seed(3)
    W, b = torch.randn(N, K, generator=g) * 0.05, torch.randn(N, generator=g) * 0.1
    lin = Linear(K, N, max_n=cap)
    lin.pack(W.cuda())
    guard = torch.full((1 << 20,), 7.0, device="cuda")      # a neighbour the old overflow would have written into
    for n in (40000, 25536, 300, cap):
        x = torch.randn(n, K, generator=g)
        out = torch.empty((n, N), device="cuda")
        lin.forward(x.cuda(), K, b.cuda(), True, out, N, n)
        close(out, torch.relu(x @ W.T + b))
    assert float(guard.min()) == 7.0 and float(guard.max()) == 7.0


# ---- per-sample magnitudes left by the producers (include/ddrl.h "per-sample magnitudes", round 5) ------------------------------------
@pytest.mark.parametrize("shape", [(37, 3, 48, 64, 7), (9, 64, 22, 128, 5), (11, 128, 10, 256, 3), (7, 1, 48, 64, 3), (6, 64, 24, 128, 3), (5, 128, 12, 256, 3)])
def test_pooled_forward_and_data_gradient_leave_exact_sample_magnitudes(shape):
    """ddrl_op_conv_forward_pool(out_amax) / ddrl_op_conv_dgrad_pooled(din_amax): the arrays hold EXACTLY max |value written| per sample
    (the operators raise a zeroed array with an atomic maximum; several row tiles meet in one slot), a sample decades below the rest
    keeps its own magnitude, an all-zero gradient sample stays at 0, and a pre-raised slot is never lowered."""
    from ddrl4nav_amd.ops import Conv, sample_amax
    n, cin, h, cout, ks = shape
    g = torch.Generator(device="cuda").manual_seed(sum(shape))
    x = torch.randn(n, cin, h, h, device="cuda", generator=g)
    x[1] *= 1e-4
    wt = torch.randn(cout, cin, ks, ks, device="cuda", generator=g) / (cin * ks * ks) ** 0.5
    b = 0.1 * torch.randn(cout, device="cuda", generator=g)
    conv = Conv(cin, h, h, cout, ks, ks, pad=(1, 1), max_n=n)
    conv.pack(wt)
    assert conv.has_forward_pool()
    oh = conv.oh
    pooled = torch.empty((n, cout, oh // 2, oh // 2), device="cuda")
    code = torch.empty((n, cout, oh // 2, oh // 2), dtype=torch.uint8, device="cuda")
    out_amax = torch.zeros(n, device="cuda")
    out_amax[2] = 1e30                                       # never lowered
    in_amax = sample_amax(x, n, torch.empty(n, device="cuda")) if conv.pooled_uses_scales() else None
    assert in_amax is None or torch.equal(in_amax, x.abs().amax(dim=(1, 2, 3)))
    conv.forward_pool(x, b, pooled, code, n=n, in_amax=in_amax, out_amax=out_amax)
    want = pooled.amax(dim=(1, 2, 3))
    want[2] = 1e30
    assert torch.equal(out_amax, want)
    if cin > 4:                                              # first layers have no data gradient
        dpool = torch.randn(pooled.shape, device="cuda", generator=g)
        dpool[3] = 0.0
        dpool[1] *= 1e-6
        din = torch.empty((n, cin, h, h), device="cuda")
        din_amax = torch.zeros(n, device="cuda")
        conv.dgrad_pooled(dpool, code, din=din, n=n, din_amax=din_amax)
        assert torch.equal(din_amax, din.abs().amax(dim=(1, 2, 3))) and float(din_amax[3]) == 0.0 and float(din_amax[1]) > 0.0


@pytest.mark.parametrize("n", [300, 40])       # the fp16 plane kernels (n >= 128 rows) and the f32-input kernels below that
def test_linear_data_gradient_leaves_row_magnitudes_of_a_column_slice(n):
    """ddrl_op_linear_dgrad(din_amax, amax_lo, amax_hi): exact row maxima of |din| over the slice a consumer behind a torch.cat reads
    (NavPreNet1D: fc1's data gradient feeds fc_1d through columns 0..255 and fc0 through 256..767), whole rows by default; masked
    elements count as the zeros they are written as."""
    from ddrl4nav_amd.ops import Linear
    K, N, ld = 773, 512, 776
    g = torch.Generator(device="cuda").manual_seed(n)
    w = torch.randn(N, K, device="cuda", generator=g) / K ** 0.5
    dout = torch.randn(n, N, device="cuda", generator=g)
    dout[5] *= 1e-5
    dout[6] = 0.0
    mask = torch.randn(n, ld, device="cuda", generator=g)
    lin = Linear(K, N, max_n=n)
    lin.pack(w)
    for cols in (None, (0, 256), (256, 768)):
        din = torch.zeros(n, ld, device="cuda")
        amax = torch.zeros(n, device="cuda")
        lin.dgrad(dout, N, mask, ld, din, ld, n, din_amax=amax, amax_cols=cols)
        lo, hi = cols if cols is not None else (0, K)
        assert torch.equal(amax, din[:, lo:hi].abs().amax(dim=1)), cols
        assert float(amax[6]) == 0.0
    close(din[:, :K], (dout @ w) * (mask[:, :K] > 0))


def test_conv1d_forward_leaves_sample_magnitudes():
    """The laser branch's second Conv1d (csrc/c1d.hip) leaves max |output| per sample for the dense layer that reads its rows; other
    geometries take a pass over the output they just wrote (same values)."""
    from ddrl4nav_amd.ops import Conv
    for (n, cin, w_, cout, kw, stride) in ((133, 32, 478, 32, 3, 2), (5, 4, 64, 8, 4, 2)):
        g = torch.Generator(device="cuda").manual_seed(n)
        x = torch.randn(n, cin, 1, w_, device="cuda", generator=g)
        x[2] *= 1e-3
        wt = torch.randn(cout, cin, 1, kw, device="cuda", generator=g) / (cin * kw) ** 0.5
        b = 0.1 * torch.randn(cout, device="cuda", generator=g)
        conv = Conv(cin, 1, w_, cout, 1, kw, stride=stride, max_n=n)
        conv.pack(wt)
        amax = torch.zeros(n, device="cuda")
        out = conv.forward(x, b, False, n=n, out_amax=amax)
        assert torch.equal(amax, out.abs().amax(dim=(1, 2, 3)))
        close(out, F.conv2d(x, wt, b, stride=(1, stride)))
