#!/usr/bin/env python3
"""Accuracy of the generic operators' gradients against float64, beside torch's fp32 CPU result (GPU box)."""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from ddrl4nav_amd.ops import Conv, Linear

def conv_case(n, cin, h, w, cout, k, pad, seed):
    rng = np.random.default_rng(seed)
    x = torch.from_numpy(rng.normal(size=(n, cin, h, w)).astype(np.float32)).relu()
    W = torch.from_numpy((rng.normal(size=(cout, cin, k, k)) / np.sqrt(cin * k * k)).astype(np.float32))
    oh, ow = h + 2 * pad - k + 1, w + 2 * pad - k + 1
    dz = torch.from_numpy((rng.normal(size=(n, cout, oh, ow)) * (rng.random((n, cout, oh, ow)) < 0.5)).astype(np.float32))
    conv = Conv(cin, h, w, cout, k, k, stride=1, pad=(pad, pad), max_n=n)
    conv.pack(W.cuda())
    dw = torch.empty_like(W).cuda(); db = torch.empty(cout).cuda()
    conv.wgrad(x.cuda(), dz.cuda(), dw, db, n=n)
    din = conv.dgrad(dz.cuda(), n=n).cpu().double().numpy()
    out = conv.forward(x.cuda(), torch.zeros(cout).cuda(), False, n=n).cpu().double().numpy()
    g = torch.nn.grad
    ref = g.conv2d_weight(x.double(), W.shape, dz.double(), padding=pad).numpy()
    f32 = g.conv2d_weight(x, W.shape, dz, padding=pad).double().numpy()
    mass = g.conv2d_weight(x.double().abs(), W.shape, dz.double().abs(), padding=pad).numpy().max()
    u = 2.0 ** -24 * mass
    k_ = dw.cpu().double().numpy()
    print("conv %dx%dx%d->%d k%d n=%d  wgrad: kernel %.2f units (max|g| rel %.2e)  torch-f32 %.2f units" % (
        cin, h, w, cout, k, n, np.abs(k_ - ref).max() / u, np.abs(k_ - ref).max() / np.abs(ref).max(), np.abs(f32 - ref).max() / u))
    refd = g.conv2d_input(x.shape, W.double(), dz.double(), padding=pad).numpy()
    f32d = g.conv2d_input(x.shape, W, dz, padding=pad).double().numpy()
    massd = g.conv2d_input(x.shape, W.double().abs(), dz.double().abs(), padding=pad).numpy().max()
    print("      dgrad: kernel %.2f units  torch-f32 %.2f units" % (np.abs(din - refd).max() / (2.0 ** -24 * massd), np.abs(f32d - refd).max() / (2.0 ** -24 * massd)))
    reff = torch.nn.functional.conv2d(x.double(), W.double(), padding=pad).numpy()
    f32f = torch.nn.functional.conv2d(x, W, padding=pad).double().numpy()
    massf = torch.nn.functional.conv2d(x.double().abs(), W.double().abs(), padding=pad).numpy().max()
    print("      fwd:   kernel %.2f units  torch-f32 %.2f units" % (np.abs(out - reff).max() / (2.0 ** -24 * massf), np.abs(f32f - reff).max() / (2.0 ** -24 * massf)))

def lin_case(n, K, N, seed):
    rng = np.random.default_rng(seed)
    ld = (K + 3) // 4 * 4
    x = torch.zeros(n, ld); x[:, :K] = torch.from_numpy(rng.normal(size=(n, K)).astype(np.float32)).relu()
    dz = torch.from_numpy(rng.normal(size=(n, N)).astype(np.float32))
    lin = Linear(K, N, max_n=n)
    dw = torch.empty(N, K).cuda(); db = torch.empty(N).cuda()
    lin.wgrad(x.cuda(), ld, dz.cuda(), N, dw, db, n)
    ref = (dz.double().T @ x[:, :K].double()).numpy(); f32 = (dz.T @ x[:, :K]).double().numpy()
    u = 2.0 ** -24 * (dz.double().abs().T @ x[:, :K].double().abs()).numpy().max()
    print("linear K=%d N=%d n=%d wgrad: kernel %.2f units  torch-f32 %.2f units" % (K, N, n, np.abs(dw.cpu().double().numpy() - ref).max() / u, np.abs(f32 - ref).max() / u))

torch.set_num_threads(8)
for n in (18, 256):
    conv_case(n, 4, 48, 48, 64, 3, 1, 1)
    conv_case(n, 64, 24, 24, 128, 3, 1, 2)
    conv_case(n, 128, 12, 12, 256, 3, 1, 3)
conv_case(20, 3, 48, 48, 64, 7, 1, 4)
conv_case(20, 64, 22, 22, 128, 5, 1, 5)
lin_case(18, 9216, 512, 6); lin_case(256, 9216, 512, 7); lin_case(20, 521, 512, 8)
