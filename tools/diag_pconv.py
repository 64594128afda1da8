#!/usr/bin/env python3
"""Accuracy of the plane convolution kernels (csrc/pconv.hip) against float64, beside torch's fp32 operator on the same inputs.
Test infrastructure (uses torch CPU as the yardstick).  usage (GPU box): python tools/diag_pconv.py"""
import os
import sys

import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ddrl4nav_amd.ops import Conv  # noqa: E402

for (n, cin, h, cout, ks) in ((16, 64, 22, 128, 5), (32, 128, 10, 256, 3), (8, 64, 24, 128, 3), (16, 128, 12, 256, 3)):
    for kind in ("dense", "sparse"):
        g = torch.Generator().manual_seed(n + cin)
        x = torch.randn(n, cin, h, h, generator=g)
        if kind == "sparse":   # ReLU + pooled-gradient like: many zeros, wide per-sample range
            x = torch.relu(x) * (10.0 ** torch.empty(n, 1, 1, 1).uniform_(-3, 1, generator=g))
        wt = torch.randn(cout, cin, ks, ks, generator=g) / (cin * ks * ks) ** 0.5
        b = torch.zeros(cout)
        oh = h + 2 - ks + 1
        dz = torch.randn(n, cout, oh, oh, generator=g)
        if kind == "sparse":
            dz = dz * (torch.rand(dz.shape, generator=g) < 0.25) * (10.0 ** torch.empty(n, 1, 1, 1).uniform_(-3, 1, generator=g))
        z64 = F.conv2d(x.double(), wt.double(), None, padding=1)
        d64 = torch.nn.grad.conv2d_input(x.shape, wt.double(), dz.double(), padding=1)
        z32 = F.conv2d(x, wt, None, padding=1).double()
        d32 = torch.nn.grad.conv2d_input(x.shape, wt, dz, padding=1).double()
        conv = Conv(cin, h, h, cout, ks, ks, pad=(1, 1), max_n=n)
        conv.pack(wt.cuda())
        zh = conv.forward(x.cuda(), b.cuda(), relu=False).cpu().double()
        dh = conv.dgrad(dz.cuda()).cpu().double()

        def per_sample(e, ref):   # rms error per sample relative to that sample's rms value: worst and median sample
            r = (e.reshape(n, -1).pow(2).mean(1).sqrt() / ref.reshape(n, -1).pow(2).mean(1).sqrt().clamp_min(1e-300))
            return "%.2e / %.2e" % (r.max().item(), r.median().item())
        print("%dx%d %d->%d @%d %-6s fwd hip %s  torch32 %s | dgrad hip %s  torch32 %s" % (
            ks, ks, cin, cout, h, kind, per_sample(zh - z64, z64), per_sample(z32 - z64, z64), per_sample(dh - d64, d64), per_sample(d32 - d64, d64)))
