#!/usr/bin/env python3
"""Diagnostic (GPU box): per-tensor comparison of the discriminator step against the oracle -- gradients before the clip,
and parameter deviations after the RMSprop step relative to the reference's own fp32 spread.
usage: python tools/diag_gail.py [f16_gail_classical|f17_gail_atari]"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    import parity_util as P
    import test_gail_gpu as T
    from ddrl4nav_amd.data import Experience
    from oracle import ddrl_oracle_gail as G
    name = sys.argv[1] if len(sys.argv) > 1 else "f17_gail_atari"
    golden = lambda n: np.load(os.path.join(ROOT, "tests", "golden", n + ".npz"))
    g, net, states, w = T._net(name, golden)
    exp = Experience(states=[states], advs=g["advs"], actions=g["actions"], old_logps=g["old_logps"], values=g["rets"])
    D = net.discriminator
    item = next(D.learn(exp))
    st = D.stats()
    print("HIP  D loss %.9f  norm %.6e coef %.6e" % (item[0]["Gail[D]Loss"], st["GradNorm"], st["ClipCoef"]), " ref D loss", g["d_loss"])
    # oracle gradient (fp32 and fp64)
    for dt in (torch.float32, torch.float64):
        _, onet, states_np, seed = P.gail_oracle(name)
        onet.load_weights(w)
        onet.to(dt)
        t = lambda k: torch.from_numpy(g[k]).to(dt)
        s = [torch.from_numpy(states_np).to(dt)]
        ex = [torch.from_numpy(states_np[g["expert_index"]][::-1].copy()).to(dt)]
        Dn = onet.discriminator
        loss = torch.mean(Dn((s, t("actions").reshape(-1, 1)))) - torch.mean(Dn((ex, t("expert_actions"))))
        loss.backward()
        gn = float(torch.sqrt(sum((p.grad.double() ** 2).sum() for p in Dn.parameters())))
        print("oracle", dt, "loss %.9f norm %.6e" % (loss.item(), gn))
        if dt == torch.float64:
            og = {k: p.grad.numpy() for k, p in Dn.named_parameters()}
    hip_g = {}
    for k, p in D.named_parameters():
        off = (p.data_ptr() - D.params.data_ptr()) // 4
        hip_g[k] = (D.grads[off:off + p.numel()].cpu().numpy().reshape(p.shape) / st["ClipCoef"])
    for k in og:
        d = np.abs(hip_g[k] - og[k]).max()
        print("  grad %-28s max|g| %.3e  max|dg| %.3e  rel %.2e" % (k, np.abs(og[k]).max(), d, d / (np.abs(og[k]).max() + 1e-300)))
    traj = P.gail_f64_trajectory(name)
    got = T._params(net)
    for k in got:
        key = "D1/" + k
        if float(g["upd_l2/" + key]) == 0.0:
            continue
        a, p64 = got[k].astype(np.float64), traj["params"]["D1"][k]
        d = a - p64
        print("  param %-40s |d|2 %.3e (ref %.3e, ratio %8.1f)  max %.3e (ref %.3e, ratio %8.1f)" % (
            k, np.linalg.norm(d), g["ref_l2/" + key], np.linalg.norm(d) / g["ref_l2/" + key], np.abs(d).max(), g["ref_max/" + key],
            np.abs(d).max() / g["ref_max/" + key]))


if __name__ == "__main__":
    main()
