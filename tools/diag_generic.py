import sys, numpy as np, torch
sys.path.insert(0, ".")
sys.path.insert(0, "tests")
import types
from test_generic_gpu import _make, _states
from ddrl4nav_amd.utils.recipe import hash_weights
from oracle import ddrl_oracle_nav as N
name = sys.argv[1] if len(sys.argv) > 1 else "f14_navped_shared"
g = np.load("tests/golden/%s.npz" % name)
net, w = _make(name, max_batch=256)
case = {"f13_nav1d_gauss": (lambda: N.NavPreNet1D(3), 2, True, False), "f14_navped_shared": (lambda: N.NavPedPreNet(4), 5, False, True),
        "f15_mlp_classical": (lambda: N.MLPPreNet(4, 512), 2, False, False)}[name]
onet = N.OracleNet(*case); onet.load_weights(w)
states = _states(g); B = len(g["advs"])
t = lambda k: torch.from_numpy(g[k])
total, al, vl, ent = N.losses(onet, [torch.from_numpy(s) for s in states], t("actions"), t("old_logps"), t("advs"), t("rets"))
if case[3]: total.backward()
else: al.backward(); vl.backward()
dev = lambda k: torch.from_numpy(g[k]).cuda()
net._ensure_packed()
net._iter_chunk(net._stage(states, 0, B), B, dev("actions"), dev("old_logps"), dev("advs"), dev("rets"), B)
flat = net.gtmp[:net.n_params].cpu().numpy(); off = 0
for (k, p) in onet.named_parameters():
    n = p.numel(); got = flat[off:off+n]; off += n
    want = p.grad.numpy().reshape(-1)
    d = np.abs(got - want); sc = np.abs(want).max()
    cos = (got.astype(np.float64) @ want) / (np.linalg.norm(got.astype(np.float64)) * np.linalg.norm(want.astype(np.float64)) + 1e-300)
    print("%-28s max|d|/max|g| %.2e  frac(d>1e-4*max) %.4f  1-cos %.2e  |g|max %.2e" % (k, d.max()/sc, (d > 1e-4*sc).mean(), 1-cos, sc))
