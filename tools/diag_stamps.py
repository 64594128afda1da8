"""Diagnostic only: phase shares (prologue / compute / commit+fetch / barrier / epilogue) of the
engine kernels, from s_memtime stamps of an instrumented build (see tools notes in DESIGN.md)."""
import ctypes, sys, torch
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ddrl4nav_amd import _lib
_lib.LIB_PATH = os.path.join(os.path.dirname(_lib.LIB_PATH), "libddrl_hip_diag.so")  # built by tools/make_diag_build.sh
from ddrl4nav_amd.engine import HotPath
from ddrl4nav_amd.utils.recipe import flatten, make_weights
B = 65536
hp = HotPath(max_batch=B); hp.set_params(flatten(make_weights(0)))
g = torch.Generator(device="cuda"); g.manual_seed(1)
fr = torch.randint(0, 256, (B, 4, 84, 84), dtype=torch.uint8, device="cuda", generator=g)
a = torch.randint(0, 6, (B,), device="cuda", generator=g).float(); old = torch.full((B,), -1.79, device="cuda")
adv = torch.randn(B, device="cuda", generator=g); ret = torch.randn(B, device="cuda", generator=g)
lib = _lib.load()
names = ['ConvFwd2', 'ConvFwd3', 'ConvFwd1', 'ConvDgrad3', 'ConvDgrad2', 'ConvWgrad1', 'ConvWgrad2', 'ConvWgrad3', 'FcFwd', 'FcDgrad', 'FcWgrad']
def read(reset):
    tot = [0] * 256
    for tu in ("conv2", "wgrad2", "fc2"):
        buf = (ctypes.c_ulonglong * 256)()
        getattr(lib, "ddrl_debug_stamps_" + tu)(buf, reset)
        tot = [x + y for x, y in zip(tot, buf)]
    return tot
hp.ppo_iter(fr, a, old, adv, ret); torch.cuda.synchronize(); read(1)
hp.ppo_iter(fr, a, old, adv, ret); torch.cuda.synchronize()
t = read(1)
print("%-11s %8s %7s %7s %7s %7s %7s %7s %7s   (shares of wave lifetime; kcyc = mean wave lifetime)" % ("kernel", "kcyc", "prolog", "compute", "vmwait", "commit", "fetch", "barrier", "epilog"))
for i, nm in enumerate(names):
    pro, comp, com, bar, epi, life, cnt, vw = t[8 * i:8 * i + 8]
    tf = t[200 + i]
    if cnt:
        print("%-11s %8.1f %6.1f%% %6.1f%% %6.1f%% %6.1f%% %6.1f%% %6.1f%% %6.1f%%" % (nm, life / cnt / 1e3, 100 * pro / life, 100 * comp / life, 100 * vw / life, 100 * com / life, 100 * tf / life, 100 * bar / life, 100 * epi / life))
