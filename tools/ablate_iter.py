"""Per-kernel HIP-event times of three PPO iterations at the bench shape (B = 65,536): the A/B driver of the
kernel experiments recorded in profiles/README.md.  Usage: python tools/ablate_iter.py <tag>"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ddrl4nav_amd import _lib
if os.environ.get("DDRL_ABL_LIB"): _lib.LIB_PATH = os.environ["DDRL_ABL_LIB"]  # knock-out builds of tools/ablate_engine.sh
from ddrl4nav_amd.engine import HotPath
from ddrl4nav_amd.utils.recipe import flatten, make_weights
B=65536
hp=HotPath(max_batch=B); hp.set_params(flatten(make_weights(0)))
g=torch.Generator(device="cuda"); g.manual_seed(1)
fr=torch.randint(0,256,(B,4,84,84),dtype=torch.uint8,device="cuda",generator=g)
a=torch.randint(0,6,(B,),device="cuda",generator=g).float(); old=torch.full((B,),-1.79,device="cuda"); adv=torch.randn(B,device="cuda",generator=g); ret=torch.randn(B,device="cuda",generator=g)
hp.ppo_iter(fr,a,old,adv,ret); torch.cuda.synchronize()
hp.profile(True)
for _ in range(3): hp.ppo_iter(fr,a,old,adv,ret)
torch.cuda.synchronize()
p=hp.profile_read()
print(sys.argv[1], " ".join("%s %.2f"%(k.replace("Conv",""), v[0]/3) for k,v in sorted(p.items()) if v[0]/3>0.1))
gr=hp.grads[:hp.n_params].double()
print(sys.argv[1], "grad |sum| %.9e  L2 %.9e  finite %s" % (float(gr.abs().sum()), float(gr.norm()), bool(torch.isfinite(gr).all())))
