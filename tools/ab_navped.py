"""Same-box A/B of the SHARED NavPedPreNet(4) net's PPO iteration (3x3 @48 / @24 / @12 conv stack; BASELINE config 5's generator) with a
given build of the library:  python tools/ab_navped.py <lib.so>"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from ddrl4nav_amd import _lib  # noqa: E402

_lib.LIB_PATH = os.path.abspath(sys.argv[1])
import bench_nav  # noqa: E402

r = bench_nav.run(4096, 4096, 3, encoder="navped")
print(os.path.basename(_lib.LIB_PATH), "navped iter %.2f ms" % r["ms_per_ppo_iter_wall"],
      " ".join("%s %.2f" % (k.replace("conv", "c").replace("_", ""), v["ms_per_iter"]) for k, v in r["ops"].items() if v["ms_per_iter"] > 0.8))
