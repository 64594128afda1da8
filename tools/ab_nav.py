"""Same-box A/B of the robot_nav PPO iteration (and, with DDRL_AB_NAVPED=1, of the shared NavPedPreNet(4) net's) with a given build of the
library:  python tools/ab_nav.py <lib.so> [B] [CAP]   (alternate builds in one gpurun call: A B A B)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from ddrl4nav_amd import _lib  # noqa: E402

_lib.LIB_PATH = os.path.abspath(sys.argv[1])
import bench_nav  # noqa: E402

r = bench_nav.run(int(sys.argv[2]) if len(sys.argv) > 2 else 4096, int(sys.argv[3]) if len(sys.argv) > 3 else 4096, 3)
extra = ""
if os.environ.get("DDRL_AB_NAVPED"):
    extra = "  navped iter %.2f ms" % bench_nav.run(4096, 4096, 3, encoder="navped")["ms_per_ppo_iter_wall"]
print(os.path.basename(_lib.LIB_PATH), "iter %.2f ms%s" % (r["ms_per_ppo_iter_wall"], extra),
      " ".join("%s %.2f" % (k.replace("conv", "c").replace("_", ""), v["ms_per_iter"]) for k, v in r["ops"].items() if v["ms_per_iter"] > 1.5))
