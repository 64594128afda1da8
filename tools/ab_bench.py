"""Same-box A/B of whole bench steps with a given build of the library:
    python tools/ab_bench.py tools/_scratch_abl/<name>.so
runs `bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-async` on that build and prints one line (env-steps/s,
ms per PPO iteration, acting ms per rollout, per-kernel ms).  Alternate builds in one gpurun call (A B A B):
profiles/r01_v7_same_box_ab.txt and r01_v8_same_box_ab.txt were made this way."""
import contextlib
import io
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ddrl4nav_amd import _lib  # noqa: E402

_lib.LIB_PATH = os.path.abspath(sys.argv[1])
sys.argv = ["bench.py", "--steps", "3", "--warmup", "1", "--no-cpu-baseline", "--no-async"]
import bench  # noqa: E402

buf = io.StringIO()
with contextlib.redirect_stdout(buf):
    bench.main()
d = json.loads(buf.getvalue().strip().splitlines()[-1])
print(os.path.basename(_lib.LIB_PATH), "value %.0f  ppo_iter_ms %.2f  acting %.2f" % (d["value"], d["ppo_iter_ms"], d["acting_ms_per_rollout"]),
      " ".join("%s %.2f" % (k.replace("Conv", ""), v["ms_avg"]) for k, v in d["kernels"].items() if v.get("ms_avg", 0) > 1))
