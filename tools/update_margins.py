#!/usr/bin/env python3
"""tests/golden/margins.json: a RECORD of the ratios a GPU run measured, next to the fixed limits they were asserted against.

`python -m pytest tests -m gpu` (on the MI355X box) logs every envelope-type ratio it measured to
gpurun_out/margins_measured.json (the asserts stay on).  Until round 3 this script turned those into the committed LIMITS
(measured x 1.5 under a cap of 4): a regression of up to 50 % passed silently.  The limits are now fixed
in tests/parity_util.py (PARAM_LIMIT 1.5, LOSS_LIMIT 2.0, VS_TORCH_LIMIT 1.25, ACCURACY_CAP 8 rounding units); this script only
copies what was measured, with the limit of its kind beside it, so that the headroom of every bound is on record.

usage: python tools/update_margins.py [--source gpurun_out/margins_measured.json] [--note "r04, MI355X"]
"""
import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--source", default=os.path.join(ROOT, "gpurun_out", "margins_measured.json"))
    ap.add_argument("--note", default="")
    args = ap.parse_args()
    import parity_util as P
    measured = json.load(open(args.source))
    out = {"_doc": "RECORD of measured ratios (deviation of the HIP path / the reference's own deviation, tests/parity_util.py); "
                   "the limits are fixed in tests/parity_util.py and are NOT read from this file"}
    over = []
    for test, d in sorted(measured.items()):
        slot = out.setdefault(test, {})
        for key, v in sorted(d.items()):
            if key.endswith("__vs_onednn_only"):   # recorded, never asserted: the same deviation against the oneDNN-only spread
                slot[key] = {"measured": round(float(v), 4), "limit": None,
                             "note": (args.note + "; " if args.note else "") + "record only: ratio against the reference's oneDNN runs alone"}
                continue
            limit = P.MARGINS.limit(test, key)
            slot[key] = {"measured": round(float(v), 4), "limit": limit, "note": args.note}
            if v > limit:
                over.append((test, key, v, limit))
    path = os.path.join(ROOT, "tests", "golden", "margins.json")
    json.dump(out, open(path, "w"), indent=1, sort_keys=True)
    print("wrote", path)
    for test, key, v, limit in over:
        print("ABOVE ITS LIMIT %s / %s: measured %.4g > %.4g" % (test, key, v, limit))


if __name__ == "__main__":
    main()
