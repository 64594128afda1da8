#!/usr/bin/env python3
"""Turn the rocprofv3 --pmc pass CSVs written by tools/prof_pmc.sh into the two files kept under
profiles/:

  <tag>_pmc_per_kernel.csv   per-kernel averages of every counter + average duration (us)
  <tag>_pmc_traffic.json     per-kernel HBM bytes per launch, MFMA-busy fraction, held clock;
                             bench.py copies the dominant kernel's entry into roofline.traffic

Usage: python tools/pmc_to_profiles.py gpurun_out/pmc profiles/r01_v4

Counter conventions (MI355X_MICROARCH.md, HBM / rocprofv3 section): FETCH_SIZE and WRITE_SIZE are
in KB (x1024 here) and come from separate passes; FETCH_SIZE counts 16-byte-per-lane streams at
half their bytes on gfx950 (noted, not "corrected" per kernel because the kernels mix access
widths); GRBM_GUI_ACTIVE is summed over the 8 XCDs; SQ_VALU_MFMA_BUSY_CYCLES is summed over SIMDs.
"""
import csv
import glob
import json
import os
import re
import sys
from collections import defaultdict

SIMDS = 1024  # 256 CUs x 4


def short(k):
    # templates over a compile-time geometry (nav layers): keep the geometry in the name
    m = re.search(r"ddrl::pconv::(\w+?)_kernel<ddrl::pconv::\w+<([\d, ]+)>", k)
    if m:
        g = [x.strip() for x in m.group(2).split(",")]
        tail = k[m.end():m.end() + 16]
        flag = ".pool" if tail.startswith(", true, false") else (".unpool" if (tail.startswith(", false, true") or tail.startswith(", true>")) else "")
        return "pconv_%s<%s>%s" % (m.group(1), "x".join(g[:5]), flag)      # CIN x COUT x KS x HIN x PAD
    m = re.search(r"ddrl::fconv::(\w+?)_kernel<ddrl::fconv::\w+<([\d, ]+)>, (\w+)>", k)
    if m:
        flag = "" if m.group(3) != "true" else (".unpool" if "wgrad" in m.group(1) else ".pool")
        return "fconv_%s<%s>%s" % (m.group(1), "x".join(x.strip() for x in m.group(2).split(",")), flag)
    m = re.search(r"ddrl::c1d::(\w+?)_kernel(?:<([\d, ]+)>)?", k)
    if m:
        return "c1d_" + m.group(1) + ("" if m.group(2) is None else "<%s>" % "x".join(x.strip() for x in m.group(2).split(",")))
    m = re.search(r"ddrl::plin::(\w+?)_kernel(?:<(\d)>)?", k)
    if m:
        return "plin_" + m.group(1) + ("" if m.group(2) is None else {"0": ".fwd", "1": ".dgrad"}[m.group(2)])
    m = re.search(r"engine2_kernel<ddrl::dconv::(\w+)<([\d, ]+)>", k)
    if m:
        g = [x.strip() for x in m.group(2).split(",")]
        return "dconv_%s<%s>" % (m.group(1), "x".join(g[:5]))
    m = re.search(r"engine2_kernel<ddrl::(gconv|glin)::(\w+)(?:<(\w+)>)?", k)
    if m:
        return "%s_%s%s" % (m.group(1), m.group(2), "" if m.group(3) is None else "." + m.group(3))
    m = re.search(r"engine2_kernel<ddrl::(\w+?)(?:v2|2)?(?:<(\d)>)?\s*>", k)
    if m:
        return m.group(1) + ("" if not m.group(2) or m.group(2) == "2" else ".ne" + m.group(2))
    m = re.search(r"ddrl::(\w+?)(?:_kernel)?(?:<[^>]*>)?\(", k)
    return m.group(1) if m else None


def main():
    src = sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/pmc"
    tag = sys.argv[2] if len(sys.argv) > 2 else "profiles/pmc"
    agg = defaultdict(lambda: defaultdict(lambda: [0.0, 0]))
    dur = defaultdict(lambda: [0.0, 0])
    seen = set()
    for f in sorted(glob.glob(os.path.join(src, "*counter_collection.csv"))):
        for row in csv.DictReader(open(f)):
            name = short(row["Kernel_Name"])
            if name is None:
                continue
            # acting-size launches of the forward kernels (small grids) are kept apart from the
            # training-size ones; every other kernel only runs in training (or is size-independent)
            big = ("fwd" not in name.lower() and name != "heads_act") or name == "conv_fwd1_resident" or name.startswith(("fconv_", "plin_", "c1d_")) or int(row.get("Grid_Size", 0) or 0) >= 256 * 2000  # (the resident conv1 kernel: one workgroup per CU, training launches only)
            key = name + ("" if big else ":acting")
            c = agg[key][row["Counter_Name"]]
            c[0] += float(row["Counter_Value"])
            c[1] += 1
            did = (f, row["Dispatch_Id"])
            if did not in seen:
                seen.add(did)
                dur[key][0] += (int(row["End_Timestamp"]) - int(row["Start_Timestamp"])) / 1000.0
                dur[key][1] += 1
    counters = sorted({c for k in agg for c in agg[k]})
    with open(tag + "_pmc_per_kernel.csv", "w") as out:
        out.write("kernel,avg_us," + ",".join(counters) + "\n")
        for k in sorted(agg):
            vals = ["%.6g" % (agg[k][c][0] / max(1, agg[k][c][1])) if c in agg[k] else "" for c in counters]
            out.write("%s,%.1f,%s\n" % (k, dur[k][0] / max(1, dur[k][1]), ",".join(vals)))
    traffic = {"kernels": {}, "source": "tools/prof_pmc.sh -> tools/pmc_to_profiles.py; one PMC pass per counter group"}
    for k in sorted(agg):
        a = {c: agg[k][c][0] / max(1, agg[k][c][1]) for c in agg[k]}
        if "FETCH_SIZE" not in a or "WRITE_SIZE" not in a:
            continue
        us = dur[k][0] / max(1, dur[k][1])
        e = {"avg_us": round(us, 1), "fetch_bytes": a["FETCH_SIZE"] * 1024.0, "write_bytes": a["WRITE_SIZE"] * 1024.0,
             "hbm_bytes": (a["FETCH_SIZE"] + a["WRITE_SIZE"]) * 1024.0,
             # the guide's gfx950 correction: FETCH_SIZE counts a wide (16 B / lane) streaming read at HALF its bytes; these
             # kernels stream with 16-byte loads / global_load_lds_dwordx4, so the corrected figure doubles the fetch side (an
             # upper estimate where part of the reads are narrower); WRITE_SIZE is exact for 16-byte streaming stores
             "hbm_bytes_corrected": (2.0 * a["FETCH_SIZE"] + a["WRITE_SIZE"]) * 1024.0,
             "fetch_note": "FETCH_SIZE as reported; 16-B-per-lane streams are counted at half their bytes on gfx950 "
                           "(MI355X_MICROARCH.md)"}
        if "GRBM_GUI_ACTIVE" in a and us > 0:
            cyc = a["GRBM_GUI_ACTIVE"] / 8.0  # summed over 8 XCDs
            e["clock_ghz"] = round(cyc / (us * 1000.0), 3)
            if "SQ_VALU_MFMA_BUSY_CYCLES" in a and cyc > 0:
                # the counter ticks once per busy cycle per SIMD -> busy fraction = sum / (SIMDs x cycles)
                e["mfma_busy_frac"] = round(a["SQ_VALU_MFMA_BUSY_CYCLES"] / (SIMDS * cyc), 4)
        # exact wave-instruction counts (SQ_INSTS_MFMA of ConvFwd1 / FcFwd equals their algorithmic count to four digits):
        # bench.py derives executed / algorithmic matrix work and VALU per MFMA from them
        if "SQ_INSTS_MFMA" in a:
            e["mfma_insts"] = a["SQ_INSTS_MFMA"]
        if "SQ_INSTS_VALU" in a:
            e["valu_insts"] = a["SQ_INSTS_VALU"]
        traffic["kernels"][k] = e
    # batch of the profiled launches and the profiled build: meta.json written by tools/prof_round.sh / prof_nav.sh next to the pmc
    # directory (or inside it), else the environment; a summary without them is refused (round 4 committed one with batch 65,536 for a
    # 4,096-sample run and an empty build)
    meta = {}
    for cand in (os.path.join(os.path.dirname(os.path.abspath(sys.argv[1])), "meta.json"), os.path.join(sys.argv[1], "meta.json")):
        if os.path.exists(cand):
            meta = json.load(open(cand))
            break
    batch = os.environ.get("DDRL_PROFILE_BATCH") or meta.get("batch")
    build = os.environ.get("DDRL_PROFILE_BUILD") or meta.get("build")
    if not batch or not build or str(build).startswith("unknown"):
        sys.exit("pmc_to_profiles.py: no batch / build for this profile (meta.json or DDRL_PROFILE_BATCH / DDRL_PROFILE_BUILD)")
    traffic["batch"] = int(batch)
    traffic["build"] = str(build)
    box = os.environ.get("DDRL_PROFILE_BOX", "")                            # host the passes ran on
    if not box:  # tools/prof_round.sh leaves hostname + product name next to the pmc directory
        for cand in (os.path.join(os.path.dirname(os.path.abspath(sys.argv[1])), "box.txt"), os.path.join(sys.argv[1], "box.txt")):
            if os.path.exists(cand):
                lines = [ln.strip() for ln in open(cand) if ln.strip() and not set(ln.strip()) <= set("=-")]
                box = "; ".join(lines[:1] + [" ".join(ln.split()) for ln in lines[1:] if "Card Model" in ln or "Card SKU" in ln or "Unique ID" in ln][:3])[:200]
                break
    traffic["box"] = box
    # what the summary is evidence OF: the SHA-1 of every kernel source file as it stands when the summary is written (the tree the
    # profiled library was built from).  bench.py compares them with the files it runs on and marks a kernel's traffic stale when its
    # source has changed since (there is no .git on a GPU box, so the hashes travel inside the summary).
    import hashlib
    # (DDRL_PROFILE_SRC: the csrc directory of the profiled build when the working tree has moved on since the passes ran)
    csrc = os.environ.get("DDRL_PROFILE_SRC") or os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "ddrl4nav_amd", "csrc")
    traffic["sources"] = {os.path.basename(f): hashlib.sha1(open(f, "rb").read()).hexdigest()
                          for f in sorted(glob.glob(os.path.join(csrc, "*.hip")) + glob.glob(os.path.join(csrc, "*.h")))}
    with open(tag + "_pmc_traffic.json", "w") as out:
        json.dump(traffic, out, indent=1, sort_keys=True)
    print("wrote", tag + "_pmc_per_kernel.csv", tag + "_pmc_traffic.json", "(%d kernels)" % len(agg))


if __name__ == "__main__":
    main()
