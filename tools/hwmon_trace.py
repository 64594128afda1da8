#!/usr/bin/env python3
"""Socket power / shader clock / junction temperature of EVERY GPU hwmon node of the host, sampled every ~25 ms until STOPFILE appears
(tools/prof_round.sh runs it beside the bench line).  A GPU box shows the hwmon nodes of all eight GPUs of its host but runs the bench
on one, and the neighbours may be busy with other tenants' work: every node is recorded with its PCI address, `--summary` then picks
the node of the GPU the bench ran on (PCI address from `rocm-smi --showbus`, which only sees that one) -- never "the node that draws
the most" (round 6: a neighbour sat at the cap for the whole run).
    python3 tools/hwmon_trace.py OUT.txt STOPFILE            # sampler (reads sysfs only: never touches the GPU runtime)
    python3 tools/hwmon_trace.py --summary OUT.txt PCI_ADDR  # -> JSON: cap, per-phase power / clock of that node"""
import glob
import json
import os
import sys
import time


def nodes():
    out = []
    for h in sorted(glob.glob("/sys/class/drm/card*/device/hwmon/hwmon*")):
        pci = os.path.basename(os.path.realpath(os.path.join(h, "..", "..")))
        out.append((h, pci))
    return out


def rd(path):
    try:
        return open(path).read().strip()
    except OSError:
        return "0"


def sample(out_path, stop_path, limit_s=900.0):
    ns = nodes()
    with open(out_path, "w") as f:
        f.write("# nodes: %s\n" % " ".join("%s=%s" % (os.path.basename(h), p) for h, p in ns))
        f.write("# power1_cap_uW: %s\n" % " ".join(rd(os.path.join(h, "power1_cap")) for h, _ in ns))
        f.write("# columns: t_ms then per node power_uW,sclk_Hz,tj_mC\n")
        t0 = time.time()
        while not os.path.exists(stop_path) and time.time() - t0 < limit_s:
            row = ["%d" % (time.time() * 1000)]
            for h, _ in ns:
                row.append("%s,%s,%s" % (rd(os.path.join(h, "power1_input")), rd(os.path.join(h, "freq1_input")), rd(os.path.join(h, "temp2_input"))))
            f.write(" ".join(row) + "\n")
            time.sleep(0.02)


def summary(path, pci):
    lines = open(path).read().splitlines()
    names = dict(kv.split("=") for kv in lines[0].split(":", 1)[1].split())
    order = list(names)
    caps = lines[1].split(":", 1)[1].split()
    pci = pci.lower()
    col = next((i for i, n in enumerate(order) if names[n].lower() == pci or names[n].lower().endswith(pci)), None)
    if col is None:
        return {"error": "no hwmon node with PCI address %s among %s" % (pci, names)}
    ts, pw, ck = [], [], []
    for ln in lines[3:]:
        parts = ln.split()
        if len(parts) <= col + 1:
            continue
        p, c, _ = parts[col + 1].split(",")
        ts.append(int(parts[0]))
        pw.append(int(p) / 1e6)
        ck.append(int(c) / 1e9)
    busy = [i for i, p in enumerate(pw) if p > 0.7 * max(pw)]
    med = lambda v: sorted(v)[len(v) // 2] if v else None
    others = {}
    for i, n in enumerate(order):
        if i != col:
            vals = [int(ln.split()[i + 1].split(",")[0]) / 1e6 for ln in lines[3:] if len(ln.split()) > i + 1]
            others[names[n]] = round(med(vals), 0) if vals else None
    cap = int(caps[col]) / 1e6
    at_cap = [i for i, p in enumerate(pw) if p >= 0.96 * cap]
    best, cur = (0, -1), None                      # longest contiguous run at the cap
    for i, p in enumerate(pw):
        if p >= 0.96 * cap:
            cur = (cur[0], i) if cur else (i, i)
            if cur[1] - cur[0] > best[1] - best[0]:
                best = cur
        else:
            cur = None
    run = list(range(best[0], best[1] + 1))
    mean = lambda v: round(sum(v) / len(v), 3) if v else None
    cap_stats = {"threshold_W": round(0.96 * cap, 0), "samples": len(at_cap), "seconds": round(len(at_cap) * ((ts[-1] - ts[0]) / max(1, len(ts) - 1)) / 1000.0, 2),
                 "power_W_mean": mean([pw[i] for i in at_cap]), "sclk_GHz_mean": mean([ck[i] for i in at_cap]),
                 "sclk_GHz_min": min([ck[i] for i in at_cap]) if at_cap else None,
                 "longest_run": {"seconds": round((ts[best[1]] - ts[best[0]]) / 1000.0, 2) if run else 0, "power_W_mean": mean([pw[i] for i in run]),
                                 "sclk_GHz_mean": mean([ck[i] for i in run])}}
    return {"node": order[col], "pci": names[order[col]], "power_cap_W": int(caps[col]) / 1e6, "samples": len(pw), "at_cap": cap_stats,
            "seconds": round((ts[-1] - ts[0]) / 1000.0, 1) if ts else 0,
            "idle_W_min": min(pw) if pw else None, "max_W": max(pw) if pw else None,
            "under_load": {"samples": len(busy), "power_W_median": med([pw[i] for i in busy]), "sclk_GHz_median": med([ck[i] for i in busy]),
                           "sclk_GHz_min": min([ck[i] for i in busy]) if busy else None},
            "idle_sclk_GHz_max": max(ck) if ck else None,
            "neighbours_median_W": others,
            "note": "under_load = samples above 70 % of the node's maximum power; the bench's GPU phases (acting + PPO iterations)"}


def own_trace(path, pci):
    """The bench GPU's own column of an all-nodes trace: `t_ms power_W sclk_GHz tj_C` per sample (what gets committed under profiles/)."""
    lines = open(path).read().splitlines()
    names = dict(kv.split("=") for kv in lines[0].split(":", 1)[1].split())
    order = list(names)
    col = next(i for i, n in enumerate(order) if names[n].lower() == pci.lower())
    out = ["# %s (%s) of %s; power1_cap %s uW; columns: t_ms power_W sclk_GHz tj_C" % (order[col], names[order[col]], os.path.basename(path), lines[1].split(":", 1)[1].split()[col])]
    t0 = None
    for ln in lines[3:]:
        parts = ln.split()
        if len(parts) <= col + 1:
            continue
        p, c, tj = parts[col + 1].split(",")
        t0 = int(parts[0]) if t0 is None else t0
        out.append("%d %.0f %.3f %.0f" % (int(parts[0]) - t0, int(p) / 1e6, int(c) / 1e9, int(tj) / 1e3))
    return "\n".join(out) + "\n"


if __name__ == "__main__":
    if sys.argv[1] == "--summary":
        print(json.dumps(summary(sys.argv[2], sys.argv[3]), indent=1))
    elif sys.argv[1] == "--own":
        sys.stdout.write(own_trace(sys.argv[2], sys.argv[3]))
    else:
        sample(sys.argv[1], sys.argv[2])
