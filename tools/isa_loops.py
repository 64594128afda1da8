#!/usr/bin/env python3
"""Instruction mix of the loops of a kernel in a hipcc -S listing (device assembly): for every backward branch, the
instruction classes between its target label and the branch.  Shows what a k loop issues per matrix instruction.
usage: python tools/isa_loops.py file.s kernel_name_substring
Produce file.s with the Makefile's flags, INCLUDING  -Xclang -target-feature -Xclang -load-store-opt  (hipcc -S --cuda-device-only):
the "not a recognized feature" warning of a normal build comes from the HOST pass; the device pass honours it (no ds_read2 merging)."""
import re
import sys
from collections import Counter


def classify(op):
    if op.startswith("v_mfma") or op.startswith("v_smfmac"):
        return "mfma"
    if op.startswith("ds_"):
        return "lds"
    if op.startswith(("global_", "buffer_", "flat_", "scratch_")):
        return "vmem"
    if op.startswith("v_"):
        return "valu"
    if op.startswith("s_waitcnt"):
        return "wait"
    if op.startswith("s_barrier"):
        return "barrier"
    if op.startswith(("s_load", "s_buffer_load")):
        return "smem"
    if op.startswith("s_"):
        return "salu"
    return "other"


def main():
    path, want = sys.argv[1], sys.argv[2]
    lines = open(path).read().split("\n")
    start = None
    for i, l in enumerate(lines):
        if re.match(r"^[A-Za-z_]\w*:", l) and want in l.split(":")[0]:
            start = i
            break
    assert start is not None, "kernel not found"
    body = []
    for l in lines[start + 1:]:
        body.append(l)
        if l.strip().startswith("s_endpgm"):
            break
    labels = {}
    insts = []
    for l in body:
        t = l.strip()
        if not t or t.startswith(";") or t.startswith("."):
            m = re.match(r"(\.LBB\w+):", t)
            if m:
                labels[m.group(1)] = len(insts)
            continue
        m = re.match(r"(\.LBB\w+):", t)
        if m:
            labels[m.group(1)] = len(insts)
            continue
        insts.append(t.split(";")[0].strip())
    total = Counter(classify(x.split()[0]) for x in insts)
    print(lines[start][:100])
    print("whole kernel:", dict(total))
    loops = []
    for i, x in enumerate(insts):
        p = x.split()
        if p[0].startswith("s_cbranch") or p[0] == "s_branch":
            tgt = p[-1]
            if tgt in labels and labels[tgt] <= i:
                loops.append((labels[tgt], i))
    for a, b in sorted(loops, key=lambda ab: ab[1] - ab[0]):
        c = Counter(classify(x.split()[0]) for x in insts[a:b + 1])
        if c["mfma"] == 0 and b - a < 20:
            continue
        vops = Counter(x.split()[0] for x in insts[a:b + 1] if classify(x.split()[0]) == "valu")
        print("loop [%d..%d] %d insts: %s" % (a, b, b - a + 1, dict(c)))
        print("    valu/mfma %.2f  top valu: %s" % (c["valu"] / max(1, c["mfma"]), vops.most_common(12)))
        lops = Counter(x.split()[0] for x in insts[a:b + 1] if classify(x.split()[0]) in ("lds", "vmem"))
        print("    mem: %s" % (lops.most_common(8),))


if __name__ == "__main__":
    main()
