#!/usr/bin/env python3
"""Cost model of the plane-product kernels, from two committed measurements:

  profiles/*_mfma16_mix.txt       tools/mfma16_mix.hip: what v_mfma_f32_32x32x16_f16 sustains on this part next to V vector-ALU
                                  instructions and B bytes of HBM traffic per MFMA (a synthetic loop, all CUs, socket power limit)
  profiles/*_pmc_per_kernel / *_pmc_traffic.json + *_bench.json    MFMA / VALU instructions and HBM bytes per launch, ms per launch
                                  (SQ_INSTS_VALU includes the MFMAs: vector-ALU instructions proper = SQ_INSTS_VALU - SQ_INSTS_MFMA)

A launch costs  MFMA + C_VALU x VALU + C_BYTE x bytes  "MFMA equivalents", executed at the rate the bare MFMA loop sustains:

  t = (mfma + C_VALU * valu + C_BYTE * bytes) * 32768 flop / RATE

RATE, C_VALU, C_BYTE are fitted to the microbenchmark's lines only (not to the kernels); the table shows how close each kernel is to
what the synthetic loop with ITS mix sustains.  usage: python tools/cost_model.py [--tag r03_v1] [--mix profiles/r03_mfma16_mix.txt]
"""
import argparse
import glob
import json
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
NAMES = {"conv_fwd1_planes": "ConvFwd1", "conv_fwd2_planes": "ConvFwd2", "conv_fwd3_planes": "ConvFwd3", "fc_fwd_planes": "FcFwd",
         "fc_dgrad_planes": "FcDgrad", "conv_dgrad3_planes": "ConvDgrad3", "conv_dgrad2_both": "ConvDgrad2",
         "conv_wgrad1_planes": "ConvWgrad1", "fc_wgrad_planes": "FcWgrad", "conv_wgrad3_planes": "ConvWgrad3",
         "conv_wgrad2_planes": "ConvWgrad2"}
FLOP_PER_MFMA = 2 * 32 * 32 * 16


def fit(mix_path):
    """(rate TFLOP/s, cost of one vector-ALU instruction, cost of one HBM byte) in MFMA equivalents, from the second pass."""
    text = open(mix_path).read().split("pass 1")[-1]
    rows = {}
    for line in text.splitlines():
        m = re.match(r"\s+(.*?)\s+([\d.]+) TFLOP/s\s+([\d.]+) TB/s", line)
        if m:
            rows[m.group(1)] = float(m.group(2))
    base = rows["MFMA + LDS operands only"]
    # the library's mix of vector-ALU instructions is mostly single-rate (moves, adds, max, conversions) with a fifth of
    # packed / mixed-precision ones: 0.8 x the mean of the single-rate lines + 0.2 x the mean of the packed lines
    single = [base / rows["+ 4 %s per MFMA" % op] - 1.0 for op in ("v_mov_b32", "v_add_u32", "v_perm_b32", "v_cvt_f32_ubyte1", "v_max_f32", "v_fma_f32", "v_cvt_pk_f16_f32")]
    packed = [base / rows["+ 4 %s per MFMA" % op] - 1.0 for op in ("v_fma_mixlo_f16", "v_pk_mul_f32", "v_pk_fma_f32")]
    c_valu = (0.8 * sum(single) / len(single) + 0.2 * sum(packed) / len(packed)) / 4.0
    c_byte = (base / rows["+ 128 B read per MFMA"] - 1.0 + base / rows["+ 64 B read + 64 B written per MFMA"] - 1.0) / 2.0 / 128.0
    return base, c_valu, c_byte, rows


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--tag", default=None)
    ap.add_argument("--mix", default=None)
    a = ap.parse_args()
    mix = a.mix or sorted(glob.glob(os.path.join(ROOT, "profiles", "*_mfma16_mix.txt")))[-1]
    tag = a.tag or os.path.basename(sorted(glob.glob(os.path.join(ROOT, "profiles", "*_pmc_traffic.json")))[-1]).replace("_pmc_traffic.json", "")
    rate, c_valu, c_byte, _ = fit(mix)
    pmc = json.load(open(os.path.join(ROOT, "profiles", tag + "_pmc_traffic.json")))["kernels"]
    bench = json.load(open(os.path.join(ROOT, "profiles", tag + "_bench.json")))["kernels"]
    print("from %s: bare MFMA loop %.0f TFLOP/s; one vector-ALU instruction = %.3f MFMA, one HBM byte = %.5f MFMA (128 B = %.2f)" % (
        os.path.basename(mix), rate, c_valu, c_byte, 128 * c_byte))
    print("%-11s %8s %10s %8s | %7s %7s %7s | %8s %9s %6s" % ("kernel", "MFMA M", "VALU/MFMA", "B/MFMA", "mfma ms", "valu ms", "hbm ms", "model ms", "measured", "ratio"))
    tot = [0.0] * 5
    for k, n in NAMES.items():
        if k == "conv_fwd1_planes" and "conv_fwd1_resident" in pmc:
            k = "conv_fwd1_resident"  # training launches of round 3 and later (csrc/conv2.hip)
        v = pmc[k]
        mf = v["mfma_insts"]
        va = v["valu_insts"] - mf  # SQ_INSTS_VALU counts the matrix instructions too (checked on tools/mfma16_mix.hip: MFMA-only loop 2.66e8 against 2.62e8 MFMAs)
        by = v["write_bytes"] + 2.0 * v["fetch_bytes"]  # FETCH_SIZE counts 16-byte-per-lane streams at half their bytes on gfx950
        ms = lambda eq: eq * FLOP_PER_MFMA / (rate * 1e12) * 1e3
        parts = (ms(mf), ms(c_valu * va), ms(c_byte * by))
        model, meas = sum(parts), bench[n]["ms_avg"]
        for i, x in enumerate(parts + (model, meas)):
            tot[i] += x
        print("%-11s %8.1f %10.1f %8.0f | %7.2f %7.2f %7.2f | %8.2f %9.2f %6.2f" % (n, mf / 1e6, va / mf, by / mf, *parts, model, meas, meas / model))
    print("%-11s %8s %10s %8s | %7.2f %7.2f %7.2f | %8.2f %9.2f %6.2f" % ("sum", "", "", "", *tot, tot[4] / tot[3]))


if __name__ == "__main__":
    main()
