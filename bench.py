#!/usr/bin/env python3
"""Benchmark of the DDRL4NAV actor-learner hot path on MI355X.

One "step" = one full pass of the path over one rollout of synthetic Pong-shaped input, per GPU:
  T=256 acting steps (forward both encoders + heads + sample) over N=256 envs,
  one bootstrap forward, the GAE scan, and one PPO update = TRAINING_ITER_TIME=10 full-batch
  iterations (forward + loss + backward + [all-reduce] + grad-norm clip + 2x Adam) on
  B = N*T = 65,536 samples.  Frames are already resident in HBM (uint8 [T+1,N,4,84,84]).
The loop is driven through the product surface: runner.create_net -> nn.PPO,
agent.DeviceRollout (device-resident experience pool) and net.learn() (the reference's generator
protocol, one host sync per iteration for the loss dict).

metric: env-steps/s for the whole job = n_gpus * N * T * steps / wall time (max over ranks).
Prints ONE JSON line on rank 0 (see the contract in the task statement).
"""
import argparse
import json
import os
import sys
import time
import types

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

# algorithmic work per sample and encoder (restated in DESIGN.md), in MACs
MAC = {"ConvFwd1": 32 * 400 * 256, "ConvFwd2": 64 * 81 * 512, "ConvFwd3": 64 * 49 * 576, "FcFwd": 512 * 3136,
       "ConvWgrad1": 32 * 400 * 256, "ConvWgrad2": 64 * 81 * 512, "ConvWgrad3": 64 * 49 * 576, "FcWgrad": 512 * 3136,
       "ConvDgrad2": 64 * 81 * 512, "ConvDgrad3": 64 * 49 * 576, "FcDgrad": 512 * 3136}
FLOP_ACT_PER_STEP = 37_379_072          # per env-step, both encoders + heads
FLOP_TRAIN_PER_SAMPLE = 99_030_016      # per sample per PPO iteration
PEAK_F32_MFMA_TFLOPS = 157.3            # MI355X_MICROARCH.md: dense f32-input MFMA = fp32 vector peak
PEAK_BF16_MFMA_TFLOPS = 2500.0          # MI355X_MICROARCH.md: dense bf16 MFMA (no sparsity)
# Matrix pipe and plane products of every GEMM kernel of a TRAINING launch (csrc/*.hip): "f16x3" = both fp32 operands as two
# scaled fp16 planes each, three products (csrc/engine2.h "plane scheme"); "f16x2" = one operand exact in fp16 (the uint8
# pixels), the other as two scaled fp16 planes, two products.  The ceiling a kernel is priced against is the dense 16-bit MFMA
# peak (2.5 PFLOP/s, bf16 and fp16 alike) / products, in fp32-equivalent (algorithmic) TFLOP/s.
PIPE = {"ConvFwd1": ("f16x2", 2), "ConvWgrad1": ("f16x2", 2), "ConvFwd2": ("f16x3", 3), "ConvFwd3": ("f16x3", 3),
        "FcFwd": ("f16x3", 3), "FcDgrad": ("f16x3", 3), "ConvDgrad3": ("f16x3", 3), "ConvDgrad2": ("f16x3", 3),
        "FcWgrad": ("f16x3", 3), "ConvWgrad3": ("f16x3", 3), "ConvWgrad2": ("f16x3", 3)}
# executed / algorithmic MFMA work of the kernels that walk padded operands: the DESIGN figures, used only
# when the newest committed PMC profile carries no SQ_INSTS_MFMA for a kernel (executed_over_algorithmic() below)
EXECUTED_OVER_ALGORITHMIC = {"ConvDgrad3": 81.0 / 49.0, "ConvDgrad2": 1.23, "ConvFwd3": 1.11, "ConvWgrad3": 112.0 / 98.0,
                             "ConvWgrad2": 96.0 / 81.0, "ConvWgrad1": 48.0 / 40.0}
PEAK_HBM_GBPS = 8000.0                  # MI355X_MICROARCH.md: HBM3E ~8 TB/s
# algorithmic HBM bytes per training launch of the GEMM kernels, both encoders (what a launch must read and
# write once -- fp32 activations / gradients, u8 frames; weights and masks are noise), f(B samples)
ALG_BYTES = {
    "ConvFwd1": lambda B: B * (28224 + 2 * 51200), "ConvFwd2": lambda B: B * 2 * (51200 + 20736), "ConvFwd3": lambda B: B * 2 * (20736 + 12544),
    "FcFwd": lambda B: B * 2 * (12544 + 2048), "FcDgrad": lambda B: B * 2 * (2048 + 12544), "FcWgrad": lambda B: B * 2 * (2048 + 12544),
    "ConvDgrad3": lambda B: B * 2 * (12544 + 20736), "ConvDgrad2": lambda B: B * 2 * (20736 + 51200),
    "ConvWgrad3": lambda B: B * 2 * (12544 + 20736), "ConvWgrad2": lambda B: B * 2 * (20736 + 51200), "ConvWgrad1": lambda B: B * (28224 + 2 * 51200),
}
# algorithmic HBM bytes per launch of the HBM-bound kernels, f(N envs, B samples, P params)
HBM_BYTES = {
    "heads_loss": lambda N, B, P: B * (2 * 2 * 512 * 4 + 72),       # read h, write dh (both heads) + loss operands
    "heads_act": lambda N, B, P: N * (2 * 512 * 4 + 36),            # read h of both encoders, write probs/value/action/logp
    "clip_adam": lambda N, B, P: P * 32,                            # norm read + p,g,m,v read + p,m,v write
    "pack_weights": lambda N, B, P: P * 4 * 3,                      # read params, write two derived layouts (approx.)
}


def _host_cpu():
    """(model string, logical CPUs, physical cores) of this host from /proc/cpuinfo."""
    model, cores = "unknown", set()
    try:
        phys = core = None
        for ln in open("/proc/cpuinfo"):
            if ln.startswith("model name") and model == "unknown":
                model = ln.split(":", 1)[1].strip()
            elif ln.startswith("physical id"):
                phys = ln.split(":", 1)[1].strip()
            elif ln.startswith("core id"):
                core = ln.split(":", 1)[1].strip()
            elif not ln.strip():
                if phys is not None and core is not None:
                    cores.add((phys, core))
                phys = core = None
    except OSError:
        pass
    logical = os.cpu_count() or 1
    return model, logical, (len(cores) or logical)


def _timed(fn, warm, reps, budget_s):
    """median / p95 (ms) of fn() over up to `reps` repetitions, at least 2, stopping early when `budget_s` is spent."""
    for _ in range(warm):
        fn()
    ts, t_start = [], time.perf_counter()
    while len(ts) < reps and (len(ts) < 2 or time.perf_counter() - t_start < budget_s):
        t0 = time.perf_counter()
        fn()
        ts.append((time.perf_counter() - t0) * 1e3)
    return {"median_ms": round(float(np.median(ts)), 4), "p95_ms": round(float(np.percentile(ts, 95)), 4), "reps": len(ts)}


def cpu_baseline():
    """Reference-equivalent CPU path (the oracle: torch-CPU restatement pinned to the reference by tests/golden) timed on
    this host as BASELINE.md section 3 lays out: PPO iterations at B = 1024 and 2048, forward at n = 4 and 256, GAE at
    N = 8 and 256 (T = 256), with k = all physical cores and k = 1, median + p95 of >= 10 repetitions where the bounded
    sample allows (every figure states its repetition count; about a minute of CPU work in total)."""
    from ddrl4nav_amd.utils.recipe import make_weights
    from oracle import ddrl_oracle as O
    model, logical, physical = _host_cpu()
    rng = np.random.default_rng(1234)
    Bmax = 2048
    x = O.frames_to_f32(rng.integers(0, 256, size=(Bmax, 4, 84, 84), dtype=np.uint8))
    acts = torch.from_numpy(rng.integers(0, 6, size=Bmax).astype(np.float32))
    old = torch.full((Bmax,), -1.79, dtype=torch.float32)
    adv = torch.from_numpy(rng.normal(size=Bmax).astype(np.float32))
    ret = torch.from_numpy(rng.normal(size=Bmax).astype(np.float32))
    T = 256
    gae_in = {}
    for N in (8, 256):
        u = rng.random((T, N))
        gae_in[N] = (rng.normal(size=(T + 1, N)).astype(np.float32),
                     np.where(u < 0.01, -1.0, np.where(u > 0.99, 1.0, 0.0)).astype(np.float32),
                     (rng.random((T, N)) < 1.0 / 800).astype(np.uint8))

    def leg(threads, full):
        torch.set_num_threads(threads)
        net = O.OraclePPO()
        net.load_weights(make_weights(0))
        out = {"threads": threads}
        with torch.no_grad():
            for n, reps in ((4, 20), (256, 10 if full else 3)):
                r = _timed(lambda: net(x[:n]), 2 if full else 1, reps, 6.0)
                r["samples_per_s"] = round(n / (r["median_ms"] * 1e-3), 1)
                out["forward_n%d" % n] = r
        for B, reps in ((1024, 10 if full else 2), (2048, 5 if full else 0)):
            if reps == 0:
                continue
            gen = O.learn(net, net.make_optims(), x[:B], acts[:B], old[:B], adv[:B], ret[:B], iters=10 ** 6)
            r = _timed(lambda: next(gen), 1, reps, 12.0 if full else 16.0)
            r["update_ms_10_iters"] = round(10 * r["median_ms"], 2)
            r["env_steps_per_s_equiv"] = round(B / (10 * r["median_ms"] * 1e-3), 1)   # B / update time
            out["ppo_iter_B%d" % B] = r
        if full:
            for N in (8, 256):
                out["gae_T256_N%d" % N] = _timed(lambda: O.gae(*gae_in[N]), 1, 10, 4.0)
        return out

    k_all = physical                 # k = all physical host cores (and k = 1 for a per-core figure)
    full, one = leg(k_all, True), leg(1, False)

    def per_env_step(d):
        return d["forward_n256"]["median_ms"] * 1e-3 / 256 + 10 * d["ppo_iter_B1024"]["median_ms"] * 1e-3 / 1024

    # torch's CPU convolutions do not scale to a whole 128-core host at these batch sizes (measured: 128 threads are slower
    # than one): a short sweep finds the thread count at which the reference-equivalent code is FASTEST, and `value` uses it
    sweep = {}
    for k in sorted({k for k in (8, 16, 32, 64) if k < k_all}):
        torch.set_num_threads(k)
        net = O.OraclePPO()
        net.load_weights(make_weights(0))
        with torch.no_grad():
            f = _timed(lambda: net(x[:256]), 1, 3, 2.0)
        gen = O.learn(net, net.make_optims(), x[:1024], acts[:1024], old[:1024], adv[:1024], ret[:1024], iters=10 ** 6)
        sweep[k] = {"threads": k, "forward_n256": f, "ppo_iter_B1024": _timed(lambda: next(gen), 1, 3, 4.0)}
    cands = {**sweep, k_all: full, 1: one}
    best = min(cands, key=lambda k: per_env_step(cands[k]))
    # the headline of the CPU baseline is k = ALL PHYSICAL HOST CORES (k = 1 beside it); the fastest thread count of
    # the sweep is a sub-key (torch's CPU convolutions do not scale to a 128-core host at these batch sizes)
    out = {"value": round(1.0 / per_env_step(full), 2), "unit": "env-steps/s", "cores": k_all, "kind": "port",
           "sample": "oracle (torch-CPU fp32 restatement of PPO.forward / learn / GAE, pinned to the reference by tests/golden) on "
                     "synthetic inputs (rng 1234): value = 1 / (forward n=256 per sample + 10 x PPO iteration B=1024 per sample) at "
                     "k = all %d physical host cores (SURVEY.md section 8d; k = 1 in value_at_one_core); best_of_sweep = the fastest "
                     "thread count of {1, 8, 16, 32, 64, all} = %d; every figure is a median over `reps` repetitions" % (k_all, best),
           "cpu_model": model, "host_logical_cpus": logical, "host_physical_cores": physical,
           "best_of_sweep": {"value": round(1.0 / per_env_step(cands[best]), 2), "threads": best},
           "torch_threads_used": k_all, "value_at_all_physical_cores": round(1.0 / per_env_step(full), 2),
           "value_at_one_core": round(1.0 / per_env_step(one), 2), "k_all": full, "k_1": one,
           "k_sweep": {str(k): {"forward_n256_ms": v["forward_n256"]["median_ms"], "ppo_iter_B1024_ms": v["ppo_iter_B1024"]["median_ms"],
                                "env_steps_per_s": round(1.0 / per_env_step(v), 2)} for k, v in sweep.items()}}
    torch.set_num_threads(min(best, 16))
    out["same_gpu_torch"] = torch_rocm_baseline_child()
    return out


def torch_rocm_baseline_child(timeout_s=240):
    """Run torch_rocm_baseline in a child process: MIOpen / rocBLAS are third-party code paths this
    library never uses, and a fault in them must not be able to take the bench line down."""
    import subprocess
    try:
        r = subprocess.run([sys.executable, os.path.abspath(__file__), "--torch-leg"], capture_output=True, text=True,
                           timeout=timeout_s)
        lines = [ln for ln in r.stdout.strip().splitlines() if ln.startswith("{")]
        return json.loads(lines[-1]) if lines else {"error": "rc=%d %s" % (r.returncode, r.stderr.strip()[-160:])}
    except Exception as e:
        return {"error": repr(e)[:200]}


def torch_leg_main():
    from ddrl4nav_amd.utils.recipe import make_weights
    from oracle import ddrl_oracle as O
    rng = np.random.default_rng(1234)
    net = O.OraclePPO()
    net.load_weights(make_weights(0))
    B = 1024
    x = O.frames_to_f32(rng.integers(0, 256, size=(B, 4, 84, 84), dtype=np.uint8))
    acts = torch.from_numpy(rng.integers(0, 6, size=B).astype(np.float32))
    old = torch.full((B,), -1.79, dtype=torch.float32)
    adv = torch.from_numpy(rng.normal(size=B).astype(np.float32))
    ret = torch.from_numpy(rng.normal(size=B).astype(np.float32))
    print(json.dumps(torch_rocm_baseline(net, x, acts, old, adv, ret)))


def torch_rocm_baseline(net, x, acts, old, adv, ret):
    """The same oracle modules executed by PyTorch-ROCm on this GPU (MIOpen / rocBLAS, fp32, TF32
    off): what the reference's own torch code would do on an MI355X.  Reported next to the CPU
    figure for orientation only; bounded to a few seconds."""
    from oracle import ddrl_oracle as O
    torch.backends.cudnn.allow_tf32 = False
    torch.backends.cuda.matmul.allow_tf32 = False
    dev = torch.device("cuda:0")
    net = net.to(dev)
    Bg = 8192
    reps = Bg // x.shape[0]
    xg = x.to(dev).repeat(reps, 1, 1, 1)
    a, o, ad, r = (t.to(dev).repeat(reps) for t in (acts, old, adv, ret))
    with torch.no_grad():
        for _ in range(3):
            net(xg[:256])
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(20):
            net(xg[:256])
        torch.cuda.synchronize()
        t_fwd = (time.perf_counter() - t0) / 20 / 256
    gen = O.learn(net, net.make_optims(), xg, a, o, ad, r, iters=100)
    for _ in range(2):
        next(gen)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(4):
        next(gen)
    torch.cuda.synchronize()
    t_iter = (time.perf_counter() - t0) / 4 / Bg
    return {"value": round(1.0 / (t_fwd + 10 * t_iter), 1), "unit": "env-steps/s", "ppo_iter_ms_B8192": round(t_iter * Bg * 1e3, 2),
            "forward_samples_per_s_n256": round(1.0 / t_fwd, 1),
            "sample": "oracle modules on cuda:0 through PyTorch-ROCm (fp32, TF32 off): forward n=256 x20, PPO iteration "
                      "B=8192 x4 (4 .item() syncs per iteration as in the reference)"}


_PMC_DOCS = {}
# names in the PMC summaries follow the device functions (tools/pmc_to_profiles.py), bench names the launch sites
_PMC_ALIAS = {"ConvFwd1": ("conv_fwd1_resident", "conv_fwd1_planes"), "ConvWgrad1": ("conv_wgrad1_planes",),
              "ConvFwd2": ("conv_fwd2_planes",), "ConvFwd3": ("conv_fwd3_planes",), "FcFwd": ("fc_fwd_planes",),
              "FcDgrad": ("fc_dgrad_planes",), "FcWgrad": ("fc_wgrad_planes",), "ConvDgrad3": ("conv_dgrad3_planes", "conv_dgrad3_exact"),
              "ConvDgrad2": ("conv_dgrad2_both",), "ConvWgrad3": ("conv_wgrad3_pipe", "conv_wgrad3_planes"),
              "ConvWgrad2": ("conv_wgrad2_pipe", "conv_wgrad2_planes")}
# the source file every profiled kernel lives in (ddrl4nav_amd/csrc): what `traffic_stale` is judged by
_KERNEL_SOURCE = {"ConvFwd1": "conv2.hip", "ConvFwd2": "conv2.hip", "ConvFwd3": "conv2.hip", "ConvDgrad3": "conv2.hip", "ConvDgrad2": "conv2.hip",
                  "ConvWgrad1": "wgrad2.hip", "ConvWgrad2": "wgrad2.hip", "ConvWgrad3": "wgrad2.hip", "FcFwd": "fc2.hip", "FcDgrad": "fc2.hip",
                  "FcWgrad": "fc2.hip", "heads_loss": "heads.hip", "clip_adam": "optim.hip", "sqnorm": "optim.hip", "reduce_partials": "optim.hip"}


def _source_of(kernel):
    if kernel in _KERNEL_SOURCE:
        return _KERNEL_SOURCE[kernel]
    for pre, f in (("pconv_", "pconv.hip"), ("fconv_", "fconv.hip"), ("plin_", "plin.hip"), ("c1d_", "c1d.hip"), ("gconv_", "gconv.hip"),
                   ("glin_", "glinear.hip")):
        if kernel.startswith(pre):
            return f
    return None


def evidence_age(doc, kernel):
    """Dates a committed PMC summary against the source the bench is RUNNING: {"traffic_build": the summary's build,
    "traffic_source_file": the kernel's .hip, "traffic_stale": True / False / None}.  Stale = the kernel's source file differs from the
    one the summary was taken on: by the SHA-1 the summary carries (tools/pmc_to_profiles.py `sources`, round 6 on), else -- older
    summaries -- by git: the last commit that touched the file is not an ancestor of the summary's build.  None: undecidable here (an
    old summary on a box without .git)."""
    import hashlib
    import subprocess
    src = _source_of(kernel)
    out = {"traffic_build": (doc or {}).get("build", ""), "traffic_source_file": src, "traffic_stale": None}
    if not doc or not src:
        return out
    path = os.path.join(ROOT, "ddrl4nav_amd", "csrc", src)
    try:
        if "sources" in doc:
            now = hashlib.sha1(open(path, "rb").read()).hexdigest()
            out["traffic_stale"] = doc["sources"].get(src) != now
            return out
        build = str(doc.get("build", "")).split()[0]
        if build and os.path.isdir(os.path.join(ROOT, ".git")):
            last = subprocess.run(["git", "-C", ROOT, "log", "-1", "--format=%H", "--", os.path.relpath(path, ROOT)], capture_output=True,
                                  text=True, timeout=20).stdout.strip()
            dirty = subprocess.run(["git", "-C", ROOT, "status", "--porcelain", "--", os.path.relpath(path, ROOT)], capture_output=True,
                                   text=True, timeout=20).stdout.strip()
            if last:
                anc = subprocess.run(["git", "-C", ROOT, "merge-base", "--is-ancestor", last, build], capture_output=True, timeout=20)
                out["traffic_stale"] = bool(dirty) or anc.returncode != 0
                out["traffic_source_last_commit"] = last[:7]
    except Exception:
        pass
    return out


def _pmc_files(family):
    """Committed PMC summaries of one profile FAMILY, newest last: "atari" = profiles/r<round>_v<version>_pmc_traffic.json (the Pong
    training kernels, tools/prof_round.sh), "nav" = profiles/r<round>_nav<version>_pmc_traffic.json (tools/prof_nav.sh).  The two
    families share kernel names (clip_adam, sqnorm) at different sizes, so a lookup never crosses them; ordering is by the
    integers (round, version) of the name ("v9" sorts after "v15" as text)."""
    import glob
    import re
    pat = re.compile(r"^r(\d+)_%s(\d*)_pmc_traffic\.json$" % ("v" if family == "atari" else "nav"))
    out = []
    for f in glob.glob(os.path.join(ROOT, "profiles", "*_pmc_traffic.json")):
        m = pat.match(os.path.basename(f))
        if m:
            out.append(((int(m.group(1)), int(m.group(2) or 0)), f))
    return [f for _, f in sorted(out)]


def _pmc_lookup(kernel, family="atari"):
    """(document, kernel entry, file name) from the NEWEST committed PMC summary of `family` THAT CONTAINS `kernel` -- never the
    newest file by name alone (round 4: a nav profile sorted after the Pong one and every Pong lookup came back empty)."""
    names = (kernel,) + _PMC_ALIAS.get(kernel, ())
    for f in reversed(_pmc_files(family)):
        try:
            doc = _PMC_DOCS.get(f)
            if doc is None:
                doc = _PMC_DOCS[f] = json.load(open(f))
            for nm in names:
                if nm in doc.get("kernels", {}):
                    return doc, doc["kernels"][nm], os.path.basename(f)
        except Exception:
            continue
    return None, None, None


def pmc_traffic(kernel):
    """HBM bytes per training launch of `kernel` from the committed rocprofv3 PMC passes
    (FETCH_SIZE and WRITE_SIZE in separate passes, tools/prof_pmc.sh -> profiles/*_pmc_traffic.json);
    null when no profile of this build is committed.  bench.py itself never runs the profiler."""
    doc, k, src = _pmc_lookup(kernel)
    if k is None:
        return None
    try:
        out = {"hbm_bytes_per_launch": k["hbm_bytes"], "fetch_bytes": k["fetch_bytes"], "write_bytes": k["write_bytes"],
               # gfx950: FETCH_SIZE tallies wide (16 B / lane) streaming reads at half their bytes (MI355X_MICROARCH.md)
               "hbm_bytes_per_launch_corrected": k.get("hbm_bytes_corrected", 2.0 * k["fetch_bytes"] + k["write_bytes"]),
               "mfma_busy_frac": k["mfma_busy_frac"], "clock_ghz": k["clock_ghz"], "source": src,
               # NOT this run: the committed rocprofv3 --pmc passes of the named build on the named box (bench.py never profiles)
               "measured_on": {"build": doc.get("build", ""), "box": doc.get("box", ""), "batch": doc.get("batch", 65536)}}
        out.update(evidence_age(doc, kernel))
        if "mfma_insts" in k and kernel in MAC and kernel in PIPE:
            # executed / algorithmic matrix work from SQ_INSTS_MFMA: one v_mfma_f32_32x32x16_f16 = 16,384 MAC per wave
            alg = MAC[kernel] * doc.get("batch", 65536) * 2 * PIPE[kernel][1] / 16384.0
            out["executed_over_algorithmic"] = round(k["mfma_insts"] / alg, 3)
            if "valu_insts" in k:
                out["valu_per_mfma"] = round(k["valu_insts"] / k["mfma_insts"] - 1.0, 2)  # SQ_INSTS_VALU counts the MFMAs too
        return out
    except Exception:
        return None


def power_evidence():
    """The newest committed hwmon summary of a bench run (tools/hwmon_trace.py via tools/prof_round.sh): socket power cap, what the
    bench GPU drew during the PPO updates and the shader clock it held there -- NOT measured by this run (an unprivileged bench
    process does not sample sysfs beside itself); None without one."""
    import glob
    import re
    best = None
    for f in glob.glob(os.path.join(ROOT, "profiles", "r*_v*_hwmon_summary.json")):
        m = re.match(r"^r(\d+)_v(\d+)_hwmon_summary\.json$", os.path.basename(f))
        if m and (best is None or (int(m.group(1)), int(m.group(2))) > best[0]):
            best = ((int(m.group(1)), int(m.group(2))), f)
    if best is None:
        return None
    try:
        d = json.load(open(best[1]))
        c = d["at_cap"]
        return {"cap_W": d["power_cap_W"], "update_power_W": c["longest_run"]["power_W_mean"], "update_sclk_GHz": c["longest_run"]["sclk_GHz_mean"],
                "idle_sclk_GHz": d["idle_sclk_GHz_max"], "seconds_at_cap": c["seconds"], "source": os.path.basename(best[1]),
                "note": "the PPO update runs at the socket's power cap: kernel time follows issued work (energy), DESIGN.md section 3.1"}
    except Exception:
        return None


TRAIN_KERNELS = ("ConvFwd1", "ConvFwd2", "ConvFwd3", "FcFwd", "FcDgrad", "FcWgrad", "ConvDgrad3", "ConvDgrad2", "ConvWgrad3",
                 "ConvWgrad2", "ConvWgrad1")


def iteration_traffic():
    """Corrected HBM bytes of ONE PPO iteration: the eleven GEMM kernels + heads_loss + clip_adam of the newest committed PMC
    profile (each launched once per iteration); None without a profile."""
    tot = 0.0
    for k in TRAIN_KERNELS + ("heads_loss", "clip_adam", "reduce_partials", "sqnorm"):
        t = pmc_traffic(k)
        if t is None:
            if k in TRAIN_KERNELS:
                return None
            continue
        tot += t["hbm_bytes_per_launch_corrected"]
    return tot


def mix_model(kernel, measured_ms):
    """Time the kernel's instruction / traffic MIX would take in the synthetic loop of tools/mfma16_mix.hip (matrix instructions with V
    vector-ALU instructions and B HBM bytes each), from the committed microbenchmark and PMC counts: tools/cost_model.py, DESIGN.md
    section 8.  `frac` above prices the kernel against the paper peak; this prices it against what the part sustains for the same mix."""
    try:
        sys.path.insert(0, os.path.join(ROOT, "tools"))
        import glob
        import cost_model
        mixes = sorted(glob.glob(os.path.join(ROOT, "profiles", "*_mfma16_mix.txt")))
        t = pmc_traffic(kernel)
        if not mixes or not t:
            return None
        rate, c_valu, c_byte, _ = cost_model.fit(mixes[-1])
        _, k, _ = _pmc_lookup(kernel)
        by = k["write_bytes"] + 2.0 * k["fetch_bytes"]  # FETCH_SIZE counts 16-byte-per-lane streams at half their bytes on gfx950
        ms = lambda eq: eq * cost_model.FLOP_PER_MFMA / (rate * 1e12) * 1e3
        parts = [ms(k["mfma_insts"]), ms(c_valu * (k["valu_insts"] - k["mfma_insts"])), ms(c_byte * by)]  # SQ_INSTS_VALU includes the MFMAs
        return {"model_ms": round(sum(parts), 3), "mfma_ms": round(parts[0], 3), "valu_ms": round(parts[1], 3), "hbm_ms": round(parts[2], 3),
                "measured_over_model": round(measured_ms / sum(parts), 3), "bare_mfma_loop_tflops": round(rate, 1),
                "valu_cost_in_mfma": round(c_valu, 4), "hbm_byte_cost_in_mfma": round(c_byte, 6),
                "sources": [os.path.basename(mixes[-1]), t["source"]],
                "note": "what v_mfma_f32_32x32x16_f16 sustains on this part next to this kernel's vector-ALU instructions and HBM bytes per "
                        "MFMA (synthetic loop, constants fitted to the microbenchmark only); counts from the committed PMC passes"}
    except Exception as e:  # never let a diagnostic break the bench line
        return {"error": repr(e)[:160]}


def executed_over_algorithmic(kernel):
    """From the newest committed PMC profile (SQ_INSTS_MFMA) where it has the counter, else the design figure."""
    t = pmc_traffic(kernel) or {}
    return t.get("executed_over_algorithmic", round(EXECUTED_OVER_ALGORITHMIC.get(kernel, 1.0), 3))


def async_actor_leg(net, config_nn, N, T, ITERS, dev, steps=3):
    """Asynchronous actor / learner on ONE GPU, as the reference deploys them by default (SYNC = False,
    base_config.py:37: Forward servers keep acting while Backward trains and pick the new weights up
    when the update tag moves, forward.py:121-126): the rollout of step i runs on a second HIP stream,
    from a replica of the weights published after update i-2, while update i-1 runs on the main
    stream.  Same work per step as the sequential headline (one 256 x 256 rollout + GAE + 10 PPO
    iterations); reported next to it, never instead of it."""
    from ddrl4nav_amd.agent import DeviceRollout
    actor, _ = build_net(N, T, ITERS, max_batch=N)
    ahp, hp = actor.hot_path, net.hot_path
    ros = [DeviceRollout(actor, N, horizon=T, gamma=config_nn.EXTRINSIC_DISCOUNT, landa=config_nn.LANDA, seed=7 + i)
           for i in range(2)]
    g = torch.Generator(device=dev)
    g.manual_seed(99)
    for ro in ros:
        ro.frames.copy_(torch.randint(0, 256, ro.frames.shape, dtype=torch.uint8, device=dev, generator=g))
        u = torch.rand((T, N), device=dev, generator=g)
        ro.rewards.copy_(torch.where(u < 0.01, -1.0, torch.where(u > 0.99, 1.0, 0.0)))
        ro.dones.copy_((torch.rand((T, N), device=dev, generator=g) < (1.0 / 800)).to(torch.uint8))
    snapshot = hp.params.clone()
    # acting launches are small and latency-bound: a high-priority stream lets their workgroups slot in
    # between the learner's long-running ones
    s_act, cur = torch.cuda.Stream(device=dev, priority=-1), torch.cuda.current_stream()
    ev_published, ev_taken = torch.cuda.Event(), torch.cuda.Event()
    ev_acted = [torch.cuda.Event(), torch.cuda.Event()]
    ev_published.record(cur)
    ev_taken.record(cur)

    def enqueue_rollout(i):
        ro = ros[i % 2]
        with torch.cuda.stream(s_act):
            s_act.wait_event(ev_published)
            ahp.params.copy_(snapshot)
            ev_taken.record(s_act)
            ahp.params_changed()
            for t in range(T):
                ro.act(t)
            ro.bootstrap()
            ro.finish()
            ev_acted[i % 2].record(s_act)

    def learn_on(i):
        cur.wait_event(ev_acted[i % 2])
        for _ in net.learn(ros[i % 2].batch()):
            pass
        cur.wait_event(ev_taken)       # the actor has copied the previous publication
        snapshot.copy_(hp.params)      # publish (the reference: nn2redis after the last iteration, backward.py:196-199)
        ev_published.record(cur)

    enqueue_rollout(0)
    enqueue_rollout(1)
    learn_on(0)                        # warm-up step
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(1, steps + 1):
        enqueue_rollout(i + 1)
        learn_on(i)
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    actor.hot_path.close()
    return {"value": round(steps * N * T / elapsed, 1), "unit": "env-steps/s", "ms_per_step": round(elapsed / steps * 1e3, 2),
            "steps": steps,
            "note": "rollout i+1 (second stream, weights published after update i-1) overlaps update i on one GPU: the "
                    "reference's default asynchronous deployment (SYNC=False); same work per step as `value`"}


def async_ingest_leg(net, config_nn, N, T, ITERS, dev, steps=10, host_memcpy=False):
    """SURVEY.md section 8d (i) + (iv) with the H2D on the clock AND hidden: the reference's default asynchronous deployment
    (SYNC = False, base_config.py:37; forward.py:121-126) with its transport (multiqueue.py:83-130) replaced by the pinned ring.
    While update i runs on the main stream, rollout i+1 runs on a high-priority acting stream from the weights published after
    update i-1, and EVERY acting step's frames arrive through the pinned-host ring: a producer thread commits one [N,4,84,84]
    uint8 slot per step, the consumer issues hipMemcpyAsync on the copy stream into the second device pool, the forward of step t
    waits for copy t while copy t+1 is already in flight.  The PCIe time of a rollout (1.86 GB) lies under the learner's kernels
    instead of in series with them.  Same work per step as `value`: one N x T rollout + bootstrap + GAE + ITERS PPO iterations."""
    import threading
    from ddrl4nav_amd.agent import DeviceRollout
    from ddrl4nav_amd.data import PinnedRing
    actor, _ = build_net(N, T, ITERS, max_batch=N)
    ahp, hp = actor.hot_path, net.hot_path
    ros = [DeviceRollout(actor, N, horizon=T, gamma=config_nn.EXTRINSIC_DISCOUNT, landa=config_nn.LANDA, seed=17 + i)
           for i in range(2)]
    g = torch.Generator(device=dev)
    g.manual_seed(199)
    for ro in ros:
        u = torch.rand((T, N), device=dev, generator=g)
        ro.rewards.copy_(torch.where(u < 0.01, -1.0, torch.where(u > 0.99, 1.0, 0.0)))
        ro.dones.copy_((torch.rand((T, N), device=dev, generator=g) < (1.0 / 800)).to(torch.uint8))
    slot = N * 4 * 84 * 84
    ring = PinnedRing(slot, n_slots=32)
    rng = np.random.default_rng(4321)
    pool = [rng.integers(0, 256, size=slot, dtype=np.uint8) for _ in range(4)] if host_memcpy else None
    scratch = torch.empty(slot, dtype=torch.uint8, device=dev)
    for _ in range(32):              # every slot holds frames before the clock starts (env workers write into the slots themselves)
        buf = ring.acquire(timeout_ms=10000)
        buf[:] = rng.integers(0, 256, size=slot, dtype=np.uint8)
        ring.commit()
        ring.pop_to(scratch)
    torch.cuda.synchronize()
    total = (steps + 2) * (T + 1)     # rollouts 0 .. steps + 1 are enqueued below: the producer ends exactly when the consumer does
    err = []

    def producer():
        try:
            for i in range(total):
                buf = ring.acquire(timeout_ms=120000)
                if pool is not None:
                    buf[:] = pool[i % 4]
                ring.commit()
        except Exception as e:  # surfaced by the consumer's timeout
            err.append(e)

    th = threading.Thread(target=producer, daemon=True)
    th.start()
    snapshot = hp.params.clone()
    s_act, cur = torch.cuda.Stream(device=dev, priority=-1), torch.cuda.current_stream()
    ev_published, ev_taken = torch.cuda.Event(), torch.cuda.Event()
    taken, taken_n = threading.Condition(), [0]
    ev_acted = [torch.cuda.Event(), torch.cuda.Event()]
    copy_span = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(2)]
    ev_published.record(cur)
    ev_taken.record(cur)

    def enqueue_rollout(i):
        ro = ros[i % 2]
        with torch.cuda.stream(s_act):
            s_act.wait_event(ev_published)
            ahp.params.copy_(snapshot)
            ev_taken.record(s_act)
            with taken:                 # host-side: ev_taken now stands for rollout i's copy of the snapshot
                taken_n[0] = i + 1
                taken.notify_all()
            ahp.params_changed()
            copy_span[i % 2][0].record(ro.copy_stream)
            for t in range(T + 1):
                ro.put_frames_from_ring(t, ring)     # hipMemcpyAsync on the copy stream; this (acting) stream waits for it
                if t < T:
                    ro.act(t)
            copy_span[i % 2][1].record(ro.copy_stream)
            ro.bootstrap()
            ro.finish()
            ev_acted[i % 2].record(s_act)

    def learn_on(i):
        cur.wait_event(ev_acted[i % 2])
        for _ in net.learn(ros[i % 2].batch()):
            pass
        with taken:                    # rollout i + 1 (the one acting under this update) has enqueued its copy of the snapshot
            taken.wait_for(lambda: taken_n[0] >= i + 2, timeout=120)
        cur.wait_event(ev_taken)       # ... and the copy has run: the snapshot may be overwritten
        snapshot.copy_(hp.params)      # publish (the reference: nn2redis after the last iteration, backward.py:196-199)
        ev_published.record(cur)

    # The acting side has its own HOST thread, as the reference's Forward server has its own process: popping 257 ring slots per
    # rollout paces the enqueueing thread to PCIe speed (a slot is reused only when its copy has left it), and the learner's ten
    # iterations must not queue up behind that (measured with one thread: 229.5k against 252k env-steps/s).
    act_err = []

    def act_thread_fn(i):
        try:
            torch.cuda.set_device(dev)
            enqueue_rollout(i)
        except BaseException as e:   # surfaced by the main thread after the join: never a silent half-enqueued rollout
            act_err.append(e)
            with taken:
                taken_n[0] = 1 << 30
                taken.notify_all()

    def start_rollout(i):
        t = threading.Thread(target=act_thread_fn, args=(i,), daemon=True)
        t.start()
        return t

    enqueue_rollout(0)
    t_act = start_rollout(1)
    learn_on(0)                        # warm-up step
    t_act.join()
    if act_err:
        raise act_err[0]
    torch.cuda.synchronize()
    spans = []
    t0 = time.perf_counter()
    for i in range(1, steps + 1):
        t_act = start_rollout(i + 1)   # rollout i+1: ring -> copy stream -> acting stream, on its own host thread
        learn_on(i)                    # ends with the update's one host synchronisation
        t_act.join()                   # everything of rollout i+1 is enqueued (its GPU work may still run: ev_acted orders it)
        if act_err:
            raise act_err[0]
        a, b = copy_span[i % 2]
        b.synchronize()
        spans.append(a.elapsed_time(b))
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    th.join(150)
    if th.is_alive():                  # never free the ring under a producer that is still inside ddrl_ring_acquire
        raise RuntimeError("ring producer did not finish")
    ring.close()
    actor.hot_path.close()
    span = float(np.mean(spans))
    nbytes = slot * (T + 1)
    return {"value": round(steps * N * T / elapsed, 1), "unit": "env-steps/s", "ms_per_step": round(elapsed / steps * 1e3, 2),
            "steps": steps, "h2d_bytes_per_rollout": nbytes, "h2d_span_ms_per_rollout": round(span, 2),
            "h2d_gbps_over_copy_span": round(nbytes / (span * 1e-3) / 1e9, 2),
            "h2d_gbps_over_step": round(nbytes * steps / elapsed / 1e9, 2), "host_memcpy_into_slot": bool(host_memcpy),
            "producer_error": repr(err[0])[:120] if err else None,
            "note": "rollout i+1 acts on a second stream from ring-fed frames (every step's [N,4,84,84] uint8 crosses PCIe inside the "
                    "timed region) while update i runs: the reference's asynchronous deployment (SYNC=False) on the pinned ring; "
                    "`with_ingest_serial` is the same ingest strictly in series with the learner"}


def ingest_leg(net, ro, N, T, steps=2, host_memcpy=False):
    """SURVEY.md section 8d(i) with the H2D on the clock: the frames of every acting step arrive through the pinned-host
    ring (ddrl_ring_*: a producer thread commits one [N,4,84,84] uint8 slot per step, the consumer issues hipMemcpyAsync
    on a copy stream into the device pool, the forward waits for that copy and overlaps the next one).  By default the
    producer commits slots that already hold frames -- env workers write into the pinned slots themselves --;
    host_memcpy=True adds a host-side copy of the 7.2 MB per step into the slot.  Same step otherwise."""
    import threading
    from ddrl4nav_amd.data import PinnedRing
    slot = N * 4 * 84 * 84
    ring = PinnedRing(slot, n_slots=16)
    rng = np.random.default_rng(4321)
    pool = [rng.integers(0, 256, size=slot, dtype=np.uint8) for _ in range(4)] if host_memcpy else None
    for _ in range(16):              # every slot holds frames before the clock starts
        buf = ring.acquire(timeout_ms=10000)
        buf[:] = rng.integers(0, 256, size=slot, dtype=np.uint8)
        ring.commit()
    scratch = torch.empty(slot, dtype=torch.uint8, device=ro.frames.device)
    for _ in range(16):
        ring.pop_to(scratch)
    torch.cuda.synchronize()
    total = (steps + 1) * (T + 1)
    err = []

    def producer():
        try:
            for i in range(total):
                buf = ring.acquire(timeout_ms=60000)
                if pool is not None:
                    buf[:] = pool[i % 4]
                ring.commit()
        except Exception as e:  # surfaced by the consumer's timeout
            err.append(e)

    th = threading.Thread(target=producer, daemon=True)
    th.start()
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    act_ms = []

    def step():
        ev0.record()
        for t in range(T + 1):
            ro.put_frames_from_ring(t, ring)
            if t < T:
                ro.act(t)
        ro.bootstrap()
        ev1.record()
        ro.finish()
        for _ in net.learn(ro.batch()):
            pass
        torch.cuda.synchronize()
        act_ms.append(ev0.elapsed_time(ev1))

    step()                           # warm-up
    act_ms.clear()
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    elapsed = time.perf_counter() - t0
    th.join(10)
    ring.close()
    a = float(np.mean(act_ms))
    return {"value": round(steps * N * T / elapsed, 1), "unit": "env-steps/s", "ms_per_step": round(elapsed / steps * 1e3, 2),
            "steps": steps, "acting_ms_per_rollout": round(a, 2), "h2d_bytes_per_rollout": slot * (T + 1),
            "h2d_gbps_over_acting_phase": round(slot * (T + 1) / (a * 1e-3) / 1e9, 2), "host_memcpy_into_slot": bool(host_memcpy),
            "producer_error": repr(err[0])[:120] if err else None,
            "note": "frames of every acting step cross PCIe through the pinned ring inside the timed region; `value` (the headline) "
                    "has them resident in HBM as the metric defines"}


def nav_leg(whole_loop=True):
    """BASELINE config 4's network (robot_nav: NavPreNet1D x2 + GaussionActor(2) + Critic, reference nn/nav_encoder.py:82-128) through
    the operator-composed path (nn/generic.py; csrc/pconv.hip, fconv.hip, plin.hip, c1d.hip): one PPO iteration on B = 4,096 samples in
    ONE micro-batch, per-operator HIP events (the two encoders run on two streams: the operator times overlap and sum to more than the
    iteration).  `shared_navped_ppo_iter_ms`: the same for the shared NavPedPreNet(4) net of the GAIL nav configuration (config 5's
    encoder).  A sub-record: it never touches the headline `value`."""
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import bench_nav
    # the first run also times ONE whole actor-learner loop at config 4's own size: 512 envs x T = 256 steps of acting + bootstrap +
    # GAE, then 10 PPO iterations over the 131,072 samples in micro-batches of 4,096 (about 8 s; one loop after a one-iteration warm-up)
    r = bench_nav.run(4096, 4096, 3, 256 if whole_loop else None, "nav1d", 10)
    ped = bench_nav.run(4096, 4096, 3, None, "navped")
    # per-operator times from a one-stream run (with the critic's encoder beside the actor's the operators' events overlap)
    r1 = bench_nav.run(4096, 4096, 3, encoder_streams=False)
    ops = r1["ops"]
    dom = max(ops, key=lambda k: ops[k]["ms_per_iter"])
    # conv and dense layers run on the 16-bit matrix pipe as three fp16 plane products (ceiling 2.5 PF / 3); the Conv1d pair of the laser
    # branch as fp32 vector kernels
    planes = {k for k in ops if k.startswith("conv") and not k.startswith("conv1x")} | {k for k in ops if k.startswith("linear")}
    peak = (PEAK_BF16_MFMA_TFLOPS / 3) if dom in planes else PEAK_F32_MFMA_TFLOPS
    return {"workload": r["workload"], "B": r["B"], "micro_batch": r["micro_batch"], "ppo_iter_ms": r["ms_per_ppo_iter_wall"],
            "samples_per_s": r["samples_per_s"], "ppo_iter_ms_one_stream": r1["ms_per_ppo_iter_wall"],
            "shared_navped_ppo_iter_ms": ped["ms_per_ppo_iter_wall"], "gemm_ops_ms_per_iter": r1["gemm_ops_ms_per_iter"],
            "algorithmic_tflops_over_gemm_ops": r1["algorithmic_tflops_over_gemm_ops"],
            "dominant_kernel": dom, "dominant_ms_per_iter": ops[dom]["ms_per_iter"], "dominant_tflops": ops[dom]["tflops"],
            "dominant_pipe": "f16x3" if dom in planes else "f32-input MFMA", "dominant_peak_tflops": round(peak, 1),
            "dominant_roofline_frac": round(ops[dom]["tflops"] / peak, 4),
            # the same figure under the key the headline's `roofline` uses, and the dominant kernel's PMC traffic from the newest committed
            # NAV profile (profiles/r<round>_nav<version>_pmc_traffic.json: a family of its own, bench.py:_pmc_files)
            "roofline_frac": round(ops[dom]["tflops"] / peak, 4), "roofline_traffic": nav_traffic(dom, r["micro_batch"]),
            "whole_loop": r.get("whole_loop"),
            "ops": {k: v for k, v in list(ops.items())[:10]}}


def nav_traffic(op_name, micro_batch):
    """Corrected HBM bytes per launch of the device kernel behind the operator `op_name` (tools/bench_nav.py names, e.g.
    conv5x5_64->128_fwd) from the newest committed nav PMC summary; None when the profile has no such kernel."""
    import re
    m = re.match(r"conv(\d+)x(\d+)_(\d+)->(\d+)_(fwd|dgrad|wgrad)$", op_name)
    files = _pmc_files("nav")
    if not m or not files:
        return None
    kh, _, cin, cout, kind = int(m.group(1)), int(m.group(2)), int(m.group(3)), int(m.group(4)), m.group(5)
    try:
        doc = json.load(open(files[-1]))
        for name, k in doc["kernels"].items():
            # template arguments (csrc/pconv.hip, fconv.hip): pconv_*<CIN x COUT x KS x HIN x PAD> of the convolution the kernel RUNS (a data
            # gradient runs the transposed convolution: CIN = the layer's cout); fconv_first_*<CIN x KS x HIN x PAD>
            t = re.match(r"(pconv_direct_planes|pconv_wgrad_planes)<(\d+)x(\d+)x(\d+)x", name)
            f = re.match(r"(fconv_first_fwd|fconv_first_wgrad)<(\d+)x(\d+)x", name)
            if t:
                fam, a, b, ks = t.group(1), int(t.group(2)), int(t.group(3)), int(t.group(4))
                ok = ks == kh and ((kind == "wgrad") == (fam == "pconv_wgrad_planes")) and \
                    ((a, b) == ((cout, cin) if kind == "dgrad" else (cin, cout)))
            elif f:
                ok = int(f.group(2)) == cin and int(f.group(3)) == kh and kind == ("fwd" if f.group(1).endswith("fwd") else "wgrad")
            else:
                continue
            if ok:
                out = {"kernel": name, "hbm_bytes_per_launch_corrected": k.get("hbm_bytes_corrected", 2.0 * k["fetch_bytes"] + k["write_bytes"]),
                       "mfma_busy_frac": k.get("mfma_busy_frac"), "clock_ghz": k.get("clock_ghz"), "source": os.path.basename(files[-1]),
                       "measured_on": {"build": doc.get("build", ""), "box": doc.get("box", ""), "batch": doc.get("batch")}}
                out.update(evidence_age(doc, name))
                return out
    except Exception:
        pass
    return None


def _rccl_version():
    """RCCL's version as torch reports it ("nccl" IS RCCL on ROCm); None when the build has none."""
    try:
        return ".".join(str(x) for x in torch.cuda.nccl.version())
    except Exception:
        return None


def preflight_static():
    """What a rank can say about the node WITHOUT any collective and without initialising a device context beyond counting: visible
    devices and the environment that selects them, the peer-access matrix (hipDeviceCanAccessPeer through torch), the RCCL torch links
    and the one csrc/comm.cpp resolves (ddrl_comm_info), the environment switches that steer either."""
    import ctypes
    out = {"rank": int(os.environ.get("RANK", "0")), "local_rank": int(os.environ.get("LOCAL_RANK", "0")),
           "world": int(os.environ.get("WORLD_SIZE", "1")), "host": os.uname().nodename,
           "env": {k: os.environ[k] for k in ("HIP_VISIBLE_DEVICES", "ROCR_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES", "HSA_ENABLE_IPC_MODE_LEGACY",
                                              "NCCL_DEBUG", "NCCL_ALGO", "NCCL_PROTO", "NCCL_P2P_DISABLE", "RCCL_MSCCL_ENABLE", "DDRL_ALLREDUCE",
                                              "DDRL_DIST_BACKEND", "MASTER_ADDR", "MASTER_PORT") if k in os.environ},
           "torch": torch.__version__, "torch_rccl_version": _rccl_version()}
    try:
        n = torch.cuda.device_count()
        out["device_count"] = n
        out["devices"] = [torch.cuda.get_device_name(i) or "(name string empty)" for i in range(n)]
        out["peer_access"] = [[1 if i == j else int(torch.cuda.can_device_access_peer(i, j)) for j in range(n)] for i in range(n)]
    except Exception as e:
        out["device_error"] = repr(e)[:200]
    try:
        from ddrl4nav_amd import _lib
        buf, ver = ctypes.create_string_buffer(1024), ctypes.c_int32()
        st = _lib.load().ddrl_comm_info(buf, 1024, ctypes.byref(ver))
        out["ddrl_comm_rccl"] = {"status": int(st), "path": buf.value.decode(errors="replace"), "version_code": int(ver.value)}
    except Exception as e:
        out["ddrl_comm_rccl"] = {"error": repr(e)[:200]}
    return out


def preflight_main(args):
    """`bench.py --gpus N --preflight` (under torchrun, or self-launched like the bench): per rank (1) preflight_static() to stderr BEFORE
    anything collective, (2) the process group with NCCL_DEBUG=INFO into a per-rank file, from which RCCL's choice of algorithm /
    protocol / channels for the gradient all-reduce is quoted, (3) ONE checked and timed SUM all-reduce of the 13,487,420-byte gradient
    arena through torch.distributed and (4) the same through the C-ABI communicator (ddrl_comm_*, csrc/comm.cpp), each compared
    with the analytic sum.  With one rank every step still runs (a one-rank RCCL communicator); ranks that share a device
    (--share-gpu, gloo) skip the RCCL steps and say so.  Rank 0 prints ONE JSON line; exit code 1 when a check fails.
    Reference: the multi-GPU path USTC_lab/server/backward.py:167 leaves as a TODO."""
    import tempfile
    import torch.distributed as dist
    t_start = time.perf_counter()
    rank, world = int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))
    # RCCL reads its debug switches ONCE, the first time anything in the process makes it log: set them before the first call into it
    # (preflight_static asks it for its version).  The GPU boxes export NCCL_DEBUG=VERSION: raised to INFO for this run.
    logdir = tempfile.mkdtemp(prefix="ddrl_preflight_")
    if os.environ.get("NCCL_DEBUG", "").upper() not in ("INFO", "TRACE"):
        os.environ["NCCL_DEBUG"] = "INFO"
    os.environ.setdefault("NCCL_DEBUG_SUBSYS", "INIT,ENV,TUNING,COLL")
    os.environ["NCCL_DEBUG_FILE"] = os.path.join(logdir, "rccl_rank%d.log" % rank)
    static = preflight_static()
    sys.stderr.write("preflight rank %d static (t = %.1f s): %s\n" % (rank, time.perf_counter() - t_start, json.dumps(static)))
    sys.stderr.flush()
    shared = os.environ.get("DDRL_DIST_BACKEND") == "gloo"
    res = {"static": static, "shared_device": shared}
    n_floats = 3371847 + 8                       # the gradient arena + its loss tail: what every PPO iteration reduces
    ok = True
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    torch.cuda.set_device(local_rank % max(1, torch.cuda.device_count()))
    dev = torch.device("cuda", torch.cuda.current_device())
    pattern = (torch.arange(n_floats, device=dev, dtype=torch.float32) % 251) * 0.5 - 31.0      # exact in fp32, sums of <= 8 ranks exact too
    want = pattern * (world * (world + 1) / 2.0)

    def timed(fn, reps=5):
        buf = pattern * float(rank + 1)
        fn(buf)                                    # first call: connection set-up
        torch.cuda.synchronize()
        good = bool(torch.equal(buf, want))
        ts = []
        for _ in range(reps):
            buf = pattern * float(rank + 1)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            fn(buf)
            torch.cuda.synchronize()
            ts.append((time.perf_counter() - t0) * 1e3)
            good = good and bool(torch.equal(buf, want))
        ms = float(np.median(ts))
        return {"matches_analytic_sum": good, "median_ms": round(ms, 4), "first_to_last_ms": [round(t, 4) for t in ts],
                "algbw_gbps": round(n_floats * 4 / (ms * 1e-3) / 1e9, 2), "bytes": n_floats * 4}

    # ---- torch.distributed: "nccl" IS RCCL on ROCm (one rank too: a one-rank communicator still goes through RCCL's init)
    try:
        if not dist.is_initialized():
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            if "MASTER_PORT" not in os.environ:
                import socket
                with socket.socket() as sk:
                    sk.bind(("127.0.0.1", 0))
                    os.environ["MASTER_PORT"] = str(sk.getsockname()[1])
            backend = "gloo" if shared else "nccl"
            kw = {"device_id": dev} if backend == "nccl" else {}
            dist.init_process_group(backend, rank=rank, world_size=world, **kw)
        res["torch_backend"] = dist.get_backend()
        if shared:
            from ddrl4nav_amd.dist import allreduce_flat
            res["torch_distributed"] = dict(timed(lambda b: allreduce_flat(b)), path="gloo through the host (ranks share a device)")
        else:
            res["torch_distributed"] = dict(timed(lambda b: dist.all_reduce(b, op=dist.ReduceOp.SUM)), path="nccl (RCCL)")
        ok = ok and res["torch_distributed"]["matches_analytic_sum"]
    except Exception as e:
        res["torch_distributed"] = {"error": repr(e)[:300]}
        ok = False
    # ---- the C-ABI communicator
    try:
        if shared or world > torch.cuda.device_count():
            res["ddrl_comm"] = {"skipped": "ranks share a device: ncclCommInitRank would fail or hang on duplicate devices"}
        else:
            from ddrl4nav_amd.dist import RcclComm
            comm = RcclComm(rank, world)
            res["ddrl_comm"] = dict(timed(lambda b: comm.allreduce(b)), path="ddrl_comm_* (csrc/comm.cpp, dlopen'ed librccl)")
            comm.close()
            ok = ok and res["ddrl_comm"]["matches_analytic_sum"]
    except Exception as e:
        res["ddrl_comm"] = {"error": repr(e)[:300]}
        ok = False
    # ---- what RCCL said it chose (NCCL_DEBUG=INFO): the tuning / channel / transport lines of this rank
    try:
        lines = open(os.environ["NCCL_DEBUG_FILE"], errors="replace").read().splitlines()
        keys = ("Algo", "algo", "Proto", "proto", "Channel", "channel", "via", "P2P", "SHM", "NET/", "comm 0x", "nranks", "Trees", "Ring ", "XGMI", "version")
        pick = [ln[-220:] for ln in lines if any(k in ln for k in keys)]
        res["rccl_debug"] = {"file_lines": len(lines), "quoted": pick[:40] + (["... %d more" % (len(pick) - 40)] if len(pick) > 40 else [])}
    except Exception as e:
        res["rccl_debug"] = {"error": repr(e)[:200]}
    res["ok"] = ok
    res["seconds"] = round(time.perf_counter() - t_start, 1)
    sys.stderr.write("preflight rank %d result: %s\n" % (rank, json.dumps({k: v for k, v in res.items() if k != "static"})[:4000]))
    sys.stderr.flush()
    allres = [res]
    if world > 1 and dist.is_initialized():
        try:
            allres = [None] * world
            dist.all_gather_object(allres, res)
        except Exception as e:
            allres = [res, {"gather_error": repr(e)[:200]}]
    if rank == 0:
        print(json.dumps({"preflight": True, "n_gpus": world, "ok": all(bool(r and r.get("ok")) for r in allres if isinstance(r, dict) and "ok" in r),
                          "ranks": allres}))
    if dist.is_initialized():
        dist.destroy_process_group()
    return 0 if ok else 1


def build_net(n_envs, horizon, iters, max_batch=None):
    from ddrl4nav_amd.config import BaseConfig, ConfigNN
    from ddrl4nav_amd.runner import create_net
    env = {"env_type": "gym", "env_name": "PongNoFrameskip-v4", "env_num": n_envs, "int_frame_stack": 4,
           "discrete_action": True, "discrete_actions": list(range(6)), "agent_num_per_env": 1, "batch_num_per_env": n_envs}
    parse = types.SimpleNamespace(task="bench", ip="127.0.0.1")
    config_nn = ConfigNN(env)
    config_nn.TRAINING_ITER_TIME = iters
    config = BaseConfig(parse, env)
    config.TIME_MAX = horizon
    return create_net({"config": config, "config_nn": config_nn, "config_env": env},
                      max_batch=max_batch if max_batch is not None else n_envs * horizon), config_nn


def self_launch(args):
    """`python bench.py --gpus N` without a launcher: start N rank processes -- fresh interpreters, one per GPU, RCCL ("nccl")
    between them -- BEFORE this process makes any GPU call (it never does: a process that has initialised the GPU must not be
    replaced or forked into ranks), wait for all of them, pass rank 0's JSON line through and fail when any rank fails.  The
    driver's own form (python -m torch.distributed.run ... bench.py --gpus N) sets WORLD_SIZE and takes the other path."""
    import socket
    import subprocess
    n = args.gpus
    ndev = torch.cuda.device_count()  # counting devices does not initialise the GPU on this image
    env_base = dict(os.environ)
    if ndev < n:
        if not args.share_gpu:
            sys.stderr.write("bench.py: --gpus %d but this host shows %d GPU(s); pass --share-gpu for a gloo rehearsal of the "
                             "rank plumbing (not a scaling measurement)\n" % (n, ndev))
            return 2
        env_base["DDRL_DIST_BACKEND"] = "gloo"
    argv = [a for a in sys.argv[1:]]
    import tempfile
    import time as _t
    deadline_s = float(os.environ.get("DDRL_BENCH_LAUNCH_TIMEOUT_S", "3000"))
    codes, out0 = [1] * n, ""
    for attempt in range(3):  # a rendezvous port picked by bind-then-close can be taken before the ranks bind it: new port, again
        with socket.socket() as sk:
            sk.bind(("127.0.0.1", 0))
            port = sk.getsockname()[1]
        procs, t0 = [], _t.time()
        out_f = tempfile.TemporaryFile()   # rank 0's stdout: a file, so that polling never blocks on a full pipe
        err_f = tempfile.TemporaryFile()   # rank 0's stderr is passed through AND searched for a rendezvous failure
        for r in range(n):
            env = dict(env_base, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                       DDRL_BENCH_LAUNCHER="self", HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
            procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + argv, env=env,
                                          stdout=out_f if r == 0 else subprocess.DEVNULL, stderr=err_f if r == 0 else None))
        # poll ALL ranks: the first one that exits non-zero (or the deadline) ends the others -- a rank that died during init or
        # inside a collective would otherwise leave rank 0 waiting for the collective's own timeout, if it has one
        failed = None
        while True:
            codes = [p.poll() for p in procs]
            if all(c is not None for c in codes):
                break
            bad = [i for i, c in enumerate(codes) if c not in (None, 0)]
            if bad or _t.time() - t0 > deadline_s:
                failed = "rank %s exited with %s" % (bad[0], codes[bad[0]]) if bad else "no result after %.0f s" % deadline_s
                for p in procs:
                    if p.poll() is None:
                        p.terminate()
                for p in procs:
                    try:
                        p.wait(timeout=10)
                    except subprocess.TimeoutExpired:
                        p.kill()
                        p.wait()
                codes = [p.returncode for p in procs]
                break
            _t.sleep(0.05)
        out_f.seek(0)
        out0 = out_f.read().decode(errors="replace")
        err_f.seek(0)
        err0 = err_f.read().decode(errors="replace")
        out_f.close()
        err_f.close()
        sys.stderr.write(err0)
        if failed:
            sys.stderr.write("bench.py: %s; the other ranks were terminated\n" % failed)
        rendezvous = any(k in err0 for k in ("Address already in use", "EADDRINUSE", "address already in use"))
        if not (failed and rendezvous and attempt < 2):
            break
        sys.stderr.write("bench.py: rendezvous port %d was taken, retrying with a new one\n" % port)
    line = [ln for ln in out0.splitlines() if ln.startswith("{")]
    if line and '"preflight": true' in line[-1]:   # diagnostics: the line is the product, also (above all) when a check failed
        print(line[-1])
        return 0 if not any(codes) and json.loads(line[-1]).get("ok") else 1
    if any(codes) or not line:
        sys.stderr.write("bench.py: rank exit codes %s, %d JSON line(s) from rank 0\n" % (codes, len(line)))
        return 1
    got = json.loads(line[-1])
    if got.get("n_gpus") != n or got.get("ranks_joined") != n:
        sys.stderr.write("bench.py: %s of %d ranks joined\n" % (got.get("ranks_joined"), n))
        return 1
    print(line[-1])
    return 0


def main():
    if "--torch-leg" in sys.argv:  # child process of cpu_baseline(): the same-GPU PyTorch-ROCm comparison
        return torch_leg_main()
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--envs", type=int, default=256, help="envs per GPU (BASELINE config 2/3)")
    ap.add_argument("--horizon", type=int, default=256, help="TIME_MAX")
    ap.add_argument("--iters", type=int, default=10, help="TRAINING_ITER_TIME")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-async", action="store_true", help="skip the asynchronous actor/learner leg")
    ap.add_argument("--no-ingest", action="store_true", help="skip the leg that feeds the frames through the pinned-host ring")
    ap.add_argument("--no-nav", action="store_true", help="skip the robot_nav (BASELINE config 4 network) sub-record")
    ap.add_argument("--no-nav-loop", action="store_true", help="nav sub-record without the whole loop at config 4's size (about 10 s)")
    ap.add_argument("--ingest-memcpy", action="store_true", help="ingest leg: the producer also copies 7.2 MB per step into the slot")
    ap.add_argument("--preflight", action="store_true",
                    help="first-contact diagnostics of a multi-GPU node instead of the bench: devices, peer-access matrix, the RCCL this "
                         "process and csrc/comm.cpp resolve, RCCL's chosen algorithm / protocol for the 13.5 MB gradient all-reduce, one "
                         "timed + checked all-reduce through torch.distributed AND through ddrl_comm; prints one JSON line")
    ap.add_argument("--share-gpu", action="store_true",
                    help="rehearsal on a box with fewer GPUs than ranks: the ranks share the devices round-robin and reduce over gloo "
                         "(DDRL_DIST_BACKEND=gloo); never a measurement of scaling")
    args = ap.parse_args()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        return self_launch(args)
    if "WORLD_SIZE" in os.environ and int(os.environ["WORLD_SIZE"]) != args.gpus:
        sys.stderr.write("bench.py: --gpus %d but the launcher set WORLD_SIZE=%s\n" % (args.gpus, os.environ["WORLD_SIZE"]))
        return 2

    if args.preflight:
        return preflight_main(args)
    if int(os.environ.get("WORLD_SIZE", "1")) > 1:
        # the cheap half of --preflight on EVERY N > 1 run, on stderr, BEFORE the first collective: if the rendezvous or the first
        # all-reduce hangs on a node nobody has run on yet, what this rank saw is already on record
        sys.stderr.write("bench.py rank %s pre-collective: %s\n" % (os.environ.get("RANK", "?"), json.dumps(preflight_static())))
        sys.stderr.flush()
    from ddrl4nav_amd.dist import broadcast_params, init_from_env
    rank, world, local_rank = init_from_env()
    if world == 1:
        torch.cuda.set_device(0)
    import torch.distributed as dist
    dev = torch.device("cuda", torch.cuda.current_device())

    from ddrl4nav_amd.agent import DeviceRollout
    from ddrl4nav_amd.engine import Timer
    from ddrl4nav_amd.utils.recipe import make_weights

    N, T, ITERS = args.envs, args.horizon, args.iters
    B = N * T
    net, config_nn = build_net(N, T, ITERS)
    net.load_state_dict({k: torch.from_numpy(v.copy()) for k, v in make_weights(0).items()})
    net.deferred_stats = True   # one host sync per update instead of one per iteration (nn/ppo.py:learn); same values, same keys
    hp = net.hot_path
    broadcast_params(hp.params)
    hp.params_changed()
    hp.profile(True)

    # ---- synthetic rollout inputs, resident in HBM before the timed region ----
    ro = DeviceRollout(net, N, horizon=T, gamma=config_nn.EXTRINSIC_DISCOUNT, landa=config_nn.LANDA, seed=rank * 1000003)
    g = torch.Generator(device=dev)
    g.manual_seed(1234 + rank)
    ro.frames.copy_(torch.randint(0, 256, ro.frames.shape, dtype=torch.uint8, device=dev, generator=g))
    u = torch.rand((T, N), device=dev, generator=g)
    ro.rewards.copy_(torch.where(u < 0.01, -1.0, torch.where(u > 0.99, 1.0, 0.0)))
    ro.dones.copy_((torch.rand((T, N), device=dev, generator=g) < (1.0 / 800)).to(torch.uint8))

    t_act, t_gae, t_upd = Timer(), Timer(), Timer()
    phase = {"act_ms": 0.0, "gae_ms": 0.0, "update_ms": 0.0}
    last = {}

    def one_step(timed):
        t_act.start()
        for t in range(T):
            ro.act(t)
        ro.bootstrap()
        t_act.stop()
        t_gae.start()
        ro.finish()
        t_gae.stop()
        t_upd.start()
        for loss_items, _, _ in net.learn(ro.batch()):
            last.update(loss_items)
        t_upd.stop()
        if timed:
            torch.cuda.synchronize()
            phase["act_ms"] += t_act.elapsed_ms()
            phase["gae_ms"] += t_gae.elapsed_ms()
            phase["update_ms"] += t_upd.elapsed_ms()

    for _ in range(args.warmup):
        one_step(False)
    torch.cuda.synchronize()
    hp.profile(False)
    hp.profile_read()
    # Reset the accumulators: per-kernel times cover the timed region only.  The training kernels (ms each) are
    # timed live; the acting launches (10-40 us each, 1,285 per rollout) are NOT -- event records around them cost
    # about as much as they do (measured: 44.5 -> 34.6 ms per rollout without) -- they get their own pass below.
    hp.profile(True, acting=False)
    if world > 1:
        hp.time_allreduce(True)
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    step_ms = []                         # per-step wall time: one_step(True) ends with a device synchronisation
    for _ in range(args.steps):
        ts = time.perf_counter()
        one_step(True)
        step_ms.append((time.perf_counter() - ts) * 1e3)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    per_rank = [elapsed]
    if world > 1:
        mine = torch.tensor([elapsed], dtype=torch.float64, device=dev if dist.get_backend() == "nccl" else "cpu")
        allr = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(allr, mine)
        per_rank = [float(t.item()) for t in allr]
        elapsed = max(per_rank)
    stats = hp.stats()
    # N > 1: the run proves its own replicas.  Every rank's parameters after the timed updates as a 64-bit checksum (the fp32 bit
    # patterns summed as integers) and its step count, gathered on all ranks: clip + Adam run on every rank from the SAME reduced
    # gradient, so the replicas must be bit-identical (the reference has no multi-GPU path to compare with,
    # server/backward.py:167).  A collective that dropped or re-ordered a contribution on one rank shows up here.
    replicas = None
    if world > 1:
        on_dev = dist.get_backend() == "nccl"
        mine = torch.stack([hp.params.view(torch.int32).to(torch.int64).sum(), torch.tensor(hp.step, dtype=torch.int64, device=dev)])
        mine = mine if on_dev else mine.cpu()
        allc = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(allc, mine)
        sums, stepsr = [int(t[0].item()) for t in allc], [int(t[1].item()) for t in allc]
        replicas = {"identical": len(set(sums)) == 1 and len(set(stepsr)) == 1, "param_checksums": ["%016x" % (c & (2 ** 64 - 1)) for c in sums],
                    "optimizer_steps": stepsr}
    ar_ms = hp.allreduce_ms() if world > 1 else []
    hp.time_allreduce(False)
    prof = hp.profile_read()
    # per-kernel times of the acting launches: a separate, untimed pass of 64 forwards with events around them
    hp.profile(False)
    hp.profile(True, acting=True)
    for t in range(min(T, 64)):
        ro.act(t)
    torch.cuda.synchronize()
    hp.profile(False)
    for k, v in hp.profile_read().items():
        prof.setdefault(k, v)

    if rank == 0:
        steps = args.steps
        env_steps = world * N * T * steps
        value = env_steps / elapsed
        kernels = {}
        for k, (ms, calls) in sorted(prof.items(), key=lambda kv: -kv[1][0]):
            ent = {"ms_total": round(ms, 3), "calls": calls, "ms_avg": round(ms / max(calls, 1), 4)}
            base = k[:-4] if k.endswith(".act") else k
            if base in MAC:
                # training launches process B samples each, acting launches (".act", FcFwdSplit) N samples
                per_launch = 2 * 2 * MAC[base] * (N if k.endswith(".act") else B)
                ent["tflops"] = round(per_launch * calls / (ms * 1e-3) / 1e12, 2)
                ent["flop_per_launch"] = per_launch
                if k in PIPE:
                    pipe, products = PIPE[k]
                    ent["pipe"] = pipe
                    ent["pipe_ceiling_tflops"] = round(PEAK_BF16_MFMA_TFLOPS / products, 1)
                    ent["frac_of_pipe_ceiling"] = round(ent["tflops"] / (PEAK_BF16_MFMA_TFLOPS / products), 4)
                    ent["executed_over_algorithmic"] = executed_over_algorithmic(k)
            elif k == "ActConvs":
                # conv1 + conv2 + conv3 of an acting forward in one launch (csrc/act.hip), N samples
                per_launch = 2 * 2 * (MAC["ConvFwd1"] + MAC["ConvFwd2"] + MAC["ConvFwd3"]) * N
                ent["tflops"] = round(per_launch * calls / (ms * 1e-3) / 1e12, 2)
                ent["flop_per_launch"] = per_launch
            elif k in HBM_BYTES:
                # HBM-bound kernels: algorithmic bytes per launch / launch time
                per_launch = HBM_BYTES[k](N, B, hp.n_params)
                ent["gbps"] = round(per_launch * calls / (ms * 1e-3) / 1e9, 1)
                ent["bytes_per_launch"] = per_launch
                ent["frac_of_hbm_peak"] = round(ent["gbps"] / PEAK_HBM_GBPS, 4)
            kernels[k] = ent
        # dominant kernel = largest accumulated time among the GEMM-shaped kernels
        gemm = {k: v for k, v in kernels.items() if k in MAC}
        roofline = None
        if gemm:
            dom = max(gemm, key=lambda k: gemm[k]["ms_total"])
            d = gemm[dom]
            pipe, products = PIPE.get(dom, ("f32", None))
            peak = PEAK_BF16_MFMA_TFLOPS / products if products else PEAK_F32_MFMA_TFLOPS
            td = pmc_traffic(dom)
            # the committed PMC passes ran at their own batch (65,536); a run at another B scales the per-launch bytes linearly
            tscale = (B / float(td["measured_on"]["batch"])) if td else 1.0
            tbytes = td["hbm_bytes_per_launch_corrected"] * tscale if td else None
            roofline = {"kernel": dom, "bound": "mfma", "achieved": d["tflops"], "peak": round(peak, 1),
                        "unit": "TFLOP/s", "frac": round(d["tflops"] / peak, 4),
                        # HBM bytes per launch from the PMC passes WITH the guide's gfx950 correction (FETCH_SIZE counts wide streaming
                        # reads at half their bytes: 2 x FETCH_SIZE + WRITE_SIZE); the uncorrected sum is in traffic_detail
                        "traffic": tbytes, "traffic_scaled_by_batch": round(tscale, 6),
                        # traffic is NOT measured by this run (bench.py never profiles): it is the newest committed PMC summary that holds
                        # the kernel, dated here -- its build, and whether the kernel's source file has changed since (evidence_age)
                        "traffic_build": td["traffic_build"] if td else None, "traffic_stale": td["traffic_stale"] if td else None,
                        "traffic_detail": td,
                        # what actually binds the update: the socket's power cap (committed hwmon trace of a bench run)
                        "power": power_evidence(),
                        "mix_model": mix_model(dom, d["ms_avg"] / tscale),
                        "avg_launch_ms": d["ms_avg"], "launches": d["calls"],
                        # the HBM side of the same kernel (north_star asks for the HBM fraction): corrected PMC bytes per launch /
                        # this run's launch time / 8 TB/s
                        "hbm_bytes_per_launch_corrected": tbytes,
                        "hbm_frac": round(tbytes / (d["ms_avg"] * 1e-3) / 1e9 / PEAK_HBM_GBPS, 4) if td else None,
                        "algorithmic_bytes_per_launch": ALG_BYTES[dom](B) if dom in ALG_BYTES else None,
                        "algorithmic_flop_per_launch": d["flop_per_launch"], "pipe": pipe,
                        "executed_over_algorithmic": executed_over_algorithmic(dom),
                        "note": "dominant training kernel (largest accumulated time).  achieved = ALGORITHMIC fp32 FLOP (2*2*MAC per "
                                "sample, both encoders, x B) / launch time from HIP events on the launch stream; peak = the ceiling of the "
                                "pipe the kernel runs on: dense 16-bit MFMA 2.5 PFLOP/s / plane products (3 for two fp32 operands as two "
                                "scaled fp16 planes each, 2 when one operand -- the pixels -- is exact in fp16), or the "
                                "f32-input MFMA peak 157.3"}
            # every GEMM kernel against ITS pipe's ceiling, and the time-weighted mean over the training kernels
            tw = sum(v["ms_total"] for k, v in gemm.items() if "frac_of_pipe_ceiling" in v)
            if tw > 0:
                roofline["time_weighted_frac_all_gemm_kernels"] = round(
                    sum(v["ms_total"] * v["frac_of_pipe_ceiling"] for v in gemm.values() if "frac_of_pipe_ceiling" in v) / tw, 4)
        upd_ms = phase["update_ms"] / steps
        it_bytes = iteration_traffic()
        if roofline is not None and it_bytes:
            it_bytes *= roofline["traffic_scaled_by_batch"]
            roofline["iteration_hbm_bytes_corrected"] = round(it_bytes)
            roofline["iteration_hbm_frac"] = round(it_bytes / (upd_ms / ITERS * 1e-3) / 1e9 / PEAK_HBM_GBPS, 4)
            roofline["iteration_algorithmic_bytes"] = 28296 * B     # frames re-read + loss operands per sample
        total_flop = env_steps / world * (FLOP_ACT_PER_STEP + ITERS * FLOP_TRAIN_PER_SAMPLE)
        out = {
            "metric": "env-steps/sec (whole node) + PPO update ms, Pong 256 envs at 1/2/4/8 GPUs",
            "value": round(value, 1), "unit": "env-steps/s", "n_gpus": world, "steps": steps, "warmup": args.warmup,
            "ms_per_step": round(elapsed / steps * 1e3, 2), "higher_is_better": True, "scaling": "weak",
            # median + p95 over the timed steps (rank 0's own steps; `value` stays total work / total time)
            "step_ms": {"median": round(float(np.median(step_ms)), 3), "p95": round(float(np.percentile(step_ms, 95)), 3),
                        "min": round(float(np.min(step_ms)), 3), "max": round(float(np.max(step_ms)), 3), "n": len(step_ms)},
            "value_at_median_step": round(world * N * T / (float(np.median(step_ms)) * 1e-3), 1),
            "vs_baseline": None, "dtype": "f32 (f16x3 MFMA planes, fp32 accumulate)",
            "arithmetic": "f16x3 split-plane MFMA (f16x2 where the pixels are exact), fp32 accumulate, per-sample power-of-two scales in the backward",
            "data": "synthetic",
            "config": {"workload": "Pong %d envs/GPU x T=%d, %d PPO iters on B=%d/GPU, 2 AtariPreNet encoders, A=6" % (N, T, ITERS, B),
                       "envs_per_gpu": N, "horizon": T, "ppo_iters": ITERS, "parallelism": "dp%d" % world,
                       "arithmetic": "fp32 operands as two scaled fp16 planes on the 16-bit MFMA, fp32 accumulate",
                       # the phase figures of the headline, where the driver's parser keeps them
                       "ppo_update_ms": round(upd_ms, 2), "ppo_iter_ms": round(upd_ms / ITERS, 3),
                       "acting_ms_per_rollout": round(phase["act_ms"] / steps, 2)},
            "ranks_joined": world if world == 1 else dist.get_world_size(),
            "launcher": os.environ.get("DDRL_BENCH_LAUNCHER") or ("torchrun" if world > 1 else "none"),
            "elapsed_s_per_rank": [round(t, 4) for t in per_rank],
            "devices": {"visible": torch.cuda.device_count(), "rank0": torch.cuda.get_device_name(dev),
                        "shared_by_ranks": bool(world > torch.cuda.device_count())},
            "ppo_update_ms": round(upd_ms, 2), "ppo_iter_ms": round(upd_ms / ITERS, 3),
            "acting_ms_per_rollout": round(phase["act_ms"] / steps, 2), "gae_ms": round(phase["gae_ms"] / steps, 4),
            "acting_env_steps_per_s_per_gpu": round(N * (T + 1) / (phase["act_ms"] / steps * 1e-3), 1),
            # fp32-equivalent (algorithmic) work of the whole step per second.  NOT a roofline fraction: the GEMM kernels run on
            # the bf16 matrix pipe (fp32-accurate plane products); each kernel's fraction of its pipe's ceiling is in `kernels`
            "whole_step_tflops_per_gpu": round(total_flop / elapsed / 1e12, 2),
            # section 8e: one SUM all-reduce of the 13,487,420-byte gradient arena (+ loss tail) per PPO iteration, bracketed by
            # events on the compute stream of rank 0 (includes waiting for the slowest rank to arrive)
            "allreduce": None if world == 1 else {
                "bytes": int(hp.grads.numel() * 4), "calls": len(ar_ms), "ms_avg": round(float(np.mean(ar_ms)), 4),
                "ms_p50": round(float(np.median(ar_ms)), 4), "ms_max": round(float(np.max(ar_ms)), 4),
                "ms_per_update": round(float(np.sum(ar_ms)) / steps, 3), "backend": dist.get_backend(),
                "rccl_version": _rccl_version(), "path": ("ddrl_comm (C ABI, csrc/comm.cpp)" if hp.comm is not None else "torch.distributed"),
                # overlapped = layer buckets reduced on a second stream under the rest of the backward: the span above is then the
                # EXPOSED part (what the compute stream waited for before clip + Adam)
                "overlapped_with_backward": bool(hp._overlap), "exposed_ms_per_iteration": round(float(np.mean(ar_ms)), 4),
                "algbw_gbps": round(hp.grads.numel() * 4 / (float(np.median(ar_ms)) * 1e-3) / 1e9, 2)},
            "replicas_identical": None if replicas is None else replicas["identical"], "replicas": replicas,
            "last_losses": dict(stats, **{k: v for k, v in last.items() if k != "PpoBackUpTime"}),
            "roofline": roofline, "kernels": kernels,
            "dtype_note": "fp32 operands in HBM, fp32 accumulation; every operand enters the 16-bit MFMA as TWO scaled fp16 planes (22 bits, not "
                          "24) and the three plane products that matter are summed in fp32 (f16x3; conv1's forward and weight gradient: "
                          "exact-fp16 pixels 0..255 x two planes of the other operand, f16x2).  Measured against float64: every parameter "
                          "tensor's gradient has 0.54-0.93 x the rms error of torch's own fp32 evaluation (profiles/"
                          "r06_accuracy_attribution_f21.txt), each operator <= 1.25 x torch-fp32's mean error (tests/test_gpu_parity.py::"
                          "*_is_at_least_fp32_accurate; worst 1.21).  The error is a FLOOR relative to a tensor's largest products, not to "
                          "the element: on the 1-2 % of gradient elements with |g| near Adam's eps the first optimiser step differs from "
                          "a float64 step by 1.1-2 x what an fp32 evaluation's does (tests/golden/margins.json __vs_onednn_only: up to "
                          "1.74).  'tflops' is fp32-equivalent (algorithmic) work; acting launches of at most 512 envs run conv1-conv3 in "
                          "one kernel (csrc/act.hip), same arithmetic",
            "kernel_timing": "training kernels: HIP events around every launch inside the timed region; acting launches "
                             "(ActConvs, FcFwdSplit, heads_act): a separate, untimed pass of 64 forwards after it",
        }
        if world == 1 and not args.no_async:
            hp.profile(False)
            try:
                out["async_actor_learner"] = async_actor_leg(net, config_nn, N, T, ITERS, dev)
            except Exception as e:  # an extra, never allowed to take the headline line down
                out["async_actor_learner"] = {"error": repr(e)[:200]}
        if world == 1 and not args.no_ingest:
            hp.profile(False)
            try:
                # the section-8d(i)/(iv) figure: H2D of every acting step inside the timed region, overlapped with the learner
                out["with_ingest"] = async_ingest_leg(net, config_nn, N, T, ITERS, dev, steps=max(10, min(args.steps, 20)),
                                                      host_memcpy=args.ingest_memcpy)
                out["value_with_ingest"] = out["with_ingest"]["value"]
                out["value_with_ingest_over_value"] = round(out["value_with_ingest"] / out["value"], 4)
                out["config"].update(value_with_ingest=out["value_with_ingest"], value_with_ingest_over_value=out["value_with_ingest_over_value"],
                                     h2d_gbps_over_copy_span=out["with_ingest"]["h2d_gbps_over_copy_span"])
            except Exception as e:
                out["with_ingest"] = {"error": repr(e)[:200]}
            try:
                out["with_ingest_serial"] = ingest_leg(net, ro, N, T, host_memcpy=args.ingest_memcpy)
            except Exception as e:
                out["with_ingest_serial"] = {"error": repr(e)[:200]}
        if world == 1 and not args.no_nav:
            try:
                out["nav"] = nav_leg(whole_loop=not args.no_nav_loop)
                out["config"].update(nav_ppo_iter_ms=out["nav"]["ppo_iter_ms"], nav_samples_per_s=out["nav"]["samples_per_s"],
                                     nav_shared_navped_ppo_iter_ms=out["nav"]["shared_navped_ppo_iter_ms"],
                                     nav_roofline_frac=out["nav"]["roofline_frac"],
                                     nav_whole_loop_env_steps_per_s=(out["nav"]["whole_loop"] or {}).get("env_steps_per_s"))
            except Exception as e:
                out["nav"] = {"error": repr(e)[:200]}
        if not args.no_cpu_baseline and world == 1:
            out["cpu_baseline"] = cpu_baseline()
        print(json.dumps(out))
    hp.close()
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    sys.exit(main() or 0)
