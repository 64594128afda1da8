"""Shared machinery of the learn-sequence parity tests (test infrastructure).

* ``f64_trajectory(mode)``: the oracle's ``learn`` in FLOAT64 on the fixture batch of a learner mode --
  the "true" trajectory p64 (parameters after every iteration, losses).  The oracle's float64 run is itself
  pinned against the reference's float64 run (checksums in tests/golden/f*_spread_*.npz, made by importing the
  reference: tests/golden/make_golden_spread.py; checked in tests/test_oracle_golden.py).
* ``param_deviation``: per parameter tensor, how far a flat fp32 parameter arena sits from p64 -- as a RATIO
  to how far the reference's own fp32 evaluations (1 thread, 8 threads, three batch orders) sit from it.
* ``Margins``: every envelope-type bound of the tests is  measured_ratio <= limit  with the limits read from
  tests/golden/margins.json (= the ratios measured on an MI355X + 50 %, rounded up).  ``DDRL_RECORD_MARGINS=1``
  switches the asserts off and writes what was measured to gpurun_out/margins_measured.json instead.
"""
import json
import os

import numpy as np
import torch

from ddrl4nav_amd.utils.recipe import make_weights, param_specs
from oracle import ddrl_oracle as O

HERE = os.path.dirname(os.path.abspath(__file__))
GOLDEN = os.path.join(HERE, "golden")
_MARGINS = os.path.join(GOLDEN, "margins.json")
_RECORD = os.environ.get("DDRL_RECORD_MARGINS") == "1"
_MEASURED_OUT = os.path.join(os.path.dirname(HERE), "gpurun_out", "margins_measured.json")

MODES = {
    # mode: (spread fixture, batch fixture for actions/old_logps/advs, fixture holding rets, shared, smooth_l1)
    "default": ("f4b_spread_default", "f4_learn", "f4_learn", False, False),
    "shared": ("f10b_spread_shared", "f10_shared", "f10_shared", True, False),
    "smooth": ("f11c_spread_smooth", "f3_loss", "f11_smooth_l1", False, True),
}


def _load(name):
    return np.load(os.path.join(GOLDEN, name + ".npz"))


def mode_batch(mode):
    """(frames u8, actions, old_logps, advs, rets) of a learner mode's fixture."""
    _, src, rsrc, _, _ = MODES[mode]
    g, r = _load(src), _load(rsrc)
    return _load("f3_loss")["frames"], g["actions"], g["old_logps"], g["advs"], r["rets"]


_TRAJ = {}


def f64_trajectory(mode, iters=10):
    """{"params": {it: {name: float64 array}}, "losses": [iters, 4], "p0": {...}} of the oracle in float64."""
    if mode in _TRAJ:
        return _TRAJ[mode]
    _, _, _, shared, smooth = MODES[mode]
    frames, actions, old_logps, advs, rets = mode_batch(mode)
    w = make_weights(0, shared=shared)
    threads = torch.get_num_threads()
    torch.set_num_threads(min(8, max(1, os.cpu_count() or 1)))
    try:
        net = (O.OracleSharedPPO() if shared else O.OraclePPO())
        net.load_weights(w)
        net.double()
        x = O.frames_to_f32(frames).double()  # float64(float32(u8 / 255.0)), what the reference's f64 run sees
        t = lambda a: torch.from_numpy(np.asarray(a)).double()
        params, losses = {}, []
        for it, (ld, _, _) in enumerate(O.learn(net, net.make_optims(), x, t(actions), t(old_logps), t(advs), t(rets),
                                                iters=iters, smooth_l1=smooth), 1):
            losses.append([ld["PpoTotalLoss"], ld["ActorLoss"], ld["VLoss"], ld["EntLoss"]])
            params[it] = {k: p.detach().numpy().copy() for k, p in net.named_parameters()}
    finally:
        torch.set_num_threads(threads)
    _TRAJ[mode] = {"params": params, "losses": np.asarray(losses), "p0": {k: np.asarray(v, np.float64) for k, v in w.items()}}
    return _TRAJ[mode]


def split_flat(flat, shared=False, n_actions=6):
    out, off = {}, 0
    for name, shape, _ in param_specs(n_actions=n_actions, shared=shared):
        n = int(np.prod(shape))
        out[name] = np.asarray(flat[off:off + n]).reshape(shape)
        off += n
    return out


def param_deviation(mode, it, flat_params):
    """Largest ratio over the parameter tensors of  dev(candidate, p64) / dev(reference fp32, p64)  for the three
    deviation measures (L2, max-abs, 1 - cosine of the accumulated update), plus where each maximum sits."""
    fixture, _, _, shared, _ = MODES[mode]
    sp = _load(fixture)
    traj = f64_trajectory(mode)
    got = split_flat(np.asarray(flat_params, np.float64), shared)
    worst = {"l2": (0.0, ""), "max": (0.0, ""), "1mcos": (0.0, "")}
    for name, a in got.items():
        p64, p0 = traj["params"][it][name], traj["p0"][name]
        d = (a - p64).ravel()
        u, u64 = (a - p0).ravel(), (p64 - p0).ravel()
        key = "it%d/%s" % (it, name)
        cos = float(u @ u64 / (np.linalg.norm(u) * np.linalg.norm(u64) + 1e-300))
        vals = {"l2": float(np.sqrt(d @ d)) / float(sp["ref_l2/" + key]),
                "max": float(np.abs(d).max()) / float(sp["ref_max/" + key]),
                "1mcos": (1.0 - cos) / max(float(sp["ref_1mcos/" + key]), 1e-12)}
        for k, v in vals.items():
            if v > worst[k][0]:
                worst[k] = (v, name)
    return worst


class Margins:
    """limit lookup + measurement log.  check(test, key, measured) asserts measured <= margins.json[test][key]."""

    def __init__(self):
        self.limits = json.load(open(_MARGINS)) if os.path.exists(_MARGINS) else {}
        self.measured = {}

    def check(self, test, key, measured, where=""):
        measured = float(measured)
        slot = self.measured.setdefault(test, {})
        slot[key] = max(slot.get(key, 0.0), measured)
        if _RECORD:
            self._flush()
            return
        limit = self.limits.get(test, {}).get(key)
        assert limit is not None, "tests/golden/margins.json has no limit for %s / %s (measured %.4g)" % (test, key, measured)
        assert measured <= limit["limit"], "%s / %s: measured ratio %.4g exceeds the limit %.4g %s" % (
            test, key, measured, limit["limit"], where)

    def _flush(self):
        os.makedirs(os.path.dirname(_MEASURED_OUT), exist_ok=True)
        old = {}
        if os.path.exists(_MEASURED_OUT):
            try:
                old = json.load(open(_MEASURED_OUT))
            except ValueError:
                old = {}
        for t, d in self.measured.items():
            o = old.setdefault(t, {})
            for k, v in d.items():
                o[k] = max(o.get(k, 0.0), v)
        json.dump(old, open(_MEASURED_OUT, "w"), indent=1, sort_keys=True)


MARGINS = Margins()


def loss_envelope(ref, *others):
    """Running maximum over the iterations of the reference's own loss spread (|ref - other run|)."""
    spread = np.zeros_like(ref)
    for o in others:
        o = np.asarray(o)
        spread = np.maximum(spread, np.abs(o - ref[None]).max(axis=0) if o.ndim == 3 else np.abs(o - ref))
    return np.maximum.accumulate(spread, axis=0)


def check_sequence(test, mode, step_fn, flat_params_fn, ref_losses, envelope, single_rtol=1e-5, single_atol=2e-6, iters=10):
    """Drive `step_fn()` -> 4 losses for `iters` iterations.  Loss bound: |got - ref| <= single-step tolerance +
    c_loss x envelope (c_loss from margins.json).  Parameter bounds after iterations 1 and 10: see param_deviation."""
    for it in range(1, iters + 1):
        got = np.asarray(step_fn(), np.float64)
        row = ref_losses[it - 1]
        excess = np.abs(got - row) - (single_rtol * np.abs(row) + single_atol)
        ratio = float(np.max(excess / np.maximum(envelope[it - 1], 1e-12)))
        MARGINS.check(test, "loss_env", max(ratio, 0.0), "(iteration %d: got %s, reference %s)" % (it, got, row))
        if it in (1, 10):
            worst = param_deviation(mode, it, flat_params_fn())
            for k, (v, name) in worst.items():
                MARGINS.check(test, "param_%s_it%d" % (k, it), v, "(%s)" % name)
