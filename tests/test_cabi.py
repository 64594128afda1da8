"""CPU-side checks of the drop-in boundary: the C-ABI library loads and exports every symbol
that include/ddrl.h declares (no compute calls without a GPU)."""
import ctypes
import os
import re

import pytest

from ddrl4nav_amd import _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    src = open(os.path.join(ROOT, "include", "ddrl.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(ddrl_[a-z0-9_]+)\s*\(", src)))


def test_library_built():
    assert os.path.exists(_lib.LIB_PATH), "run __graft_entry__.build() first"


def test_every_declared_symbol_is_exported():
    lib = ctypes.CDLL(_lib.LIB_PATH)
    names = _declared()
    assert len(names) >= 30
    for n in names:
        assert hasattr(lib, n), "symbol %s declared in include/ddrl.h is not exported" % n
    # and the binding table covers the header
    assert set(names) == set(_lib.SIGNATURES.keys())


def test_load_and_host_only_calls():
    lib = _lib.load()
    assert lib.ddrl_abi_version() == 1
    assert lib.ddrl_status_string(-3) == b"workspace too small"
    cfg = _lib.default_config(max_batch=64)
    assert (cfg.n_actions, cfg.in_channels) == (6, 4)
    assert abs(cfg.actor_lr - 5e-5) < 1e-10 and abs(cfg.critic_lr - 1e-3) < 1e-9
    n, na = ctypes.c_int64(), ctypes.c_int64()
    _lib.check(lib.ddrl_param_count(ctypes.byref(cfg), ctypes.byref(n), ctypes.byref(na)))
    assert n.value == 3371847 and na.value == 1687206
    wb = ctypes.c_int64()
    _lib.check(lib.ddrl_workspace_bytes(ctypes.byref(cfg), ctypes.byref(wb)))
    assert wb.value > 64 * 86528 * 4


def test_unsupported_configs_fail_loudly():
    lib = _lib.load()
    wb = ctypes.c_int64()
    for bad in (dict(n_actions=19), dict(in_channels=3)):
        cfg = _lib.default_config(max_batch=64, **bad)
        assert lib.ddrl_workspace_bytes(ctypes.byref(cfg), ctypes.byref(wb)) == -2
    cfg = _lib.default_config(max_batch=64, share_cnn_net=2)
    assert lib.ddrl_workspace_bytes(ctypes.byref(cfg), ctypes.byref(wb)) == -1
    with pytest.raises(_lib.DdrlError):
        _lib.check(-2)


def test_shared_prenet_layout():
    """SHARE_CNN_NET=True: prenet + two linear heads, one Adam group (reference ppo.py:39)."""
    lib = _lib.load()
    cfg = _lib.default_config(max_batch=64, share_cnn_net=1)
    assert abs(cfg.learning_rate - 2e-4) < 1e-9 and cfg.smooth_l1_loss == 0
    n, na = ctypes.c_int64(), ctypes.c_int64()
    _lib.check(lib.ddrl_param_count(ctypes.byref(cfg), ctypes.byref(n), ctypes.byref(na)))
    assert n.value == 1684128 + 6 * 512 + 6 + 512 + 1 and na.value == n.value


def test_no_gpu_is_an_error_not_a_fallback():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from ddrl4nav_amd.engine import HotPath
    with pytest.raises(_lib.DdrlError):
        HotPath(max_batch=8)
