"""Host -> device staging that cannot outlive its source.

An asynchronous copy (``non_blocking=True``) out of PAGEABLE host memory keeps reading the caller's buffer after the
call returns; the callers of this package own their inputs (numpy arrays decoded from a message, temporaries) and may free
them at once -- the copy then reads freed pages: stale data, or a GPU memory-access fault once the pages are unmapped
(seen as a flaky micro-batching test and a process abort).  Only sources whose lifetime the runtime itself guarantees are
copied asynchronously: device tensors and pinned host tensors (the pinned ring, ``data/ring.py``)."""
import torch


def async_ok(t):
    """True when `t` (a tensor) may be the source of a non_blocking copy."""
    return bool(t.is_cuda or t.is_pinned())


def to_device(x, device, dtype=None):
    """torch.as_tensor(x) on `device` (and `dtype`), asynchronously only when that is safe."""
    t = torch.as_tensor(x)
    nb = async_ok(t)
    if dtype is None:
        return t.to(device, non_blocking=nb)
    return t.to(device, dtype, non_blocking=nb)


def copy_into(dst, src):
    """dst.copy_(src) for a device `dst`; asynchronous only when `src` is a device or pinned tensor."""
    s = torch.as_tensor(src)
    dst.copy_(s, non_blocking=async_ok(s))
    return dst
