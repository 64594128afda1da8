"""Host -> device staging that cannot outlive its source, and does not block the host for the H2D copy either.

An asynchronous copy (``non_blocking=True``) out of PAGEABLE host memory keeps reading the caller's buffer after the
call returns; the callers of this package own their inputs (numpy arrays decoded from a message, temporaries) and may free
them at once -- the copy then reads freed pages: stale data, or a GPU memory-access fault once the pages are unmapped
(seen as a flaky micro-batching test and a process abort).  Sources whose lifetime the runtime itself guarantees -- device
tensors and pinned host tensors (the pinned ring, ``data/ring.py``) -- are copied asynchronously as they are.  A pageable
source goes through a PERSISTENT PINNED buffer (one per shape / dtype / device, kept by this module): the host memcpy into it
is synchronous (the caller's buffer is free again when the call returns), the H2D copy out of it is asynchronous on the
current stream, and an event per buffer makes the next use of the same buffer wait until that copy has left it.  Callers on
a hot loop should still hand over pinned tensors (or use the ring): that saves the host memcpy as well."""
import threading
from collections import OrderedDict

import torch

_MAX_BUFFERS = 64
_lock = threading.Lock()
_buffers = OrderedDict()  # (shape, dtype, device index) -> [pinned tensor, event or None]


def async_ok(t):
    """True when `t` (a tensor) may be the source of a non_blocking copy."""
    return bool(t.is_cuda or t.is_pinned())


def _staged(s, device):
    """Pinned copy of the pageable tensor `s` and the slot whose event must be recorded after the H2D copy was enqueued."""
    dev = torch.device(device)
    key = (tuple(s.shape), s.dtype, dev.index if dev.index is not None else torch.cuda.current_device())
    slot = _buffers.get(key)      # (the callers hold _lock from here until the H2D copy is enqueued and marked)
    if slot is None:
        slot = [torch.empty(s.shape, dtype=s.dtype, pin_memory=True), None]
        _buffers[key] = slot
        while len(_buffers) > _MAX_BUFFERS:
            old = _buffers.popitem(last=False)[1]
            if old[1] is not None:
                old[1].synchronize()
    else:
        _buffers.move_to_end(key)
    if slot[1] is not None:
        slot[1].synchronize()  # the previous H2D copy out of this buffer has finished (normally long ago)
    slot[0].copy_(s)            # synchronous host memcpy: the caller's buffer is free again after this line
    return slot


def _mark(slot, device):
    """Record the slot's event on the DESTINATION device's current stream -- the stream the H2D copy was enqueued on.  (The
    calling thread's current device may be another one when one process drives several GPUs: an event recorded there would not
    cover the copy, and the next fill of the pinned buffer could overwrite it under the DMA.)"""
    dev = torch.device(device)
    idx = dev.index if dev.index is not None else torch.cuda.current_device()
    with torch.cuda.device(idx):
        if slot[1] is None:
            slot[1] = torch.cuda.Event()
        slot[1].record(torch.cuda.current_stream(idx))


def to_device(x, device, dtype=None):
    """torch.as_tensor(x) on `device` (and `dtype`); never blocks on the H2D copy, never reads `x` after returning."""
    t = torch.as_tensor(x)
    if async_ok(t) or t.numel() == 0:
        return t.to(device, non_blocking=True) if dtype is None else t.to(device, dtype, non_blocking=True)
    with _lock:   # one user of a pinned buffer at a time: fill, enqueue the H2D copy, record its event
        slot = _staged(t, device)
        out = slot[0].to(device, non_blocking=True)
        _mark(slot, device)
    return out if dtype is None else out.to(dtype)


def copy_into(dst, src):
    """dst.copy_(src) for a device `dst`; never blocks on the H2D copy, never reads `src` after returning."""
    s = torch.as_tensor(src)
    if async_ok(s) or s.numel() == 0 or not dst.is_cuda:
        dst.copy_(s, non_blocking=async_ok(s))
        return dst
    if tuple(s.shape) != tuple(dst.shape):
        s = torch.broadcast_to(s, dst.shape)  # dst.copy_'s own broadcasting rule (raises on a mismatch)
    with _lock:
        slot = _staged(s, dst.device)
        dst.copy_(slot[0], non_blocking=True)
        _mark(slot, dst.device)
    return dst
