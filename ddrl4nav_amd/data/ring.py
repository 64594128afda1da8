"""PinnedRing: pinned-host ring buffer feeding the device-resident experience pool.

Replaces the Redis LPUSH/BRPOP shuttle of frames between env workers and the learner
(USTC_lab/agent/multiqueue.py:83-130, USTC_lab/server/backward.py:145-151): producers write
uint8 frames straight into page-locked slots; the consumer issues hipMemcpyAsync on a copy
stream into the pool slot.  Thin ctypes wrapper over ddrl_ring_* (include/ddrl.h)."""
import ctypes
from ctypes import byref, c_int32, c_void_p

import numpy as np

from ddrl4nav_amd import _lib
from ddrl4nav_amd._lib import check


class PinnedRing:
    def __init__(self, slot_bytes, n_slots=4):
        self.lib = _lib.load()
        self.slot_bytes = int(slot_bytes)
        self.h = c_void_p()
        check(self.lib.ddrl_ring_create(self.slot_bytes, int(n_slots), byref(self.h)))

    def acquire(self, timeout_ms=1000):
        """Next free slot as a writable uint8 numpy view (producer side)."""
        p = c_void_p()
        check(self.lib.ddrl_ring_acquire(self.h, byref(p), int(timeout_ms)))
        buf = (ctypes.c_uint8 * self.slot_bytes).from_address(p.value)
        return np.frombuffer(buf, dtype=np.uint8)

    def commit(self):
        check(self.lib.ddrl_ring_commit(self.h))

    def pop_to(self, dst, stream=None, timeout_ms=1000):
        """hipMemcpyAsync the oldest committed slot into the device tensor `dst` on `stream` (default: the current one).  With a stream of
        the caller's own the ordering against other streams is the caller's too: work already enqueued elsewhere on `dst` (its
        allocation's fill, a kernel still reading it) must be ordered in front of the copy (`stream.wait_stream(...)`), and consumers
        behind it -- what agent/rollout.py:put_frames_from_ring does."""
        import torch
        s = stream if stream is not None else torch.cuda.current_stream()
        nbytes = dst.numel() * dst.element_size()
        check(self.lib.ddrl_ring_pop_to_device(self.h, c_void_p(dst.data_ptr()), nbytes, c_void_p(s.cuda_stream),
                                               int(timeout_ms)))

    def pending(self):
        n = c_int32()
        check(self.lib.ddrl_ring_pending(self.h, byref(n)))
        return n.value

    def close(self):
        if self.h:
            self.lib.ddrl_ring_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
