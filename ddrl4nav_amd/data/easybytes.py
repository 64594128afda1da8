"""EasyBytes: wire codec of the reference's Redis messages, over the C codec in libddrl_hip.so
(csrc/easybytes.cpp, include/ddrl.h `ddrl_eb_*`).  Same class name, method names, argument
meaning and error behaviour as USTC_lab/data/easybytes.py:18-172; the byte streams are
bit-identical (tests/test_easybytes.py checks them against KATs produced by the reference).

Extra (not in the reference): ``frames_to_u8`` decodes the frames of a batched forward-states
item straight into a uint8 buffer -- a pinned ring slot -- instead of float64 numpy arrays."""
import ctypes
import marshal
from ctypes import byref, c_int32, c_int64, c_void_p
from typing import Dict, List, Tuple

import numpy as np

from ddrl4nav_amd import _lib
from ddrl4nav_amd._lib import EbArray, EbMsg

_CODES = {np.dtype(np.uint8): 1, np.dtype(np.float16): 2, np.dtype(np.float32): 3, np.dtype(np.float64): 4}
_DTYPES = {v: k for k, v in _CODES.items()}


def _buf(b):
    """(address, length, keep-alive) of a bytes-like object without copying."""
    mv = memoryview(b)
    arr = np.frombuffer(mv, dtype=np.uint8)
    return arr.ctypes.data, arr.size, arr


class EasyBytes:
    d = {1: (1, np.uint8), 2: (2, np.float16), 3: (4, np.float32), 4: (8, np.float64)}

    def __init__(self, machine="127.0.0.1"):
        parts = [int(t) for t in machine.split(".")]
        assert len(parts) == 4 and all(0 <= t < 256 for t in parts)
        self._ip = (c_int32 * 4)(*parts)
        self.lib = _lib.load()
        hdr = np.zeros(20, np.uint8)
        _lib.check(self.lib.ddrl_eb_forward_header(self._ip, 0, 0, c_void_p(hdr.ctypes.data)))
        self.machine_bytes = hdr[8:16].tobytes()

    # ---- arrays ---------------------------------------------------------------------------------
    def _code(self, dtype):
        code = _CODES.get(np.dtype(dtype))
        if code is None:
            print("EasyBytes: Match data type error !", dtype, flush=True)
            raise ValueError
        return code

    def encode_data(self, np_list_data: List[np.ndarray]) -> bytes:
        chunks = []
        for a in np_list_data:
            a = np.ascontiguousarray(a)
            code = self._code(a.dtype)
            dims = (c_int64 * max(1, a.ndim))(*a.shape)
            need = c_int64()
            _lib.check(self.lib.ddrl_eb_array_bytes(code, a.ndim, dims, byref(need)))
            out = np.empty(need.value, np.uint8)
            wrote = c_int64()
            _lib.check(self.lib.ddrl_eb_encode_array(code, a.ndim, dims, c_void_p(a.ctypes.data), c_void_p(out.ctypes.data),
                                                     need.value, byref(wrote)))
            chunks.append(out.tobytes())
        return b"".join(chunks)

    def _scan(self, bytes_data):
        addr, n, keep = _buf(bytes_data)
        cap = 64
        arr = (EbArray * cap)()
        cnt = c_int32()
        st = self.lib.ddrl_eb_scan(c_void_p(addr), n, arr, cap, byref(cnt))
        if st == -2:  # unknown dtype code: KeyError in the reference's table lookup
            raise KeyError("EasyBytes: unknown dtype code")
        _lib.check(st)
        return [arr[i] for i in range(cnt.value)], keep

    def decode_data(self, bytes_data: bytes) -> List[np.ndarray]:
        recs, keep = self._scan(bytes_data)
        out = []
        for r in recs:
            shape = tuple(r.dims[i] for i in range(r.ndim))
            out.append(np.frombuffer(keep, dtype=_DTYPES[r.dtype], count=r.count, offset=r.data_offset).reshape(shape))
        return out

    # ---- forward states ---------------------------------------------------------------------------
    def encode_forward_states(self, process_env_id: int, list_np_states: List[np.ndarray]) -> bytes:
        payload = self.encode_data(list_np_states)
        hdr = np.empty(20, np.uint8)
        _lib.check(self.lib.ddrl_eb_forward_header(self._ip, int(process_env_id), len(payload), c_void_p(hdr.ctypes.data)))
        return hdr.tobytes() + payload

    def _scan_msgs(self, byte_states):
        addr, n, keep = _buf(byte_states)
        cap = 1024
        msgs = (EbMsg * cap)()
        cnt = c_int32()
        _lib.check(self.lib.ddrl_eb_scan_forward_states(c_void_p(addr), n, msgs, cap, byref(cnt)))
        return [msgs[i] for i in range(cnt.value)], keep

    def decode_forward_states(self, byte_states: bytes) -> Tuple[List[str], List[np.ndarray]]:
        msgs, keep = self._scan_msgs(byte_states)
        ids, per_env = [], []
        for m in msgs:
            ids.append(".".join(str(m.ip[i]) for i in range(4)) + "_" + str(m.process_env_id))
            per_env.append(self.decode_data(keep[m.payload_offset:m.payload_offset + m.payload_len]))
        states = [np.concatenate([e[i] for e in per_env], axis=0) for i in range(len(per_env[0]))]
        return ids, states

    def frames_to_u8(self, byte_states, out: np.ndarray, state_index: int = 0):
        """Frames of a batched forward-states item -> uint8 `out` (flat, e.g. a pinned ring slot).
        Returns (n_samples, elems_per_sample)."""
        addr, n, keep = _buf(byte_states)
        assert out.dtype == np.uint8 and out.flags["C_CONTIGUOUS"]
        ns, per = c_int64(), c_int64()
        _lib.check(self.lib.ddrl_eb_frames_to_u8(c_void_p(addr), n, int(state_index), c_void_p(out.ctypes.data), out.size,
                                                 byref(ns), byref(per)))
        return ns.value, per.value

    # ---- forward replies ----------------------------------------------------------------------------
    def encode_forward_return_data(self, list_forward_return_np_data, list_env_batch_num: List[int]) -> List[bytes]:
        data = []
        for x in list_forward_return_np_data:
            if hasattr(x, "detach"):
                x = x.detach().cpu().numpy()
            data.append(np.asarray(x))
        out, index = [], 0
        for m in list_env_batch_num:
            per_env = [a[:, index:index + m] if i == 2 else a[index:index + m] for i, a in enumerate(data)]
            out.append(self.encode_data(per_env))
            index += m
        return out

    # ---- backward blobs -----------------------------------------------------------------------------
    def encode_backward_data(self, list_np_data, dict_logger: Dict) -> bytes:
        states = self.encode_data(list_np_data[0])
        other4 = self.encode_data(list_np_data[1:])
        q = np.empty(8, np.uint8)
        parts = []
        for blob in (states, other4):
            _lib.check(self.lib.ddrl_eb_put_u64(len(blob), c_void_p(q.ctypes.data)))
            parts += [q.tobytes(), blob]
        return b"".join(parts) + marshal.dumps(dict_logger)

    def decode_backward_data(self, bytes_data: bytes):
        addr, n, keep = _buf(bytes_data)
        so, sl, oo, ol, to = (c_int64() for _ in range(5))
        _lib.check(self.lib.ddrl_eb_scan_backward(c_void_p(addr), n, byref(so), byref(sl), byref(oo), byref(ol), byref(to)))
        states = self.decode_data(keep[so.value:so.value + sl.value])
        other4 = self.decode_data(keep[oo.value:oo.value + ol.value])
        return states, other4, marshal.loads(bytes(keep[to.value:]))
