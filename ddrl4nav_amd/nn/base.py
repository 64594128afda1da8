"""Basenn / PreNet: weight shuttle and checkpoint surface of the reference's networks
(mirror of USTC_lab/nn/base.py:23-155 for the methods the Forward/Backward servers call).

Blob format of nn2redis / updatenn_by_redis (base.py:38-81): for every tensor in
named_parameters() order: big-endian uint32 ndim, ndim x uint32 dims, raw little-endian fp32."""
import struct
from typing import Tuple

import numpy as np
import torch


def _prod(shape):
    n = 1
    for d in shape:
        n *= int(d)
    return n


class PreNet(torch.nn.Module):
    def __init__(self):
        super().__init__()


class Basenn(torch.nn.Module):
    def __init__(self, config, config_nn):
        super().__init__()
        self.conn = self._connect_redis(getattr(config, "MIDDLE_REDIS_HOST", None), getattr(config, "MIDDLE_REDIS_PORT", None))
        self.pipe = self.conn.pipeline() if self.conn is not None else None
        self.model_key = getattr(config, "TASK_NAME", "ddrl") + getattr(config, "MODULE_KEY", "MODEL")
        self.device = getattr(config, "DEVICE", "cuda")
        self.model_dtype = config_nn.MODULE_NUMPY_DTYPE
        self.model_dtype_bytes = config_nn.MODULE_BITS // 8
        self.model_tensor_dtype = config_nn.MODULE_TENSOR_DTYPE

    # -- Redis is optional plumbing here: co-located Forward/Backward share the device arena ------
    def _connect_redis(self, host, port):
        try:
            import redis
        except ImportError:
            return None
        if host is None:
            return None
        return redis.Redis(host=host, port=port)

    def _encode_wb(self, wb_np: np.ndarray) -> bytes:
        shape = wb_np.shape
        return struct.pack(">I", len(shape)) + struct.pack(">%dI" % len(shape), *shape) + np.ascontiguousarray(wb_np).tobytes()

    def _decode_wb(self, wb_bytes: bytes) -> Tuple[torch.Tensor, int]:
        ndim = struct.unpack(">I", wb_bytes[:4])[0]
        shape = struct.unpack(">%dI" % ndim, wb_bytes[4:4 + 4 * ndim])
        count = _prod(shape)
        head = 4 + 4 * ndim
        arr = np.frombuffer(wb_bytes, dtype=self.model_dtype, offset=head, count=count).reshape(shape)
        return torch.tensor(arr, device=self.device), head + count * self.model_dtype_bytes

    def model_bytes(self) -> bytes:
        """The blob nn2redis publishes (one device->host copy of the flat arena, then slicing)."""
        return b"".join(self._encode_wb(v.detach().cpu().numpy()) for _, v in self.named_parameters())

    def nn2redis(self, pipe, update_key, key=None):
        pipe.set(key if key else self.model_key, self.model_bytes())
        pipe.incr(update_key)
        pipe.execute()

    def load_model_bytes(self, model_bytes: bytes):
        index, state = 0, {}
        for k, _ in self.named_parameters():
            t, used = self._decode_wb(model_bytes[index:])
            index += used
            state[k] = t
        self.load_state_dict(state, strict=False)

    def updatenn_by_redis(self, conn, key=None):
        self.load_model_bytes(conn.get(key if key else self.model_key))

    def updatenn_by_file(self, file_path: str):
        self.load_state_dict(torch.load(file_path, map_location=self.device))

    def updatenn(self, path: str, conn=None):
        if path.startswith("redis"):
            assert conn is not None
            self.updatenn_by_redis(conn, path.split("://")[-1])
        elif path.startswith("file"):
            self.updatenn_by_file(path.split("://")[-1])

    def init_weight(self):
        for m in self.modules():
            if isinstance(m, (torch.nn.Conv2d, torch.nn.Linear)):
                torch.nn.init.orthogonal_(m.weight, np.sqrt(2))
                m.bias.data.zero_()
        self.params_changed()

    def params_changed(self):
        """Parameters are views into the flat arena; the kernels read packed copies of it.  Anything
        that writes parameters in place (init_weight, ``p.data.copy_()``) must call this afterwards
        (load_state_dict and the optimiser step do it themselves).  PPO / GenericPPO override it."""

    def states_normalization(self, states):
        pass

    def imitation_learning(self, *args, **kwargs):
        raise NotImplementedError("imitation pre-training is outside the hot path (SURVEY.md section 2 row 5)")
