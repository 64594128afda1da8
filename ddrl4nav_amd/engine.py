"""HotPath: owner of the device arenas and the libddrl_hip context.

PyTorch is used here for device memory, streams and (optionally) torch.distributed only; all
arithmetic on the path runs in the HIP kernels behind the C ABI (include/ddrl.h).
"""
from ctypes import byref, c_float, c_int32, c_int64, c_void_p, create_string_buffer

import numpy as np
import torch

from . import _lib
from ._lib import STATS_FLOATS, Config, check


def _ptr(t):
    return c_void_p(0) if t is None else c_void_p(t.data_ptr())


def _stream():
    return c_void_p(torch.cuda.current_stream().cuda_stream)


class HotPath:
    """One per GPU.  Holds params / grads / Adam state as flat fp32 tensors in the reference's
    named_parameters() order (reference nn/base.py:60-66) and the kernel workspace."""

    def __init__(self, max_batch, device=None, n_actions=6, in_channels=4, process_group=None, **cfg_overrides):
        if not torch.cuda.is_available():
            raise _lib.DdrlError("ddrl4nav_amd needs a ROCm GPU (torch.cuda.is_available() is False); "
                                 "there is no CPU fallback")
        self.lib = _lib.load()
        self.device = torch.device(device if device is not None else "cuda:%d" % torch.cuda.current_device())
        self.cfg = _lib.default_config(max_batch=int(max_batch), n_actions=int(n_actions),
                                       in_channels=int(in_channels), **cfg_overrides)
        n, na, wb = c_int64(), c_int64(), c_int64()
        check(self.lib.ddrl_param_count(byref(self.cfg), byref(n), byref(na)))
        check(self.lib.ddrl_workspace_bytes(byref(self.cfg), byref(wb)))
        self.n_params, self.n_actor, self.workspace_bytes = n.value, na.value, wb.value
        with torch.cuda.device(self.device):
            self.params = torch.zeros(self.n_params, dtype=torch.float32, device=self.device)
            self.grads = torch.zeros(self.n_params + STATS_FLOATS, dtype=torch.float32, device=self.device)
            self.adam_m = torch.zeros(self.n_params, dtype=torch.float32, device=self.device)
            self.adam_v = torch.zeros(self.n_params, dtype=torch.float32, device=self.device)
            self.workspace = torch.empty(self.workspace_bytes, dtype=torch.uint8, device=self.device)
        ctx = c_void_p()
        check(self.lib.ddrl_ctx_create(byref(self.cfg), _ptr(self.params), _ptr(self.grads), _ptr(self.adam_m),
                                       _ptr(self.adam_v), _ptr(self.workspace), self.workspace_bytes, byref(ctx)))
        self.ctx = ctx
        self.process_group = process_group
        self.n_actions = int(n_actions)
        self.max_batch = int(max_batch)
        self._ar_events = None  # (start, stop) event pairs around the gradient all-reduce while time_allreduce(True)
        self.comm = None        # dist.RcclComm: the C-ABI all-reduce (ddrl_grad_allreduce) instead of torch.distributed
        import os
        import torch.distributed as tdist
        # DDRL_ALLREDUCE = comma-separated options of the gradient all-reduce: "rccl" (the C-ABI communicator instead of
        # torch.distributed), "overlap" (layer buckets on a second stream under the backward)
        ar_opts = {o.strip() for o in os.environ.get("DDRL_ALLREDUCE", "").split(",") if o.strip()}
        if "rccl" in ar_opts and tdist.is_available() and tdist.is_initialized() \
                and tdist.get_world_size(process_group) > 1:
            from .dist import RcclComm
            world = tdist.get_world_size(process_group)
            # one RCCL rank per device: ranks that share a GPU (the gloo rehearsal of single-GPU boxes) would hang or fail inside
            # ncclCommInitRank on duplicate devices -- refuse before creating the communicator
            if tdist.get_backend(process_group) == "gloo" or world > torch.cuda.device_count():
                raise _lib.DdrlError("DDRL_ALLREDUCE=rccl needs one GPU per rank (world %d, %d device(s), torch backend %s); "
                                     "ranks sharing a device reduce through torch.distributed" % (
                                         world, torch.cuda.device_count(), tdist.get_backend(process_group)))
            self.comm = RcclComm(tdist.get_rank(process_group), world, group=process_group)
        # layer-bucketed all-reduce that overlaps the backward: RCCL ranks only (gloo stages through the host).
        # OFF by default (DDRL_ALLREDUCE=overlap, or "rccl,overlap", turns it on): it has never run with two real RCCL ranks, and what it can hide is
        # one 13.5 MB all-reduce per 24 ms iteration; the flat reduction on the compute stream is the default until a multi-GPU
        # run has shown bit-identity of the two.
        self._overlap = False
        self._comm_stream = self._comm_done = None
        self._buckets = []
        if tdist.is_available() and tdist.is_initialized() and tdist.get_world_size(process_group) > 1 \
                and "overlap" in ar_opts \
                and (self.comm is not None or tdist.get_backend(process_group) == "nccl"):
            self.enable_overlap()

    def close(self):
        if getattr(self, "comm", None) is not None:
            self.comm.close()
            self.comm = None
        if getattr(self, "ctx", None):
            self.lib.ddrl_ctx_destroy(self.ctx)
            self.ctx = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ---- parameters -------------------------------------------------------------------------
    def set_params(self, flat):
        """Copy a flat float32 array/tensor (reference blob order) into the arena."""
        t = torch.as_tensor(flat, dtype=torch.float32).reshape(-1)
        if t.numel() != self.n_params:
            raise ValueError("expected %d parameters, got %d" % (self.n_params, t.numel()))
        self.params.copy_(t, non_blocking=False)
        self.params_changed()

    def params_changed(self):
        check(self.lib.ddrl_params_changed(self.ctx))

    def reset_optimizer(self):
        self.adam_m.zero_()
        self.adam_v.zero_()
        check(self.lib.ddrl_set_step(self.ctx, 0))

    @property
    def step(self):
        s = c_int64()
        check(self.lib.ddrl_get_step(self.ctx, byref(s)))
        return s.value

    # ---- forward ----------------------------------------------------------------------------
    def forward(self, frames, act=None, seed=0, stream_id=0, probs=None, value=None, action=None, logp=None):
        """frames uint8 [n,4,84,84] on device.  Returns (probs [n,A], value [n], action [n], logp [n])."""
        assert frames.dtype == torch.uint8 and frames.is_cuda and frames.is_contiguous()
        n = frames.shape[0]
        dev = frames.device
        probs = torch.empty((n, self.n_actions), dtype=torch.float32, device=dev) if probs is None else probs
        value = torch.empty(n, dtype=torch.float32, device=dev) if value is None else value
        logp = torch.empty(n, dtype=torch.float32, device=dev) if logp is None else logp
        if act is None:
            action = torch.empty(n, dtype=torch.float32, device=dev) if action is None else action
        else:
            assert act.dtype == torch.float32 and act.is_contiguous() and act.numel() == n
            action = act
        check(self.lib.ddrl_forward(self.ctx, _ptr(frames), n, _ptr(act), int(seed) & (2 ** 64 - 1),
                                    int(stream_id) & (2 ** 64 - 1), _ptr(probs), _ptr(value),
                                    _ptr(action) if act is None else c_void_p(0), _ptr(logp), _stream()))
        return probs, value, action, logp

    def categorical_stats(self, probs):
        n, A = probs.shape
        p_hat = torch.empty_like(probs)
        logits = torch.empty_like(probs)
        ent = torch.empty(n, dtype=torch.float32, device=probs.device)
        check(self.lib.ddrl_categorical_stats(_ptr(probs), n, A, _ptr(p_hat), _ptr(logits), _ptr(ent), _stream()))
        return p_hat, logits, ent

    def categorical_sample(self, probs, seed, stream_id):
        n, A = probs.shape
        action = torch.empty(n, dtype=torch.float32, device=probs.device)
        logp = torch.empty(n, dtype=torch.float32, device=probs.device)
        check(self.lib.ddrl_categorical_sample(_ptr(probs), n, A, int(seed) & (2 ** 64 - 1),
                                               int(stream_id) & (2 ** 64 - 1), _ptr(action), _ptr(logp), _stream()))
        return action, logp

    def last_features(self, n):
        ha = torch.empty((n, 512), dtype=torch.float32, device=self.device)
        hc = torch.empty((n, 512), dtype=torch.float32, device=self.device)
        check(self.lib.ddrl_last_features(self.ctx, n, _ptr(ha), _ptr(hc), _stream()))
        return ha, hc

    # ---- GAE --------------------------------------------------------------------------------
    def gae(self, values, rewards, dones, gamma=0.99, landa=0.95, adv=None, ret=None):
        """values [T+1,N] f32, rewards [T,N] f32, dones [T,N] u8 (device) -> adv, ret [T,N]."""
        T, N = rewards.shape
        assert values.shape == (T + 1, N) and dones.shape == (T, N)
        assert values.dtype == torch.float32 and rewards.dtype == torch.float32 and dones.dtype == torch.uint8
        assert values.is_contiguous() and rewards.is_contiguous() and dones.is_contiguous()
        adv = torch.empty((T, N), dtype=torch.float32, device=values.device) if adv is None else adv
        ret = torch.empty((T, N), dtype=torch.float32, device=values.device) if ret is None else ret
        check(self.lib.ddrl_gae(_ptr(values), _ptr(rewards), _ptr(dones), T, N, float(np.float32(gamma)),
                                float(np.float32(landa)), _ptr(adv), _ptr(ret), _stream()))
        return adv, ret

    # ---- learner ----------------------------------------------------------------------------
    def ppo_iter(self, frames, actions, old_logps, advs, rets, b_global=None):
        B = frames.shape[0]
        for t in (actions, old_logps, advs, rets):
            assert t.dtype == torch.float32 and t.is_contiguous() and t.numel() == B
        assert frames.dtype == torch.uint8 and frames.is_contiguous()
        check(self.lib.ddrl_ppo_iter(self.ctx, _ptr(frames), _ptr(actions), _ptr(old_logps), _ptr(advs), _ptr(rets),
                                     B, int(b_global if b_global is not None else B), _stream()))

    def enable_overlap(self):
        """Record the per-layer bucket events in every ppo_iter and reduce bucket by bucket on a second stream."""
        from ctypes import c_int32
        check(self.lib.ddrl_grad_buckets_enable(self.ctx))
        n = c_int32()
        check(self.lib.ddrl_grad_bucket_count(self.ctx, byref(n)))
        self._buckets = []
        for b in range(n.value):
            off, cnt, nr = (c_int64 * 2)(), (c_int64 * 2)(), c_int32()
            check(self.lib.ddrl_grad_bucket_info(self.ctx, b, off, cnt, byref(nr)))
            self._buckets.append([(int(off[r]), int(cnt[r])) for r in range(nr.value)])
        self._comm_stream = torch.cuda.Stream(device=self.device)
        self._comm_done = torch.cuda.Event()
        self._overlap = True

    def grad_buckets(self):
        """[[(offset, count), ...] per bucket] in completion order of the backward (after enable_overlap)."""
        return [list(b) for b in self._buckets]

    def _allreduce_overlapped(self):
        """Bucket b's ranges are reduced on the communication stream as soon as the compute stream has passed the event the
        backward recorded for it; the compute stream then waits for the last one (only the small conv2 bucket is exposed)."""
        cs = self._comm_stream
        if self.comm is not None:
            check(self.lib.ddrl_grad_allreduce_overlapped(self.ctx, self.comm.h, c_void_p(cs.cuda_stream), _stream()))
            return
        from .dist import allreduce_flat
        # ordering against whatever the compute stream holds at this call; stale bucket events (no ppo_iter since the last
        # reduction) make the communication stream wait for the compute stream up front (no race, no overlap)
        fresh = c_int32()
        check(self.lib.ddrl_grad_buckets_begin(self.ctx, c_void_p(cs.cuda_stream), _stream(), byref(fresh)))
        last = len(self._buckets) - 1
        for b, ranges in enumerate(self._buckets):
            if fresh.value:
                check(self.lib.ddrl_grad_bucket_wait(self.ctx, b, c_void_p(cs.cuda_stream)))
            if b == last:
                check(self.lib.ddrl_grad_bucket_wait_last(self.ctx, c_void_p(cs.cuda_stream)))
            with torch.cuda.stream(cs):
                for off, cnt in ranges:
                    # RCCL ranks: in place on the device; ranks that share a GPU (gloo rehearsal): staged through the host per bucket
                    allreduce_flat(self.grads[off:off + cnt], self.process_group)
        self._comm_done.record(cs)
        torch.cuda.current_stream().wait_event(self._comm_done)

    def allreduce_grads(self):
        """The SUM all-reduce of the flat gradient arena + loss tail of one PPO iteration; gradients were
        pre-scaled by 1/B_global.  RCCL ranks reduce in layer buckets that overlap the rest of the backward; the timed span
        (time_allreduce) is then what the compute stream actually waits for."""
        from .dist import allreduce_flat

        def run():
            if self._overlap:
                self._allreduce_overlapped()
            elif self.comm is not None:
                check(self.lib.ddrl_grad_allreduce(self.ctx, self.comm.h, _stream()))
            else:
                allreduce_flat(self.grads, self.process_group)

        if self._ar_events is None:
            run()
            return
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        run()
        e1.record()
        self._ar_events.append((e0, e1))

    def time_allreduce(self, on=True):
        """Bracket every gradient all-reduce with events on the compute stream (what the learner waits for)."""
        self._ar_events = [] if on else None

    def allreduce_ms(self):
        """Durations (ms) of the all-reduces since time_allreduce(True); synchronises."""
        torch.cuda.synchronize()
        out = [a.elapsed_time(b) for a, b in (self._ar_events or [])]
        if self._ar_events is not None:
            self._ar_events = []
        return out

    def clip_adam_step(self):
        check(self.lib.ddrl_clip_adam_step(self.ctx, _stream()))

    def stats_async(self, out_row):
        """Enqueue the copy of the 8-float statistics tail into `out_row` (a pinned host tensor row); no sync.  The values are
        those of stats() once the stream has passed this point."""
        out_row.copy_(self.grads[self.n_params:self.n_params + STATS_FLOATS], non_blocking=True)

    @staticmethod
    def stats_dict(row):
        s = row.numpy() if hasattr(row, "numpy") else row
        return {"ActorLoss": float(s[0]), "VLoss": float(s[1]), "EntLoss": float(s[2]), "PpoTotalLoss": float(s[3]),
                "GradNorm": float(s[4]), "ClipCoef": float(s[5])}

    def stats(self):
        """Host copy of (actor_loss, v_loss, entropy, total, grad_norm, clip_coef) -- one sync."""
        s = self.grads[self.n_params:self.n_params + 6].cpu().numpy()
        return {"ActorLoss": float(s[0]), "VLoss": float(s[1]), "EntLoss": float(s[2]), "PpoTotalLoss": float(s[3]),
                "GradNorm": float(s[4]), "ClipCoef": float(s[5])}

    # ---- diagnostics ------------------------------------------------------------------------
    def keep_activations(self, on=True):
        """forward() of at most 512 samples keeps a1 / a2 on chip (csrc/act.hip); on=True makes it store them as well, for
        debug_buffer(0 / 1) after an acting forward.  Outputs are bit-identical either way."""
        check(self.lib.ddrl_debug_keep_activations(self.ctx, 1 if on else 0))
        return self

    def debug_buffer(self, which, shape_per_sample, n, enc, raw=False):
        """Copy of a workspace tensor: [n, *shape_per_sample] for encoder `enc`.  The gradient tensors of the data-gradient
        chain (4 dz1, 5 dz2, 6 dz3, 7 dh) are stored NORMALISED per sample (csrc/common.h Workspace::gsc); unless `raw`
        they are returned multiplied by their per-sample power-of-two scale, i.e. as the true gradients."""
        p, es = c_void_p(), c_int64()
        check(self.lib.ddrl_debug_buffer(self.ctx, which, byref(p), byref(es)))
        count = int(np.prod(shape_per_sample)) * n
        off = (p.value - self.workspace.data_ptr()) // 4 + enc * es.value
        flat = self.workspace.view(torch.float32)[off:off + count]
        out = flat.clone().reshape((n,) + tuple(shape_per_sample))
        if which in (4, 5, 6, 7) and not raw:
            g = self.debug_buffer(13, (), n, enc)
            out = out * g.reshape((n,) + (1,) * len(tuple(shape_per_sample)))
        return out

    def debug_view(self, which, shape_per_sample, n, enc):
        """WRITABLE view (no copy, no rescaling) of a workspace tensor -- tests use it to hand the encoder-only entry points
        a crafted dh (ddrl_encoder_backward reads the context's dh buffer)."""
        p, es = c_void_p(), c_int64()
        check(self.lib.ddrl_debug_buffer(self.ctx, which, byref(p), byref(es)))
        count = int(np.prod(shape_per_sample)) * n
        off = (p.value - self.workspace.data_ptr()) // 4 + enc * es.value
        return self.workspace.view(torch.float32)[off:off + count].view((n,) + tuple(shape_per_sample))

    def plane_maxima(self):
        """{slot name: [per encoder]} of the running maxima / bounds behind the fp16 plane scales (csrc/common.h AMAX_*)."""
        names = ("wl", "w2", "w3", "w1", "a1", "a2", "a3", "dh", "dz3", "dz2", "dz1", "gmax")
        v = self.debug_buffer(14, (), 2 * len(names), 0).cpu().numpy()
        return {k: v[2 * i:2 * i + 2].copy() for i, k in enumerate(names)}

    def u8_table(self):
        out = torch.empty(256, dtype=torch.float32, device=self.device)
        check(self.lib.ddrl_u8_table(_ptr(out), _stream()))
        return out

    def profile(self, on=True, acting=True):
        """Per-kernel HIP-event timing; acting=False leaves the ddrl_forward launches untimed (the event
        records around those short launches slow them by about a quarter)."""
        check(self.lib.ddrl_profile_enable(self.ctx, (1 if acting else 2) if on else 0))

    def profile_read(self):
        cap = 64
        names = create_string_buffer(48 * cap)
        ms = (c_float * cap)()
        calls = (c_int32 * cap)()
        n = c_int32()
        check(self.lib.ddrl_profile_read(self.ctx, names, ms, calls, cap, byref(n)))
        out = {}
        for i in range(min(n.value, cap)):
            nm = names.raw[48 * i:48 * (i + 1)].split(b"\0")[0].decode()
            out[nm] = (float(ms[i]), int(calls[i]))
        return out


class Timer:
    """HIP-event pair recorded on the caller's stream (bench.py roofline measurement)."""

    def __init__(self):
        self.lib = _lib.load()
        self.h = c_void_p()
        check(self.lib.ddrl_timer_create(byref(self.h)))

    def start(self):
        check(self.lib.ddrl_timer_start(self.h, _stream()))

    def stop(self):
        check(self.lib.ddrl_timer_stop(self.h, _stream()))

    def elapsed_ms(self):
        ms = c_float()
        check(self.lib.ddrl_timer_elapsed_ms(self.h, byref(ms)))
        return ms.value

    def __del__(self):
        try:
            self.lib.ddrl_timer_destroy(self.h)
        except Exception:
            pass
