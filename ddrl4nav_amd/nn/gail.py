"""Discriminator / GAIL: drop-in for USTC_lab.nn.GAIL (reference nn/GAIL.py:19-160) on the HIP operators.

Same constructors, module trees (``mlp_layer`` before ``pre``; ``generator``, ``discriminator``, ``actor`` alias,
``gail_critic``), ``forward`` routing (a 2-tuple ``(states, actions)`` goes to the discriminator, anything else to the
generator) and ``learn`` generator protocol (discriminator items with ``last=False``, then the generator's PPO items
with ``last=True``).  The arithmetic -- encoder, dense layers, WGAN loss terms, grad-norm clip + RMSprop, the PPO update
with the extra value head -- runs in the kernels behind include/ddrl.h.

What the reference leaves open, and what this file does about it (see also tests/golden/make_golden_gail.py):
  * ``config.GAN_D_MLP_LIST`` / ``config.ACTIONS_DIM`` are read (GAIL.py:23,47) but defined by no config class:
    BaseConfig here supplies ``GAN_D_MLP_LIST = [(512 + ACTIONS_DIM, 256, "relu"), (256, 1, None)]`` and copies
    ``ACTIONS_DIM`` from ConfigNN; both can be overridden.
  * the expert batches: ``expert_data=`` (any iterable of ``(states, actions)`` batches) takes the place of the
    DataLoader over ``MIMIC_START_LOAD_PATH``; without it the directory is read with data.mimic_exp (same file format).
    A batch's states may be a list of arrays (what every PreNet indexes) or one array.
  * the GAIL critic is in no optimiser in the reference (PPO.add_critic appends to a plain list after the Adam was built):
    it is never trained.  Kept as is -- identical results are the contract.
"""
import time
from ctypes import byref, c_int64

import numpy as np
import torch

from ddrl4nav_amd import _lib
from ddrl4nav_amd._lib import STATS_FLOATS, check
from ddrl4nav_amd.data import Experience
from ddrl4nav_amd.nn.base import Basenn
from ddrl4nav_amd.nn.generic import FEAT, _pad4, dense_layer, mlp
from ddrl4nav_amd.ops import _p, _st
from ddrl4nav_amd.utils.staging import to_device


class Discriminator(Basenn):
    def __init__(self, **kwargs):
        config, config_nn = kwargs['config'], kwargs['config_nn']
        super().__init__(config, config_nn)
        if not torch.cuda.is_available():
            raise _lib.DdrlError("ddrl4nav_amd needs a ROCm GPU; there is no CPU fallback")
        pre = kwargs['pre']
        if pre is None:
            raise ValueError("the discriminator needs an encoder: the reference passes deepcopy(prenet), which exists only with "
                             "SHARE_CNN_NET=True (runner/utils.py:164); without it GAIL.py:68 concatenates a list with a tensor")
        self.lib = _lib.load()
        self.mlp_layer = mlp(config.GAN_D_MLP_LIST)      # registered BEFORE pre (GAIL.py:26-27): parameter order
        self.pre = pre
        self.device = torch.device(config.DEVICE if str(config.DEVICE) != "cuda" else "cuda:%d" % torch.cuda.current_device())
        self.dtype = config_nn.MODULE_TENSOR_DTYPE
        self.config, self.config_nn = config, config_nn
        # WGAN recommends RMSprop (GAIL.py:28-30): RMSprop(lr, alpha=0.9) + StepLR(step_size=250, gamma=0.95)
        self.lr = float(config_nn.GAN_D_LEARNING_RATE)
        self.alpha, self.eps = 0.9, 1e-8
        self.decay_step, self.decay_gamma = 250, 0.95
        self.epochs = config_nn.GAN_D_EPOCH
        self.accumulation_steps = config_nn.GRAD_ACCUMULATION_STEP
        self.WGAN_clip_grad_num = float(config_nn.WGAN_CLIP_GRAD_NUM)
        self.update_time = 0
        self.action_dim = int(config.ACTIONS_DIM)
        self.cap = int(kwargs.get('max_batch') or 4096)
        # data-parallel ranks (one per GPU): every rank holds a shard of the policy batch and its own expert batches; the two WGAN
        # means run over the UNION of the shards (1 / n_total inside ddrl_op_wgan_terms) and the flat gradient + loss are summed
        # across ranks before the clip, exactly as the generator's (BASELINE config 5 is an 8-GPU config)
        self._process_group = kwargs.get('process_group')
        self.expert_data = kwargs.get('expert_data')
        if self.expert_data is None:
            self.expert_data = self._get_data(config.MIMIC_START_LOAD_PATH, config.TASK_TYPE, config_nn.GAN_D_BATCH_SIZE)
        self._bind_arena()
        self._build()

    def _get_data(self, data_path, task_type, batch_size):
        from ddrl4nav_amd.data.mimic_exp import MimicExpFactory, batches
        return batches(MimicExpFactory().mimic_reader(task_type, data_path), batch_size)

    # ---- flat arenas: [mlp_layer.*][pad to 16 floats][pre.*], gradients beside them ------------------------------------
    def _bind_arena(self):
        params = list(self.named_parameters())
        n_mlp = sum(p.numel() for k, p in params if k.startswith("mlp_layer."))
        pad = (-n_mlp) % 16
        total = sum(p.numel() for _, p in params) + pad
        f = dict(dtype=torch.float32, device=self.device)
        self.n_params = total
        self.params = torch.zeros(total, **f)
        self.grads = torch.zeros(total + STATS_FLOATS, **f)
        self.gtmp = torch.zeros(total + STATS_FLOATS, **f)
        self.square_avg = torch.zeros(total, **f)
        off, seen_pre = 0, False
        with torch.no_grad():
            for name, p in params:
                if name.startswith("pre.") and not seen_pre:
                    off, seen_pre = off + pad, True
                n = p.numel()
                view = self.params[off:off + n].view(p.shape)
                view.copy_(p.detach().to(self.device, torch.float32))
                p.data = view
                p.requires_grad_(False)
                p.grad_view = self.gtmp[off:off + n].view(p.shape)
                off += n
        assert off == total or not seen_pre

    def _build(self):
        f = dict(dtype=torch.float32, device=self.device)
        self.pre.build(self.cap, self.device)
        if self.pre.h.shape[1] != FEAT:
            raise NotImplementedError("the discriminator expects %d-wide encoder features" % FEAT)
        mods = [m for m in self.mlp_layer if isinstance(m, torch.nn.Linear)]
        relus = []
        for i, m in enumerate(self.mlp_layer):
            if isinstance(m, torch.nn.Linear):
                relus.append(i + 1 < len(self.mlp_layer) and isinstance(self.mlp_layer[i + 1], torch.nn.ReLU))
        if mods[0].in_features != FEAT + self.action_dim or mods[-1].out_features != 1:
            raise ValueError("GAN_D_MLP_LIST must map %d (features + action) to 1 score" % (FEAT + self.action_dim))
        self._layers = [dense_layer(m, r, self.cap, self.device) for m, r in zip(mods, relus)]
        self._ld = [_pad4(mods[0].in_features)] + [l.N for l in self._layers]
        self._act = [torch.zeros((self.cap, ld), **f) for ld in self._ld]      # cat(features, action), then every layer's output
        self._dact = [torch.zeros((self.cap, ld), **f) for ld in self._ld]
        self._dh = getattr(self.pre, "_dh", None)
        if self._dh is None:
            self._dh = torch.empty((self.cap, FEAT), **f)
        self._loss = torch.zeros(1, **f)
        cb = c_int64()
        check(self.lib.ddrl_op_clip_adam_ws_bytes(byref(cb)))
        self._opt_ws = torch.empty(cb.value, dtype=torch.uint8, device=self.device)
        self._raw_u8 = bool(getattr(self.pre, "raw_u8", False))
        self._dirty = True

    def params_changed(self):
        self._dirty = True

    def load_state_dict(self, state_dict, strict=True, **kw):
        out = super().load_state_dict(state_dict, strict=strict, **kw)
        self._dirty = True
        return out

    def to(self, *args, **kwargs):
        return self

    def _ensure_packed(self):
        if self._dirty:
            self.pre.pack()
            for l in self._layers:
                l.pack()
            self._dirty = False

    # ---- D(s, a) (GAIL.py:63-71) -----------------------------------------------------------------------------------------
    def _stage(self, states, lo, hi):
        states = states if isinstance(states, (list, tuple)) else [states]
        if self._raw_u8:
            return [torch.as_tensor(s)[lo:hi] for s in states]
        return [to_device(torch.as_tensor(s)[lo:hi], self.device, torch.float32) for s in states]

    def _forward_chunk(self, st, action, n):
        """scores of one micro-batch -> self._act[-1][:n, 0]; activations stay in place for the backward."""
        h = self.pre.forward_dev(st, n)
        cat = self._act[0]
        cat[:n, :FEAT].copy_(h[:n])                                           # merge(state, action) = torch.cat(dim=-1)
        cat[:n, FEAT:FEAT + self.action_dim].copy_(action.reshape(n, self.action_dim))
        for i, l in enumerate(self._layers):
            l.forward(self._act[i], self._ld[i], self._act[i + 1], self._ld[i + 1], n)
        return self._act[-1]

    def _backward_chunk(self, n):
        """d(loss)/d(score) sits in self._dact[-1]; parameter gradients land in self.gtmp (overwritten)."""
        for i in reversed(range(len(self._layers))):
            l = self._layers[i]
            # the gradient that reaches layer i's INPUT is masked by that input's ReLU (the previous layer's activation);
            # layer 0's input is cat(features, action): no activation of this module in between
            mask = self._act[i] if i > 0 and self._layers[i - 1].relu else None
            l.backward(self._act[i], self._ld[i], self._dact[i + 1], self._ld[i + 1], n, din=self._dact[i], ld_din=self._ld[i],
                       mask_src=mask, ld_mask=self._ld[i])
        self._dh[:n].copy_(self._dact[0][:n, :FEAT])
        self.pre.backward_dev(self._dh, n)

    def forward(self, x):
        """x = (states, actions [n, ACTIONS_DIM]) -> scores [n, 1]."""
        states, action = x
        states = states if isinstance(states, (list, tuple)) else [states]
        n = int(torch.as_tensor(states[0]).shape[0])
        self._ensure_packed()
        action = torch.as_tensor(action, dtype=torch.float32, device=self.device).reshape(n, self.action_dim)
        out = torch.empty((n, 1), dtype=torch.float32, device=self.device)
        for lo in range(0, n, self.cap):
            hi = min(n, lo + self.cap)
            score = self._forward_chunk(self._stage(states, lo, hi), action[lo:hi], hi - lo)
            out[lo:hi, 0].copy_(score[:hi - lo, 0])
        return out

    # ---- one WGAN step (GAIL.py:73-94) ------------------------------------------------------------------------------------
    def _pass(self, states, action, sign, first, n_total=None):
        """forward + backward of one term  sign * mean(D(states, action)); gradients are accumulated into self.grads.
        n_total: the size of the whole batch the mean runs over (default: the sum of the ranks' shard sizes)."""
        from ddrl4nav_amd.dist import global_batch
        states = states if isinstance(states, (list, tuple)) else [states]
        n = int(torch.as_tensor(states[0]).shape[0])
        if n_total is None:
            n_total = global_batch(n, self._process_group)   # one collective per term: every rank runs the same two terms per step
        action = torch.as_tensor(action, dtype=torch.float32, device=self.device).reshape(n, self.action_dim)
        total = self.n_params + STATS_FLOATS
        for lo in range(0, n, self.cap):
            hi = min(n, lo + self.cap)
            score = self._forward_chunk(self._stage(states, lo, hi), action[lo:hi], hi - lo)
            check(self.lib.ddrl_op_wgan_terms(_p(score), self._ld[-1], hi - lo, n_total, float(sign), _p(self._dact[-1]), self._ld[-1],
                                              self._ld[-1], _p(self._loss), 0 if first else 1, _st()))
            self._backward_chunk(hi - lo)
            if first:
                self.grads.copy_(self.gtmp)
            else:
                check(self.lib.ddrl_op_accumulate(_p(self.grads), _p(self.gtmp), total, _st()))
            first = False

    @staticmethod
    def _expert_states(batch0):
        """An expert batch's states: a list of arrays (what every PreNet indexes with [0]) or one array; a leading axis
        of length 1 is the reference-shaped way of saying the same thing (tests/golden/make_golden_gail.py)."""
        if isinstance(batch0, (list, tuple)):
            return list(batch0)
        t = torch.as_tensor(batch0)
        return [t[0]] if t.dim() >= 3 and t.shape[0] == 1 else [t]

    def learn(self, data: Experience):
        for epoch in range(1, self.epochs + 1):
            start_time = time.time()
            for expert_batch in self.expert_data:
                self._ensure_packed()
                # g_loss = mean(D(data.states, data.actions)); expert_loss = -mean(D(expert)) (GAIL.py:78-80)
                self._pass(data.states, torch.as_tensor(data.actions), +1.0, True)
                self._pass(self._expert_states(expert_batch[0]), expert_batch[1], -1.0, False)
                import torch.distributed as tdist
                if tdist.is_available() and tdist.is_initialized() and tdist.get_world_size(self._process_group) > 1:
                    from ddrl4nav_amd.dist import allreduce_flat
                    allreduce_flat(self.grads, self._process_group)     # SUM of shard gradients pre-scaled by 1 / n_total
                    allreduce_flat(self._loss, self._process_group)
                check(self.lib.ddrl_op_clip_rmsprop(_p(self.params), _p(self.grads), _p(self.square_avg), self.n_params, self.lr,
                                                    self.alpha, self.eps, self.WGAN_clip_grad_num, _p(self._opt_ws), _st()))
                self._dirty = True
                self.update_time += 1
                if self.update_time % self.decay_step == 0:       # StepLR.step() after every optimiser step (GAIL.py:85)
                    self.lr = self.lr * self.decay_gamma
                yield {"Gail[D]BackUpTime": time.time() - start_time, "Gail[D]Loss": float(self._loss.item())}, self.update_time, True
                break

    def stats(self):
        s = self.grads[self.n_params + 4:self.n_params + 6].cpu().numpy()
        return {"GradNorm": float(s[0]), "ClipCoef": float(s[1]), "lr": self.lr}


class GAIL(Basenn):
    def __init__(self, generator, discriminator: Discriminator, gail_critic):
        super().__init__(discriminator.config, discriminator.config_nn)
        self.device = discriminator.device
        self.generator = generator
        self.discriminator = discriminator
        self.actor = generator.actor
        self.rnd = self.generator.rnd
        self.gail_critic = gail_critic
        self.generator.add_critic(self.gail_critic)
        self.generator.gail_critic = True

    def _train_generator(self, data: Experience):
        return self.generator.learn(data)

    def _train_discriminator(self, data: Experience):
        return self.discriminator.learn(data)

    def get_rnd(self, states):
        return self.rnd(states)

    def to(self, *args, **kwargs):
        return self

    def params_changed(self):
        self.generator.params_changed()
        self.discriminator.params_changed()

    def load_state_dict(self, state_dict, strict=True, **kw):
        out = super().load_state_dict(state_dict, strict=strict, **kw)
        self.params_changed()
        return out

    def forward(self, x, act=None, play_mode=False):
        if not isinstance(x, tuple):                       # G (GAIL.py:140-143)
            return self.generator(x, act, play_mode)
        if len(x) == 2:                                    # D (GAIL.py:145-147)
            return self.discriminator(x)

    def learn(self, data: Experience):
        for loss_item, update_time, last in self._train_discriminator(data):
            yield loss_item, update_time, False
        for loss_item, update_time, last in self._train_generator(data):
            yield loss_item, update_time, True
