"""PPO: drop-in for USTC_lab.nn.PPO on the Pong / AtariPreNet path (reference ppo.py:17-146).

Same constructor arguments, same ``forward(states, act=None, play_mode=False)`` return shape,
same ``learn(data)`` generator protocol and loss-dict keys, same ``state_dict()`` key names and
nn2redis blob -- but every tensor op of the reference is replaced by the HIP kernels behind
include/ddrl.h.  The module's parameters are views into one flat fp32 arena (the reference's
named_parameters() order), which is what the kernels, the Adam step and the RCCL all-reduce use.
"""
import time

import torch

from ddrl4nav_amd.data import Experience
from ddrl4nav_amd.engine import HotPath
from ddrl4nav_amd.nn.base import Basenn
from ddrl4nav_amd.nn.distribution import HipCategorical


from ddrl4nav_amd.nn.atari_encoder import frames_u8 as _frames_u8  # noqa: E402


class PPO(Basenn):
    def __new__(cls, actor, critic, prenet=None, *args, **kwargs):
        """``PPO(...)`` over anything but the Atari encoder is the operator-composed GenericPPO
        (nn/generic.py): same constructor, same protocol."""
        from ddrl4nav_amd.nn.atari_encoder import AtariPreNet
        enc = prenet if prenet is not None else getattr(actor, "pre", None)
        if cls is PPO and not isinstance(enc, AtariPreNet):
            from ddrl4nav_amd.nn.generic import GenericPPO
            return GenericPPO(actor, critic, prenet, *args, **kwargs)
        return super().__new__(cls)

    def __init__(self, actor, critic, prenet=None, rnd=None, config=None, config_nn=None, max_batch=None,
                 process_group=None):
        super().__init__(config, config_nn)
        if hasattr(actor, "log_std"):
            raise NotImplementedError("the Atari fast path has a Categorical actor only (reference atari.yaml)")
        if bool(config_nn.SHARE_CNN_NET) != (prenet is not None):
            raise ValueError("SHARE_CNN_NET=True needs a shared prenet (and pre-less actor / critic); "
                             "SHARE_CNN_NET=False needs prenet=None (reference runner/utils.py:122-143)")
        if rnd is not None:
            raise NotImplementedError("RND is disabled in the reference defaults (USE_RND=False) and out of scope")
        self.device = torch.device(config.DEVICE if str(config.DEVICE) != "cuda" else "cuda:%d" % torch.cuda.current_device())
        self.prenet = prenet
        self.actor = actor
        self.critic = critic
        self._critics = [self.critic]
        self.rnd = rnd
        self.gail_critic = False
        self.share_cnn_net = config_nn.SHARE_CNN_NET
        self.training_iter_time = config_nn.TRAINING_ITER_TIME
        self.update_time = 0
        self._cfg_nn = config_nn
        self._process_group = process_group
        self._seed = int(torch.initial_seed()) & (2 ** 63 - 1)
        self._calls = 0
        self._hp = None
        # learn(): False = one host sync per iteration (the reference's protocol: weights match update_time at every yield);
        # True = all iterations enqueued, one sync, then the yields (config_nn.DEFERRED_LOSS_READBACK; bench.py sets it)
        self.deferred_stats = bool(getattr(config_nn, "DEFERRED_LOSS_READBACK", False))
        self._stats_rows = None
        n_actions = actor.action_output_dim
        in_ch = (prenet if prenet is not None else actor.pre).conv1.in_channels
        cap = int(max_batch if max_batch is not None else max(2 * config_nn.TRAINING_MIN_BATCH, 2048))
        self._build(cap, n_actions, in_ch)
        actor._bind_owner(self)    # net.actor(x) / net.critic(x) as the reference offers them (actor.py:27-40, critic.py:14-21)
        critic._bind_owner(self)

    # ---- arena binding --------------------------------------------------------------------------
    def _hot_path_kwargs(self):
        c = self._cfg_nn
        return dict(share_cnn_net=1 if c.SHARE_CNN_NET else 0, learning_rate=float(c.LEARNING_RATE),
                    smooth_l1_loss=1 if c.SMOOTH_L1_LOSS else 0, clip_grad=1 if c.CLIP_GRID else 0, clip_grad_norm=float(c.CLIP_GRID_NUM),
                    actor_lr=float(c.ACTOR_LEARNING_RATE), critic_lr=float(c.CRITIC_LEARNING_RATE),
                    ppo_clip=float(c.PPO_CLIP), dual_clip=float(c.DUEL_PPO_CLIP), v_loss_theta=float(c.V_LOSS_THETA),
                    ent_loss_theta=float(c.ENTROPY_LOSS_THETA))

    def _build(self, max_batch, n_actions, in_ch, old=None):
        hp = HotPath(max_batch=max_batch, device=self.device, n_actions=n_actions, in_channels=in_ch,
                     process_group=self._process_group, **self._hot_path_kwargs())
        params = list(self.named_parameters())
        total = sum(p.numel() for _, p in params)
        if total != hp.n_params:
            raise ValueError("module tree has %d parameters, the HIP path expects %d" % (total, hp.n_params))
        off = 0
        with torch.no_grad():
            for _, p in params:
                n = p.numel()
                view = hp.params[off:off + n].view(p.shape)
                view.copy_(p.detach().to(self.device, torch.float32))
                p.data = view            # the module now aliases the flat arena
                p.requires_grad_(False)
                off += n
        if old is not None:
            hp.adam_m.copy_(old.adam_m)
            hp.adam_v.copy_(old.adam_v)
            from ddrl4nav_amd._lib import check
            check(hp.lib.ddrl_set_step(hp.ctx, old.step))
            old.close()
        hp.params_changed()
        self._hp = hp

    def _ensure_capacity(self, n):
        if n > self._hp.max_batch:
            self._build(int(n), self._hp.n_actions, int(self._hp.cfg.in_channels), old=self._hp)

    @property
    def hot_path(self):
        return self._hp

    def load_state_dict(self, state_dict, strict=True, **kw):
        out = super().load_state_dict(state_dict, strict=strict, **kw)
        self._hp.params_changed()
        return out

    def params_changed(self):
        """Rebuild the packed weight layouts after an in-place write to the parameter views."""
        if self._hp is not None:
            self._hp.params_changed()

    def to(self, *args, **kwargs):  # the arena already lives on the GPU; keep `.to(DEVICE)` callers happy
        return self

    # ---- PPO.forward (ppo.py:72-75) ----------------------------------------------------------------
    def forward(self, states, act=None, play_mode=False):
        frames = _frames_u8(states, self.device)
        n = frames.shape[0]
        self._ensure_capacity(n)
        self._calls += 1
        a_in = None if act is None else torch.as_tensor(act, dtype=torch.float32, device=self.device).contiguous()
        probs, value, action, logp = self._hp.forward(frames, act=a_in, seed=self._seed, stream_id=self._calls)
        values = [value.view(n, 1)]
        if play_mode:
            return (probs, logp if act is not None else None), values
        if act is None:
            dist = HipCategorical(self._hp, probs, self._seed, self._calls, action, logp)
            return (dist, None), values
        dist = HipCategorical(self._hp, probs, self._seed, self._calls, a_in, logp)
        return (dist, logp), values

    def add_critic(self, critic):
        raise NotImplementedError("extra critics (RND / GAIL) are out of scope on this path")

    def get_rnd(self, states):
        raise NotImplementedError("RND is out of scope on this path")

    def states_normalization(self, states):
        return states / 255

    # ---- PPO.learn (ppo.py:77-146) -----------------------------------------------------------------
    def learn(self, data: Experience):
        frames = _frames_u8(data.states, self.device)
        B = frames.shape[0]
        self._ensure_capacity(B)
        f32 = lambda t: torch.as_tensor(t, dtype=torch.float32, device=self.device).contiguous()
        actions, old_logps, advs = f32(data.actions), f32(data.old_logps), f32(data.advs)
        rets = f32(data.values)[0].contiguous()
        assert rets.shape == (B,)
        # data-parallel: every rank scales by 1 / (sum of the ranks' batch sizes) -- shards may be uneven
        from ddrl4nav_amd.dist import global_batch
        b_global = global_batch(B, self._process_group)
        if self.deferred_stats:
            # All TRAINING_ITER_TIME iterations are enqueued back to back; every iteration's 8-float statistics tail goes to its own
            # pinned host row by an asynchronous copy and ONE synchronisation precedes the yields (the reference syncs four times per
            # iteration, ppo.py:132-137).  Same keys, same values, same update_time per yield; what differs is that the weights are
            # already those of the LAST iteration when the first item is yielded -- the reference's consumer (backward.py:189-209)
            # publishes at update_time % 10 == 0 only, i.e. after the last one either way.  With N > 1 ranks this also keeps a slow
            # host from stalling the other ranks' collectives once per iteration.
            k = self.training_iter_time
            if self._stats_rows is None or self._stats_rows.shape[0] < k:
                self._stats_rows = torch.empty((k, 8), dtype=torch.float32).pin_memory()
            t0 = time.time()
            for i in range(k):
                self._hp.ppo_iter(frames, actions, old_logps, advs, rets, b_global=b_global)
                self._hp.allreduce_grads()
                self._hp.clip_adam_step()
                self._hp.stats_async(self._stats_rows[i])
            torch.cuda.current_stream().synchronize()
            dt = (time.time() - t0) / max(k, 1)
            for i in range(k):
                self.update_time += 1
                s = self._hp.stats_dict(self._stats_rows[i])
                yield ({"PpoTotalLoss": s["PpoTotalLoss"], "ActorLoss": s["ActorLoss"], "VLoss": s["VLoss"], "EntLoss": s["EntLoss"],
                        "PpoBackUpTime": dt}, self.update_time, True)
            return
        for _ in range(self.training_iter_time):
            t0 = time.time()
            self._hp.ppo_iter(frames, actions, old_logps, advs, rets, b_global=b_global)
            self._hp.allreduce_grads()
            self._hp.clip_adam_step()
            self.update_time += 1
            s = self._hp.stats()  # one device->host copy (the reference does four .item() syncs)
            loss_log = {"PpoTotalLoss": s["PpoTotalLoss"], "ActorLoss": s["ActorLoss"], "VLoss": s["VLoss"],
                        "EntLoss": s["EntLoss"], "PpoBackUpTime": time.time() - t0}
            yield loss_log, self.update_time, True
