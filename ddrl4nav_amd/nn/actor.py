"""Actor heads (mirror of USTC_lab/nn/actor.py:10-101): parameter holders + the
log_prob_from_distribution hook the Forward server calls (forward.py:138)."""
from torch import nn


class Actor(nn.Module):
    def __init__(self, **kwargs):
        super().__init__()
        self.pre = kwargs['pre']
        self.device = kwargs['device']

    def _log_prob_from_distribution(self, pi, act):
        raise NotImplementedError

    def log_prob_from_distribution(self, pi, act):
        return self._log_prob_from_distribution(pi, act)

    def _bind_owner(self, net):
        """The net whose fused forward evaluates this head (a weak reference: the net owns the module, not the reverse)."""
        import weakref
        object.__setattr__(self, "_owner", weakref.ref(net))

    def forward(self, x, act=None, play_mode=False):
        """actor.py:27-40: ``(pi, log_p)`` for the raw observation ``x`` when the actor has its own encoder (``pre``).  The
        arithmetic is the owning net's fused forward (the HIP kernels evaluate both heads in one pass; the value is dropped
        here).  With a shared prenet the reference feeds FEATURES to the bare head: not offered on this path."""
        owner = getattr(self, "_owner", lambda: None)()
        if owner is None:
            raise RuntimeError("this actor head is not bound to a ddrl4nav_amd net (its arithmetic lives in the net's HIP kernels)")
        if self.pre is None:
            raise NotImplementedError("net.actor(features) with a shared prenet: call net(states) -- the heads are fused into it")
        return owner.forward(x, act, play_mode)[0]


class CategoricalActor(Actor):
    def __init__(self, action_output_dim, device='cpu', soft_max_grid=True, last_input_dim=512, pre=None, nn_dtype=None):
        super().__init__(pre=pre, device=device)
        self.logits_net = None
        self.action_output_dim = action_output_dim
        self.actor_linear = nn.Linear(last_input_dim, action_output_dim)
        self.soft_max_grid = soft_max_grid

    def _log_prob_from_distribution(self, pi, act):
        return pi.log_prob(act)


class GaussionActor(Actor):
    """actor.py:43-70: mu = actor_linear(h), std = exp(log_std), Normal(mu, std); log-prob summed
    over the action dims.  log_std is registered after ``pre`` is stored but, being a direct
    Parameter of the module, comes FIRST in named_parameters() (as in the reference)."""

    def __init__(self, action_output_dim=1, device='cpu', soft_max_grid=True, last_input_dim=512, pre=None, nn_dtype=None):
        super().__init__(pre=pre, device=device)
        import torch
        self.action_output_dim = action_output_dim
        self.actor_linear = nn.Linear(last_input_dim, action_output_dim)
        self.log_std = nn.Parameter(-0.5 * torch.ones(action_output_dim, dtype=torch.float32))

    def _log_prob_from_distribution(self, pi, act):
        if hasattr(pi, "summed_log_prob"):
            return pi.summed_log_prob(act)
        return pi.log_prob(act).sum(axis=-1)
