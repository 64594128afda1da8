"""Value head: parameter holder for the state-value estimator V(s).

Interface parity with the reference's critic (USTC_lab/nn/critic.py:6-21): constructor keywords
``device``, ``last_input_dim``, ``pre`` and the attribute names ``critic_linear`` / ``pre`` -- the
latter fix the ``state_dict`` keys (``critic.critic_linear.weight`` ...) and, because the linear
layer is registered BEFORE the optional private encoder, the position of both inside the flat
parameter arena (csrc/common.h: make_layout).

There is no arithmetic here.  The dot product with the 512-wide features, the squared-error /
smooth-L1 loss and their backward run in the head kernels (csrc/heads.hip, csrc/gheads.hip) of the
PPO object that owns this module.
"""
from torch import nn

__all__ = ["Critic"]

_FEATURES = 512  # AC_INPUT_DIM


class Critic(nn.Module):
    def __init__(self, device="cpu", last_input_dim=_FEATURES, pre=None):
        nn.Module.__init__(self)
        # registration order matters: head first, encoder second (see module docstring)
        self.add_module("critic_linear", nn.Linear(int(last_input_dim), 1))
        self.add_module("pre", pre) if pre is not None else setattr(self, "pre", None)
        self.device = device

    def _bind_owner(self, net):
        import weakref
        object.__setattr__(self, "_owner", weakref.ref(net))

    def forward(self, x):
        """critic.py:14-21: V(x) ``[n, 1]`` for the raw observation when the critic has its own encoder.  Evaluated by the owning
        net's fused forward (play mode: nothing is sampled); with a shared prenet the reference passes features -- not offered."""
        owner = getattr(self, "_owner", lambda: None)()
        if owner is None:
            raise RuntimeError("this critic is not bound to a ddrl4nav_amd net (its arithmetic lives in the net's HIP kernels)")
        if self.pre is None:
            raise NotImplementedError("net.critic(features) with a shared prenet: call net(states) -- the heads are fused into it")
        return owner.forward(x, None, True)[1][0]
