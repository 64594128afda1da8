"""HipCategorical: the object PPO.forward hands back where the reference returns a
torch.distributions.Categorical (actor.py:97).  Same attribute surface for the callers on the
path (forward.py:132-138, ppo.py:82-83,106): .probs .logits .sample() .log_prob(a) .entropy()."""
import torch


class HipCategorical:
    def __init__(self, hot_path, softmax_probs, seed, stream_id, sampled_action=None, sampled_logp=None):
        self._hp = hot_path
        self._p = softmax_probs            # raw softmax output [n, A]
        self._seed, self._stream = seed, stream_id
        self._action, self._logp = sampled_action, sampled_logp
        self._stats = None
        self._draws = 0

    def _derive(self):
        if self._stats is None:
            self._stats = self._hp.categorical_stats(self._p)
        return self._stats

    @property
    def probs(self):          # Categorical.probs = p / sum(p)
        return self._derive()[0]

    @property
    def logits(self):         # log(clamp(probs, eps, 1-eps))
        return self._derive()[1]

    def entropy(self):
        return self._derive()[2]

    def sample(self):
        """First call returns the draw fused into the forward kernel; later calls draw again from
        the same counter-based stream family (draw index in bits 48.. of the stream id; all 64 bits
        of the id are mixed into the generator, csrc/common.h:hash_uniform)."""
        if self._draws == 0 and self._action is not None:
            self._draws += 1
            return self._action
        self._draws += 1
        a, lp = self._hp.categorical_sample(self._p, self._seed, self._stream ^ (self._draws << 48))
        self._action, self._logp = a, lp
        return a

    def log_prob(self, value):
        if value is self._action and self._logp is not None:
            return self._logp
        idx = value.to(torch.int64).unsqueeze(-1)
        return self.logits.gather(-1, idx).squeeze(-1)
