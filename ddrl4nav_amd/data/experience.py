"""Experience: the sample container handed to ``net.learn`` (mirror of the slots and methods
of USTC_lab/data/experience.py:23-148 that callers on the hot path use).

Layout contract into the learner:
  states    list whose [0] is [B, 4, 84, 84]  (uint8 frames, or float32 = uint8/255 as the
            reference ships them; the HIP path consumes uint8)
  advs, actions, old_logps   [B]
  values    [n_value_heads, B]   (row 0 = PPO return target)
"""
import math
from typing import Generator, List

import numpy as np
import torch


class Experience:
    __slots__ = ['states', 'actions', 'old_logps', 'rewards', 'dones', 'advs', 'values', 'durations', 'is_clean']

    def __init__(self, states, advs=None, actions=None, old_logps=None, values=None, rewards=None, dones=None,
                 durations=None, is_clean=None):
        self.states = states
        self.advs = advs
        self.actions = actions
        self.old_logps = old_logps
        self.values = values
        self.rewards = rewards
        self.dones = dones
        self.durations = durations
        self.is_clean = is_clean

    def __len__(self):
        return len(self.states[0])

    def get_xrapv(self):
        return [self.states, self.advs, self.actions, self.old_logps, self.values]

    def to_tensor(self, dtype=torch.float32, device='cpu'):
        """In-place conversion like experience.py:56-62.  uint8 frames stay uint8 (that is what
        the kernels read); everything else becomes `dtype`."""
        for i, s in enumerate(self.states):
            keep_u8 = (isinstance(s, np.ndarray) and s.dtype == np.uint8) or (torch.is_tensor(s) and s.dtype == torch.uint8)
            self.states[i] = torch.as_tensor(s, device=device) if keep_u8 else torch.as_tensor(s, dtype=dtype, device=device)
        self.advs = torch.as_tensor(self.advs, dtype=dtype, device=device)
        self.actions = torch.as_tensor(self.actions, dtype=dtype, device=device)
        self.old_logps = torch.as_tensor(self.old_logps, dtype=dtype, device=device)
        self.values = torch.as_tensor(self.values, dtype=dtype, device=device)

    # ---- host-side batching helpers (numpy) --------------------------------------------------
    @classmethod
    def concat_state(cls, states: List[List[np.ndarray]]) -> List[np.ndarray]:
        n_inputs = len(states[0])
        return [np.concatenate([s[i] for s in states], axis=0) for i in range(n_inputs)]

    @classmethod
    def index_state(cls, states: List[np.ndarray], index: np.ndarray) -> List[np.ndarray]:
        return [s[index] for s in states]

    @classmethod
    def batch_data(cls, exps: List["Experience"], clean: bool = True) -> "Experience":
        states = cls.concat_state([e.states for e in exps])
        advs = np.concatenate([e.advs for e in exps], axis=0)
        actions = np.concatenate([e.actions for e in exps], axis=0)
        old_logps = np.concatenate([e.old_logps for e in exps], axis=0)
        values = np.concatenate([e.values for e in exps], axis=1)
        if clean:
            return Experience(states=states, advs=advs, actions=actions, old_logps=old_logps, values=values)
        keep = np.concatenate([e.is_clean for e in exps], axis=0)
        assert keep.dtype == np.bool_
        return Experience(states=cls.index_state(states, keep), advs=advs[keep], actions=actions[keep],
                          old_logps=old_logps[keep], values=values[:, keep],
                          is_clean=np.ones(int(keep.sum()), dtype=np.bool_))

    @classmethod
    def batch_data_gene(cls, exps: List["Experience"]) -> Generator["Experience", None, None]:
        for start in range(0, len(exps), 64):  # 64-step chunks, experience.py:88-98
            yield cls.batch_data(exps[start:start + 64], clean=False)

    def concatenate(self, data: "Experience"):
        if not data:
            return
        self.states = np.concatenate((self.states, data.states), axis=0)
        self.advs = np.concatenate((self.advs, data.advs), axis=0)
        self.actions = np.concatenate((self.actions, data.actions), axis=0)
        self.old_logps = np.concatenate((self.old_logps, data.old_logps), axis=0)
        self.values = np.concatenate((self.values, data.values), axis=0)

    def split(self, batch_size: int) -> Generator["Experience", None, None]:
        pieces = math.ceil(len(self.states) / batch_size)
        cuts = [i * batch_size for i in range(1, pieces)]
        parts = [np.split(item, cuts, axis=0) for item in self.get_xrapv()]
        for i in range(pieces):
            yield Experience(*[p[i] for p in parts])
