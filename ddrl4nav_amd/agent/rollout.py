"""DeviceRollout: the device-resident experience pool that replaces the per-step Experience
records + Redis training blobs of the reference (agent.py:217-296, multiqueue.py:83-105,
backward.py:48-62) when Forward, Env bookkeeping and Backward are co-located on one GPU.

Pool layout (all on the GPU, [time][env] major so that GAE loads are coalesced across envs):
  frames  uint8 [T+1, N, C, 84, 84]    values f32 [T+1, N]     rewards f32 [T, N]
  actions f32 [T, N]   logps f32 [T, N]   dones u8 [T, N]      adv / ret f32 [T, N]
Frames arrive either as device tensors or through a pinned-host ring (data/ring.py) with
hipMemcpyAsync on a copy stream, overlapping the previous step's forward."""
import torch

from ddrl4nav_amd.agent.agent import gae_device
from ddrl4nav_amd.data import Experience
from ddrl4nav_amd.utils.staging import copy_into


class DeviceRollout:
    def __init__(self, net, n_envs, horizon=256, channels=4, gamma=0.99, landa=0.95, device=None, seed=0):
        self.hp = net.hot_path if hasattr(net, "hot_path") else net
        self.N, self.T, self.C = int(n_envs), int(horizon), int(channels)
        self.gamma, self.landa = gamma, landa
        dev = torch.device(device if device is not None else self.hp.device)
        self.device = dev
        N, T = self.N, self.T
        self.frames = torch.empty((T + 1, N, self.C, 84, 84), dtype=torch.uint8, device=dev)
        self.values = torch.zeros((T + 1, N), dtype=torch.float32, device=dev)
        self.rewards = torch.zeros((T, N), dtype=torch.float32, device=dev)
        self.dones = torch.zeros((T, N), dtype=torch.uint8, device=dev)
        self.actions = torch.zeros((T, N), dtype=torch.float32, device=dev)
        self.logps = torch.zeros((T, N), dtype=torch.float32, device=dev)
        self.adv = torch.empty((T, N), dtype=torch.float32, device=dev)
        self.ret = torch.empty((T, N), dtype=torch.float32, device=dev)
        self._probs = torch.empty((N, self.hp.n_actions), dtype=torch.float32, device=dev)
        self._scratch_a = torch.empty(N, dtype=torch.float32, device=dev)
        self._scratch_l = torch.empty(N, dtype=torch.float32, device=dev)
        self.seed, self.rollouts, self.t = int(seed), 0, 0
        self.copy_stream = torch.cuda.Stream(device=dev)

    # ---- ingest ---------------------------------------------------------------------------------
    def put_frames(self, t, frames):
        """Device (or pinned host) uint8 frames [N,C,84,84] -> pool slot t."""
        copy_into(self.frames[t], frames)

    def put_frames_from_ring(self, t, ring):
        """hipMemcpyAsync from the pinned ring on the copy stream; the compute stream waits on it."""
        ring.pop_to(self.frames[t], stream=self.copy_stream)
        torch.cuda.current_stream().wait_stream(self.copy_stream)

    # ---- acting ---------------------------------------------------------------------------------
    def act(self, t):
        """Forward + sample on slot t: fills values[t], actions[t], logps[t]; returns actions[t]."""
        self.hp.forward(self.frames[t], seed=self.seed + self.rollouts, stream_id=t, probs=self._probs,
                        value=self.values[t], action=self.actions[t], logp=self.logps[t])
        return self.actions[t]

    def bootstrap(self):
        """Value of the (T+1)-th stored step (agent.py:130); its action/logp are not kept."""
        self.hp.forward(self.frames[self.T], seed=self.seed + self.rollouts, stream_id=self.T, probs=self._probs,
                        value=self.values[self.T], action=self._scratch_a, logp=self._scratch_l)

    def record(self, t, rewards, dones):
        copy_into(self.rewards[t], rewards)
        copy_into(self.dones[t], dones)

    # ---- GAE + learner batch ----------------------------------------------------------------------
    def finish(self):
        gae_device(self.values, self.rewards, self.dones, self.gamma, self.landa, adv=self.adv, ret=self.ret)
        self.rollouts += 1

    def carry_over(self):
        """The last stored step becomes step 0 of the next rollout (agent.py:289-291)."""
        self.frames[0].copy_(self.frames[self.T])

    def batch(self):
        """Zero-copy views in the layout net.learn expects (sample order is irrelevant to the maths)."""
        B = self.N * self.T
        return Experience(states=[self.frames[:self.T].view(B, self.C, 84, 84)], advs=self.adv.view(B),
                          actions=self.actions.view(B), old_logps=self.logps.view(B), values=self.ret.view(1, B))


class StateRollout:
    """DeviceRollout for the operator-composed nets (nn/generic.py): the observation is a LIST of
    float tensors per env (e.g. robot_nav: laser [1,960], vector [5], pedestrian image [3,48,48]) and
    the action may be continuous.  Same pool layout idea ([time][env] major, everything on the GPU),
    same life cycle: put_states(t, ...) -> act(t) -> record(t, ...) ... bootstrap() -> finish() -> batch()."""

    def __init__(self, net, n_envs, state_shapes, horizon=256, gamma=0.99, landa=0.95, device=None):
        self.net = net
        self.N, self.T = int(n_envs), int(horizon)
        self.gamma, self.landa = gamma, landa
        dev = torch.device(device if device is not None else net.device)
        self.device = dev
        N, T = self.N, self.T
        f = dict(dtype=torch.float32, device=dev)
        self.states = [torch.empty((T + 1, N) + tuple(int(d) for d in shape), **f) for shape in state_shapes]
        self.values = torch.zeros((T + 1, N), **f)
        self.rewards = torch.zeros((T, N), **f)
        self.dones = torch.zeros((T, N), dtype=torch.uint8, device=dev)
        act_shape = (T, N, net.n_actions) if getattr(net, "continuous", False) else (T, N)
        self.actions = torch.zeros(act_shape, **f)
        self.logps = torch.zeros((T, N), **f)
        self.adv = torch.empty((T, N), **f)
        self.ret = torch.empty((T, N), **f)
        self.copy_stream = torch.cuda.Stream(device=dev)

    def put_states(self, t, states):
        """Device or pinned-host tensors, one per observation component -> pool slot t."""
        for pool, s in zip(self.states, states):
            copy_into(pool[t], torch.as_tensor(s).reshape(pool[t].shape))

    def put_state_from_ring(self, t, index, ring):
        """Component `index` of slot t from a pinned ring (raw fp32 bytes), on the copy stream."""
        ring.pop_to(self.states[index][t], stream=self.copy_stream)
        torch.cuda.current_stream().wait_stream(self.copy_stream)

    def act(self, t):
        (dist, _), values = self.net([p[t] for p in self.states])
        a = dist.sample()
        self.values[t].copy_(values[0][:, 0])
        self.actions[t].copy_(a)
        self.logps[t].copy_(self.net.actor.log_prob_from_distribution(dist, a))
        return self.actions[t]

    def bootstrap(self):
        (_, _), values = self.net([p[self.T] for p in self.states], play_mode=True)
        self.values[self.T].copy_(values[0][:, 0])

    def record(self, t, rewards, dones):
        copy_into(self.rewards[t], rewards)
        copy_into(self.dones[t], dones)

    def finish(self):
        gae_device(self.values, self.rewards, self.dones, self.gamma, self.landa, adv=self.adv, ret=self.ret)

    def carry_over(self):
        for p in self.states:
            p[0].copy_(p[self.T])

    def batch(self):
        B = self.N * self.T
        return Experience(states=[p[:self.T].reshape((B,) + tuple(p.shape[2:])) for p in self.states],
                          advs=self.adv.view(B), actions=self.actions.view((B,) + tuple(self.actions.shape[2:])),
                          old_logps=self.logps.view(B), values=self.ret.view(1, B))
