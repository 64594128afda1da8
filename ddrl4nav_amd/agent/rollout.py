"""DeviceRollout: the device-resident experience pool that replaces the per-step Experience
records + Redis training blobs of the reference (agent.py:217-296, multiqueue.py:83-105,
backward.py:48-62) when Forward, Env bookkeeping and Backward are co-located on one GPU.

Pool layout (all on the GPU, [time][env] major so that GAE loads are coalesced across envs):
  frames  uint8 [T+1, N, C, 84, 84]    values f32 [T+1, N]     rewards f32 [T, N]
  actions f32 [T, N]   logps f32 [T, N]   dones u8 [T, N]      adv / ret f32 [T, N]
Frames arrive either as device tensors or through a pinned-host ring (data/ring.py) with
hipMemcpyAsync on a copy stream, overlapping the previous step's forward.

Carry-over between rollouts (reference agent.py:286-292: ``rewards_step[0] = rewards_step[T]; exps = [exps[-1]]``): the
reference keeps the WHOLE (T+1)-th Experience -- state, the action already sent to the env, its old log-prob, its value
under the weights of that moment, its reward and done -- as step 0 of the next rollout and acts from step 1 on.
``carry_over(keep_step=True)`` does exactly that (rows T of every pool, which ``bootstrap()`` / ``record(T, ...)`` fill,
move to row 0; the next rollout starts at ``t0 == 1``).  The default, ``carry_over()``, carries the FRAME only and lets
``act(0)`` re-evaluate it with the current weights (a fresh action, log-prob and value for slot 0): a deviation from the
reference, listed in DESIGN.md section 7 -- it needs no env step between ``bootstrap()`` and ``finish()`` and its slot-0
sample is on-policy for the weights the update starts from."""
import torch

from ddrl4nav_amd.agent.agent import gae_device
from ddrl4nav_amd.agent.statistics import EpisodeReturns
from ddrl4nav_amd.data import Experience
from ddrl4nav_amd.utils.staging import copy_into


class DeviceRollout:
    def __init__(self, net, n_envs, horizon=256, channels=4, gamma=0.99, landa=0.95, device=None, seed=0, track_returns=False):
        self.hp = net.hot_path if hasattr(net, "hot_path") else net
        self.N, self.T, self.C = int(n_envs), int(horizon), int(channels)
        self.gamma, self.landa = gamma, landa
        dev = torch.device(device if device is not None else self.hp.device)
        self.device = dev
        N, T = self.N, self.T
        self.frames = torch.empty((T + 1, N, self.C, 84, 84), dtype=torch.uint8, device=dev)
        self.values = torch.zeros((T + 1, N), dtype=torch.float32, device=dev)
        # rows 0..T-1 are the rollout (contiguous [T, N] views: what GAE and the learner batch read); row T holds the (T+1)-th
        # step's reward / done / action / log-prob for carry_over(keep_step=True)
        self._rewards = torch.zeros((T + 1, N), dtype=torch.float32, device=dev)
        self._dones = torch.zeros((T + 1, N), dtype=torch.uint8, device=dev)
        self._actions = torch.zeros((T + 1, N), dtype=torch.float32, device=dev)
        self._logps = torch.zeros((T + 1, N), dtype=torch.float32, device=dev)
        self.rewards, self.dones = self._rewards[:T], self._dones[:T]
        self.actions, self.logps = self._actions[:T], self._logps[:T]
        self.adv = torch.empty((T, N), dtype=torch.float32, device=dev)
        self.ret = torch.empty((T, N), dtype=torch.float32, device=dev)
        self._probs = torch.empty((N, self.hp.n_actions), dtype=torch.float32, device=dev)
        self.seed, self.rollouts, self.t = int(seed), 0, 0
        self.t0 = 0  # first step the next rollout has to act on: 1 after carry_over(keep_step=True), else 0
        self.copy_stream = torch.cuda.Stream(device=dev)
        self.returns = EpisodeReturns(N, dev) if track_returns else None

    # ---- ingest ---------------------------------------------------------------------------------
    def put_frames(self, t, frames):
        """Device (or pinned host) uint8 frames [N,C,84,84] -> pool slot t."""
        copy_into(self.frames[t], frames)

    def put_frames_from_ring(self, t, ring):
        """hipMemcpyAsync from the pinned ring on the copy stream; the compute stream waits on it.  The FIRST slot a rollout fills
        (t <= t0) also orders the copy stream behind the compute stream: whatever was enqueued there on the pool before this rollout
        (its zero fill, the previous update still reading the frames) is through before the first DMA overwrites a slot.  Later slots
        do not wait again -- copy t + 1 runs under forward t (bench.py async_ingest_leg: waiting per slot cost 10 % of the overlapped
        ingest rate)."""
        if t <= self.t0:
            self.copy_stream.wait_stream(torch.cuda.current_stream())
        ring.pop_to(self.frames[t], stream=self.copy_stream)
        torch.cuda.current_stream().wait_stream(self.copy_stream)

    # ---- acting ---------------------------------------------------------------------------------
    def act(self, t):
        """Forward + sample on slot t: fills values[t], actions[t], logps[t]; returns actions[t].  A slot below t0 was carried
        over as a complete step (its action has already been sent to the env): nothing is evaluated, the kept action returns."""
        if t < self.t0:
            return self._actions[t]
        self.hp.forward(self.frames[t], seed=self.seed + self.rollouts, stream_id=t, probs=self._probs,
                        value=self.values[t], action=self._actions[t], logp=self._logps[t])
        return self._actions[t]

    def bootstrap(self):
        """The (T+1)-th stored step: its value is the GAE bootstrap (agent.py:130); its action / log-prob land in row T, for a
        host that steps the env with them and keeps the step (carry_over(keep_step=True)).  Returns the action."""
        return self.act(self.T)

    def record(self, t, rewards, dones):
        """Reward / done of step t (0..T; T = the (T+1)-th step, only needed for keep_step)."""
        if t < self.t0:
            raise ValueError("step %d was carried over from the previous rollout with its reward and done" % t)
        copy_into(self._rewards[t], rewards)
        copy_into(self._dones[t], dones)

    # ---- GAE + learner batch ----------------------------------------------------------------------
    def finish(self):
        gae_device(self.values, self.rewards, self.dones, self.gamma, self.landa, adv=self.adv, ret=self.ret)
        if self.returns is not None:  # Status.update_reward_status over rows 0..T-1: every step once, also with keep_step
            self.returns.update(self.rewards, self.dones)
        self.rollouts += 1

    def carry_over(self, keep_step=False):
        """The last stored step becomes step 0 of the next rollout (agent.py:286-292).  keep_step=True: the whole step, as the
        reference keeps it (frame, action, old log-prob, value, reward, done of row T; the next rollout acts from t0 = 1);
        False (default): the frame only, slot 0 is evaluated again by act(0) under the current weights."""
        T = self.T
        self.frames[0].copy_(self.frames[T])
        if keep_step:
            self.values[0].copy_(self.values[T])
            for pool in (self._rewards, self._dones, self._actions, self._logps):
                pool[0].copy_(pool[T])
            self.t0 = 1
        else:
            self.t0 = 0

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

    def __init__(self, net, n_envs, state_shapes, horizon=256, gamma=0.99, landa=0.95, device=None, track_returns=False):
        self.net = net
        self.N, self.T = int(n_envs), int(horizon)
        self.gamma, self.landa = gamma, landa
        dev = torch.device(device if device is not None else net.device)
        self.device = dev
        N, T = self.N, self.T
        f = dict(dtype=torch.float32, device=dev)
        self.states = [torch.empty((T + 1, N) + tuple(int(d) for d in shape), **f) for shape in state_shapes]
        self.values = torch.zeros((T + 1, N), **f)
        self._rewards = torch.zeros((T + 1, N), **f)
        self._dones = torch.zeros((T + 1, N), dtype=torch.uint8, device=dev)
        act_shape = (T + 1, N, net.n_actions) if getattr(net, "continuous", False) else (T + 1, N)
        self._actions = torch.zeros(act_shape, **f)
        self._logps = torch.zeros((T + 1, N), **f)
        self.rewards, self.dones = self._rewards[:T], self._dones[:T]
        self.actions, self.logps = self._actions[:T], self._logps[:T]
        self.adv = torch.empty((T, N), **f)
        self.ret = torch.empty((T, N), **f)
        self.t0 = 0
        self.copy_stream = torch.cuda.Stream(device=dev)
        self.returns = EpisodeReturns(N, dev) if track_returns else None

    def put_states(self, t, states):
        """Device or pinned-host tensors, one per observation component -> pool slot t."""
        for pool, s in zip(self.states, states):
            copy_into(pool[t], torch.as_tensor(s).reshape(pool[t].shape))

    def put_state_from_ring(self, t, index, ring):
        """Component `index` of slot t from a pinned ring (raw fp32 bytes), on the copy stream (the first slot of a rollout orders the
        copy stream behind the compute stream, as put_frames_from_ring)."""
        if t <= self.t0:
            self.copy_stream.wait_stream(torch.cuda.current_stream())
        ring.pop_to(self.states[index][t], stream=self.copy_stream)
        torch.cuda.current_stream().wait_stream(self.copy_stream)

    def act(self, t):
        if t < self.t0:  # carried over as a complete step (DeviceRollout.act)
            return self._actions[t]
        (dist, _), values = self.net([p[t] for p in self.states])
        a = dist.sample()
        self.values[t].copy_(values[0][:, 0])
        self._actions[t].copy_(a)
        self._logps[t].copy_(self.net.actor.log_prob_from_distribution(dist, a))
        return self._actions[t]

    def bootstrap(self, sample=False):
        """Value of the (T+1)-th stored step (agent.py:130).  sample=True also draws its action / log-prob into row T, for a
        host that steps the env with them and keeps the step (carry_over(keep_step=True)); returns that action."""
        if sample:
            return self.act(self.T)
        (_, _), values = self.net([p[self.T] for p in self.states], play_mode=True)
        self.values[self.T].copy_(values[0][:, 0])
        return None

    def record(self, t, rewards, dones):
        if t < self.t0:
            raise ValueError("step %d was carried over from the previous rollout with its reward and done" % t)
        copy_into(self._rewards[t], rewards)
        copy_into(self._dones[t], dones)

    def finish(self):
        gae_device(self.values, self.rewards, self.dones, self.gamma, self.landa, adv=self.adv, ret=self.ret)
        if self.returns is not None:
            self.returns.update(self.rewards, self.dones)

    def carry_over(self, keep_step=False):
        """DeviceRollout.carry_over: keep_step=True keeps the whole (T+1)-th step as the reference does (agent.py:286-292)."""
        T = self.T
        for p in self.states:
            p[0].copy_(p[T])
        if keep_step:
            self.values[0].copy_(self.values[T])
            for pool in (self._rewards, self._dones, self._actions, self._logps):
                pool[0].copy_(pool[T])
            self.t0 = 1
        else:
            self.t0 = 0

    def batch(self):
        B = self.N * self.T
        return Experience(states=[p[:self.T].reshape((B,) + tuple(p.shape[2:])) for p in self.states],
                          advs=self.adv.view(B), actions=self.actions.view((B,) + tuple(self.actions.shape[2:])),
                          old_logps=self.logps.view(B), values=self.ret.view(1, B))
