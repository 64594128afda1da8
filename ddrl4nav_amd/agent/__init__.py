from ddrl4nav_amd.agent.agent import Agents, gae_device
from ddrl4nav_amd.agent.rollout import DeviceRollout, StateRollout

__all__ = ["Agents", "gae_device", "DeviceRollout", "StateRollout"]
