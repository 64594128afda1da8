from ddrl4nav_amd.agent.agent import Agents, gae_device
from ddrl4nav_amd.agent.rollout import DeviceRollout, StateRollout
from ddrl4nav_amd.agent.statistics import EpisodeReturns

__all__ = ["Agents", "gae_device", "DeviceRollout", "StateRollout", "EpisodeReturns"]
