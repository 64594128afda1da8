"""ddrl4nav_amd.nn -- mirror of the USTC_lab.nn names the Pong hot path uses."""
from ddrl4nav_amd.nn.base import Basenn, PreNet
from ddrl4nav_amd.nn.critic import Critic
from ddrl4nav_amd.nn.actor import Actor, CategoricalActor, GaussionActor
from ddrl4nav_amd.nn.atari_encoder import AtariPreNet
from ddrl4nav_amd.nn.distribution import HipCategorical
from ddrl4nav_amd.nn.generic import GenericPPO, HipNormal, MLPPreNet, NavPedPreNet, NavPreNet, NavPreNet1D, mlp
from ddrl4nav_amd.nn.ppo import PPO
from ddrl4nav_amd.nn.gail import GAIL, Discriminator

NETWORK_MAP = {"ppo": PPO, "gail": GAIL}   # nn/__init__.py:19-22

__all__ = ["PPO", "GenericPPO", "Basenn", "PreNet", "NETWORK_MAP", "CategoricalActor", "GaussionActor", "Critic",
           "AtariPreNet", "MLPPreNet", "NavPreNet", "NavPedPreNet", "NavPreNet1D", "mlp", "Actor", "HipCategorical",
           "HipNormal", "GAIL", "Discriminator"]
