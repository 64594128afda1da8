from ddrl4nav_amd.config.utils import game_type
from ddrl4nav_amd.config.base_config import BaseConfig
from ddrl4nav_amd.config.config_nn import ConfigNN

__all__ = ["game_type", "BaseConfig", "ConfigNN"]
