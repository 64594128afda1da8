from ddrl4nav_amd.runner.utils import create_net, ini_config, read_yaml

__all__ = ["create_net", "ini_config", "read_yaml"]
