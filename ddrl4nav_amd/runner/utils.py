"""ini_config / create_net: mirror of USTC_lab/runner/utils.py:50-170 for the atari branch.

``read_yaml`` does not call gym.make (gym is not needed to learn Pong's action space on this
path): the YAML may carry ``discrete_actions`` itself, otherwise the Atari minimal action-set
size is looked up from a small table (Pong: 6)."""
import yaml

from ddrl4nav_amd.config import BaseConfig, ConfigNN
from ddrl4nav_amd.nn import AtariPreNet, Basenn, Critic, PPO

_ATARI_ACTIONS = {"Pong": 6, "Breakout": 4, "SpaceInvaders": 6, "Seaquest": 18, "Qbert": 6, "BeamRider": 9,
                  "Enduro": 9, "MsPacman": 9, "Boxing": 18, "Freeway": 3}


def update_cfg_env(cfg):
    if cfg.get('env_type') == 'gym' and 'discrete_action' not in cfg:
        name = cfg['env_name']
        n = next((v for k, v in _ATARI_ACTIONS.items() if name.startswith(k)), None)
        if n is None:
            raise KeyError("put discrete_actions into the YAML for %s" % name)
        cfg['discrete_action'] = True
        cfg['discrete_actions'] = list(range(n))
        cfg['input_dim'] = 84 * cfg.get('int_frame_stack', 4)


def read_yaml(parse) -> dict:
    with open(parse.yaml_f, 'r', encoding="utf-8") as f:
        cfg = yaml.load(f.read(), Loader=yaml.FullLoader)
    update_cfg_env(cfg)
    return cfg


def ini_config(parse):
    dict_config_env = read_yaml(parse)
    config_nn = ConfigNN(dict_config_env)
    config = BaseConfig(parse, dict_config_env)
    config.TEST = dict_config_env.get('test', False)
    return config, config_nn, dict_config_env


def create_net(configs, max_batch=None, process_group=None) -> Basenn:
    """atari branches of create_net (runner/utils.py:122-143,159-160): two encoders
    (SHARE_CNN_NET=False, the default) or one shared prenet."""
    config, config_nn, config_env = configs['config'], configs['config_nn'], configs['config_env']
    if config.TASK_TYPE != 'atari':
        raise NotImplementedError("only the atari task type is built (SURVEY.md section 8); got %s" % config.TASK_TYPE)
    if config_nn.NETWORK_TYPE != "ppo":
        raise NotImplementedError("NETWORK_TYPE=%s is not built (GAIL: SURVEY.md section 8f row 4)" % config_nn.NETWORK_TYPE)
    frames = config_env['int_frame_stack']
    if config_nn.SHARE_CNN_NET:
        actor = config_nn.ACTOR_CLASS(action_output_dim=config_nn.ACTION_OUTPUT_DIM, device=config_nn.DEVICE,
                                      last_input_dim=config_nn.AC_INPUT_DIM, soft_max_grid=config_nn.SOFT_MAX_GRID,
                                      nn_dtype=config_nn.MODULE_TENSOR_DTYPE)
        critic = Critic(device=config_nn.DEVICE)
        prenet = AtariPreNet(frames, last_output_dim=config_nn.AC_INPUT_DIM, device=config_nn.DEVICE)
        return PPO(actor, critic, prenet, None, config, config_nn, max_batch=max_batch, process_group=process_group)
    pre_actor = AtariPreNet(frames, last_output_dim=config_nn.AC_INPUT_DIM, device=config_nn.DEVICE)
    pre_critic = AtariPreNet(frames, last_output_dim=config_nn.AC_INPUT_DIM, device=config_nn.DEVICE)
    actor = config_nn.ACTOR_CLASS(action_output_dim=config_nn.ACTION_OUTPUT_DIM, device=config_nn.DEVICE,
                                  soft_max_grid=config_nn.SOFT_MAX_GRID, last_input_dim=config_nn.AC_INPUT_DIM,
                                  pre=pre_actor, nn_dtype=config_nn.MODULE_TENSOR_DTYPE)
    critic = Critic(device=config_nn.DEVICE, last_input_dim=config_nn.AC_INPUT_DIM, pre=pre_critic)
    return PPO(actor, critic, None, None, config, config_nn, max_batch=max_batch, process_group=process_group)
