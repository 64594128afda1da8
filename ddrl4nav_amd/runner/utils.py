"""ini_config / create_net: mirror of USTC_lab/runner/utils.py:50-170 for the atari branch.

``read_yaml`` does not call gym.make (gym is not needed to learn Pong's action space on this
path): the YAML may carry ``discrete_actions`` itself, otherwise the Atari minimal action-set
size is looked up from a small table (Pong: 6)."""
import yaml

from ddrl4nav_amd.config import BaseConfig, ConfigNN
import copy

from ddrl4nav_amd.nn import (AtariPreNet, Basenn, Critic, Discriminator, GAIL, GenericPPO, MLPPreNet, NavPedPreNet, NavPreNet,
                             NavPreNet1D, PPO)

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
    if config_nn.NETWORK_TYPE == "gail":            # config_nn.py:131
        config_nn.MODEL_TO_REDIS_FREQUENCY = config_nn.GAN_D_EPOCH
    config = BaseConfig(parse, dict_config_env)
    config.TEST = dict_config_env.get('test', False)
    return config, config_nn, dict_config_env


def create_net(configs, max_batch=None, process_group=None, expert_data=None) -> Basenn:
    """create_net (runner/utils.py:59-170): the atari branches (two encoders -- SHARE_CNN_NET=False, the default -- or one
    shared prenet), the generic branches, and NETWORK_TYPE "ppo" or "gail"."""
    config, config_nn, config_env = configs['config'], configs['config_nn'], configs['config_env']
    if config_nn.NETWORK_TYPE == "gail":
        return _create_gail_net(config, config_nn, config_env, max_batch, process_group, expert_data)
    if config_nn.NETWORK_TYPE != "ppo":
        raise NotImplementedError("NETWORK_TYPE=%s does not exist in the reference either (nn/__init__.py:19-22)"
                                  % config_nn.NETWORK_TYPE)
    if config.TASK_TYPE != 'atari':
        return _create_generic_net(config, config_nn, config_env, max_batch, process_group)
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


def _create_gail_net(config, config_nn, config_env, max_batch, process_group, expert_data):
    """The gail branch (runner/utils.py:161-168): gail_critic = deepcopy(critic), generator = PPO(actor, critic, prenet),
    D_net = Discriminator(pre=deepcopy(prenet)), GAIL(generator, D_net, gail_critic).  The reference's version only works
    with a shared prenet (D_prenet is None otherwise and GAIL.py:68 fails); so does this one, loudly."""
    if not config_nn.SHARE_CNN_NET:
        raise NotImplementedError("NETWORK_TYPE='gail' needs SHARE_CNN_NET=True: the reference builds the discriminator's "
                                  "encoder as deepcopy(prenet) (runner/utils.py:164)")
    dim = config_nn.AC_INPUT_DIM
    actor = config_nn.ACTOR_CLASS(action_output_dim=config_nn.ACTION_OUTPUT_DIM, device=config_nn.DEVICE, last_input_dim=dim,
                                  soft_max_grid=config_nn.SOFT_MAX_GRID, nn_dtype=config_nn.MODULE_TENSOR_DTYPE)
    critic = Critic(device=config_nn.DEVICE, last_input_dim=dim)
    if config.TASK_TYPE == 'atari':
        prenet = AtariPreNet(config_env['int_frame_stack'], last_output_dim=dim, device=config_nn.DEVICE)
    elif config.TASK_TYPE in ("mujoco", "classical"):
        prenet = MLPPreNet(config_env.get('input_dim', 4), dim)
    elif config.TASK_TYPE in ("robot_nav", "gazebo_env", "real_env"):
        if config_env['ped_sim']['total'] > 0:
            prenet = NavPedPreNet(image_channel=config_env["image_batch"] + 3, last_output_dim=dim)
        else:
            prenet = NavPreNet(image_channel=config_env["image_batch"], last_output_dim=dim)
    else:
        raise NotImplementedError("task type %s has no network in the reference either" % config.TASK_TYPE)
    gail_critic = copy.deepcopy(critic)
    d_prenet = copy.deepcopy(prenet)
    # the generator is the operator-composed PPO for every encoder (the fused Atari iteration has no slot for a second value head)
    ppo_net = GenericPPO(actor, critic, prenet, None, config, config_nn, max_batch=max_batch, process_group=process_group)
    d_net = Discriminator(pre=d_prenet, config=config, config_nn=config_nn, max_batch=max_batch, expert_data=expert_data,
                          process_group=process_group)
    return GAIL(generator=ppo_net, discriminator=d_net, gail_critic=gail_critic)


def _create_generic_net(config, config_nn, config_env, max_batch, process_group):
    """mujoco / classical (MLPPreNet) and robot_nav / gazebo_env / real_env (nav encoders) branches of
    create_net (runner/utils.py:61-121), on the operator-composed GenericPPO."""
    dim = config_nn.AC_INPUT_DIM

    def heads(pre_a, pre_c):
        actor = config_nn.ACTOR_CLASS(action_output_dim=config_nn.ACTION_OUTPUT_DIM, device=config_nn.DEVICE,
                                      soft_max_grid=config_nn.SOFT_MAX_GRID, last_input_dim=dim, pre=pre_a,
                                      nn_dtype=config_nn.MODULE_TENSOR_DTYPE)
        return actor, Critic(device=config_nn.DEVICE, last_input_dim=dim, pre=pre_c)

    if config.TASK_TYPE in ("mujoco", "classical"):
        make = lambda: MLPPreNet(config_env.get('input_dim', 4), dim)
        if config_nn.SHARE_CNN_NET:
            actor, critic = heads(None, None)
            prenet = make()
        else:
            actor, critic = heads(make(), make())
            prenet = None
    elif config.TASK_TYPE in ("robot_nav", "gazebo_env", "real_env"):
        if config_nn.SHARE_CNN_NET:
            actor, critic = heads(None, None)
            if config_env['ped_sim']['total'] > 0:   # pedestrian map as extra image channels (utils.py:98-102)
                prenet = NavPedPreNet(image_channel=config_env["image_batch"] + 3, last_output_dim=dim)
            else:
                prenet = NavPreNet(image_channel=config_env["image_batch"], last_output_dim=dim)
        else:                                         # utils.py:110-121
            actor, critic = heads(NavPreNet1D(image_channel=3, last_output_dim=dim),
                                  NavPreNet1D(image_channel=3, last_output_dim=dim))
            prenet = None
    else:
        raise NotImplementedError("task type %s has no network in the reference either (runner/utils.py:144-145)"
                                  % config.TASK_TYPE)
    return PPO(actor, critic, prenet, None, config, config_nn, max_batch=max_batch, process_group=process_group)
