"""BaseConfig: process-level constants the hot path's callers read (mirror of the attribute
names in USTC_lab/config/base_config.py; Redis hosts are kept only so that key names resolve)."""
import logging
import math

import torch

from ddrl4nav_amd.config.utils import game_type


class BaseConfig:
    def __init__(self, parse, config_env: dict):
        if config_env['env_type'] == 'gym':                      # base_config.py:10-13
            self.TASK_TYPE = game_type(config_env['env_name'])
        else:
            self.TASK_TYPE = config_env['env_type']
        self.ENV_NUM = int(config_env['env_num'])
        for attr, key, default in (("TRAINER_REDIS_HOST", "th", "127.0.0.1"), ("TRAINER_REDIS_PORT", "tp", 6379),
                                   ("PREDICTOR_REDIS_HOST", "ph", "127.0.0.1"), ("PREDICTOR_REDIS_PORT", "pp", 6379),
                                   ("MIDDLE_REDIS_HOST", "mh", "127.0.0.1"), ("MIDDLE_REDIS_PORT", "mp", 6379),
                                   ("CONTROL_REDIS_HOST", "ch", "127.0.0.1"), ("CONTROL_REDIS_PORT", "cp", 6379)):
            setattr(self, attr, getattr(parse, key, default))
        self.SAVE_MODEL_PATH = getattr(parse, "model_dir", "./model")
        self.PREDICTING_MIN_BATCH = math.ceil(self.ENV_NUM / 2)  # :29
        ip = getattr(parse, "ip", "127.0.0.1")
        self.MACHINE_IP = "127.0.0.1" if ip == "localhost" else ip
        assert len(self.MACHINE_IP.split(".")) == 4
        self.TASK_NAME = getattr(parse, "task", "ddrl") + "-" + self.MACHINE_IP
        self.TEST = config_env.get('test', False)
        # The reference's Discriminator reads config.ACTIONS_DIM and config.GAN_D_MLP_LIST (nn/GAIL.py:23,47); neither is
        # defined by its config classes.  ACTIONS_DIM follows ConfigNN (config_nn.py:14,16); the score network defaults to
        # one hidden layer on cat(512 features, action).
        if config_env.get('discrete_action', True):
            self.ACTIONS_DIM = 1
        else:
            self.ACTIONS_DIM = int(config_env.get('act_dim', 1))
        self.GAN_D_MLP_LIST = [(512 + self.ACTIONS_DIM, 256, "relu"), (256, 1, None)]

    SYNC = False                 # :37
    PLAY_MODE = False
    DEMONSTRATE_MODE = False
    MIMIC_START = False
    MIMIC_START_LOAD_PATH = "./mimic/"   # base_config.py:49 points into a developer's home directory
    LOAD_CHECKPOINT = False
    LOAD_CHECKPOINT_PATH = ""
    LOAD_EPISODE = 0
    USE_RND = False              # :58
    PREDICTORS = 1               # :61
    TRAINERS = 1                 # :63
    TIME_MAX = 256               # :66
    TIME_OUT = 10                # :69
    DEVICE = 'cuda' if torch.cuda.is_available() else 'cpu'
    SAVE_MODELS = True           # :91
    SAVE_FREQUENCY = 2000        # :93
    INFO_LEVEL = logging.INFO
    LOG_REWARD_FREQUENCY = 1
    LOG_LOSS_FREQUENCY = 1
    # redis key names (:113-139), kept for drop-in callers
    TRAINING_DATA_KEY = "TRAIN"
    EXIT_KEY = "EXIT"
    PREDICTING_STATES_KEY = "FORWARD_STATES"
    PRE_ACTIONS_KEY = "PRE_ACTION_{}"
    MODULE_KEY = "MODEL"
    TRAIN_LOCK_KEY = "LOCK_KEY"
    ENV_NUM_DICT_KEY = "ENV_DICT"
    UPDATE_TAG_KEY = "UPDATE_TAG"
    RENDER = 0
