"""ConfigNN: the hyper-parameter contract of the hot path.

Attribute names and default values are those of the reference's USTC_lab/config/config_nn.py
(line numbers in comments); the HIP kernels are parameterised by them through ddrl_config
(include/ddrl.h).  Editing this class is the flag system, as in the reference.
"""
import numpy
import torch


class ConfigNN:
    def __init__(self, dict_config_env: dict):
        from ddrl4nav_amd.nn import CategoricalActor, GaussionActor
        if dict_config_env['discrete_action']:                      # config_nn.py:10-14
            self.ACTOR_CLASS = CategoricalActor
            self.ACTION_OUTPUT_DIM = len(dict_config_env['discrete_actions'])
            self.ACTIONS_DIM = 1
        else:                                                        # config_nn.py:15-17
            self.ACTION_OUTPUT_DIM = self.ACTIONS_DIM = dict_config_env['act_dim']
            self.ACTOR_CLASS = GaussionActor

    NETWORK_TYPE = "ppo"            # :19
    USE_RND = False                 # :21
    AC_INPUT_DIM = 512              # :23
    DEVICE = 'cuda' if torch.cuda.is_available() else 'cpu'   # :25
    EXTRINSIC_DISCOUNT = 0.99       # :27
    LANDA = 0.95                    # :29
    LEARNING_RATE = 2e-4            # :32 (shared-encoder mode only)
    ACTOR_LEARNING_RATE = 5e-5      # :33
    CRITIC_LEARNING_RATE = 1e-3     # :34
    V_LOSS_THETA = 1.0              # :36
    ENTROPY_LOSS_THETA = 0.05       # :38
    PPO_CLIP = 0.2                  # :40
    DUEL_PPO_CLIP = 3               # :43
    TRAINING_ITER_TIME = 10         # :45
    TRAINING_MIN_BATCH = 1024       # :47
    SOFT_MAX_GRID = True            # :50
    CLIP_GRID = True                # :52
    CLIP_GRID_NUM = 0.5             # :53
    SMOOTH_L1_LOSS = False          # :55
    SHARE_CNN_NET = False           # :57
    GRAD_ACCUMULATION_STEP = 5      # :60
    EXTRINSIC_REWARD_COFF = 0       # :62
    HALF = False                    # :67
    MODULE_TENSOR_DTYPE = torch.float32
    MODULE_NUMPY_DTYPE = numpy.float32
    MODULE_BITS = 32
    # GAIL / RND switches the agent constructor reads (config_nn.py:92-126); both are off on this path
    GAN_VALUE_TRICK = True          # :90
    GAN_DISCOUNT = 0.99             # :91
    GAN_D_LEARNING_RATE = 5e-5      # :93
    GAN_D_EPOCH = 1                 # :95
    GAN_D_BATCH_SIZE = 128          # :97
    WGAN_CLIP_GRAD_NUM = 0.01       # :99
    D_REWARD_DECAY = 1 / 10000
    D_REWARD_COFF = 1
    RND_VALUE_TRICK = False
    RND_DISCOUNT = 0.99
    RND_REWARD_COFF = 1
    # :131 `GAN_D_EPOCH if NETWORK_TYPE == "gail" else TRAINING_ITER_TIME`, evaluated when the class body runs (i.e. with the
    # class default NETWORK_TYPE = "ppo"); runner.ini_config re-evaluates it for instances whose NETWORK_TYPE was changed
    MODEL_TO_REDIS_FREQUENCY = TRAINING_ITER_TIME
