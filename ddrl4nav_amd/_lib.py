"""ctypes binding of libddrl_hip.so (include/ddrl.h).  There is NO fallback: if the HIP
library is missing or a call fails this module raises."""
import ctypes
import os
from ctypes import (POINTER, Structure, byref, c_char, c_char_p, c_double, c_float, c_int32, c_int64, c_uint64,
                    c_void_p)

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "csrc", "libddrl_hip.so")

STATS_FLOATS = 8
ABI_VERSION = 3  # include/ddrl.h DDRL_ABI_VERSION this binding was written against


class DdrlError(RuntimeError):
    pass


class Config(Structure):
    """ddrl_config (include/ddrl.h) == the ConfigNN contract (reference config/config_nn.py)."""
    _fields_ = [
        ("n_actions", c_int32), ("in_channels", c_int32), ("max_batch", c_int32), ("share_cnn_net", c_int32),
        ("clip_grad", c_int32), ("clip_grad_norm", c_float), ("actor_lr", c_float), ("critic_lr", c_float),
        ("adam_beta1", c_float), ("adam_beta2", c_float), ("adam_eps", c_float), ("ppo_clip", c_float),
        ("dual_clip", c_float), ("v_loss_theta", c_float), ("ent_loss_theta", c_float),
        ("learning_rate", c_float), ("smooth_l1_loss", c_int32),
    ]


class EbArray(Structure):
    _fields_ = [("dtype", c_int32), ("ndim", c_int32), ("dims", c_int64 * 8), ("count", c_int64),
                ("data_offset", c_int64), ("nbytes", c_int64)]


class EbMsg(Structure):
    _fields_ = [("ip", c_int32 * 4), ("process_env_id", ctypes.c_uint32), ("payload_offset", c_int64),
                ("payload_len", c_int64)]


class ConvDesc(Structure):
    """ddrl_conv_desc (include/ddrl.h)."""
    _fields_ = [("n", c_int32), ("cin", c_int32), ("h", c_int32), ("w", c_int32), ("cout", c_int32), ("kh", c_int32),
                ("kw", c_int32), ("stride", c_int32), ("pad_h", c_int32), ("pad_w", c_int32), ("in_sn", c_int64),
                ("out_sn", c_int64)]


class HeadsDesc(Structure):
    """ddrl_heads_desc (include/ddrl.h)."""
    _fields_ = [("continuous", c_int32), ("n_actions", c_int32), ("shared", c_int32), ("reserved", c_int32),
                ("actor_w", c_int64), ("actor_b", c_int64), ("log_std", c_int64), ("critic_w", c_int64),
                ("critic_b", c_int64), ("n_params", c_int64)]


# name -> (restype, argtypes); every symbol include/ddrl.h declares
SIGNATURES = {
    "ddrl_abi_version": (c_int32, []),
    "ddrl_status_string": (c_char_p, [c_int32]),
    "ddrl_config_default": (c_int32, [POINTER(Config)]),
    "ddrl_param_count": (c_int32, [POINTER(Config), POINTER(c_int64), POINTER(c_int64)]),
    "ddrl_workspace_bytes": (c_int32, [POINTER(Config), POINTER(c_int64)]),
    "ddrl_ctx_create": (c_int32, [POINTER(Config), c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int64,
                                  POINTER(c_void_p)]),
    "ddrl_ctx_destroy": (c_int32, [c_void_p]),
    "ddrl_params_changed": (c_int32, [c_void_p]),
    "ddrl_get_step": (c_int32, [c_void_p, POINTER(c_int64)]),
    "ddrl_set_step": (c_int32, [c_void_p, c_int64]),
    "ddrl_forward": (c_int32, [c_void_p, c_void_p, c_int32, c_void_p, c_uint64, c_uint64, c_void_p, c_void_p,
                               c_void_p, c_void_p, c_void_p]),
    "ddrl_categorical_stats": (c_int32, [c_void_p, c_int32, c_int32, c_void_p, c_void_p, c_void_p, c_void_p]),
    "ddrl_categorical_sample": (c_int32, [c_void_p, c_int32, c_int32, c_uint64, c_uint64, c_void_p, c_void_p, c_void_p]),
    "ddrl_last_features": (c_int32, [c_void_p, c_int32, c_void_p, c_void_p, c_void_p]),
    "ddrl_gae": (c_int32, [c_void_p, c_void_p, c_void_p, c_int32, c_int32, c_float, c_float, c_void_p, c_void_p,
                           c_void_p]),
    "ddrl_episode_returns": (c_int32, [c_void_p, c_void_p, c_int32, c_int32, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    "ddrl_ppo_iter": (c_int32, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int32, c_int64,
                                c_void_p]),
    "ddrl_clip_adam_step": (c_int32, [c_void_p, c_void_p]),
    "ddrl_u8_table": (c_int32, [c_void_p, c_void_p]),
    "ddrl_debug_buffer": (c_int32, [c_void_p, c_int32, POINTER(c_void_p), POINTER(c_int64)]),
    "ddrl_debug_keep_activations": (c_int32, [c_void_p, c_int32]),
    "ddrl_ring_create": (c_int32, [c_int64, c_int32, POINTER(c_void_p)]),
    "ddrl_ring_destroy": (c_int32, [c_void_p]),
    "ddrl_ring_acquire": (c_int32, [c_void_p, POINTER(c_void_p), c_int32]),
    "ddrl_ring_commit": (c_int32, [c_void_p]),
    "ddrl_ring_pop_to_device": (c_int32, [c_void_p, c_void_p, c_int64, c_void_p, c_int32]),
    "ddrl_ring_pending": (c_int32, [c_void_p, POINTER(c_int32)]),
    "ddrl_eb_array_bytes": (c_int32, [c_int32, c_int32, POINTER(c_int64), POINTER(c_int64)]),
    "ddrl_eb_encode_array": (c_int32, [c_int32, c_int32, POINTER(c_int64), c_void_p, c_void_p, c_int64, POINTER(c_int64)]),
    "ddrl_eb_scan": (c_int32, [c_void_p, c_int64, POINTER(EbArray), c_int32, POINTER(c_int32)]),
    "ddrl_eb_forward_header": (c_int32, [POINTER(c_int32), ctypes.c_uint32, c_uint64, c_void_p]),
    "ddrl_eb_scan_forward_states": (c_int32, [c_void_p, c_int64, POINTER(EbMsg), c_int32, POINTER(c_int32)]),
    "ddrl_eb_frames_to_u8": (c_int32, [c_void_p, c_int64, c_int32, c_void_p, c_int64, POINTER(c_int64), POINTER(c_int64)]),
    "ddrl_eb_scan_backward": (c_int32, [c_void_p, c_int64, POINTER(c_int64), POINTER(c_int64), POINTER(c_int64),
                                        POINTER(c_int64), POINTER(c_int64)]),
    "ddrl_eb_put_u64": (c_int32, [c_uint64, c_void_p]),
    "ddrl_timer_create": (c_int32, [POINTER(c_void_p)]),
    "ddrl_timer_destroy": (c_int32, [c_void_p]),
    "ddrl_timer_start": (c_int32, [c_void_p, c_void_p]),
    "ddrl_timer_stop": (c_int32, [c_void_p, c_void_p]),
    "ddrl_timer_elapsed_ms": (c_int32, [c_void_p, POINTER(c_float)]),
    "ddrl_profile_enable": (c_int32, [c_void_p, c_int32]),
    "ddrl_profile_read": (c_int32, [c_void_p, c_void_p, c_void_p, c_void_p, c_int32, POINTER(c_int32)]),
    "ddrl_op_conv_out_shape": (c_int32, [POINTER(ConvDesc), POINTER(c_int32), POINTER(c_int32)]),
    "ddrl_op_conv_pack_floats": (c_int32, [POINTER(ConvDesc), POINTER(c_int64)]),
    "ddrl_op_conv_pack": (c_int32, [POINTER(ConvDesc), c_void_p, c_void_p, c_void_p]),
    "ddrl_op_conv_forward": (c_int32, [POINTER(ConvDesc), c_void_p, c_void_p, c_void_p, c_int32, c_void_p, c_void_p, c_void_p, c_void_p]),
    "ddrl_op_conv_scratch_floats": (c_int32, [POINTER(ConvDesc), POINTER(c_int64)]),
    "ddrl_op_conv_dgrad_pooled": (c_int32, [POINTER(ConvDesc), c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    "ddrl_op_conv_wgrad_pooled": (c_int32, [POINTER(ConvDesc), c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                            c_void_p, c_void_p, c_void_p]),
    "ddrl_op_sample_amax": (c_int32, [c_void_p, c_int64, c_int32, c_int32, c_void_p, c_void_p]),
    "ddrl_op_conv_pooled_uses_scales": (c_int32, [POINTER(ConvDesc)]),
    "ddrl_op_conv_has_forward_pool": (c_int32, [POINTER(ConvDesc)]),
    "ddrl_op_conv_forward_pool": (c_int32, [POINTER(ConvDesc), c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    "ddrl_op_conv_dgrad": (c_int32, [POINTER(ConvDesc), c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    "ddrl_op_conv_ws_floats": (c_int32, [POINTER(ConvDesc), POINTER(c_int64)]),
    "ddrl_op_conv_wgrad": (c_int32, [POINTER(ConvDesc), c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    "ddrl_op_maxpool2_forward": (c_int32, [c_void_p, c_int64, c_int32, c_int32, c_void_p, c_void_p]),
    "ddrl_op_maxpool2_relu_backward": (c_int32, [c_void_p, c_void_p, c_int64, c_int32, c_int32, c_void_p, c_void_p]),
    "ddrl_op_maxpool2_forward_idx": (c_int32, [c_void_p, c_int64, c_int32, c_int32, c_void_p, c_void_p, c_void_p]),
    "ddrl_op_maxpool2_backward_idx": (c_int32, [c_void_p, c_void_p, c_int64, c_int32, c_int32, c_void_p, c_void_p]),
    "ddrl_op_linear_pack_floats": (c_int32, [c_int32, c_int32, POINTER(c_int64), POINTER(c_int64)]),
    "ddrl_op_linear_pack": (c_int32, [c_void_p, c_int32, c_int32, c_void_p, c_void_p, c_void_p]),
    "ddrl_op_linear_uses_planes": (c_int32, [c_int32, c_int32, c_int32]),
    "ddrl_op_row_amax": (c_int32, [c_void_p, c_int64, c_int32, c_int32, c_void_p, c_int32, c_void_p]),
    "ddrl_op_linear_forward": (c_int32, [c_void_p, c_int64, c_void_p, c_void_p, c_int32, c_void_p, c_int64, c_int32,
                                         c_int32, c_int32, c_void_p, c_void_p, c_void_p]),
    "ddrl_op_linear_dgrad": (c_int32, [c_void_p, c_int64, c_void_p, c_void_p, c_int64, c_void_p, c_int64, c_int32, c_int32,
                                       c_int32, c_void_p, c_void_p, c_void_p, c_int32, c_int32, c_void_p]),
    "ddrl_op_linear_ws_floats": (c_int32, [c_int32, c_int32, c_int32, POINTER(c_int64)]),
    "ddrl_op_linear_wgrad": (c_int32, [c_void_p, c_int64, c_void_p, c_int64, c_void_p, c_void_p, c_void_p, c_int32, c_int32,
                                       c_int32, c_void_p, c_void_p, c_void_p]),
    "ddrl_op_heads_ws_floats": (c_int32, [POINTER(HeadsDesc), c_int32, POINTER(c_int64)]),
    "ddrl_op_heads_act": (c_int32, [POINTER(HeadsDesc), c_void_p, c_void_p, c_void_p, c_int32, c_void_p, c_uint64, c_uint64,
                                    c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    "ddrl_op_heads_loss": (c_int32, [POINTER(HeadsDesc), POINTER(Config), c_void_p, c_void_p, c_void_p, c_int32, c_void_p,
                                     c_void_p, c_void_p, c_void_p, c_int64, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    "ddrl_op_clip_adam_ws_bytes": (c_int32, [POINTER(c_int64)]),
    "ddrl_op_clip_adam": (c_int32, [POINTER(Config), c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_int64, c_int32,
                                    c_int64, c_void_p, c_void_p]),
    "ddrl_op_relu_mask": (c_int32, [c_void_p, c_int64, c_void_p, c_int64, c_int32, c_int32, c_void_p]),
    "ddrl_op_accumulate": (c_int32, [c_void_p, c_void_p, c_int64, c_void_p]),
    "ddrl_comm_unique_id": (c_int32, [c_void_p]),
    "ddrl_comm_info": (c_int32, [c_void_p, c_int64, POINTER(c_int32)]),
    "ddrl_comm_create": (c_int32, [c_void_p, c_int32, c_int32, POINTER(c_void_p)]),
    "ddrl_comm_destroy": (c_int32, [c_void_p]),
    "ddrl_allreduce_f32": (c_int32, [c_void_p, c_void_p, c_int64, c_void_p]),
    "ddrl_broadcast_f32": (c_int32, [c_void_p, c_void_p, c_int64, c_int32, c_void_p]),
    "ddrl_grad_allreduce": (c_int32, [c_void_p, c_void_p, c_void_p]),
    "ddrl_params_broadcast": (c_int32, [c_void_p, c_void_p, c_int32, c_void_p]),
    "ddrl_encoder_forward": (c_int32, [c_void_p, c_void_p, c_int32, c_void_p]),
    "ddrl_encoder_backward": (c_int32, [c_void_p, c_void_p, c_int32, c_void_p]),
    "ddrl_encoder_buffers": (c_int32, [c_void_p, POINTER(c_void_p), POINTER(c_void_p)]),
    "ddrl_op_value_head_forward": (c_int32, [c_void_p, c_void_p, c_void_p, c_int64, c_int32, c_void_p, c_void_p]),
    "ddrl_op_value_head_ws_floats": (c_int32, [POINTER(c_int64)]),
    "ddrl_op_value_head_loss": (c_int32, [POINTER(Config), c_int32, c_void_p, c_void_p, c_void_p, c_int64, c_int32, c_void_p,
                                          c_int64, c_void_p, c_int64, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    "ddrl_op_wgan_terms": (c_int32, [c_void_p, c_int64, c_int32, c_int64, c_float, c_void_p, c_int64, c_int32, c_void_p,
                                     c_int32, c_void_p]),
    "ddrl_op_colsum": (c_int32, [c_void_p, c_int64, c_int32, c_int32, c_void_p, c_void_p]),
    "ddrl_grad_buckets_enable": (c_int32, [c_void_p]),
    "ddrl_grad_bucket_count": (c_int32, [c_void_p, POINTER(c_int32)]),
    "ddrl_grad_bucket_info": (c_int32, [c_void_p, c_int32, POINTER(c_int64), POINTER(c_int64), POINTER(c_int32)]),
    "ddrl_grad_bucket_wait": (c_int32, [c_void_p, c_int32, c_void_p]),
    "ddrl_grad_buckets_begin": (c_int32, [c_void_p, c_void_p, c_void_p, POINTER(c_int32)]),
    "ddrl_grad_bucket_wait_last": (c_int32, [c_void_p, c_void_p]),
    "ddrl_grad_allreduce_overlapped": (c_int32, [c_void_p, c_void_p, c_void_p, c_void_p]),
    "ddrl_op_clip_rmsprop": (c_int32, [c_void_p, c_void_p, c_void_p, c_int64, c_float, c_double, c_float, c_float, c_void_p,
                                       c_void_p]),
}

_lib = None


def load():
    """Load the shared library (once).  Raises DdrlError when it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise DdrlError("HIP extension %s is missing: run `python -c 'import __graft_entry__ as g; g.build()'` "
                        "(or `make -C ddrl4nav_amd/csrc`).  There is no CPU fallback." % LIB_PATH)
    # PyTorch ships its own libamdhip64; load it FIRST so that this library's libamdhip64.so.7
    # dependency resolves to the same runtime (device pointers and streams are shared between
    # the two).  Loading in the other order leaves two HIP runtimes in one process.
    import torch  # noqa: F401
    lib = ctypes.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError if the symbol is not exported
        fn.restype = res
        fn.argtypes = args
    got = lib.ddrl_abi_version()
    if got != ABI_VERSION:
        raise DdrlError("libddrl_hip.so ABI version %d, this binding needs %d: rebuild with `make -C ddrl4nav_amd/csrc`"
                        % (got, ABI_VERSION))
    _lib = lib
    _warn_removed_switches()
    return lib


# environment switches of earlier rounds that no longer do anything: a launch script that still sets one keeps running,
# so say once what replaced it instead of ignoring it silently
REMOVED_SWITCHES = {
    "DDRL_ALLREDUCE_OVERLAP": "use DDRL_ALLREDUCE=overlap (or rccl,overlap)",
    "DDRL_ENC_STREAMS": "set net.encoder_streams = False on the GenericPPO instance",
    "DDRL_NAV_F32": "the f32-input direct-convolution family was removed",
    "DDRL_FIRST_F32": "the f32-input direct-convolution family was removed",
    "DDRL_LIN_F32": "the f32-input dense kernels are chosen by shape only",
    "DDRL_POOL_UNFUSED": "pooling is fused wherever the layer has a plane kernel",
    "DDRL_POOL_BWD_UNFUSED": "pooling is fused wherever the layer has a plane kernel",
    "DDRL_SCALES_PER_OP": "set encoder.producer_amax = False (nn/generic.py PRODUCER_AMAX) for the pre-pass arrangement",
    "DDRL_C1D_GATHER": "the laser branch always runs csrc/c1d.hip",
}


def _warn_removed_switches():
    import warnings
    for k, hint in REMOVED_SWITCHES.items():
        if k in os.environ:
            warnings.warn("%s is set but was removed and has no effect: %s" % (k, hint), RuntimeWarning, stacklevel=3)


def check(status):
    if status != 0:
        msg = load().ddrl_status_string(status)
        raise DdrlError("libddrl_hip: %s (status %d)" % (msg.decode() if msg else "?", status))


def default_config(**overrides):
    cfg = Config()
    check(load().ddrl_config_default(byref(cfg)))
    for k, v in overrides.items():
        if not hasattr(cfg, k):
            raise AttributeError(k)
        setattr(cfg, k, v)
    return cfg
