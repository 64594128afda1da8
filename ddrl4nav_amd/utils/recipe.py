"""Deterministic weight recipe for the default Pong PPO net.

The reference initialises its layers with torch defaults (``nn.Conv2d`` / ``nn.Linear``:
uniform in +-1/sqrt(fan_in) for weight and bias; reference: USTC_lab/nn/atari_encoder.py:16-21,
USTC_lab/nn/actor.py:86, USTC_lab/nn/critic.py:11).  This recipe draws from the same
distribution with a counter-based integer hash, so that the GPU box, the oracle and the
golden-vector generator all build bit-identical weights without shipping a 13 MB blob.

Tensor order == ``named_parameters()`` order of the reference ``PPO`` module
(USTC_lab/nn/base.py:62-63 walks it for the Redis blob).
"""
import numpy as np

_M64 = np.uint64(0xFFFFFFFFFFFFFFFF)


def param_specs(num_inputs=4, n_actions=6, shared=False):
    """[(name, shape, fan_in)] in the reference's ``named_parameters()`` order; ``shared`` =
    SHARE_CNN_NET=True (one ``prenet``, pre-less actor / critic, runner/utils.py:136-143)."""
    def enc(prefix):
        return [
            (prefix + "conv1.weight", (32, num_inputs, 8, 8), num_inputs * 64),
            (prefix + "conv1.bias", (32,), num_inputs * 64),
            (prefix + "conv2.weight", (64, 32, 4, 4), 32 * 16),
            (prefix + "conv2.bias", (64,), 32 * 16),
            (prefix + "conv3.weight", (64, 64, 3, 3), 64 * 9),
            (prefix + "conv3.bias", (64,), 64 * 9),
            (prefix + "linear.weight", (512, 3136), 3136),
            (prefix + "linear.bias", (512,), 3136),
        ]
    if shared:
        return enc("prenet.") + [("actor.actor_linear.weight", (n_actions, 512), 512),
                                 ("actor.actor_linear.bias", (n_actions,), 512),
                                 ("critic.critic_linear.weight", (1, 512), 512),
                                 ("critic.critic_linear.bias", (1,), 512)]
    specs = enc("actor.pre.")
    specs += [("actor.actor_linear.weight", (n_actions, 512), 512),
              ("actor.actor_linear.bias", (n_actions,), 512)]
    specs += [("critic.critic_linear.weight", (1, 512), 512),
              ("critic.critic_linear.bias", (1,), 512)]
    specs += enc("critic.pre.")
    return specs


def _splitmix64(x):
    x = (x + np.uint64(0x9E3779B97F4A7C15)) & _M64
    z = x
    z = ((z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)) & _M64
    z = ((z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)) & _M64
    return z ^ (z >> np.uint64(31))


def hash_uniform(seed, stream, n):
    """n float32 values in [0,1) (24-bit mantissa) from (seed, stream, index): the WEIGHT recipe the
    golden fixtures were generated with (stream = tensor index, far below 2^24 -- only the low 24
    bits of `stream` take part, so this is not the samplers' generator; see sample_uniform)."""
    with np.errstate(over="ignore"):
        idx = np.arange(n, dtype=np.uint64)
        base = _splitmix64(np.uint64(seed) ^ (np.uint64(stream) << np.uint64(40)))
        z = _splitmix64(base + idx)
    return ((z >> np.uint64(40)).astype(np.float32) * np.float32(1.0 / (1 << 24))).astype(np.float32)


def sample_uniform(seed, stream, n):
    """The samplers' counter stream (csrc/common.h:hash_uniform): n float32 values in [0,1) from
    (seed, stream, index) with ALL 64 bits of `stream` mixed in, so that streams which differ only
    in high bits (rollout counter, draw index) are independent."""
    with np.errstate(over="ignore"):
        idx = np.arange(n, dtype=np.uint64)
        base = _splitmix64(_splitmix64(np.uint64(int(seed) & 0xFFFFFFFFFFFFFFFF)) ^ np.uint64(int(stream) & 0xFFFFFFFFFFFFFFFF))
        z = _splitmix64(base + idx)
    return ((z >> np.uint64(40)).astype(np.float32) * np.float32(1.0 / (1 << 24))).astype(np.float32)


def make_weights(seed=0, num_inputs=4, n_actions=6, shared=False):
    """dict name -> float32 ndarray, uniform in +-1/sqrt(fan_in)."""
    out = {}
    for ti, (name, shape, fan_in) in enumerate(param_specs(num_inputs, n_actions, shared)):
        n = int(np.prod(shape))
        u = hash_uniform(seed, ti, n)
        bound = np.float32(1.0 / np.sqrt(np.float64(fan_in)))
        out[name] = ((u * np.float32(2.0) - np.float32(1.0)) * bound).astype(np.float32).reshape(shape)
    return out


def flatten(weights, num_inputs=4, n_actions=6, shared=None):
    """Concatenate into the flat fp32 arena (reference blob order)."""
    if shared is None:
        shared = any(k.startswith("prenet.") for k in weights)
    return np.concatenate([weights[n].reshape(-1)
                           for n, _, _ in param_specs(num_inputs, n_actions, shared)]).astype(np.float32)


def hash_weights(named_shapes, seed=0):
    """Deterministic weights for ANY module tree: ``named_shapes`` = [(name, shape)] in
    named_parameters() order.  Weights and biases: uniform in +-1/sqrt(fan_in) like torch's default
    init (a bias takes the fan-in of the weight before it); ``log_std`` tensors: -0.5 + uniform +-0.2."""
    out, fan_in = {}, 1
    for ti, (name, shape) in enumerate(named_shapes):
        shape = tuple(int(d) for d in shape)
        n = int(np.prod(shape)) if shape else 1
        u = hash_uniform(seed, 1000 + ti, n)
        if name.endswith("log_std"):
            out[name] = (np.float32(-0.5) + (u * np.float32(2.0) - np.float32(1.0)) * np.float32(0.2)).astype(np.float32).reshape(shape)
            continue
        if len(shape) > 1:
            fan_in = int(np.prod(shape[1:]))
        bound = np.float32(1.0 / np.sqrt(np.float64(fan_in)))
        out[name] = ((u * np.float32(2.0) - np.float32(1.0)) * bound).astype(np.float32).reshape(shape)
    return out
