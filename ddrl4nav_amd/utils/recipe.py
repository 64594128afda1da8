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


def _area_weights(dst, src):
    """[dst, src] INTEGER weights of exact area averaging (cv2.INTER_AREA's arithmetic for a non-integer
    shrink): with g = gcd(dst, src) a destination cell spans src / g units of 1 / (dst / g) source pixels, and
    entry (i, j) is the overlap of cell i with source pixel j in those units (rows sum to src / g)."""
    g = int(np.gcd(dst, src))
    a, b = src // g, dst // g          # cell i = [a i, a i + a), pixel j = [b j, b j + b)
    w = np.zeros((dst, src), np.float64)
    for i in range(dst):
        for j in range(src):
            w[i, j] = max(0, min(a * i + a, b * j + b) - max(a * i, b * j))
    return w, a


def pong_frames(seed, n, stack=4, chunk=1024):
    """Synthetic Pong observations, uint8 [n, stack, 84, 84], shaped like what the reference's frame
    wrapper emits before its ``/ 255.0`` (env/gym_env/wrapper/warputils.py:274-301: grayscale, rows
    34..193 of the 210 x 160 screen, area-resized to 84 x 84): background 87, a 4 x 16 paddle of gray
    148 (left) and 147 (right), a 2 x 4 ball of gray 236, drawn on the 160 x 160 playfield and shrunk with
    exact area averaging (integer-valued float64 products: bit-reproducible on any BLAS), so paddle and ball
    edges carry the in-between grays a real frame has.  The ``stack`` frames of a sample are consecutive: the
    ball moves with a per-sample velocity, the paddles drift.  Mostly-flat images like these are the realistic
    input of the encoder; dense random bytes are its worst case for time, not for
    numerics."""
    rng = np.random.default_rng(int(seed))
    w, a = _area_weights(84, 160)
    out = np.empty((n, stack, 84, 84), np.uint8)
    for c0 in range(0, n, chunk):
        m = min(chunk, n - c0)
        py_l = rng.integers(0, 160 - 16, size=m)
        py_r = rng.integers(0, 160 - 16, size=m)
        dpy_l = rng.integers(-4, 5, size=m)
        dpy_r = rng.integers(-4, 5, size=m)
        bx = rng.integers(20, 140, size=m)
        by = rng.integers(4, 152, size=m)
        vx = rng.integers(-6, 7, size=m)
        vy = rng.integers(-6, 7, size=m)
        no_ball = rng.random(m) < 0.1        # between points the ball is off screen
        field = np.full((m, stack, 160, 160), 87.0, np.float64)
        for f in range(stack):
            for i in range(m):
                yl = int(np.clip(py_l[i] + f * dpy_l[i], 0, 144))
                yr = int(np.clip(py_r[i] + f * dpy_r[i], 0, 144))
                field[i, f, yl:yl + 16, 16:20] = 148.0
                field[i, f, yr:yr + 16, 140:144] = 147.0
                if not no_ball[i]:
                    x = int(np.clip(bx[i] + f * vx[i], 0, 158))
                    y = int(np.clip(by[i] + f * vy[i], 0, 156))
                    field[i, f, y:y + 4, x:x + 2] = 236.0
        small = np.matmul(np.matmul(w, field), w.T)            # integers < 2^53: exact in any summation order
        q = np.floor((small + (a * a) // 2) / (a * a))           # round half up, exact
        out[c0:c0 + m] = q.astype(np.uint8)
    return out
