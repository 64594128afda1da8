"""EasyBytes wire codec: the C codec + Python mirror against byte
streams and decoded arrays produced by the reference's own EasyBytes (golden F7, F9).  CPU only."""
import numpy as np
import pytest

from ddrl4nav_amd.data import EasyBytes


def test_kat_literals_from_the_reference_smoke_block(golden):
    """The literal arrays of easybytes.py:175-198."""
    g = golden("f7_codec")
    s = EasyBytes()
    enc = s.encode_data([np.array([[1, 2, 3], [4, 5, 6]], dtype=np.uint8), np.array([4, 6], dtype=np.float32)])
    assert np.array_equal(np.frombuffer(enc, np.uint8), g["enc"])
    fwd = s.encode_forward_states(1, [np.array([[1, 2, 3, 4]], dtype=np.float32)])
    assert np.array_equal(np.frombuffer(fwd, np.uint8), g["fwd"])
    a, b = s.decode_data(enc)
    assert a.dtype == np.uint8 and a.shape == (2, 3) and b.dtype == np.float32 and list(b) == [4.0, 6.0]


def test_forward_states_batched_messages(golden):
    g = golden("f9_easybytes")
    s = EasyBytes("10.2.3.4")
    assert np.array_equal(np.frombuffer(s.machine_bytes, np.uint8), g["machine_bytes"])
    msg = s.encode_forward_states(3, [g["fr_a"] / 255.0, g["vec_a"]]) + s.encode_forward_states(70000, [g["fr_b"] / 255.0, g["vec_b"]])
    assert np.array_equal(np.frombuffer(msg, np.uint8), g["msg"])
    ids, states = s.decode_forward_states(g["msg"].tobytes())
    assert ids == list(g["ids"]) == ["10.2.3.4_3", "10.2.3.4_70000"]
    assert states[0].dtype == np.float64 and np.array_equal(states[0], g["dec_frames"])
    assert np.array_equal(states[1], g["dec_vec"])
    # hot helper: float64 frames -> the exact uint8 bytes, straight into a flat (pinned) buffer
    out = np.zeros(5 * 4 * 6 * 6 + 7, np.uint8)
    n, per = s.frames_to_u8(g["msg"].tobytes(), out)
    assert (n, per) == (5, 144)
    assert np.array_equal(out[:720].reshape(5, 4, 6, 6), np.concatenate([g["fr_a"], g["fr_b"]]))
    with pytest.raises(Exception):
        s.frames_to_u8(g["msg"].tobytes(), np.zeros(100, np.uint8))  # destination too small -> loud


def test_forward_replies_and_backward_blob(golden):
    g = golden("f9_easybytes")
    s = EasyBytes("10.2.3.4")
    replies = s.encode_forward_return_data([g["actions"], g["logps"], g["values"]], [2, 3])
    assert np.array_equal(np.frombuffer(replies[0], np.uint8), g["reply0"])
    assert np.array_equal(np.frombuffer(replies[1], np.uint8), g["reply1"])
    a, lp, v = s.decode_data(replies[1])
    assert a.shape == (3,) and v.shape == (1, 3, 1)  # values are sliced on axis 1 (easybytes.py:98-100)
    stats = {"RewardEpisode": 1.5, "steps": 7}
    blob = s.encode_backward_data([[g["st0"], g["st1"]], g["o0"], g["o1"], g["o2"], g["o3"]], stats)
    tail = int(g["tail_len"])
    assert np.array_equal(np.frombuffer(blob[:-tail], np.uint8), g["blob_head"])
    states, other4, d = s.decode_backward_data(blob)
    assert d == stats and states[1].dtype == np.float16 and np.array_equal(states[0], g["st0"])
    assert [o.shape for o in other4] == [(4,), (4,), (4,), (1, 4)] and np.array_equal(other4[3], g["o3"])
    # straight into the learner's container
    from ddrl4nav_amd.data import Experience
    exp = Experience(states, *other4)
    assert len(exp) == 4


def test_all_four_dtypes_roundtrip_and_errors():
    s = EasyBytes()
    rng = np.random.default_rng(0)
    arrs = [rng.integers(0, 256, size=(3, 2), dtype=np.uint8), rng.normal(size=(2, 2, 2)).astype(np.float16),
            rng.normal(size=7).astype(np.float32), rng.normal(size=(1, 1, 1, 3)), np.zeros((0, 4), np.float32)]
    out = s.decode_data(s.encode_data(arrs))
    for a, b in zip(arrs, out):
        assert a.dtype == b.dtype and a.shape == b.shape and np.array_equal(a, b)
    with pytest.raises(ValueError):
        s.encode_data([np.zeros(3, np.int32)])  # unsupported dtype -> ValueError (easybytes.py:42-44)
    with pytest.raises(Exception):
        s.decode_data(b"\x00\x03\x00\x00\x00\x05")  # truncated record
    # uint8 and float32 frames also map back to bytes
    fr = rng.integers(0, 256, size=(2, 4, 5, 5), dtype=np.uint8)
    for payload in (fr, (fr / 255.0).astype(np.float32), (fr / 255.0).astype(np.float64)):
        buf = np.zeros(fr.size, np.uint8)
        n, per = s.frames_to_u8(s.encode_forward_states(0, [payload]), buf)
        assert n == 2 and per == 100 and np.array_equal(buf.reshape(fr.shape), fr)


def test_codec_fuzz_roundtrip_and_truncation():
    """Property test of the C codec behind the Python mirror: any list of supported arrays round-trips
    exactly; any strict prefix of a valid stream is rejected with an exception (never a crash or a
    silently shorter result)."""
    from hypothesis import given, settings, strategies as st
    dtypes = [np.uint8, np.float16, np.float32, np.float64]

    @st.composite
    def arrays(draw):
        dt = draw(st.sampled_from(dtypes))
        shape = tuple(draw(st.lists(st.integers(0, 5), min_size=1, max_size=4)))
        seed = draw(st.integers(0, 2 ** 31 - 1))
        rng = np.random.default_rng(seed)
        if dt is np.uint8:
            return rng.integers(0, 256, size=shape, dtype=np.uint8)
        return rng.normal(size=shape).astype(dt)

    s = EasyBytes("192.168.1.77")

    @settings(max_examples=150, deadline=None)
    @given(st.lists(arrays(), min_size=1, max_size=5), st.integers(0, 10 ** 6))
    def check(arrs, cut_seed):
        enc = s.encode_data(arrs)
        out = s.decode_data(enc)
        assert len(out) == len(arrs)
        for a, b in zip(arrs, out):
            assert a.dtype == b.dtype and a.shape == b.shape and np.array_equal(a, b)
        cut = cut_seed % len(enc)
        if cut > 0:
            try:
                short = s.decode_data(enc[:cut])
            except Exception:
                return
            # a prefix that happens to end on a record boundary decodes to fewer arrays: still consistent
            assert len(short) < len(arrs) and all(np.array_equal(x, y) for x, y in zip(short, arrs))

    check()


_MALFORMED = r"""
import struct, sys
import numpy as np
from ddrl4nav_amd.data import EasyBytes
from ddrl4nav_amd._lib import DdrlError
s = EasyBytes()
bad = 0
def expect_loud(fn, *a):
    global bad
    try:
        fn(*a)
    except (DdrlError, KeyError, ValueError):
        bad += 1
big = [0x7FFFFFFFFFFFFFF0, 0x7FFFFFFFFFFFFFFF, 0xFFFFFFFFFFFFFFF0, 0x8000000000000000, 21]
out = np.zeros(64, np.uint8)
for L in big:
    hdr = struct.pack(">Q4HI", L, 127, 0, 0, 1, 7)
    expect_loud(s.decode_forward_states, hdr)
    expect_loud(s.frames_to_u8, hdr, out)
    expect_loud(s.decode_backward_data, struct.pack(">QQ", L, 0))
    expect_loud(s.decode_backward_data, struct.pack(">QQ", 0, L))
# array record whose dims multiply past 2^64 (wraps to the stated count without the guard)
rec = struct.pack(">hII", 1, 0, 8) + struct.pack(">8I", *([0x10000] * 8))
expect_loud(s.decode_data, rec)
rec = struct.pack(">hII", 4, 0xFFFFFFFF, 1) + struct.pack(">I", 0xFFFFFFFF)
expect_loud(s.decode_data, rec)
# a benign header still decodes
ok = s.encode_forward_states(1, [np.zeros((1, 4), np.float32)])
ids, st = s.decode_forward_states(ok)
assert ids == ["127.0.0.1_1"] and st[0].shape == (1, 4)
print("LOUD", bad)
"""


def test_malformed_lengths_fail_loudly_instead_of_reading_out_of_bounds():
    """Lengths on the wire are untrusted big-endian u64 / u32 values: near-INT64_MAX lengths and
    overflowing dim products must come back as an error status, not as a wild read (run in a child
    process so that a regression shows up as a failed test and not as a dead pytest)."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, "-c", _MALFORMED], cwd=root, capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, (r.returncode, r.stderr[-2000:])
    assert "LOUD 22" in r.stdout, r.stdout
