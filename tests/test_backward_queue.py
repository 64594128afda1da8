"""CPU tests of the learner-side gather step (server/backward.py: BackwardQueue.get, batch_logger, decode_train_blob) -- the
counterpart of USTC_lab/server/backward.py:30-62,145-151 fed by the >= 128-sample blobs of USTC_lab/agent/multiqueue.py:83-105.
The codec under it is the C codec of libddrl_hip.so (host code only: no GPU needed)."""
import queue

import numpy as np
import pytest

from ddrl4nav_amd.data import EasyBytes, Experience
from ddrl4nav_amd.server import BackwardQueue, batch_logger, decode_train_blob


def _piece(rng, n, frames_dtype=np.float64):
    fr = rng.integers(0, 256, size=(n, 4, 6, 6), dtype=np.uint8)
    states = [fr / 255.0 if frames_dtype == np.float64 else fr, rng.normal(size=(n, 3)).astype(np.float32)]
    return Experience(states=states, advs=rng.normal(size=n).astype(np.float32), actions=rng.integers(0, 6, n).astype(np.float32),
                      old_logps=rng.normal(size=n).astype(np.float32), values=rng.normal(size=(1, n)).astype(np.float32)), fr


def test_batch_logger_is_the_keywise_mean_of_the_first_dicts_keys():
    assert batch_logger([]) == {}
    got = batch_logger([{"RewardEpisode": 1.0, "Len": 10}, {"RewardEpisode": 4.0, "Len": 30, "extra": 7}])
    assert got == {"RewardEpisode": 2.5, "Len": 20.0} and all(isinstance(v, float) for v in got.values())
    with pytest.raises(KeyError):          # a later dict without a key of the first: the reference raises too (backward.py:36)
        batch_logger([{"a": 1.0}, {"b": 2.0}])


def test_get_gathers_whole_pieces_until_the_minimum_batch():
    """backward.py:48-62: pop until cur_size >= batch_size -- never fewer samples, whole pieces only, arrival order, empty logger dicts
    skipped; what was not needed stays queued for the next batch."""
    rng = np.random.default_rng(5)
    eb, q = EasyBytes("10.0.0.7"), BackwardQueue()
    sizes = [128, 150, 131, 160, 129, 144, 128, 182, 128, 140]          # >= 128 each (multiqueue.py:100); the first 8 hold 1,152
    pieces = [_piece(rng, n) for n in sizes]
    loggers = [{"RewardEpisode": float(i), "Len": 2.0 * i} if i % 3 else {} for i in range(len(sizes))]
    for (e, _), lg in zip(pieces, loggers):
        q.put_blob(eb, eb.encode_backward_data(e.get_xrapv(), lg))       # TrainingProcess.run's blob -> get_train_data
    exp, lg = q.get(1024)
    used = 8
    assert sum(sizes[:used - 1]) < 1024 <= sum(sizes[:used]) and len(exp) == sum(sizes[:used]) == 1152
    assert exp.states[0].dtype == np.float64 and exp.states[0].shape == (1152, 4, 6, 6)       # frames arrive as float64 (warputils.py:300)
    np.testing.assert_array_equal(np.rint(exp.states[0] * 255.0).astype(np.uint8), np.concatenate([f for _, f in pieces[:used]]))
    for k in ("advs", "actions", "old_logps"):
        np.testing.assert_array_equal(getattr(exp, k), np.concatenate([getattr(e, k) for e, _ in pieces[:used]]))
    np.testing.assert_array_equal(exp.values, np.concatenate([e.values for e, _ in pieces[:used]], axis=1))
    np.testing.assert_array_equal(exp.states[1], np.concatenate([e.states[1] for e, _ in pieces[:used]]))
    kept = [d for d in loggers[:used] if d]
    assert lg == {"RewardEpisode": float(np.mean([d["RewardEpisode"] for d in kept])), "Len": float(np.mean([d["Len"] for d in kept]))}
    assert q.q.qsize() == 2                                              # pieces 9 and 10 wait for the next batch
    with pytest.raises(queue.Empty):                                     # ... which is not complete yet
        q.get(1024, True, 0.05)
    # the two pieces the timed-out call had taken are not lost (the reference's local gather list would be): they head the next batch
    assert q.q.qsize() == 0
    more = [_piece(rng, n) for n in (300, 300, 200)]
    for e, _ in more:
        q.put((e, {}))
    exp2, lg2 = q.get(1024)
    assert len(exp2) == 128 + 140 + 800 and lg2 == {"RewardEpisode": 8.0, "Len": 16.0}     # piece 10's dict was empty (9 % 3 == 0), piece 9's is the only one
    np.testing.assert_array_equal(exp2.advs[:268], np.concatenate([pieces[8][0].advs, pieces[9][0].advs]))


def test_decode_train_blob_roundtrips_u8_and_float64_frames():
    rng = np.random.default_rng(6)
    eb = EasyBytes()
    for dt in (np.uint8, np.float64):
        e, fr = _piece(rng, 130, dt)
        got, lg = decode_train_blob(eb, eb.encode_backward_data(e.get_xrapv(), {"RewardEpisode": -3.5}))
        assert isinstance(got, Experience) and len(got) == 130 and lg == {"RewardEpisode": -3.5}
        assert got.states[0].dtype == dt
        np.testing.assert_array_equal(got.states[0], e.states[0])
        np.testing.assert_array_equal(got.values, e.values)
