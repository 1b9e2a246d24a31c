"""Host side of babyai_amd.demos: the per-chunk scan that finds every stream's first solved episode in the history a device
rollout returns (scripts/make_agent_demos.py:84-123: a failed episode or a bot crash moves the stream on to its next level,
the first solved one -- within `filter_steps` if set -- is the demonstration).  Pure numpy: checked here against a plain
per-stream loop, across chunk boundaries."""
import numpy as np
import pytest

from babyai_amd.demos import scan_chunk


def plain(done, gave, rew, filter_steps):
    T, n = done.shape
    out = np.full((n, 2), -1, np.int64)
    for i in range(n):
        start = 0
        for t in range(T):
            if done[t, i]:
                if not gave[t, i] and rew[t, i] > 0 and (not filter_steps or t - start + 1 <= filter_steps):
                    out[i] = (start, t)
                    break
                start = t + 1
    return out


@pytest.mark.parametrize("filter_steps", [0, 7])
@pytest.mark.parametrize("chunk", [1, 5, 16, 64])
def test_scan_chunk_matches_a_plain_per_stream_loop(chunk, filter_steps):
    rng = np.random.RandomState(chunk + 100 * filter_steps)
    n, T = 97, 192
    done = rng.rand(T, n) < 0.09
    done[:, 0] = False                                         # a stream that never finishes stays open
    gave = (rng.rand(T, n) < 0.25) & done
    rew = np.where(done & ~gave & (rng.rand(T, n) < 0.45), 0.37, 0.0).astype(np.float32)
    last_done = np.full(n, -1, np.int32)
    span = np.full((n, 2), -1, np.int64)
    open_ = np.ones(n, bool)
    for g0 in range(0, T - T % chunk, chunk):
        scan_chunk(done[g0:g0 + chunk].astype(np.uint8), gave[g0:g0 + chunk].astype(np.uint8), rew[g0:g0 + chunk], g0, filter_steps,
                   last_done, open_, span)
    want = plain(done[:T - T % chunk], gave, rew, filter_steps)
    assert np.array_equal(span, want)
    assert np.array_equal(open_, want[:, 0] < 0)
    assert open_[0]
