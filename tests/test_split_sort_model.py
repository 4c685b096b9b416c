"""The split sort's ranking rule (csrc/split_sort.hpp), restated in numpy (no GPU): every 512-slot piece of a list sorted on its own
by (key, position), an element's final position = its position in its run + for every OTHER run the number of keys a binary search
finds below it — or, in runs of EARLIER list positions, not above it.  That is the total order (key, position) — the order
sort/item_rank_score.go:26-32's comparison gives once ties are pinned to the input position — whatever the ties, also
when equal keys straddle run boundaries and when the last run is short."""
import numpy as np
import pytest

RUN = 512


def split_sort_positions(keys):
    n = len(keys)
    parts = (n + RUN - 1) // RUN
    runs = []
    for p in range(parts):
        idx = np.arange(p * RUN, min(n, (p + 1) * RUN))
        order = np.lexsort((idx, keys[idx]))               # by key, then position
        runs.append((keys[idx][order], idx[order]))
    out = np.empty(n, dtype=np.int64)
    for p, (rk, ri) in enumerate(runs):
        rank = np.arange(len(rk))
        for r, (ok, _) in enumerate(runs):
            if r == p:
                continue
            # lower run index = lower positions: its equal keys come first (count keys <= mine); later runs: keys < mine
            rank = rank + np.searchsorted(ok, rk, side="right" if r < p else "left")
        out[rank] = ri
    return out


@pytest.mark.parametrize("n", [1, 2, 511, 512, 513, 1025, 5000, 8192])
@pytest.mark.parametrize("levels", [1, 2, 17, 10**9])
def test_rank_rule_is_the_stable_order(n, levels):
    rng = np.random.default_rng(n * 31 + levels % 97)
    keys = rng.integers(0, levels, n).astype(np.uint64)
    if n > 600:
        keys[500:530] = 3                                    # equal keys across the first run boundary
    got = split_sort_positions(keys)
    want = np.lexsort((np.arange(n), keys))
    assert np.array_equal(got, want)
