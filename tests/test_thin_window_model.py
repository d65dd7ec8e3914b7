"""Model of k_flatten_thin's index arithmetic (csrc/vgx_flatten.hip), in numpy: the 256-ary search a workgroup starts with, the staged
LDS window (256 entries at a time, ended by the first block that reaches past the chunk), the per-lane search bounded by the number of
draws the chunk touches, and the hand-over of the owner draw from chunk to chunk. The kernel itself is checked on the GPU
(tests/test_gpu_parity.py); this pins the invariants its comments state -- every command instance finds its draw, no window entry that
was not loaded for the chunk is ever read -- for any mix of path lengths, chunk sizes and grids."""
import numpy as np
import pytest

STALE = -12345  # what an entry left over from an earlier chunk would look like (reading it must never happen)


def _first_owner(pref, nd, key, T):
    lo, hi = 0, nd  # pref[lo] <= key < pref[hi]
    rounds = 0
    while hi - lo > 1:
        step = (hi - lo + T - 1) // T
        idx = np.minimum(lo + (np.arange(T) + 1) * step, hi)
        cnt = int((pref[idx] <= key).sum())
        assert cnt < T  # the last sample is hi's (or beyond it): never <= key
        lo, hi = lo + cnt * step, min(lo + (cnt + 1) * step, hi)
        rounds += 1
    return lo, rounds


def _run(rs, nd, maxc, T, items, grid):
    chunk = T * items
    cnts = rs.randint(1, maxc + 1, size=nd)  # every draw of a thin set has at least one command
    pref = np.concatenate([[0], np.cumsum(cnts)]).astype(np.int64)
    total = int(pref[-1])
    owner = np.full(total, -1, np.int64)
    nchunks = (total + chunk - 1) // chunk
    per = (nchunks + grid - 1) // grid
    big = np.iinfo(np.int64).max
    for b in range(grid):
        ch0, ch1 = b * per, min(b * per + per, nchunks)
        if ch0 >= ch1:
            continue
        dcur, rounds = _first_owner(pref, nd, ch0 * chunk, T)
        assert dcur == int(np.searchsorted(pref, ch0 * chunk, side="right") - 1)
        assert rounds <= max(1, int(np.ceil(np.log(max(nd, 2)) / np.log(T))) + 1)
        sp = np.full(chunk + 1, STALE, np.int64)
        for ch in range(ch0, ch1):
            c0, key_next = ch * chunk, ch * chunk + chunk
            sp[:] = STALE
            sp[0] = pref[dcur]
            next_own = 0
            for b0 in range(0, chunk, T):
                i = b0 + 1 + np.arange(T)
                idx = dcur + i
                v = np.where(idx <= nd, pref[np.minimum(idx, nd)], big)
                sp[i] = v
                c = int((v <= key_next).sum())
                next_own += c
                if c < T:
                    break
            assert sp[0] <= c0 < sp[1]
            for ci in range(c0, min(c0 + chunk, total)):
                lo, hi = 0, next_own
                while lo < hi:
                    mid = (lo + hi + 1) >> 1
                    assert sp[mid] != STALE
                    if sp[mid] <= ci:
                        lo = mid
                    else:
                        hi = mid - 1
                assert owner[ci] == -1
                owner[ci] = dcur + lo
            dcur += next_own
    assert np.array_equal(owner, np.repeat(np.arange(nd), cnts))


@pytest.mark.parametrize("seed", range(6))
def test_every_command_instance_finds_its_draw(seed):
    rs = np.random.RandomState(seed)
    for _ in range(40):
        _run(rs, int(rs.randint(1, 700)), int(rs.choice([1, 2, 7, 40, 300])), int(rs.choice([2, 4, 16])), int(rs.choice([1, 2, 4])), int(rs.choice([1, 3, 8, 64])))


def test_polylines_of_a_thousand_commands_need_one_window_block():
    """BASELINE configs[3]'s shape: a chunk of 1024 command instances touches one or two draws -- the first block of 256 entries ends the
    window, and the per-lane search is one or two steps."""
    rs = np.random.RandomState(1)
    _run(rs, 40, 1, 16, 4, 8)  # one command per draw: every block is full, next_own == chunk
    nd, T, items = 64, 256, 4
    pref = np.arange(nd + 1, dtype=np.int64) * 1001
    for ch in range(0, 60):
        c0 = ch * T * items
        dcur = int(np.searchsorted(pref, c0, side="right") - 1)
        idx = dcur + 1 + np.arange(T)
        v = np.where(idx <= nd, pref[np.minimum(idx, nd)], np.iinfo(np.int64).max)
        assert int((v <= c0 + T * items).sum()) in (1, 2)
