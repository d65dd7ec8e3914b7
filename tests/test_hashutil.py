"""The digests of tests/hashutil.py: numpy (reference side) and torch (device side) forms agree, uniform and ragged."""
import numpy as np

import hashutil as hu


def test_digest_forms_agree():
    import torch
    rs = np.random.RandomState(0)
    w = rs.randint(0, 1 << 32, size=12 * 100, dtype=np.uint64).astype(np.uint32)
    a = hu.digest_uniform_np(w, 12)
    assert (a == hu.digest_ragged_np(w, np.arange(12) * 100, np.full(12, 100))).all()
    t = torch.from_numpy(w.view(np.int32))
    assert (a == hu.digest_uniform_torch(t, 12)).all()
    assert (a == hu.digest_ragged_torch(t, torch.arange(12) * 100, torch.full((12,), 100, dtype=torch.int64))).all()
    i16 = rs.randint(0, 1 << 16, size=600).astype(np.uint16)
    f = hu.digest_uniform_np(i16.astype(np.uint32), 6)
    assert (f == hu.digest_uniform_torch(torch.from_numpy(i16.view(np.int16)), 6)).all()
    assert (f == hu.digest_ragged_torch(torch.from_numpy(i16.view(np.int16)), torch.arange(6) * 100, torch.full((6,), 100, dtype=torch.int64), is_u16=True)).all()
    # sensitivity: one changed word, one transposition
    w2 = w.copy(); w2[517] ^= 1
    assert (hu.digest_uniform_np(w2, 12) != a).any()
    w3 = w.copy(); w3[10], w3[11] = w[11], w[10]
    assert (hu.digest_uniform_np(w3, 12) != a).any()
