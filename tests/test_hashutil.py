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


def test_reference_worker_rows_of_the_cubic_box_sweep(tmp_path, wl, oracle):
    """tests/ref_hash_worker.py in its `cubics@<box>:<paths>` mode (the reference side of the GPU box-sweep test): the rows of a
    sub-range equal the digests of the oracle's polylines computed here path by path, and the ranges of two workers concatenate."""
    import importlib
    import os
    import subprocess
    import sys
    here = os.path.dirname(os.path.abspath(__file__))
    paths, box = 300, 10.0
    outs = []
    for first, count in ((0, 200), (200, 100)):
        out = str(tmp_path / ("rows%d.npy" % first))
        subprocess.check_call([sys.executable, os.path.join(here, "ref_hash_worker.py"), "cubics@%g:%d" % (box, paths), str(first), str(count), out])
        outs.append(np.load(out))
    rows = np.concatenate(outs)
    assert rows.shape == (paths, 5)
    pyoracle = importlib.import_module("pyoracle")
    ps, d = wl.random_cubics(paths, seed=1234, box=box)
    r = pyoracle.flatten(ps, d, apply_transform=True)
    di = r.draw_info
    fv, nv = di["first_poly_vertex"].astype(np.int64), di["num_poly_vertices"].astype(np.int64)
    assert (rows[:, 4] == nv).all() and 4 < nv.mean() < 8  # ~4.7 segments per cubic at box 10
    want = hu.digest_ragged_np(r.poly.view(np.uint32).reshape(-1), 2 * fv, 2 * nv)
    assert (rows[:, :4] == want).all()
