"""EAVDataSplit: bit-exact against index vectors captured from the imported
reference (EAV_datasplit.py:26-40) - tests/golden/datasplit.npz."""
import os

import numpy as np

from eav_amd.datasplit import EAVDataSplit


def _g(golden_dir):
    return np.load(os.path.join(golden_dir, "datasplit.npz"))


def test_split_indices_bit_exact(golden_dir):
    g = _g(golden_dir)
    x = np.arange(400)
    for i in range(6):
        y = g[f"y{i}"]
        for h in (40, 56):
            tr, try_, te, tey = EAVDataSplit(x, y).get_split(h_idx=h)
            for got, key in ((tr, f"tr{i}_{h}"), (te, f"te{i}_{h}"), (try_, f"try{i}_{h}"), (tey, f"tey{i}_{h}")):
                assert got.dtype == g[key].dtype
                assert np.array_equal(got, g[key]), key
            itr, ite = EAVDataSplit(x, y).split_indices(h)
            assert np.array_equal(itr, g[f"tr{i}_{h}"]) and np.array_equal(ite, g[f"te{i}_{h}"])


def test_split_sizes_default_and_70_30(golden_dir):
    y = _g(golden_dir)["y3"]
    tr, _, te, _ = EAVDataSplit(np.arange(400), y).get_split()
    assert tr.shape == (200,) and te.shape == (200,)
    tr, _, te, _ = EAVDataSplit(np.arange(400), y).get_split(56)
    assert tr.shape == (280,) and te.shape == (120,)


def test_squeeze_behaviour(golden_dir):
    g = _g(golden_dir)
    x3 = np.arange(400 * 1 * 3, dtype=np.float32).reshape(400, 1, 3)
    tr, _, te, _ = EAVDataSplit(x3, g["y1"]).get_split(40)
    assert tr.shape == g["x3_tr"].shape == (200, 3)
    assert np.array_equal(tr, g["x3_tr"]) and np.array_equal(te, g["x3_te"])


def test_ragged_and_empty_classes():
    # a class with fewer than h_idx members contributes all of them to train, none to test
    y = np.array([0] * 50 + [1] * 10 + [2] * 45 + [4] * 41)
    x = np.arange(len(y))
    tr, try_, te, tey = EAVDataSplit(x, y).get_split(40)
    assert np.array_equal(np.bincount(try_, minlength=5), [40, 10, 40, 0, 40])
    assert np.array_equal(np.bincount(tey, minlength=5), [10, 0, 5, 0, 1])
    assert np.array_equal(np.sort(np.concatenate([tr, te])), x)


def test_get_loaders():
    y = np.repeat(np.arange(5), 80)
    x = np.random.RandomState(0).randn(400, 4).astype(np.float32)
    ltr, lte = EAVDataSplit(x, y, batch_size=32).get_loaders()
    assert len(ltr.dataset) == 200 and len(lte.dataset) == 200
    xb, yb = next(iter(lte))
    assert xb.shape == (32, 4) and yb.dtype.is_floating_point is False
