"""CPU tests of the data-format helpers against a literal, loop-for-loop evaluation of the
reference lines they restate (main.py:1325, 1336-1346) on random episodes."""
import numpy as np
import pytest

import cgs_amd  # noqa: F401
from cgs_amd import dataformat as df


def ref_trunk(reward):   # main.py:1325 evaluated as written
    return np.array([True] + [np.sum(reward[max(0, i - 35):i]) == 0 for i in range(1, len(reward))])


def ref_discount(reward, gamma):   # main.py:1340-1344 evaluated as written
    add = len(reward)
    local = reward.copy()
    for i in range(2, add + 1):
        last = gamma * local[-i + 1]
        local[-i] = min(local[-i] + last, 1)
    return local


def test_trunk_mask_and_discounting_match_reference_lines():
    rs = np.random.RandomState(0)
    for _ in range(20):
        T = rs.randint(1, 200)
        reward = (rs.rand(T) > 0.97).astype(np.float64)
        np.testing.assert_array_equal(df.trunk_mask(reward), ref_trunk(reward))
        for g in (0.98, 0.95, 0.5):
            np.testing.assert_array_equal(df.discounted_rewards(reward, g), ref_discount(reward, g))
    r = np.array([0, 0, 0, 1.0])
    np.testing.assert_allclose(df.discounted_rewards(r, 0.5), [0.125, 0.25, 0.5, 1.0])
    assert df.discounted_rewards(np.array([1.0, 1.0]), 0.98).max() == 1.0   # clipped at 1


def test_build_write_read_roundtrip(tmp_path):
    rs = np.random.RandomState(1)
    eps = []
    for _ in range(5):
        T = rs.randint(50, 120)
        eps.append((rs.randint(0, 256, (T, 64, 64, 3)).astype(np.uint8), (rs.rand(T) > 0.96).astype(np.float64)))
    X, Y, I = df.build_dataset(eps, size=300)
    assert X.dtype == np.uint8 and Y.shape == (7, len(X)) and not Y[5:].any() and I.dtype == np.uint16 and len(X) <= 300
    assert set(np.unique(Y[0])) <= {0.0, 1.0} and (Y[1:5] >= Y[0]).all() and Y.max() <= 1.0
    # after the trunk filter no kept frame (except an episode's first) follows a reward within 35 frames
    path = df.dataset_path(datasize=len(X), data_dir=str(tmp_path) + "/")
    assert path.endswith(f"Treechop-trunk-{len(X)}-[0.98-0.97-0.96-0.95].pickle")
    df.write_dataset(path, X, Y, I)
    X2, Y2, I2 = df.read_dataset(path)
    np.testing.assert_array_equal(X, X2); np.testing.assert_array_equal(Y, Y2); np.testing.assert_array_equal(I, I2)
    with pytest.raises(ValueError):
        df.write_dataset(path, X.astype(np.float32), Y, I)
