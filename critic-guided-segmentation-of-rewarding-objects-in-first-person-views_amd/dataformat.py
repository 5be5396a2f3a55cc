"""Data format either side of the training path, pure numpy:

* the gz-pickle the reference trains from: ``(X uint8 [N,64,64,3], Y float64 [7,N], I uint16 [N])`` at
  ``runs/data/straight/{env}-{mode}-{datasize}-[{gammas}].pickle`` (main.py:1277-1284, 1353-1354);
* the episode -> rows labelling of ``Handler.collect_data`` (main.py:1317-1348): the "trunk" frame filter and the
  clipped discounted rewards for each gamma.

MineRL download / decoding itself stays out of scope: ``build_dataset`` takes already-decoded episodes
``(pov uint8 [T,64,64,3], reward float [T])``."""
import gzip
import os
import pickle
from typing import Iterable, Sequence, Tuple

import numpy as np

DATA_DIR = "runs/data/straight/"
Y_ROWS = 7      # main.py:1296: the target array always has 7 rows (0/1 reward + up to 6 discount rows; unused rows stay zero) -- G11


def dataset_path(envname="Treechop", datamode="trunk", datasize=100000, gammas="0.98-0.97-0.96-0.95", data_dir=DATA_DIR) -> str:
    return data_dir + f"{envname}-{datamode}-{datasize}-[{gammas}].pickle"


def write_dataset(path: str, X: np.ndarray, Y: np.ndarray, I: np.ndarray) -> None:
    if X.dtype != np.uint8 or X.ndim != 4 or X.shape[1:] != (64, 64, 3):
        raise ValueError(f"X must be uint8 [N,64,64,3], got {X.dtype} {X.shape}")
    if Y.ndim != 2 or Y.shape[1] != len(X) or len(I) != len(X):
        raise ValueError("Y must be [rows,N] and I [N]")
    os.makedirs(os.path.dirname(path) or ".", exist_ok=True)
    with gzip.GzipFile(path, "wb") as fp:
        pickle.dump((X, Y, I), fp)


def read_dataset(path: str) -> Tuple[np.ndarray, np.ndarray, np.ndarray]:
    with gzip.open(path, "rb") as fp:
        X, Y, I = pickle.load(fp)
    return X, Y, I


def trunk_mask(reward: np.ndarray) -> np.ndarray:
    """main.py:1325: keep frame 0 and every frame whose previous (up to) 35 frames carry no reward."""
    reward = np.asarray(reward)
    keep = np.ones(len(reward), dtype=bool)
    for i in range(1, len(reward)):
        keep[i] = np.sum(reward[max(0, i - 35):i]) == 0
    return keep


def discounted_rewards(reward01: np.ndarray, gamma: float) -> np.ndarray:
    """main.py:1340-1344: backward pass r[t] = min(r[t] + gamma * r[t+1], 1) over a 0/1 reward row."""
    out = np.array(reward01, dtype=np.float64)
    for i in range(len(out) - 2, -1, -1):
        out[i] = min(out[i] + gamma * out[i + 1], 1.0)
    return out


def build_dataset(episodes: Iterable[Tuple[np.ndarray, np.ndarray]], size: int, mode: str = "trunk",
                  gammas: Sequence[float] = (0.98, 0.97, 0.96, 0.95)):
    """Episode list -> (X, Y, I) exactly as collect_data fills them (main.py:1293-1350), for modes "trunk" and "begin"."""
    X = np.zeros((size, 64, 64, 3), dtype=np.uint8)
    if 1 + len(gammas) > Y_ROWS:
        raise ValueError(f"at most {Y_ROWS - 1} discount factors fit the reference's [{Y_ROWS}, N] target array (main.py:1296)")
    Y = np.zeros((Y_ROWS, size), dtype=np.float64)
    I = np.zeros(size, dtype=np.uint16)
    run = 0
    add = 0
    for pov, reward in episodes:
        pov, reward = np.asarray(pov), np.asarray(reward, dtype=np.float64)
        if mode == "begin":
            add = int(np.argmax(reward > 0)) + 1 if reward.any() else add
            if add > 1000:
                continue
            reward = reward[:add]
        elif mode == "trunk":
            keep = trunk_mask(reward)
            pov, reward = pov[keep], reward[keep]
        add = min(size - run, len(pov))
        r01 = (reward[:add] > 0).astype(np.float64)
        X[run:run + add] = pov[:add]
        Y[0, run:run + add] = r01
        I[run:run + add] = np.arange(len(pov))[:add]
        for k, g in enumerate(gammas):
            Y[k + 1, run:run + add] = discounted_rewards(r01, float(g))
        run += add
        if run >= size:
            break
    return X[:run], Y[:, :run], I[:run]
