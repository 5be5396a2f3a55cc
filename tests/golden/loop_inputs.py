"""Synthetic inputs of the loop-level captures G9 / G10 / G11 (make_golden_loops.py), shared by the generator (build container, next to the
reference) and the tests (CPU here, GPU box): seeded numpy only, no reference code, nothing imported from /root/reference."""
import numpy as np

SEED_P1, SEED_P2 = 91, 92          # RNG seeds set at the top of critic_pipe / segmentation_training
DATASIZE, TESTSIZE = 3072, 1024


def synthetic_frames(n, seed):
    """Three brightness populations (dark / middle / bright noise frames with a few flat patches) and a target row that follows the
    brightness: a critic can learn the split in one epoch, and its values leave gaps to put the two thresholds in."""
    rs = np.random.RandomState(seed)
    pop = rs.randint(0, 3, n)
    gain = np.array([0.12, 0.5, 0.95])[pop] * (0.9 + 0.2 * rs.rand(n))
    X = (rs.randint(0, 256, (n, 64, 64, 3)) * gain[:, None, None, None]).astype(np.uint8)
    for i in range(0, n, 7):
        y0, x0 = rs.randint(0, 40, 2)
        X[i, y0:y0 + 20, x0:x0 + 24] = rs.randint(0, 256, 3).astype(np.uint8)
    Y = rs.rand(7, n)
    Y[1] = np.clip(np.array([0.04, 0.5, 0.96])[pop] + 0.03 * rs.randn(n), 0, 1)
    I = (np.arange(n) % 2 ** 16).astype(np.uint16)
    return X, Y, I



def synthetic_eval_set(n, seed):
    """A stand-in for red-trees/X.npy + Y.npy (main.py:920-925): noise frames with one red-ish box each; Y is an RGB label image whose
    all-channels-non-zero pixels are the ground truth (some frames lose a channel: np.all(..., -1) then drops their object)."""
    rs = np.random.RandomState(seed)
    X = rs.randint(0, 256, (n, 64, 64, 3)).astype(np.uint8)
    X[::3] = (X[::3] * 0.3).astype(np.uint8)
    Yrgb = np.zeros((n, 64, 64, 3), dtype=np.uint8)
    for i in range(n):
        y0, x0 = rs.randint(0, 44, 2)
        h, w = rs.randint(6, 20, 2)
        Yrgb[i, y0:y0 + h, x0:x0 + w] = 255
        X[i, y0:y0 + h, x0:x0 + w, 0] = 220          # the "object" is red-ish: masks and saliency are not constant
    Yrgb[5::11, :, :, 1] = 0
    return X, Yrgb


def synthetic_episodes(seed=11, n_eps=9):
    """(name, pov uint8 [T,64,64,3], reward float [T]) episodes: rewards in bursts (the trunk filter drops the 35 frames after each), one
    episode without any reward, one shorter than the filter window."""
    rs = np.random.RandomState(seed)
    eps = []
    for e in range(n_eps):
        T = int(rs.randint(20, 260)) if e != 3 else 12
        reward = np.zeros(T)
        if e != 5:
            for t in rs.randint(0, T, max(1, T // 60)):
                reward[t:t + int(rs.randint(1, 4))] = rs.choice([1.0, 2.0])
        pov = rs.randint(0, 256, (T, 64, 64, 3)).astype(np.uint8)
        eps.append((f"v3_episode_{e}", pov, reward))
    return eps
