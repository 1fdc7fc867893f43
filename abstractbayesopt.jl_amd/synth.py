"""Counter-based synthetic design matrices: SplitMix64 → U[0,1)^d, so the GPU box
regenerates bit-identical inputs from a seed instead of shipping arrays.

    X[i,c] = u(seed=1, i·d + c)      Z[j,c] = u(seed=2, j·d + c)
    y_i    = Σ_c sin(2π x_ic)/√d + σ_n·n(seed=3, i)       (Box–Muller on the same stream)
"""
import numpy as np

_M64 = np.uint64(0xFFFFFFFFFFFFFFFF)


def splitmix64(seed: int, counter: np.ndarray) -> np.ndarray:
    with np.errstate(over="ignore"):
        z = (counter.astype(np.uint64) + np.uint64(seed) * np.uint64(0x632BE59BD9B4E019)) * np.uint64(0x9E3779B97F4A7C15)
        z = z + np.uint64(0x9E3779B97F4A7C15)
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        z = z ^ (z >> np.uint64(31))
    return z


def uniform(seed: int, start: int, count: int) -> np.ndarray:
    """U[0,1) doubles for counters start .. start+count−1 (53-bit mantissa)."""
    ctr = np.arange(start, start + count, dtype=np.uint64)
    return (splitmix64(seed, ctr) >> np.uint64(11)).astype(np.float64) * (1.0 / 9007199254740992.0)


def points(seed: int, n: int, d: int, first: int = 0) -> np.ndarray:
    """n points in [0,1)^d, point-major; `first` = index of the first point (for sharding)."""
    return uniform(seed, first * d, n * d).reshape(n, d)


def normal(seed: int, start: int, count: int) -> np.ndarray:
    u1 = uniform(seed, 2 * start, 2 * count)[0::2]
    u2 = uniform(seed, 2 * start, 2 * count)[1::2]
    return np.sqrt(-2.0 * np.log(1.0 - u1)) * np.cos(2.0 * np.pi * u2)


def objective(X: np.ndarray, noise_std: float = 0.0, seed: int = 3) -> np.ndarray:
    n, d = X.shape
    y = np.sin(2.0 * np.pi * X).sum(axis=1) / np.sqrt(d)
    if noise_std > 0.0:
        y = y + noise_std * normal(seed, 0, n)
    return y


def standardized_problem(n: int, d: int, noise_std: float = 0.0):
    """(X, y) with y standardised to mean 0 / std 1 as the BO driver does (src/BO_utils.jl:44-64)."""
    X = points(1, n, d)
    y = objective(X, noise_std)
    y = (y - y.mean()) / y.std(ddof=1)
    return X, y
