"""CPU restatement of the int8-residue contraction engine (abstractbayesopt.jl_amd/csrc/ozaki.hip) — TEST INFRASTRUCTURE ONLY.

Only tests/ may import this.  It restates, in NumPy and Python big integers, what the engine computes for posterior_var's product
V = L⁻¹·K_XZ (reference: src/surrogates/StandardGP.jl:377-379 → [upstream AbstractGPs] diag_Xt_invA_X, an fp64 BLAS product there):
fixed-point images of the two operands, residues modulo pairwise-coprime moduli ≤ 256, exact integer products per modulus,
Chinese-remainder reconstruction with the constants split for fp64.  The scheme is Ozaki, Uchino, Imamura, "Ozaki scheme II"
(2025), restated from the published algorithm.  The reference holds no vectors for this (it has no such engine): the pin is
Python's exact integer arithmetic — `exact_dot` below — which the reconstruction must reproduce digit for digit."""
import math

import numpy as np


def moduli(n: int):
    """the first n pairwise-coprime integers descending from 256"""
    out, c = [], 256
    while len(out) < n:
        if all(math.gcd(c, m) == 1 for m in out):
            out.append(c)
        c -= 1
    return out


def plan(n: int):
    """constants of an n-modulus plan exactly as oz_make_plan derives them (41-bit heads on a common grid)"""
    ps = moduli(n)
    P = math.prod(ps)
    t = P.bit_length() - 41
    s1, s2, c26 = [], [], []
    for p in ps:
        Mi = P // p
        s = Mi * pow(Mi, -1, p)
        hi = (s >> t) << t
        s1.append(float(hi))
        s2.append(float(s - hi))
        c = (1 << 26) % p
        c26.append(float(c - p if 2 * c > p else c))
    P1 = (P >> t) << t
    return {"p": ps, "P": P, "s1": np.array(s1), "s2": np.array(s2), "c26": np.array(c26), "P1": float(P1), "P2": float(P - P1),
            "invP": 1.0 / float(P), "eP": P.bit_length() - 3}


def sym_residue(x: np.ndarray, p: int) -> np.ndarray:
    """symmetric residue of integer-valued doubles |x| < 2^53 through the split x = xh·2^26 + xl (the device's five operations)"""
    xh = np.rint(x * 2.0 ** -26)
    xl = x - xh * 2.0 ** 26
    c = (1 << 26) % p
    c = c - p if 2 * c > p else c
    t = xh * c + xl
    q = np.rint(t * (1.0 / p))
    return (t - q * p).astype(np.int64)


def row_scales(W: np.ndarray, eP: int) -> np.ndarray:
    """s_i = min(eP − 53 − e(L1_i), 52 − e(max_i)), e = frexp exponent (oz_rowscale_kernel)"""
    l1 = np.abs(W).sum(1)
    mx = np.abs(W).max(1)
    s = np.zeros(W.shape[0], dtype=np.int64)
    ok = mx > 0
    s[ok] = np.minimum(eP - 53 - np.frexp(l1[ok])[1], 52 - np.frexp(mx[ok])[1])
    return s


def k_scale(kmax: float) -> int:
    return 52 - math.frexp(kmax * (1.0 + 1e-12))[1]


def contract(W: np.ndarray, K: np.ndarray, kmax: float, n: int = 14):
    """V = W·K (W [N][N] lower-triangular, K [N][M]) the engine's way.  Returns V and the fixed-point images (Wq, Kq, s_i, sK)."""
    pl = plan(n)
    sK = k_scale(kmax)
    si = row_scales(W, pl["eP"])
    Wq = np.rint(W * 2.0 ** si[:, None].astype(np.float64))
    Kq = np.rint(K * 2.0 ** sK)
    c1 = np.zeros((W.shape[0], K.shape[1]))
    c2 = np.zeros_like(c1)
    for l, p in enumerate(pl["p"]):
        a, b = sym_residue(Wq, p), sym_residue(Kq, p)
        assert np.abs(a).max() <= 128 and np.abs(b).max() <= 128
        u = sym_residue((a @ b).astype(np.float64), p).astype(np.float64)
        c1 += u * pl["s1"][l]
        c2 += u * pl["s2"][l]
    Q = np.rint((c1 + c2) * pl["invP"])
    cp = (c1 - Q * pl["P1"]) + (c2 - Q * pl["P2"])
    V = cp * 2.0 ** (-(si[:, None] + sK)).astype(np.float64)
    return V, (Wq, Kq, si, sK)


def exact_dot(Wq_row, Kq_col) -> int:
    """the pin: Python integers"""
    return sum(int(a) * int(b) for a, b in zip(Wq_row, Kq_col))
