"""CPU tests (-m "not gpu") of the int8-residue engine's host side: the constants libabo_hip.so derives (abo_test_oz_plan, no GPU
call) against Python big-integer arithmetic, and the scheme itself — restated in oracle/ozaki_oracle.py — against exact integer dot
products and a long-double product of the fp64 operands."""
import ctypes as C
import math

import numpy as np
import pytest

import abstractbayesopt.jl_amd as abo
from oracle import ozaki_oracle as Z


@pytest.mark.parametrize("n", [8, 10, 12, 13, 14, 15, 16])
def test_library_plan_constants_against_big_integers(n):
    p = (C.c_int32 * 16)()
    tab = (C.c_double * 64)()
    scal = (C.c_double * 3)()
    eP = C.c_int32()
    abo._lib.check(abo._lib.lib().abo_test_oz_plan(n, p, tab, scal, C.byref(eP)))
    ref = Z.plan(n)
    assert list(p)[:n] == ref["p"] == Z.moduli(n)
    assert all(math.gcd(a, b) == 1 for i, a in enumerate(ref["p"]) for b in ref["p"][:i])
    t = np.array(tab).reshape(4, 16)[:, :n]
    np.testing.assert_array_equal(t[0], 1.0 / np.array(ref["p"], dtype=np.float64))
    np.testing.assert_array_equal(t[1], ref["c26"])
    np.testing.assert_array_equal(t[2], ref["s1"])
    np.testing.assert_array_equal(t[3], ref["s2"])
    assert scal[0] == ref["P1"] and scal[1] == ref["P2"] and scal[2] == ref["invP"] and eP.value == ref["eP"]
    # the heads really are on a 41-bit grid, so that 16 terms of |U| ≤ 128 sum exactly in fp64, and the split is exact enough
    P = ref["P"]
    tgrid = P.bit_length() - 41
    for l, q in enumerate(ref["p"]):
        s = (P // q) * pow(P // q, -1, q)
        assert int(t[2][l]) % (1 << tgrid) == 0 and int(t[2][l]) < (1 << (tgrid + 41))
        assert s % q == 1 and all(s % o == 0 for o in ref["p"] if o != q)
        assert abs(int(t[2][l]) + int(t[3][l]) - s) <= (1 << max(tgrid - 53, 0))
    assert (1 << eP.value) <= P // 4 < (1 << (eP.value + 1))


def test_unsupported_moduli_counts_are_refused():
    lib = abo._lib.lib()
    buf = (C.c_double * 64)()
    for n in (0, 1, 7, 17, -3):
        assert lib.abo_test_oz_plan(n, (C.c_int32 * 16)(), buf, (C.c_double * 3)(), C.byref(C.c_int32())) == abo._lib.ABO_EINVAL
    assert lib.abo_set_contraction(None, 5, 0) == abo._lib.ABO_EINVAL
    assert lib.abo_set_contraction(None, abo._lib.CONTRACT_INT8, 7) == abo._lib.ABO_EINVAL
    assert lib.abo_set_contraction(None, abo._lib.CONTRACT_AUTO, 0) == abo._lib.ABO_OK


def test_symmetric_residue_split_is_exact():
    rng = np.random.default_rng(3)
    x = np.rint(rng.uniform(-1, 1, 20000) * 2.0 ** rng.integers(0, 53, 20000))
    x = np.concatenate([x, [0.0, 2.0 ** 53 - 1, -(2.0 ** 53 - 1), 2.0 ** 26, 2.0 ** 26 - 1, 127.0, 128.0, -128.0]])
    for p in Z.moduli(16):
        r = Z.sym_residue(x, p)
        want = np.array([int(v) % p for v in x])
        want = np.where(want > p // 2, want - p, want)
        if p == 256:      # any representative whose low byte is the residue
            assert np.all((r - want) % 256 == 0) and np.abs(r).max() <= 128
        else:
            np.testing.assert_array_equal(r, want)
            assert np.abs(r).max() <= (p - 1) // 2


@pytest.mark.parametrize("n,tol", [(14, 3e-15), (16, 3e-15), (12, 3e-9), (10, 3e-4)])
def test_scheme_reproduces_exact_integer_products(n, tol):
    rng = np.random.default_rng(n)
    N, M = 96, 40
    W = np.tril(rng.standard_normal((N, N)) * np.exp(-0.05 * np.abs(np.subtract.outer(np.arange(N), np.arange(N)))) *
                (1 + np.arange(N)[:, None] % 5) ** 2)
    K = rng.uniform(0, 1, (N, M)) ** 3 * 2.5
    V, (Wq, Kq, si, sK) = Z.contract(W, K, 2.5, n)
    assert np.abs(Wq).max() < 2.0 ** 52 and Kq.max() < 2.0 ** 53
    for i, j in [(N - 1, 0), (N // 2, 3), (5, 7), (0, 1), (N - 1, M - 1)]:
        ex = Z.exact_dot(Wq[i], Kq[:, j])
        got = V[i, j] * 2.0 ** float(si[i] + sK)
        assert abs(got - ex) <= abs(ex) * 2.0 ** -52, (i, j, ex, got)       # the reconstruction is the exact integer, rounded once
    Vld = W.astype(np.longdouble) @ K.astype(np.longdouble)
    err = float(np.max(np.abs(V - Vld)) / np.max(np.abs(Vld)))
    assert err < tol, err
    if n >= 14:           # at the default the engine is as close to the long-double product as an fp64 product is
        assert err <= 4 * float(np.max(np.abs(W @ K - Vld)) / np.max(np.abs(Vld))) + 1e-16
