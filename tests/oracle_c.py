"""ctypes binding of the plain-C oracle (oracle/gp_oracle.c).  Test infrastructure only."""
import ctypes as C
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
_SO = os.path.join(ROOT, "oracle", "_build", "libgp_oracle.so")
_dp = np.ctypeslib.ndpointer(dtype=np.float64, flags="C_CONTIGUOUS")


def load():
    if not os.path.exists(_SO):
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle")])
    lib = C.CDLL(_SO)
    lib.oracle_fit.restype = C.c_int64
    lib.oracle_fit.argtypes = [C.c_int, C.c_double, C.c_double, C.c_double, C.c_double, _dp,
                               C.c_int64, C.c_int, _dp, _dp, _dp]
    lib.oracle_predict.restype = None
    lib.oracle_predict.argtypes = [C.c_int, C.c_double, C.c_double, C.c_double, _dp, C.c_int64,
                                   C.c_int, _dp, _dp, _dp, C.c_int64, _dp, _dp]
    lib.oracle_nlml.restype = C.c_double
    lib.oracle_nlml.argtypes = [_dp, _dp, _dp, C.c_double, C.c_int64]
    lib.oracle_acq.restype = None
    lib.oracle_acq.argtypes = [C.c_int, _dp, _dp, C.c_int64, C.c_double, C.c_double, _dp]
    return lib


def fit(lib, family, ell, sf2, noise, mean_c, X, y):
    X = np.ascontiguousarray(X, dtype=np.float64)
    y = np.ascontiguousarray(y, dtype=np.float64)
    n, d = X.shape
    L = np.zeros((n, n))
    alpha = np.zeros(n)
    info = lib.oracle_fit(family, ell, sf2, noise, mean_c, X, n, d, y, L, alpha)
    return info, L, alpha


def predict(lib, family, ell, sf2, mean_c, X, L, alpha, Z):
    X = np.ascontiguousarray(X, dtype=np.float64)
    Z = np.ascontiguousarray(Z, dtype=np.float64)
    m = Z.shape[0]
    mu = np.zeros(m)
    var = np.zeros(m)
    lib.oracle_predict(family, ell, sf2, mean_c, X, X.shape[0], X.shape[1], L, alpha, Z, m, mu, var)
    return mu, var


def acq(lib, kind, mu, var, p0, best_y):
    out = np.zeros_like(mu)
    lib.oracle_acq(kind, np.ascontiguousarray(mu), np.ascontiguousarray(var), mu.shape[0], p0, best_y, out)
    return out
