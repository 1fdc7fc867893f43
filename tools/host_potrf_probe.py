"""Why is the host Cholesky of the CPU baseline slow?  numpy.linalg.cholesky (NumPy's OpenBLAS) against scipy.linalg.cholesky and the raw
LAPACK wrapper (SciPy's OpenBLAS), the symmetric rank-k update and the triangular solve the blocked algorithm is made of, at 16 pinned
threads: python tools/host_potrf_probe.py [N]"""
import os, sys, time
os.sched_setaffinity(0, set(sorted(os.sched_getaffinity(0))[:16]))
os.environ.setdefault("OPENBLAS_NUM_THREADS", "16")
os.environ.setdefault("OMP_NUM_THREADS", "16")
import numpy as np, scipy.linalg as sl
from scipy.linalg import blas, lapack
from threadpoolctl import threadpool_info
N = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
rng = np.random.default_rng(0)
A = rng.standard_normal((N, N)); K = A @ A.T / N + np.eye(N)
for lib in threadpool_info():
    print(lib.get("internal_api"), lib.get("version"), lib.get("num_threads"), lib.get("architecture"), lib.get("filepath", "")[-60:])
def t(f, flop, what):
    f(); t0 = time.perf_counter(); f(); dt = time.perf_counter() - t0
    print(f"{what:46s} {dt*1e3:9.1f} ms  {flop/dt/1e9:8.1f} GFLOP/s")
t(lambda: np.linalg.cholesky(K), N**3 / 3, "numpy.linalg.cholesky")
t(lambda: sl.cholesky(K, lower=True, check_finite=False), N**3 / 3, "scipy.linalg.cholesky")
Kf = np.asfortranarray(K)
t(lambda: lapack.dpotrf(Kf, lower=1, overwrite_a=0), N**3 / 3, "scipy.linalg.lapack.dpotrf (Fortran order)")
B = np.asfortranarray(A[:, : N // 2])
t(lambda: blas.dsyrk(1.0, B), N * N * (N // 2), "scipy dsyrk N x N/2")
t(lambda: A @ A.T, 2 * N**3, "numpy dgemm N^3")
L = np.linalg.cholesky(K)
t(lambda: sl.solve_triangular(L, A[:, :2048], lower=True, check_finite=False), N * N * 2048, "scipy solve_triangular N x 2048")
Lf = np.asfortranarray(L); Bf = np.asfortranarray(A[:, :2048])
t(lambda: blas.dtrsm(1.0, Lf, Bf, lower=1), N * N * 2048, "scipy blas.dtrsm (Fortran order) N x 2048")
t(lambda: sl.solve_triangular(Lf, Bf, lower=True, check_finite=False), N * N * 2048, "scipy solve_triangular (Fortran order)")
