"""ctypes binding of libabo_hip.so (include/abo_hip.h).  There is no CPU fallback: if the HIP
library is missing or a call fails, this module raises."""
import ctypes as C
import os

PKG = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(PKG, "lib", "libabo_hip.so")
# the same library with the abo_test_* building blocks compiled in (-DABO_TEST_HOOKS; include/abo_hip.h, last section): what the test
# suite loads (tests/conftest.py sets ABO_LIB_TEST_HOOKS=1 before the package is imported).  The shipped library exports none of them.
LIB_TEST_PATH = os.path.join(PKG, "lib", "libabo_hip_test.so")

ABO_OK, ABO_ENOTPD, ABO_EDIM, ABO_EINVAL, ABO_EHIP, ABO_ENOMEM = range(6)
HOST, DEVICE = 0, 1

EXPORTS = ["abo_create", "abo_set_contraction", "abo_create_grad", "abo_predict_grad", "abo_predict_grad_cov", "abo_retain", "abo_destroy", "abo_fit", "abo_append", "abo_append_grad", "abo_cand_create", "abo_cand_destroy",
           "abo_cand_refresh", "abo_cand_downdate", "abo_cand_save", "abo_cand_restore", "abo_cand_acq", "abo_cand_get", "abo_cand_point", "abo_cand_exclude", "abo_predict", "abo_acq", "abo_nlml", "abo_nlml_grad", "abo_lhs", "abo_score",
           "abo_get_factor", "abo_get_n", "abo_get_data", "abo_get_timings", "abo_last_error", "abo_abi_version", "abo_pool_trim",
           "abo_mgpu_create", "abo_mgpu_clone", "abo_mgpu_destroy", "abo_mgpu_info", "abo_mgpu_get", "abo_mgpu_fit",
           "abo_mgpu_predict", "abo_mgpu_acq", "abo_mgpu_acq_lhs", "abo_mgpu_append", "abo_mgpu_cand_create",
           "abo_mgpu_cand_create_lhs", "abo_mgpu_cand_refresh", "abo_mgpu_cand_destroy", "abo_mgpu_cand_acq",
           "abo_mgpu_cand_qei", "abo_refine", "abo_optimize_acquisition", "abo_mgpu_optimize_acquisition", "abo_fit_acq", "abo_mgpu_create_grad", "abo_mgpu_append_grad", "abo_mgpu_cand_get", "abo_acq_terms", "abo_acq_lhs", "abo_refine_terms",
           "abo_optimize_acquisition_terms", "abo_mgpu_optimize_acquisition_terms",
           "abo_set_qei_block", "abo_cand_qei", "abo_cand_qei_begin", "abo_cand_qei_top", "abo_cand_qei_block", "abo_cand_qei_pick",
           "abo_cand_qei_end", "abo_cand_qei_has", "abo_cand_qei_stats", "abo_mgpu_cand_qei_stats", "abo_cand_qei_eligible", "abo_fill_distance"]
TEST_EXPORTS = ["abo_test_gemm_nt", "abo_test_kappa", "abo_test_oz_plan", "abo_test_oz_contract", "abo_test_acq_grad",
                "abo_test_acq_grad_terms"]
ABI_VERSION = 7
CONTRACT_AUTO, CONTRACT_FP64, CONTRACT_INT8 = 0, 1, 2


class AboParams(C.Structure):
    _fields_ = [("family", C.c_int32), ("device", C.c_int32), ("ell", C.c_double), ("sigma_f2", C.c_double),
                ("noise_var", C.c_double), ("mean_c", C.c_double), ("jitter", C.c_double),
                ("n_max", C.c_int64), ("chunk", C.c_int64)]


class AboTimings(C.Structure):
    _fields_ = [(n, C.c_double) for n in ("fit_kernel_matrix_ms", "fit_cholesky_ms", "fit_inverse_ms",
                                          "fit_alpha_ms", "fit_total_ms", "acq_kxz_ms", "acq_var_gemm_ms",
                                          "acq_finalize_ms", "acq_topk_ms", "acq_total_ms")] + \
               [("var_gemm_launches", C.c_int64), ("var_gemm_flop", C.c_double), ("downdate_ms", C.c_double),
                ("downdate_bytes", C.c_double), ("contraction_engine", C.c_int64), ("oz_nmod", C.c_int64)] + \
               [(n, C.c_double) for n in ("oz_prepare_ms", "oz_quant_ms", "oz_gemm_ms", "oz_crt_ms", "oz_gemm_ops", "refine_ms")] + \
               [("refine_starts", C.c_int64), ("refine_evals", C.c_int64), ("downdate_from_chain", C.c_int64),
                ("nlml_kinv_ms", C.c_double), ("nlml_trace_ms", C.c_double), ("append_trmv_ms", C.c_double),
                ("append_trmv_bytes", C.c_double)]

    def as_dict(self):
        return {n: getattr(self, n) for n, _ in self._fields_}


class AboQeiStats(C.Structure):
    """statistics of a block-form greedy q-EI batch (include/abo_hip.h: abo_qei_stats)"""
    _fields_ = [("picks", C.c_int32), ("block", C.c_int32), ("block_builds", C.c_int32), ("block_hits", C.c_int32),
                ("total_ms", C.c_double), ("block_ms", C.c_double), ("pass_ms", C.c_double), ("pass_bytes", C.c_double),
                ("pass_flop", C.c_double)]

    def as_dict(self):
        return {n: getattr(self, n) for n, _ in self._fields_}


class AboRefineOpts(C.Structure):
    _fields_ = [("max_iter", C.c_int32), ("linesearch_max", C.c_int32), ("history", C.c_int32), ("reserved", C.c_int32),
                ("g_tol", C.c_double), ("f_abstol", C.c_double), ("x_abstol", C.c_double)]


class AboAcqTerm(C.Structure):
    """one term of a weighted-sum objective (include/abo_hip.h: abo_acq_term)"""
    _fields_ = [("kind", C.c_int32), ("reserved", C.c_int32), ("p0", C.c_double), ("best_y", C.c_double), ("weight", C.c_double)]


class PosDefException(Exception):
    """Mirror of LinearAlgebra.PosDefException(info) — the exception the BO driver's rollback
    catches (src/bayesian_opt.jl:126-141)."""

    def __init__(self, info, msg=""):
        super().__init__(msg or f"matrix is not positive definite; Cholesky factorization failed at {info}")
        self.info = int(info)


class DimensionMismatch(Exception):
    """Mirror of Julia's DimensionMismatch (test/test_bayesian_opt.jl:788-817)."""


class AboError(RuntimeError):
    pass


_lib = None


def lib():
    global _lib
    if _lib is not None:
        return _lib
    hooks = os.environ.get("ABO_LIB_TEST_HOOKS") == "1"
    path = LIB_TEST_PATH if hooks else LIB_PATH
    if not os.path.exists(path):
        raise AboError(f"{path} not found: the HIP library has not been built "
                       "(run `python -c 'import __graft_entry__ as g; g.build()'`). There is no CPU fallback.")
    # libabo_hip.so has NEEDED libamdhip64.so.7 + RUNPATH /opt/rocm-*/lib: a host without PyTorch (the Julia `ccall`
    # host; tests/c_abi_harness.c, run as a child process by tests/test_gpu_c_abi.py, which asserts that the system
    # runtime and nothing of PyTorch is mapped) needs nothing beyond the dynamic loader's defaults.  THIS host is the
    # special case: PyTorch-ROCm bundles its own libamdhip64.so.7, and a process must hold exactly one HIP runtime —
    # importing torch first makes the loader resolve our NEEDED entry to the copy already in the process.
    try:
        import torch  # noqa: F401
    except ImportError:
        pass
    L = C.CDLL(path)
    L.has_test_hooks = hooks
    vp, i32, i64, f64 = C.c_void_p, C.c_int32, C.c_int64, C.c_double
    L.abo_create.argtypes = [C.POINTER(AboParams), C.POINTER(vp)]
    L.abo_set_contraction.argtypes = [vp, i32, i32]
    L.abo_create_grad.argtypes = [C.POINTER(AboParams), i32, vp, C.POINTER(vp)]
    L.abo_predict_grad.argtypes = [vp, vp, i64, i32, i32, vp, vp, i32]
    L.abo_predict_grad_cov.argtypes = [vp, vp, i64, i32, i32, f64, vp, vp, vp, i32]
    L.abo_retain.argtypes = [vp]
    L.abo_destroy.argtypes = [vp]
    L.abo_fit.argtypes = [vp, vp, i64, i32, vp, i32, C.POINTER(i64)]
    L.abo_append.argtypes = [vp, vp, i32, f64, C.POINTER(i64), C.POINTER(vp)]
    L.abo_append_grad.argtypes = [vp, vp, i32, vp, C.POINTER(i64), C.POINTER(vp)]
    L.abo_cand_create.argtypes = [vp, vp, i64, i32, i32, C.POINTER(vp)]
    L.abo_cand_destroy.argtypes = [vp]
    L.abo_cand_refresh.argtypes = [vp, vp]
    L.abo_cand_downdate.argtypes = [vp, vp]
    L.abo_cand_save.argtypes = [vp, vp]
    L.abo_cand_restore.argtypes = [vp, vp]
    L.abo_cand_acq.argtypes = [vp, vp, i32, f64, f64, i64, vp, i32, vp, vp, i32]
    L.abo_cand_get.argtypes = [vp, vp, vp, vp, i32]
    L.abo_cand_point.argtypes = [vp, vp, i64, vp, vp, vp]
    L.abo_cand_exclude.argtypes = [vp, vp, i64]
    L.abo_predict.argtypes = [vp, vp, i64, i32, i32, vp, vp, i32]
    L.abo_acq.argtypes = [vp, vp, i64, i32, i32, i32, f64, f64, i64, vp, i32, vp, vp, i32]
    L.abo_nlml.argtypes = [vp, C.POINTER(f64)]
    L.abo_nlml_grad.argtypes = [vp, C.POINTER(f64), C.POINTER(f64), C.POINTER(f64)]
    L.abo_lhs.argtypes = [i32, i64, i32, vp, vp, C.c_uint64, i64, i64, vp]
    L.abo_score.argtypes = [i32, vp, vp, i64, i32, f64, f64, vp]
    L.abo_fill_distance.argtypes = [i32, vp, i64, i32, i32, vp, i64, i32, C.POINTER(f64)]
    L.abo_get_factor.argtypes = [vp, vp, vp, vp]
    L.abo_get_n.argtypes = [vp, C.POINTER(i64), C.POINTER(i32)]
    L.abo_get_data.argtypes = [vp, vp, vp]
    L.abo_get_timings.argtypes = [vp, C.POINTER(AboTimings)]
    L.abo_last_error.argtypes = [C.c_char_p, C.c_size_t]
    L.abo_abi_version.argtypes = []
    L.abo_pool_trim.argtypes = [i32]
    if hooks:
        L.abo_test_kappa.argtypes = [i32, i32, vp, vp, i64]
    if hooks:
        L.abo_test_oz_plan.argtypes = [i32, vp, vp, vp, vp]
    if hooks:
        L.abo_test_oz_contract.argtypes = [i32, vp, i64, i32, i32, vp, i64, i32, f64, i32, vp, i64]
    if hooks:
        L.abo_test_gemm_nt.argtypes = [i32, vp, vp, vp, i32, i32, i32, i64, i64, i64, f64, f64]
    L.abo_mgpu_create.argtypes = [C.POINTER(AboParams), i32, C.POINTER(i32), C.POINTER(vp)]
    L.abo_mgpu_create_grad.argtypes = [C.POINTER(AboParams), i32, vp, i32, C.POINTER(i32), C.POINTER(vp)]
    L.abo_mgpu_append_grad.argtypes = [vp, vp, i32, vp, C.POINTER(i64), vp]
    L.abo_mgpu_clone.argtypes = [vp, C.POINTER(vp)]
    L.abo_mgpu_destroy.argtypes = [vp]
    L.abo_mgpu_info.argtypes = [vp, C.POINTER(i32), C.POINTER(i32), C.POINTER(i32)]
    L.abo_mgpu_get.argtypes = [vp, i32, C.POINTER(vp)]
    L.abo_mgpu_fit.argtypes = [vp, vp, i64, i32, vp, C.POINTER(i64)]
    L.abo_mgpu_predict.argtypes = [vp, vp, i64, i32, vp, vp]
    L.abo_mgpu_acq.argtypes = [vp, vp, i64, i32, i32, f64, f64, vp, i32, vp, vp]
    L.abo_mgpu_acq_lhs.argtypes = [vp, i64, i32, vp, vp, C.c_uint64, i32, f64, f64, i32, vp, vp, vp]
    L.abo_mgpu_append.argtypes = [vp, vp, i32, f64, C.POINTER(i64), vp]
    L.abo_mgpu_cand_create.argtypes = [vp, vp, i64, i32, C.POINTER(vp)]
    L.abo_mgpu_cand_create_lhs.argtypes = [vp, i64, i32, vp, vp, C.c_uint64, C.POINTER(vp)]
    L.abo_mgpu_cand_refresh.argtypes = [vp, vp]
    L.abo_mgpu_cand_destroy.argtypes = [vp]
    L.abo_mgpu_cand_acq.argtypes = [vp, vp, i32, f64, f64, i32, vp, vp]
    L.abo_mgpu_cand_get.argtypes = [vp, vp, vp, vp]
    L.abo_mgpu_cand_qei.argtypes = [vp, vp, i32, f64, f64, i32, vp, vp, vp]
    L.abo_mgpu_cand_qei_stats.argtypes = [vp, vp, C.POINTER(AboQeiStats)]
    L.abo_set_qei_block.argtypes = [i32]
    L.abo_cand_qei.argtypes = [vp, vp, i32, f64, f64, i32, i64, i32, vp, vp, vp, C.POINTER(AboQeiStats)]
    L.abo_cand_qei_begin.argtypes = [vp, vp, i32, i32]
    L.abo_cand_qei_top.argtypes = [vp, vp, f64, f64, i64, i32, vp, i64]
    L.abo_cand_qei_eligible.argtypes = [vp, vp, i32, i32, C.POINTER(i32)]
    L.abo_cand_qei_block.argtypes = [vp, vp, vp, vp, i32]
    L.abo_cand_qei_pick.argtypes = [vp, vp, i64, f64, vp, i32, i64, C.POINTER(i64)]
    L.abo_cand_qei_end.argtypes = [vp, vp]
    L.abo_cand_qei_has.argtypes = [vp, vp, i64, C.POINTER(i32), C.POINTER(i32)]
    L.abo_cand_qei_stats.argtypes = [vp, vp, C.POINTER(AboQeiStats)]
    L.abo_refine.argtypes = [vp, i32, f64, f64, vp, vp, i32, vp, i32, C.POINTER(AboRefineOpts), vp, vp, vp]
    L.abo_optimize_acquisition.argtypes = [vp, i32, f64, f64, vp, vp, i32, i64, i32, C.c_uint64, C.POINTER(AboRefineOpts),
                                           vp, C.POINTER(f64), vp, vp, vp, vp]
    L.abo_mgpu_optimize_acquisition.argtypes = L.abo_optimize_acquisition.argtypes
    if hooks:
        L.abo_test_acq_grad.argtypes = [vp, i32, f64, f64, vp, i64, i32, vp, vp]
    tp = C.POINTER(AboAcqTerm)
    if hooks:
        L.abo_test_acq_grad_terms.argtypes = [vp, tp, i32, vp, i64, i32, vp, vp]
    L.abo_acq_terms.argtypes = [vp, vp, i64, i32, i32, tp, i32, i64, vp, i32, vp, vp, i32]
    L.abo_acq_lhs.argtypes = [vp, i64, i32, vp, vp, C.c_uint64, i32, f64, f64, i32, vp, vp, vp]
    L.abo_refine_terms.argtypes = [vp, tp, i32, vp, vp, i32, vp, i32, C.POINTER(AboRefineOpts), vp, vp, vp]
    L.abo_optimize_acquisition_terms.argtypes = [vp, tp, i32, vp, vp, i32, i64, i32, C.c_uint64, C.POINTER(AboRefineOpts),
                                                 vp, C.POINTER(f64), vp, vp, vp, vp]
    L.abo_mgpu_optimize_acquisition_terms.argtypes = L.abo_optimize_acquisition_terms.argtypes
    L.abo_fit_acq.argtypes = [vp, vp, i64, i32, vp, i32, C.POINTER(i64), vp, i64, i32, i32, f64, f64, i64, vp, i32, vp, vp, i32]
    for name in EXPORTS + (TEST_EXPORTS if hooks else []):
        getattr(L, name).restype = i32
    _lib = L
    return L


def last_error() -> str:
    buf = C.create_string_buffer(512)
    lib().abo_last_error(buf, 512)
    return buf.value.decode("utf-8", "replace")


def check(status: int, info: int = 0):
    if status == ABO_OK:
        return
    msg = last_error()
    if status == ABO_ENOTPD:
        raise PosDefException(info, msg)
    if status == ABO_EDIM:
        raise DimensionMismatch(msg)
    if status == ABO_EINVAL:
        raise ValueError(msg)
    if status == ABO_ENOMEM:
        raise MemoryError(msg)
    raise AboError(msg)
