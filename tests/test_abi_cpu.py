"""CPU tests of the drop-in boundary: the C-ABI library loads without a GPU, exports every symbol
include/abo_hip.h declares, and the product path fails loudly (no CPU fallback) when no device exists."""
import ctypes as C
import os
import re

import pytest

import abstractbayesopt.jl_amd as abo

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _has_gpu():
    import torch
    return torch.cuda.is_available()


def _declared():
    """(shipped surface, test hooks) as include/abo_hip.h declares them: the hooks sit between #ifdef ABO_TEST_HOOKS and its #endif"""
    hdr = open(os.path.join(ROOT, "include", "abo_hip.h")).read()
    i0, i1 = hdr.index("#ifdef ABO_TEST_HOOKS"), hdr.index("#endif /* ABO_TEST_HOOKS */")
    proto = r"^int32_t\s+(abo_\w+)\s*\("
    hooks = sorted(set(re.findall(proto, hdr[i0:i1], flags=re.M)))
    shipped = sorted(set(re.findall(proto, hdr[:i0] + hdr[i1:], flags=re.M)))
    return hdr, shipped, hooks


def test_library_exports_every_declared_symbol():
    hdr, shipped, hooks = _declared()
    assert shipped and hooks and all(h.startswith("abo_test_") for h in hooks) and not any(s.startswith("abo_test_") for s in shipped)
    assert sorted(abo._lib.EXPORTS) == shipped and sorted(abo._lib.TEST_EXPORTS) == hooks
    lib = abo._lib.lib()                 # the test suite runs on the test build (tests/conftest.py): surface + hooks
    assert lib.has_test_hooks
    for name in shipped + hooks:
        assert getattr(lib, name) is not None
    assert lib.abo_abi_version() == int(re.search(r"#define ABO_ABI_VERSION (\d+)", hdr).group(1))


def test_shipped_library_exports_the_documented_surface_and_nothing_else():
    """libabo_hip.so — what a Julia host links, what bench.py and smoke() load — defines every entry point of the header's documented
    surface and NONE of the abo_test_* building blocks (six test hooks were exported from the shipped library)."""
    import subprocess
    _, shipped, hooks = _declared()
    out = subprocess.run(["nm", "-D", "--defined-only", abo._lib.LIB_PATH], capture_output=True, text=True, check=True).stdout
    syms = {ln.split()[-1] for ln in out.splitlines() if ln.strip()}
    abo_syms = sorted(s for s in syms if s.startswith("abo_"))
    assert abo_syms == shipped, (sorted(set(abo_syms) - set(shipped)), sorted(set(shipped) - set(abo_syms)))
    assert not any(h in syms for h in hooks)
    out_t = subprocess.run(["nm", "-D", "--defined-only", abo._lib.LIB_TEST_PATH], capture_output=True, text=True, check=True).stdout
    syms_t = {ln.split()[-1] for ln in out_t.splitlines() if ln.strip()}
    assert sorted(s for s in syms_t if s.startswith("abo_")) == sorted(shipped + hooks)


def test_struct_layout_matches_header():
    # abo_params: 2×int32, 5×double, 2×int64 ; abo_timings: 10×double, int64, 3×double, (ABI 3:) 2×int64, 5×double
    assert C.sizeof(abo._lib.AboParams) == 8 + 5 * 8 + 2 * 8
    assert C.sizeof(abo._lib.AboTimings) == 29 * 8              # ABI 4: + refine_ms, refine_starts, refine_evals; ABI 6: + downdate_from_chain; ABI 7: + nlml_kinv_ms, nlml_trace_ms, append_trmv_ms, append_trmv_bytes
    assert C.sizeof(abo._lib.AboQeiStats) == 4 * 4 + 5 * 8
    assert C.sizeof(abo._lib.AboRefineOpts) == 4 * 4 + 3 * 8


def test_argument_validation_needs_no_gpu():
    lib = abo._lib.lib()
    assert lib.abo_create(None, None) == abo._lib.ABO_EINVAL
    assert "null" in abo._lib.last_error()
    bad = abo._lib.AboParams(family=9, device=0, ell=1.0, sigma_f2=1.0, noise_var=0.0, mean_c=0.0, jitter=0.0)
    hp = C.c_void_p()
    assert lib.abo_create(C.byref(bad), C.byref(hp)) == abo._lib.ABO_EINVAL
    bad = abo._lib.AboParams(family=0, device=0, ell=-1.0, sigma_f2=1.0, noise_var=0.0, mean_c=0.0, jitter=0.0)
    assert lib.abo_create(C.byref(bad), C.byref(hp)) == abo._lib.ABO_EINVAL
    assert lib.abo_destroy(None) == abo._lib.ABO_OK


@pytest.mark.skipif(_has_gpu(), reason="only meaningful on a box without a GPU")
def test_product_path_fails_loudly_without_gpu():
    gp = abo.HipStandardGP(abo.SqExponentialKernel(), 0.1)
    with pytest.raises((abo.AboError, ValueError)):
        abo.update(gp, [0.0, 0.5, 1.0], [0.0, 0.25, 1.0])
    with pytest.raises(ValueError):
        abo.posterior_mean(gp, [0.25])        # gpx === nothing


def test_no_oracle_import_in_product():
    pkg = os.path.join(ROOT, "abstractbayesopt.jl_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp")):
                src = open(os.path.join(dirpath, f)).read()
                assert "gp_oracle" not in src and "from oracle" not in src and "import oracle" not in src, f


def test_plain_c_host_links_and_loads_without_python_or_torch():
    """tests/c_abi_harness.c (the torch-free `ccall`-shaped host) compiles with gcc against libabo_hip.so, and the
    dynamic loader resolves the library and the SYSTEM HIP runtime for it; without arguments it prints its usage
    (exit 2) before any GPU call, so this runs on a box without a GPU."""
    import subprocess
    from tests import c_harness
    exe = c_harness.build(force=True)
    r = subprocess.run([exe], capture_output=True, text=True, timeout=60)
    assert r.returncode == 2 and "usage" in r.stderr
    ldd = subprocess.run(["ldd", exe], capture_output=True, text=True).stdout
    assert "libabo_hip.so" in ldd and "not found" not in ldd
    hip = [ln for ln in ldd.splitlines() if "libamdhip64" in ln]
    assert hip and "/opt/rocm" in hip[0] and "torch" not in hip[0], ldd
