"""Builds libabo_hip.so (HIP kernels + C-ABI) for gfx950 with hipcc, in-tree — and libabo_hip_test.so, the same objects with api.hip
compiled once more under -DABO_TEST_HOOKS: the abo_test_* building blocks the GPU suite drives (include/abo_hip.h, last section) are in
the test library only; the shipped library exports the header's documented surface and nothing else."""
import os
import shutil
import subprocess

PKG = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(PKG, "csrc")
LIB = os.path.join(PKG, "lib", "libabo_hip.so")
LIB_TEST = os.path.join(PKG, "lib", "libabo_hip_test.so")
HOOK_SOURCES = ["api.hip"]          # the translation units that hold #ifdef ABO_TEST_HOOKS code
SOURCES = ["kgen.hip", "kgen_res.hip", "kgen_grad_res.hip", "gemm.hip", "ozaki.hip", "chol.hip", "misc.hip", "qei.hip", "refine.hip", "api.hip", "mgpu.hip"]
# -amdgpu-mfma-vgpr-form: keep fp64 MFMA accumulators in VGPRs; the AGPR form makes hipcc shuttle
# every accumulator through v_accvgpr_read/write each k-step (2.2x slower, profiles/r01_mfma_f64_probe.txt)
# -ldl / -pthread: the multi-device driver resolves RCCL with dlopen and runs one host thread per shard
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-mllvm", "-amdgpu-mfma-vgpr-form", "-ldl", "-pthread"]


MANIFEST = os.path.join(PKG, "lib", "libabo_hip.manifest")


def _deps():
    return sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC)) + [os.path.join(PKG, "..", "include", "abo_hip.h")]


def _manifest():
    """sha256 over every file the library is compiled from and the flags: what the library in lib/ claims to be built from"""
    import hashlib
    h = hashlib.sha256(" ".join(FLAGS).encode())
    for p in _deps():
        h.update(os.path.basename(p).encode())
        with open(p, "rb") as f:
            h.update(f.read())
    return h.hexdigest()


def _lib_digest():
    """sha256 of the linked library itself: the manifest's second line — a file copied over lib/libabo_hip.so while the manifest
    stays (an A/B build put back) no longer matches it"""
    import hashlib
    h = hashlib.sha256()
    with open(LIB, "rb") as f:
        for blk in iter(lambda: f.read(1 << 20), b""):
            h.update(blk)
    return h.hexdigest()


def _newer_than_lib():
    t = os.path.getmtime(LIB)
    return any(os.path.getmtime(p) > t for p in _deps())


def _stale():
    # by CONTENT, not by time stamps alone: a library file copied over lib/libabo_hip.so (an A/B build put back, a checkout of
    # older sources) is newer than every source and would pass a time-stamp test while being built from something else
    if not os.path.exists(LIB) or not os.path.exists(LIB_TEST) or not os.path.exists(MANIFEST):
        return True
    with open(MANIFEST) as f:
        lines = f.read().split()
    return len(lines) != 2 or lines[0] != _manifest() or lines[1] != _lib_digest()


def build(force: bool = False, verbose: bool = False) -> str:
    """One hipcc per translation unit, in parallel (kgen.hip with its per-family / per-dimension instantiations is the long pole),
    then one link."""
    if not force and not _stale():
        return LIB
    if not force and os.path.exists(LIB) and not _newer_than_lib():
        force = True          # the manifest disagrees although no source is newer: the objects cannot be trusted either
    from concurrent.futures import ThreadPoolExecutor
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    os.makedirs(os.path.dirname(LIB), exist_ok=True)
    objdir = os.path.join(os.path.dirname(LIB), "obj")
    os.makedirs(objdir, exist_ok=True)
    cflags = [f for f in FLAGS if f not in ("-shared", "-ldl", "-pthread")]

    # an object is rebuilt when its own source, any header of csrc/ or the public header is newer (force: all of them)
    hdrs = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h")] + [os.path.join(PKG, "..", "include", "abo_hip.h")]
    hdr_t = max(os.path.getmtime(h) for h in hdrs)

    def compile_one(job):
        src, hooks = job
        obj = os.path.join(objdir, os.path.splitext(src)[0] + ("_hooks.o" if hooks else ".o"))
        if not force and os.path.exists(obj) and os.path.getmtime(obj) > max(hdr_t, os.path.getmtime(os.path.join(CSRC, src))):
            return obj
        cmd = [hipcc] + cflags + (["-DABO_TEST_HOOKS"] if hooks else []) + ["-c", os.path.join(CSRC, src), "-o", obj]
        if verbose:
            print(" ".join(cmd))
        subprocess.check_call(cmd)
        return obj

    jobs = [(s_, False) for s_ in SOURCES] + [(s_, True) for s_ in HOOK_SOURCES]
    with ThreadPoolExecutor(max_workers=min(len(jobs), os.cpu_count() or 1)) as ex:
        built = list(ex.map(compile_one, jobs))
    objs, hook_objs = built[:len(SOURCES)], dict(zip(HOOK_SOURCES, built[len(SOURCES):]))
    for lib, these in ((LIB, objs), (LIB_TEST, [hook_objs.get(s_, o) for s_, o in zip(SOURCES, objs)])):
        cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", lib] + these + ["-ldl", "-pthread"]
        if verbose:
            print(" ".join(cmd))
        subprocess.check_call(cmd)
    with open(MANIFEST, "w") as f:
        f.write(_manifest() + "\n" + _lib_digest() + "\n")
    return LIB


if __name__ == "__main__":
    print(build(force=True, verbose=True))
