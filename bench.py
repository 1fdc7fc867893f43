#!/usr/bin/env python3
"""bench.py — ms per BO step (GP update + acquisition evaluation over M candidates incl. top-100) on
MI355X, BASELINE.json's metric.

A "step" is what the reference's EGO loop does per iteration on the hot path
(src/bayesian_opt.jl:430,445): `update(model, xs, ys)` — full refit: K_XX assembly, Cholesky, α —
followed by `scores = acqf(model, grid)` + `sortperm(scores; rev=true)[1:100]`
(src/acquisition_functions/acq_utils.jl:50-52).  Inputs (X, y, Z) are resident in HBM before the
timed region starts; the refit is NOT cached between steps.

Default workload = BASELINE config C3 (the configuration the north-star target is quoted on):
d = 8 Matérn-5/2, N = 8192, M = 1 048 576 candidates per GPU, EI ξ = 0.01.  With --gpus G the
candidate batch is G·2²⁰ (C4 at G = 8) sharded contiguously, one process per GPU, every rank refits
redundantly, and the only collective is the all_gather of 100 (score, index) pairs per rank.

  python bench.py                       # 1 GPU, C3
  python bench.py --gpus 8              # no launcher: one process drives the 8 devices through the library's own
                                        # multi-device handle (abo_mgpu_*: worker thread per device, RCCL all-gather)
  python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 \
         --master-port 29500 bench.py --gpus 8 --steps 5 --warmup 2
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

PEAK_FP64_MFMA_TFLOPS = 78.6   # MI355X fp64 matrix peak: 256 CU × 4 SIMD × 2048 flop / 64 clk × 2.4 GHz
# MI355X dense int8 matrix peak: 256 CU × 4 SIMD × 1024 MAC/clk × 2 × 2.4 GHz (MI355X_MICROARCH.md: I8 = 2 × BF16 per clock,
# BF16 ≈ 2.5 PFLOP/s dense).  On random operands the chip holds 1.7 – 2.1 GHz under this load: tools/mfma_i8_power_probe.hip
# sustains 3.1 (32×32×32) / 3.4 POP/s (16×16×64) from registers — quoted beside the peak, never instead of it.
PEAK_INT8_MFMA_TOPS = 5033.0


def sustained_int8_tops(with_source=False):
    """what v_mfma_i32_16x16x64_i8 sustains on random operands from registers (tools/mfma_i8_power_probe.hip), read from the
    NEWEST committed probe output; None when no file is there.  with_source: (value, the file it came from)"""
    import re
    for tag in ("r06", "r05", "r04", "r03", "r02"):
        rel = os.path.join("profiles", f"{tag}_mfma_i8_power_probe.txt")
        try:
            with open(os.path.join(ROOT, rel)) as f:
                for line in f:
                    if line.startswith("random operands") and "16x16x64" in line:
                        v = float(re.search(r"([0-9.]+) TOP/s", line).group(1))
                        return (v, rel) if with_source else v
        except OSError:
            pass
    return (None, None) if with_source else None
                               # (v_mfma_f64_16x16x4_f64 measured at 64 clk/SIMD: profiles/r01_mfma_f64_probe.txt)

CONFIGS = {
    # name: (family ctor name, d, N, M per GPU, ell, sigma_f2, noise_var, acq, p0)
    "c2": ("SqExponentialKernel", 4, 1024, 65536, 0.5, 1.0, 1e-4, "ucb", 2.0),
    "c3": ("Matern52Kernel", 8, 8192, 1 << 20, 1.0, 1.0, 1e-3, "ei", 0.01),
    # C4 as a STRONG-scaling run: M = 8 388 608 candidates in total, cut into `world` shards (at --gpus 8 a rank's
    # work is exactly C3's; at --gpus 1 this is the single-GPU reference the >= 6x scaling target is stated against)
    "c4": ("Matern52Kernel", 8, 8192, 1 << 23, 1.0, 1.0, 1e-3, "ei", 0.01),
    # C5: noisy objective, incremental rank-1 update per BO step, greedy q-EI (q = 8) on a resident grid
    "c5": ("Matern52Kernel", 16, 16384, 131072, 2.0, 1.0, 1e-2, "qei", 0.01),
}


def usable_cores():
    """Host cores this process may actually use: the cgroup CPU quota when there is one (a one-GPU box hands
    out a share of the host, and BLAS threads beyond it only thrash), else the affinity mask."""
    n = len(os.sched_getaffinity(0))
    try:
        with open("/sys/fs/cgroup/cpu.max") as f:
            quota, period = f.read().split()
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except (OSError, ValueError):
        pass
    return n


def pin_threads(cores):
    """Pin every thread of this process to `cores` CPUs of the allowed set, one per physical core where the topology files say which
    logical CPUs share one.  Returns what was done (and the masks to restore)."""
    info = {"requested": cores, "pinned": False, "saved": {}}
    try:
        allowed = sorted(os.sched_getaffinity(0))
        info["allowed_logical_cpus"] = len(allowed)
        seen, pick = set(), []
        for c in allowed:
            try:
                with open(f"/sys/devices/system/cpu/cpu{c}/topology/core_id") as f:
                    core = int(f.read())
                with open(f"/sys/devices/system/cpu/cpu{c}/topology/physical_package_id") as f:
                    core = (int(f.read()), core)
            except (OSError, ValueError):
                core = ("cpu", c)
            if core in seen:
                continue
            seen.add(core)
            pick.append(c)
            if len(pick) == cores:
                break
        if len(pick) < cores:
            pick = allowed[:cores]
        for tid in os.listdir("/proc/self/task"):
            try:
                info["saved"][int(tid)] = os.sched_getaffinity(int(tid))
                os.sched_setaffinity(int(tid), pick)
            except OSError:
                pass
        info["pinned"] = True
        info["cpus"] = pick
    except (OSError, AttributeError) as e:
        info["error"] = f"{type(e).__name__}: {e}"
    return info


def unpin_threads(info):
    for tid, mask in info.get("saved", {}).items():
        try:
            os.sched_setaffinity(tid, mask)
        except OSError:
            pass


def cpu_baseline(cfg, sample_m=21846, reps=3):
    """Median of `reps` repetitions of the bounded CPU sample (median of ≥ 3 steps, up to 65 536
    candidates timed and scaled linearly in M): by default 3 × 21 846 = 65 538 candidates in all, ≈ 80 s of host time
    at C3 — each repetition is a full N-point refit + the posterior / acquisition over its candidates."""
    from threadpoolctl import threadpool_limits
    # every BLAS this baseline will call must be LOADED before the limit is set: threadpool_limits only reaches the libraries that
    # are in the process at that moment — round 3's baseline imported scipy.linalg inside the limited region, SciPy's own OpenBLAS
    # came in afterwards with its default of one thread per LOGICAL cpu of the host (64 on a 16-core share) and dpotrf ran
    # oversubscribed at 28 GFLOP/s
    import scipy.linalg  # noqa: F401
    from oracle import gp_oracle  # noqa: F401
    cores = usable_cores()
    # Round 5: the baseline's threads are PINNED to `cores` distinct physical cores for its duration (every thread of the process —
    # the BLAS pools exist since NumPy was imported — through /proc/self/task): on a 256-logical-CPU host a 16-core cgroup share
    # otherwise lets the scheduler spread 16 BLAS threads over SMT siblings and NUMA nodes.  Whether dpotrf leaves its ≈ 30 GFLOP/s
    # that way is recorded either way (`affinity`).
    aff = pin_threads(cores)
    try:
        with threadpool_limits(limits=cores):
            runs = [_cpu_baseline(cfg, sample_m) for _ in range(max(1, reps))]
    finally:
        unpin_threads(aff)
    runs.sort(key=lambda r: r["value"])
    out = runs[len(runs) // 2]
    out["affinity"] = {k: v for k, v in aff.items() if k != "saved"}
    out["repetitions"] = len(runs)
    out["all_values"] = [r["value"] for r in runs]
    out["sample"] += f"; median of {len(runs)} repetitions"
    out["cores"] = cores
    out["host_logical_cpus"] = os.cpu_count()
    return out


def _cpu_baseline(cfg, sample_m):
    """Oracle (CPU restatement, NOT the Julia reference) timed on the host cores on a bounded
    sample: the full N-point refit once, plus the posterior over `sample_m` candidates computed the
    way the reference does — K_XZ built separately for posterior_mean and posterior_var
    (ExpectedImprovement.jl:41-42) — then scaled linearly in M (cost is exactly linear in M at fixed N)."""
    import scipy.linalg as sla
    from threadpoolctl import threadpool_info

    from abstractbayesopt.jl_amd import synth
    from oracle import gp_oracle as O

    fam_name, d, N, M, ell, sf2, noise, acq, p0 = cfg
    fam = {"SqExponentialKernel": O.SE, "Matern52Kernel": O.MATERN52}[fam_name]
    X, y = synth.standardized_problem(N, d, 0.03)
    Z = synth.points(2, sample_m, d)
    t0 = time.perf_counter()
    st = O.fit(fam, ell, sf2, noise, 0.0, X, y)
    t1 = time.perf_counter()
    chunk = 4096
    mu = np.empty(sample_m)
    var = np.empty(sample_m)
    trsm_s = asm_s = 0.0
    # Round 6: the kernel-matrix ASSEMBLY runs on all the cores the BLAS runs on.  The oracle's kernel_matrix is
    # single-threaded NumPy (d passes of N x m temporaries): 9 s of the 10.4 s acquisition sample in round 5, which made the "port"
    # ratio a measure of NumPy temporaries.  Here the chunk's candidates are dealt to `workers` threads, each assembling its rows with
    # the oracle's own function (NumPy ufuncs release the GIL; rows of K_ZX are independent) — same values, bit for bit.  K_XZ is
    # still built TWICE per chunk, as the reference does (posterior_mean and posterior_var each evaluate the kernel matrix,
    # ExpectedImprovement.jl:41-42).  The refit is the oracle's own fit (its N x N assembly single-threaded: < 3 % of the extrapolated
    # step).
    from concurrent.futures import ThreadPoolExecutor
    workers = max(1, usable_cores())
    pool = ThreadPoolExecutor(max_workers=workers)

    def kzx_threaded(zc):
        out = np.empty((zc.shape[0], N))
        step = 128          # row blocks whose temporaries (128 x N doubles each) stay cache-resident: 137 against 42 Mpair/s with 512-row
                            # blocks on 8 threads, 2.9 for the oracle's whole-chunk call on one (measured in the build container)
        parts = [(a, min(a + step, zc.shape[0])) for a in range(0, zc.shape[0], step)]

        def one(ab):
            out[ab[0]:ab[1]] = O.kernel_matrix(fam, ell, sf2, zc[ab[0]:ab[1]], X)
        list(pool.map(one, parts))
        return out
    # Round 5: operands in the layout the host BLAS is fast in (tools/host_potrf_probe.py, profiles/r05_host_potrf_probe.txt: with
    # Fortran-ordered operands solve_triangular runs at 1150 GFLOP/s on 16 threads, with C-ordered ones at 130; dpotrf 120 against
    # 33) — K_ZX is generated candidate-major and its transpose VIEW is the Fortran-ordered K_XZ, L comes Fortran-ordered out of O.fit
    Lf = st.L if st.L.flags.f_contiguous else np.asfortranarray(st.L)
    for a in range(0, sample_m, chunk):
        zc = Z[a:a + chunk]
        tk = time.perf_counter()
        Kzx1 = kzx_threaded(zc)
        asm_s += time.perf_counter() - tk
        mu[a:a + chunk] = Kzx1 @ st.alpha                                               # posterior_mean
        tk = time.perf_counter()
        Kxz = kzx_threaded(zc).T                                                        # posterior_var builds it again
        asm_s += time.perf_counter() - tk
        ta = time.perf_counter()
        V = sla.solve_triangular(Lf, Kxz, lower=True, check_finite=False)
        trsm_s += time.perf_counter() - ta
        var[a:a + chunk] = sf2 - np.einsum("ij,ij->j", V, V) + 1e-18                     # posterior_var
    s = O.acquisition(O.ACQ_EI if acq == "ei" else O.ACQ_UCB, mu, var, p0, float(y.min()))
    O.top_k(s, 100)
    t2 = time.perf_counter()
    pool.shutdown()
    pools = threadpool_info()
    threads = max([p.get("num_threads", 1) for p in pools] + [1])
    fit_ms, acq_ms = (t1 - t0) * 1e3, (t2 - t1) * 1e3
    # the LAPACK/BLAS-3 part alone (dpotrf + dtrsm): a floor for ANY host implementation of the path, however the
    # kernel matrices are assembled
    K = np.asfortranarray(st.L @ st.L.T)
    tb = time.perf_counter()
    sla.lapack.dpotrf(K, lower=1, overwrite_a=1)
    potrf_ms = (time.perf_counter() - tb) * 1e3
    del K
    # what the host BLAS actually delivered: a threaded LAPACK is expected at ≥ 10 GFLOP/s per core on dpotrf / dtrsm of this
    # size; a baseline far below that is a handicapped one (a cgroup CPU share spread over many logical CPUs, a reference BLAS) and
    # the ratio against it says little — the line then flags it and does not print a speed-up
    potrf_gflops = (N ** 3 / 3.0) / (potrf_ms * 1e-3) / 1e9
    trsm_gflops = (float(N) * N * sample_m) / max(trsm_s, 1e-9) / 1e9
    # (the flag goes by dtrsm, 98 % of the step's host flops: OpenBLAS's own dpotrf delivers 60 – 125 GFLOP/s on 16 threads from box to
    # box where its dsyrk reaches 1100 — reported, not judged)
    under = trsm_gflops < 10.0 * threads
    return {
        "host_blas": {"libraries": [{k: p.get(k) for k in ("user_api", "internal_api", "version", "num_threads", "threading_layer")}
                                    for p in pools],
                      "dpotrf_gflops": potrf_gflops, "dtrsm_gflops": trsm_gflops, "threads": threads,
                      "under_threaded": bool(under),
                      "note": "achieved rate of LAPACK dpotrf (N x N, Fortran order, lower) and solve_triangular (N x N against the "
                              "sample, Fortran-ordered operands) inside this baseline; under_threaded = dtrsm (98 % of the host flops) below "
                              "10 GFLOP/s per thread"},
        "value": fit_ms + acq_ms * (M / sample_m), "unit": "ms per BO step (extrapolated)", "cores": threads,
        "kind": "port",
        "sample": f"CPU restatement (NumPy/SciPy LAPACK), not the Julia reference: full N={N} refit measured "
                  f"({fit_ms:.0f} ms) + posterior/acq over M'={sample_m} candidates measured ({acq_ms:.0f} ms), "
                  f"acq part scaled x{M / sample_m:.2f} to M={M}",
        "measured_fit_ms": fit_ms, "measured_acq_ms_sample": acq_ms, "sample_m": sample_m,
        "acq_sample_split_s": {"kernel_matrix_assembly": asm_s, "dtrsm": trsm_s, "rest": acq_ms * 1e-3 - asm_s - trsm_s,
                               "assembly_threads": workers,
                               "note": "assembly = K_ZX built twice per chunk (posterior_mean and posterior_var each evaluate it, "
                                       "ExpectedImprovement.jl:41-42) with the oracle's kernel_matrix on assembly_threads threads; rest = "
                                       "the mean's gemv, the column sums of squares, EI and the top-100"},
        "blas3_floor": {"value": potrf_ms + trsm_s * 1e3 * (M / sample_m), "unit": "ms per BO step (extrapolated)",
                        "measured_potrf_ms": potrf_ms, "measured_trsm_ms_sample": trsm_s * 1e3,
                        "note": "dpotrf + dtrsm only, kernel-matrix assembly and epilogue excluded"},
    }


def source_sha(names):
    """sha256 (16 hex digits) over the kernel sources a PMC figure belongs to"""
    import hashlib
    h = hashlib.sha256()
    for n in names:
        with open(os.path.join(ROOT, "abstractbayesopt.jl_amd", "csrc", n), "rb") as f:
            h.update(f.read())
    return h.hexdigest()[:16]


# every file the dominant kernel of a config is compiled from (tools/pmc_traffic_json.py records the same hash)
PMC_SOURCES = {"c3": ["gemm.hip", "abo_kernels.h"], "c3_int8": ["ozaki.hip", "abo_oz_dev.h", "abo_kernels.h"],
               "c5": ["gemm.hip", "abo_kernels.h"]}        # qei_pass_kernel (the block pass of greedy q-EI over the resident K_ZX)


def pmc_traffic(config, mc_per_launch):
    """HBM-side bytes per launch of the dominant kernel, from the committed rocprofv3 PMC pass
    (FETCH_SIZE/WRITE_SIZE collected in their own runs by tools/run_pmc.sh, gfx950 ×2 correction on
    FETCH_SIZE) — PMC counters cannot be read from inside this process, so the figure is the per-candidate
    traffic of that pass scaled to this run's candidates per launch.  The pass records the hash of the kernel's
    source file; when the file has changed since, the figure is STALE and `traffic` is reported as null."""
    key = config.replace("c4", "c3")                         # C4 = C3 per launch
    for tag in ("r06", "r05", "r04", "r03", "r02", "r01"):
        path = os.path.join(ROOT, "profiles", f"{tag}_{key}_pmc_traffic.json")
        if os.path.exists(path):
            break
    else:
        return None, {"note": "no PMC pass committed"}
    with open(path) as f:
        d = json.load(f)
    d["file"] = os.path.relpath(path, ROOT)
    if d.get("kernel_source_sha") != source_sha(PMC_SOURCES.get(key, ["gemm.hip"])):
        d["stale"] = True
        return None, d
    per = d.get("traffic_bytes_per_candidate")
    return (per * mc_per_launch if per is not None else d.get("traffic_bytes_per_launch")), d


def trmv_traffic(c5_pmc):
    """HBM-side bytes per launch of trmv_kernel from the committed C5 FETCH_SIZE pass (tools/run_pmc_c5.sh), null when chol.hip has
    changed since"""
    t = (c5_pmc or {}).get("trmv")
    if not t or t.get("kernel_source_sha") != source_sha(["chol.hip"]):
        return None
    return t.get("traffic_bytes_per_launch")


def contraction_label(abo, med):
    """what `config.contraction` says about the engine that ran: the int8-residue engine's only approximation is the fixed-point
    image of its operands, and the guarantee is stated per ROW of L^-1 (ozaki.hip: oz_rowscale_kernel), not per entry"""
    if int(med["contraction_engine"]) != abo._lib.CONTRACT_INT8:
        return "fp64 MFMA"
    return (f"int8-residue, {int(med['oz_nmod'])} moduli: exact integer products and sums of fixed-point images of the fp64 operands "
            "(K_XZ kept to 2^-52 of sigma_f2; each row of L^-1 kept to >= 50 bits below that row's L1 norm, i.e. an entry far below its "
            "row's L1 norm keeps fewer of its own bits - at most log2(N) fewer than 53 for a dense equal-magnitude row); results fp64, "
            "parity vs the oracle recorded in profiles/parity_r06.json")


def dominant_kernel_roofline(abo, med, config, N, M_per):
    """`roofline` object of the C2/C3/C4-shaped step from the median phase timings of one device (abo_get_timings)"""
    launches = int(med["var_gemm_launches"])
    flop = med["var_gemm_flop"]                       # N²·M_per (triangular credit), all launches of one step
    t_kernel_ms = med["acq_var_gemm_ms"]
    achieved = flop / (t_kernel_ms * 1e-3) / 1e12 if t_kernel_ms > 0 else 0.0
    int8 = int(med["contraction_engine"]) == abo._lib.CONTRACT_INT8
    traffic, pmc = pmc_traffic(config + ("_int8" if int8 else ""), M_per / max(launches, 1))
    mc = M_per / max(launches, 1)
    if int8:
        # dominant kernel = the residue GEMM (one launch per chunk covers all moduli): ALGORITHMIC int8 operations —
        # n moduli × N²·M (the triangular product, 2 operations per multiply-add; what the kernel issues beyond that on its
        # diagonal blocks and on padding is not credited) ÷ its HIP-event duration
        nmod = int(med["oz_nmod"])
        tops = med["oz_gemm_ops"] / (med["oz_gemm_ms"] * 1e-3) / 1e12 if med["oz_gemm_ms"] > 0 else 0.0
        np256 = -(-N // 256) * 256
        return {
            "kernel": f"oz_gemm16p_kernel (V = L^-1 K_XZ as {nmod} exact int8 residue GEMMs, v_mfma_i32_16x16x64_i8, persistent: one "
                      "workgroup per CU draws 256x256 tiles from per-XCD lists, triangular k-range, symmetric-mod epilogue)",
            "bound": "mfma", "achieved": tops, "peak": PEAK_INT8_MFMA_TOPS, "unit": "TOP/s", "frac": tops / PEAK_INT8_MFMA_TOPS,
            "traffic": traffic,
            "traffic_unit": "bytes per launch (PMC FETCH_SIZE x2 + WRITE_SIZE, profiles/)" if traffic else None,
            "traffic_source": {k: pmc.get(k) for k in ("file", "kernel_source_sha", "stale", "note") if pmc and k in pmc},
            # residue planes read once + U written once: n·(Np²/2 + Mc·Np) + n·Np·Mc bytes
            "algorithmic_bytes_per_launch": float(nmod) * (np256 * np256 / 2 + 2 * mc * np256),
            "ops_per_launch": med["oz_gemm_ops"] / max(launches, 1), "launches_per_step": launches,
            "avg_launch_ms": med["oz_gemm_ms"] / max(launches, 1),
            "sustained_peak_random_operands": sustained_int8_tops(),
            "frac_of_sustained": (tops / sustained_int8_tops()) if sustained_int8_tops() else None,
            "engine": {"name": "int8-residue (ABO_CONTRACT_INT8)", "moduli": nmod,
                       "pipeline_ms_per_chunk": {"quantise_K_XZ": med["oz_quant_ms"] / max(launches, 1),
                                                 "residue_gemm": med["oz_gemm_ms"] / max(launches, 1),
                                                 "reconstruct_and_square": med["oz_crt_ms"] / max(launches, 1)},
                       "residue_planes_of_W_ms": med["oz_prepare_ms"],
                       "fp64_equivalent_tflops": achieved, "fp64_equivalent_over_fp64_mfma_peak": achieved / PEAK_FP64_MFMA_TFLOPS},
            "note": "achieved = algorithmic int8 operations (moduli x N^2 x M) of the residue GEMM launches / their HIP-event duration "
                    "(library stream, median over timed steps); peak = dense int8 MFMA at 2.4 GHz; sustained_peak = the same "
                    "instruction on random operands from registers, clock as the chip holds it "
                    f"({sustained_int8_tops(True)[1]}); fp64_equivalent = N^2*M / time of the whole contraction pipeline",
        }
    return {
        "kernel": "var_gemm256s_kernel (V = L^-1 K_XZ triangular contraction + column sum of squares)",
        "bound": "mfma", "achieved": achieved, "peak": PEAK_FP64_MFMA_TFLOPS, "unit": "TFLOP/s",
        "frac": achieved / PEAK_FP64_MFMA_TFLOPS, "traffic": traffic,
        "traffic_unit": "bytes per launch (PMC FETCH_SIZE x2 + WRITE_SIZE, profiles/)" if traffic else None,
        "traffic_source": {k: pmc.get(k) for k in ("file", "kernel_source_sha", "stale", "note") if pmc and k in pmc},
        "algorithmic_bytes_per_launch": 8.0 * (N * N / 2 + mc * N + (N / 128) * mc),
        "flop_per_launch": flop / max(launches, 1), "launches_per_step": launches,
        "avg_launch_ms": t_kernel_ms / max(launches, 1),
        "note": "algorithmic flop = N^2*M (triangular credit, SURVEY 8(d)); duration = HIP events on the "
                "library stream around each launch, median over timed steps",
    }


def quick_config(abo, synth, torch, dev, local_rank, name, k_top, steps=20, warmup=3):
    """Same step on another BASELINE configuration (reported next to the headline one; C2 is the small
    configuration: d = 4 RBF, N = 1024, M = 65536, UCB)."""
    fam_name, d, N, M, ell, sf2, noise, acq_name, p0 = CONFIGS[name]
    X, y = synth.standardized_problem(N, d, 0.03)
    Xd, yd = torch.from_numpy(X).to(dev), torch.from_numpy(y).to(dev)
    Zd = torch.from_numpy(synth.points(2, M, d)).to(dev)
    gp = abo.HipStandardGP(sf2 * abo.with_lengthscale(getattr(abo, fam_name)(), ell), noise, device=local_rank)
    acq = abo.ExpectedImprovement(p0, float(y.min())) if acq_name == "ei" else abo.UpperConfidenceBound(p0)
    for step in range(warmup + steps):
        if step == warmup:
            torch.cuda.synchronize(dev)
            t0 = time.perf_counter()
        model = abo.update(gp, Xd, yd)
        abo.evaluate(acq, model, Zd, k=k_top, return_scores=False)
    torch.cuda.synchronize(dev)
    ms = (time.perf_counter() - t0) * 1e3 / steps
    return {"workload": f"{name.upper()}: d={d} {fam_name}, N={N}, M={M}, {acq_name.upper()}, top-{k_top}, full refit every step",
            "value": ms, "unit": "ms", "steps": steps, "warmup": warmup}


def quick_nlml_grad(abo, synth, torch, dev, local_rank, N=8192, d=8, reps=7):
    """One objective evaluation of the hyper-parameter search (nlml, StandardGP.jl:99-114, under
    optimize_hyperparameters, bayesian_opt.jl:196-328) at the headline's N: refit + abo_nlml_grad (value and analytic gradient
    w.r.t. log ell, log sigma_f2).  Its dominant kernel beyond the refit is K^-1 = L^-T L^-1 on the fp64 MFMA tile core (N^3/3
    flop on the lower tiles); tools/hyperparameter_latency.py prints the same at three sizes and a whole optimize_hyperparameters."""
    import ctypes as C
    X, y = synth.standardized_problem(N, d, 0.03)
    Xd, yd = torch.from_numpy(X).to(dev), torch.from_numpy(y).to(dev)
    walls, tms = [], []
    for r in range(reps + 2):
        torch.cuda.synchronize(dev)
        t0 = time.perf_counter()
        m = abo.update(abo.HipStandardGP(abo.with_lengthscale(abo.Matern52Kernel(), 1.0), 1e-3, device=local_rank), Xd, yd)
        v, d1, d2 = C.c_double(), C.c_double(), C.c_double()
        abo._lib.check(abo._lib.lib().abo_nlml_grad(m._require(), C.byref(v), C.byref(d1), C.byref(d2)))
        if r >= 2:
            walls.append((time.perf_counter() - t0) * 1e3)
            tms.append(m.timings())
    med = {k: float(np.median([t[k] for t in tms])) for k in tms[0]}
    tf = N ** 3 / 3.0 / (med["nlml_kinv_ms"] * 1e-3) / 1e12 if med["nlml_kinv_ms"] > 0 else None
    return {"workload": f"hyper-parameter objective: N={N}, d={d}, Matern52Kernel: refit + nlml value and analytic gradient (abo_nlml_grad)",
            "value": float(np.median(walls)), "unit": "ms", "steps": reps, "warmup": 2,
            "phases_ms": {"refit": med["fit_total_ms"], "kinv_gemm": med["nlml_kinv_ms"], "trace_sweep": med["nlml_trace_ms"]},
            "roofline": {"kernel": "gemm_nt_kernel (K^-1 = L^-T L^-1, lower tiles)", "bound": "mfma", "achieved": tf, "peak": 78.6,
                         "unit": "TFLOP/s", "frac": (tf / 78.6 if tf else None), "traffic": None,
                         "note": "algorithmic flop N^3/3 / HIP events around the product (abo_timings.nlml_kinv_ms)"},
            "nlml": v.value, "grad": [d1.value, d2.value]}


def quick_c1_shape(abo, synth, torch, dev, local_rank, k_top=100, steps=200, warmup=20):
    """The reference's own loop size (BASELINE config 1's shape: tens of points, acq_utils.jl:37's 10 000 grid points): a step =
    refit + EI over the grid + top-100, through the two C-ABI calls and through the fused one (abo_fit_acq); and one whole
    `optimize_acquisition` (device LHS grid → scores → top-100 → on-device L-BFGS refinement of every start → best point) in one
    call (abo_optimize_acquisition) at N = 100, d = 2."""
    N, d, M = 25, 1, 10_000
    X = synth.points(1, N, d)
    y = np.sin(10.0 * X[:, 0])
    Xd, yd, Zd = torch.from_numpy(X).to(dev), torch.from_numpy(y).to(dev), torch.from_numpy(synth.points(2, M, d)).to(dev)
    gp = abo.HipStandardGP(abo.with_lengthscale(abo.Matern52Kernel(), 0.3), 1e-6, device=local_rank)
    acq = abo.ExpectedImprovement(0.0, float(y.min()))
    out = {"workload": f"C1 shape: d={d} Matern52Kernel, N={N}, M={M} grid, EI, top-{k_top}, full refit every step", "unit": "ms",
           "steps": steps, "warmup": warmup}
    for name, fused in (("value", False), ("fused_call_ms", True)):
        for step in range(warmup + steps):
            if step == warmup:
                torch.cuda.synchronize(dev)
                t0 = time.perf_counter()
            if fused:
                abo.update_and_evaluate(acq, gp, Xd, yd, Zd, k=k_top, return_scores=False, best_y=acq.best_y)
            else:
                model = abo.update(gp, Xd, yd)
                abo.evaluate(acq, model, Zd, k=k_top, return_scores=False)
        torch.cuda.synchronize(dev)
        out[name] = (time.perf_counter() - t0) * 1e3 / steps
    t = model.timings()
    out["device_ms"] = {"fit": t["fit_total_ms"], "acq": t["acq_total_ms"]}
    out["phase_events"] = {"1": "on (ABO_PHASE_EVENTS=1)"}.get(os.environ.get("ABO_PHASE_EVENTS", ""), "automatic: off for a one-block model (N <= 128)")
    # optimize_acquisition in one call at the reference's stock sizes (n_grid = 10 000, n_local = 100)
    N2, d2 = 100, 2
    X2 = synth.points(1, N2, d2)
    y2 = np.sin(3 * X2).sum(axis=1)
    y2 = (y2 - y2.mean()) / y2.std(ddof=1)
    m2 = abo.update(abo.HipStandardGP(abo.with_lengthscale(abo.Matern52Kernel(), 0.4), 1e-4, device=local_rank), X2, y2)
    dom = abo.ContinuousDomain(np.zeros(d2), np.ones(d2))
    ucb = abo.UpperConfidenceBound(2.0)
    wall = []
    for r in range(25):
        t0 = time.perf_counter()
        abo.optimize_acquisition_device(ucb, m2, dom, 10_000, 100, seed=r)
        wall.append((time.perf_counter() - t0) * 1e3)
    t2 = m2.timings()
    out["optimize_acquisition"] = {"workload": f"N={N2}, d={d2}, UCB(2), n_grid=10000, n_local=100: grid + top-100 + on-device L-BFGS of "
                                               "every start + arg-max in ONE C-ABI call (abo_optimize_acquisition)",
                                   "value": float(np.median(wall[5:])), "unit": "ms", "device_grid_ms": t2["acq_total_ms"],
                                   "device_refine_ms": t2["refine_ms"], "acquisition_evaluations": int(t2["refine_evals"])}
    return out


def exchange_record(args, use_dist, local_rank):
    """What the collective layer itself saw: backend, world size from the process group, and the device every rank
    runs on — gathered through the group (outside the timed region), so the first multi-GPU line can be checked for N ranks on N
    distinct devices without reading logs."""
    import torch
    if not use_dist:
        return {"backend": None, "world_size": 1, "ranks": [{"rank": 0, "device": local_rank, "host": os.uname().nodename}],
                "note": "one process, no process group"}
    import torch.distributed as dist
    mine = {"rank": dist.get_rank(), "device": local_rank, "host": os.uname().nodename}
    try:
        mine["device_uuid"] = str(torch.cuda.get_device_properties(local_rank).uuid)
    except Exception:                                   # noqa: BLE001 - older torch: no uuid field
        pass
    ranks = [None] * dist.get_world_size()
    dist.all_gather_object(ranks, mine)
    rec = {"backend": dist.get_backend(), "world_size": dist.get_world_size(), "ranks": ranks,
           "distinct_devices": len({(r["host"], r.get("device_uuid", r["device"])) for r in ranks})}
    if args.backend == "nccl":
        rec["library"] = "RCCL through torch.distributed (backend nccl)"
    return rec


def run_c5(args, cfg, world, rank, local_rank, dev, use_dist=False, steps=None, warmup=None):
    """BASELINE config 5.  A step = greedy q-EI over the resident grid (q = 8 picks, each: EI + arg-max,
    fantasy bordered append, O(N·M) down-date), roll the grid posterior back, append the real (noisy)
    observation of the first pick to the parent model, down-date.  The full refresh (refit + grid
    re-evaluation: due when the factor's spare rows are used up — --refresh-every — and whenever hyper-parameters
    change; appends alone never need one, profiles/r06_c5_refresh_drift.txt) is timed once in setup and
    reported as refresh_ms / value_amortized."""
    import torch
    import torch.distributed as dist

    import abstractbayesopt.jl_amd as abo
    from abstractbayesopt.jl_amd import distributed as D
    from abstractbayesopt.jl_amd import synth

    fam_name, d, N, M_per, ell, sf2, noise, _, xi = cfg
    n_steps = args.steps if steps is None else steps
    n_warm = args.warmup if warmup is None else warmup
    Q = 8
    M_total = M_per * world
    lo, hi = D.shard_range(M_total, rank, world)
    X = synth.points(1, N, d)
    y_raw = synth.objective(X, noise_std=float(np.sqrt(noise)))
    y_mean, y_std = y_raw.mean(), y_raw.std(ddof=1)
    y = (y_raw - y_mean) / y_std
    Zd = torch.from_numpy(synth.points(2, hi - lo, d, first=lo)).to(dev)
    # Refresh cadence (full refit + grid re-evaluation).  Round 6 MEASURED what rounds 1 - 5 assumed (16): 64 real appends with no
    # refresh leave the appended factor and the down-dated grid where an independent oracle refit puts them — max |dL| 6.4e-14,
    # |dmu| 7.0e-12, |dvar| 6.9e-14 after 64 appends against 5.2e-14, 5.2e-12, 6.9e-14 after 16 (profiles/r06_c5_refresh_drift.txt):
    # there is no drift to bound, appends alone never force a refresh.  What does is the factor's capacity (n_max rows) and a change of
    # hyper-parameters (the reference re-optimises them every 10 iterations when asked to, bayesian_opt.jl:388 — a refit either way).
    # The cadence is therefore the capacity the caller gives the model: --refresh-every, n_max = N + cadence.  Default 512: the same
    # measurement run for 512 appends (|dmu| 1.5e-11, |dvar| 9.5e-14, |dL| 6.9e-14, alpha 1.2e-11, top-100 of EI unchanged).  Over such a
    # cycle the q-EI state is rebuilt (one pass over K_ZX) whenever its chain of 64 conditioning columns is used up — about every 56
    # steps: value_amortized carries that term too.
    cadence = max(1, int(getattr(args, "refresh_every", 512) or 512))
    if n_warm + n_steps + 3 > cadence:
        raise SystemExit(f"bench.py --config c5: {n_warm} + {n_steps} (+ 3 untimed) appends exceed the model's capacity N + {cadence} "
                         "(--refresh-every): a refresh inside the timed region is not what this line times")
    gp = abo.HipStandardGP(sf2 * abo.with_lengthscale(getattr(abo, fam_name)(), ell), noise, device=local_rank,
                           n_max=N + cadence, chunk=args.chunk)

    def sync():
        torch.cuda.synchronize(dev)
        if use_dist:
            dist.barrier()
            torch.cuda.synchronize(dev)

    Xd, yd = torch.from_numpy(X).to(dev), torch.from_numpy(y).to(dev)
    model = abo.update(gp, Xd, yd)
    cands = abo.ResidentCandidates(model, Zd)               # first creation: one-time allocations (17 GB of K_ZX), untimed
    # the periodic full refresh: refit + grid re-evaluation.  One untimed repetition first (a BO loop refreshes every 16 steps with
    # the buffer pool warm: the first refresh after a cold start also pays hipMalloc for a second model's 10+ GB), then the median of 3
    refresh_all = []
    for rep in range(4):
        sync()
        t0 = time.perf_counter()
        model = abo.update(gp, Xd, yd)
        cands.refresh(model)
        sync()
        refresh_all.append((time.perf_counter() - t0) * 1e3)
    refresh_ms = float(np.median(refresh_all[1:]))
    fit_t = model.timings()
    best_y = float(y.min())
    ph = {"qei_ms": [], "append_ms": [], "downdate_ms": [], "downdate_pass_ms": [], "downdate_pass_bytes": [], "block_ms": [],
          "block_pass_ms": [], "block_pass_bytes": [], "block_pass_flop": [], "append_trmv_ms": [], "append_trmv_bytes": []}
    builds_timed = 0
    obs_rng = np.random.default_rng(7)
    counts = {"block_builds": 0, "block_hits": 0, "downdates_from_chain": 0, "block": 0}
    picks = None
    blk = None if args.qei_block is None else int(args.qei_block)
    for step in range(n_warm + n_steps):
        if step == n_warm:
            sync()
            t_start = time.perf_counter()
        ta = time.perf_counter()
        st = {}
        # the batch only: model and grid are as before on return (block form: no fantasy appends, one pass over K_ZX per block)
        pts, idxs, vals, _ = abo.greedy_qei(model, cands, Q, xi, best_y, idx_base=lo, rollback=True, block=blk, stats=st)
        tb = time.perf_counter()
        x_new = pts[0]
        # the objective is NOISY (config 5): the real observation carries the noise the model assumes.  (Rounds 4 - 6a observed the
        # noise-free value: the posterior then hardly moves between steps, every pick is found in a carried-over block and the step looks
        # 0.3 ms cheaper than a BO loop on a noisy objective is — tools/c5_cycle.py.)  Same seed on every rank: same appends.
        y_new = ((np.sin(2 * np.pi * x_new).sum() / np.sqrt(d) + np.sqrt(noise) * obs_rng.standard_normal()) - y_mean) / y_std
        model = abo.append(model, x_new, float(y_new))
        td = time.perf_counter()
        cands.downdate(model)                              # block form: the column of pick 1 is in the batch's chain — no pass
        te = time.perf_counter()
        dd = model.timings()
        best_y = min(best_y, float(y_new))
        picks = (idxs, vals)
        if step >= n_warm:
            ph["qei_ms"].append((tb - ta) * 1e3)
            ph["append_ms"].append((td - tb) * 1e3); ph["downdate_ms"].append((te - td) * 1e3)
            ph["downdate_pass_ms"].append(dd["downdate_ms"]); ph["downdate_pass_bytes"].append(dd["downdate_bytes"])
            ph["append_trmv_ms"].append(dd.get("append_trmv_ms", 0.0)); ph["append_trmv_bytes"].append(dd.get("append_trmv_bytes", 0.0))
            counts["downdates_from_chain"] += int(dd.get("downdate_from_chain", 0))
            counts["block"] = int(st.get("block", 0))
            if st.get("block", 0):
                counts["block_builds"] += int(st["block_builds"]); counts["block_hits"] += int(st["block_hits"])
                ph["block_ms"].append(st["block_ms"])
        # the pass over K_ZX: every step that built a block, warm-up steps included (blocks and chain follow the model from step to
        # step: most timed steps find all their picks' columns in earlier blocks and stream nothing at all)
        if st.get("block", 0) and st.get("block_builds", 0) > 0 and st.get("pass_ms", 0.0) > 0.0:
            ph["block_pass_ms"].append(st["pass_ms"])
            ph["block_pass_bytes"].append(st["pass_bytes"]); ph["block_pass_flop"].append(st["pass_flop"])
    sync()
    elapsed = time.perf_counter() - t_start
    if use_dist:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev if args.backend == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    ms = elapsed * 1e3 / n_steps
    xchg = exchange_record(args, use_dist, local_rank)
    # The same step when NOTHING carries over from the step before (ABO_QEI_NO_REUSE: every batch builds its own block — one pass
    # over K_ZX per step): three untimed-for-`value` steps.  `value` is what the loop above measured — a BO loop, where the blocks and
    # the chain follow the model and a step streams K_ZX only when a pick falls outside every block; this is the step without that.
    fresh_ms = []
    if counts["block"]:
        os.environ["ABO_QEI_NO_REUSE"] = "1"
        for _ in range(3):
            sync()
            t0 = time.perf_counter()
            pts, idxs2, vals2, _ = abo.greedy_qei(model, cands, Q, xi, best_y, idx_base=lo, rollback=True, block=blk)
            x_new = pts[0]
            y_new = ((np.sin(2 * np.pi * x_new).sum() / np.sqrt(d) + np.sqrt(noise) * obs_rng.standard_normal()) - y_mean) / y_std
            model = abo.append(model, x_new, float(y_new))
            cands.downdate(model)
            sync()
            fresh_ms.append((time.perf_counter() - t0) * 1e3)
            best_y = min(best_y, float(y_new))
        del os.environ["ABO_QEI_NO_REUSE"]
    out = None
    del cands, model
    if rank == 0:
        med = {k: (float(np.median(v)) if len(v) else 0.0) for k, v in ph.items()}
        n_now = N + n_warm + n_steps
        append_bytes = 8.0 * n_now * n_now                 # W (lower) + WT (upper), read once each
        ach = append_bytes / (med["append_ms"] * 1e-3) / 1e9
        tr, tr_src = pmc_traffic("c5", M_per)              # committed rocprofv3 FETCH_SIZE pass (null when stale)
        tsrc = {k: tr_src.get(k) for k in ("file", "kernel_source_sha", "stale", "note") if tr_src and k in tr_src}
        block_roof = None
        if counts["block"] and med["block_pass_ms"] > 0:
            # the ONE product over the resident K_ZX that gives the covariance columns of a whole block — HBM-bound; its flop ride
            # under the stream.  In the running loop it is RARE (blocks and chain follow the model): its own launches_per_step says
            # how often the timed steps ran it, and it is the line's `roofline` only when they did.
            gbs = med["block_pass_bytes"] / (med["block_pass_ms"] * 1e-3) / 1e9
            block_roof = {"kernel": f"qei_passd_kernel<{counts['block'] // 16}> (C0 = -K_ZX . K^-1 K_XT: T = {counts['block']} "
                                    f"covariance columns from ONE pass over the resident K_ZX)", "bound": "hbm",
                          "achieved": gbs, "peak": 8000.0, "unit": "GB/s", "frac": gbs / 8000.0, "traffic": tr, "traffic_source": tsrc,
                          "algorithmic_bytes_per_launch": med["block_pass_bytes"], "avg_launch_ms": med["block_pass_ms"],
                          "launches_per_step": counts["block_builds"] / n_steps,
                          "mfma_tflops_under_the_stream": med["block_pass_flop"] / (med["block_pass_ms"] * 1e-3) / 1e12,
                          "launches_timed": len(ph["block_pass_ms"]),
                          "note": "algorithmic bytes = 8*N*M (K_ZX read once per block of T picks' columns; the plain loop reads it "
                                  "once per pick); duration = HIP events on the library stream, median over every step that built "
                                  "a block, WARM-UP STEPS INCLUDED (launches_per_step counts the timed steps only)"}
        if counts["block"] and med["append_trmv_ms"] > 0 and (block_roof is None or counts["block_builds"] < n_steps):
            # the timed step's dominant kernel BY KERNEL TIME: the bordered append's two triangular mat-vecs (trmv_kernel: l = L^-1 k,
            # v = L^-T l), each streaming one triangle of its N x N matrix once — 2 launches per step, HBM-bound
            gbs = med["append_trmv_bytes"] / (med["append_trmv_ms"] * 1e-3) / 1e9
            roof = {"kernel": "trmv_kernel x2 (bordered append: l = L^-1 k over the lower triangle of L^-1, v = L^-T l over the upper "
                              "triangle of L^-T)", "bound": "hbm", "achieved": gbs, "peak": 8000.0, "unit": "GB/s",
                    "frac": gbs / 8000.0, "traffic": trmv_traffic(tr_src),
                    "traffic_unit": "bytes per launch (PMC FETCH_SIZE x2, profiles/)",
                    "algorithmic_bytes_per_launch": med["append_trmv_bytes"] / 2.0, "avg_launch_ms": med["append_trmv_ms"] / 2.0,
                    "launches_per_step": 2,
                    "share_of_step": med["append_trmv_ms"] / ms if ms > 0 else None,
                    "note": "algorithmic bytes = 8*N^2 for the pair (one triangle of each matrix, read once); duration = HIP events "
                            "on the library stream around the two launches, median over the timed steps.  The q-EI batch of a timed "
                            "step runs q + 1 = 9 launches of qei_step_kernel (pick loop on the device, no pass over K_ZX: the "
                            "block build is block_build_roofline, with its own launches_per_step)"}
        elif block_roof is not None:
            roof, block_roof = block_roof, None
        elif med["downdate_pass_bytes"] > 0:
            # the plain loop (--qei-block 0): one O(N*M) down-date pass per pick, 8 launches per step (7 fantasies + the real point)
            gbs = med["downdate_pass_bytes"] / (med["downdate_pass_ms"] * 1e-3) / 1e9
            roof = {"kernel": "cand_gemv_kernel (c = K_ZX . [-v; 1] over the resident K_ZX) + new-column kernel", "bound": "hbm",
                    "achieved": gbs, "peak": 8000.0, "unit": "GB/s", "frac": gbs / 8000.0, "traffic": tr, "traffic_source": tsrc,
                    "algorithmic_bytes_per_launch": med["downdate_pass_bytes"], "avg_launch_ms": med["downdate_pass_ms"],
                    "launches_per_step": Q,
                    "note": "algorithmic bytes = 8*N*M (K_ZX read once); duration = HIP events on the library stream"}
        else:
            pairs = n_now * M_per / (max(med["downdate_pass_ms"], 1e-9) * 1e-3)
            roof = {"kernel": "kgen_kernel, dot-only mode (K_ZX re-evaluated: ABO_CAND_KZX_GIB budget too small)", "bound": "valu",
                    "achieved": pairs / 1e9, "peak": None, "unit": "Gpair/s", "frac": None, "traffic": None,
                    "avg_launch_ms": med["downdate_pass_ms"], "launches_per_step": Q}
        form = (f"block form, T={counts['block']}: covariance columns of the T best candidates from one pass over K_ZX, rank-1 "
                f"corrections between picks, no fantasy appends; blocks and chain follow the model from step to step; the real "
                f"append's column from the batch's chain") if counts["block"] \
            else "plain loop: fantasy append + O(N*M) down-date per pick"
        out = {
            "metric": "GP-update+acq-eval ms per BO step at N train pts x M candidates",
            "value": ms, "unit": "ms", "n_gpus": world, "steps": n_steps, "warmup": n_warm, "ms_per_step": ms,
            "higher_is_better": False, "scaling": "weak", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": f"C5: d={d} {fam_name} ell={ell} noise={noise}, N={N}(+1 per step) train, resident grid "
                                   f"M={M_per} per GPU ({M_total} total), greedy q-EI q={Q} ({form}) + 1 real bordered append per step",
                       "N": N, "M_per_gpu": M_per, "M_total": M_total, "d": d, "q": Q,
                       "sharding": f"grid x{world}, all_gather of one pick record per pick (+ T records per block)",
                       "exchange": xchg},
            "qei": {"block": counts["block"], "block_builds_per_step": counts["block_builds"] / n_steps,
                    "picks_found_in_a_block": counts["block_hits"], "picks_conditioned": (Q - 1) * n_steps,
                    "block_hit_rate": counts["block_hits"] / max(1, counts["block_hits"] + counts["block_builds"]),
                    "real_appends_from_chain": counts["downdates_from_chain"], "real_appends": n_steps,
                    "step_with_a_fresh_block_ms": float(np.median(fresh_ms)) if fresh_ms else None,
                    "note": "value = the BO loop as it runs: blocks and chain follow the model from step to step, a step streams "
                            "K_ZX (one pass per block) only when a pick falls outside every block; step_with_a_fresh_block_ms = "
                            "the same step with nothing carried over (every batch builds its block)"},
            "refresh_ms": refresh_ms, "refresh_ms_all": refresh_all,
            "value_amortized": ms + refresh_ms / cadence,
            "refresh": {"every_steps": cadence, "value_amortized_at_64": ms + refresh_ms / 64.0, "value_amortized_at_16": ms + refresh_ms / 16.0,
                        "why": "value_amortized = value + refresh_ms / every_steps: a refresh when the factor's capacity (n_max = N + every_steps) "
                               "is used up or the hyper-parameters change.  value itself is the MEAN over the timed steps, the ones that rebuild "
                               "a q-EI block (qei.block_builds_per_step: one pass over K_ZX each) included - time at least 48 steps for a "
                               "representative mix (tools/c5_cycle.py: 512 steps).  NOT numerical drift: 512 appends without a refresh stay at "
                               "1.5e-11 (mu) / 9.5e-14 (var) / 6.9e-14 (L) of an independent oracle refit, 64 appends at 8e-12 / 9.5e-14 / 6.2e-14 "
                               "(profiles/r06_c5_refresh_drift.txt, r06_c5_refresh_drift_512.txt)"},
            "refresh_phases_ms": {k: v for k, v in fit_t.items() if k.endswith("_ms")},
            "roofline": roof,
            "block_build_roofline": block_roof,
            "secondary_roofline": {"kernel": "the whole abo_append call (k-row kernel, trmv_kernel x2, two tiny kernels, host sync)",
                                   "bound": "hbm", "achieved": ach, "peak": 8000.0, "unit": "GB/s", "frac": ach / 8000.0,
                                   "note": "algorithmic bytes 8*N^2 per append; duration = host wall-clock of the synchronous "
                                           "abo_append call"},
            "phases_ms": med,
            "last_batch": {"indices": [int(i) for i in picks[0]], "ei": [float(v) for v in picks[1]]},
        }
    return out


def plan_launch(args, environ):
    """Which host drives this invocation: 'ranks' (one process per GPU; the driver's torchrun launch, WORLD_SIZE set, or any
    1-GPU run) or 'library' (a plain `python bench.py --gpus N`, N > 1: ONE process drives the N devices through the library's
    multi-device handle abo_mgpu_* — the deployment north_star describes, a Julia host has no process launcher).  No GPU is
    touched here, so the decision is testable on a CPU-only machine."""
    world = int(environ.get("WORLD_SIZE", "1"))
    if args.single_process:
        if world != 1:
            raise SystemExit("--single-process is for a plain `python bench.py --gpus N --single-process` launch, not torchrun")
        return "library"
    if world == 1 and args.gpus > 1 and "RANK" not in environ:
        return "library"
    if world != args.gpus:
        raise SystemExit(f"torchrun started {world} rank(s) but --gpus says {args.gpus}")
    return "ranks"


def run_single_process(args):
    """`python bench.py --gpus N` (no launcher) — the C3/C4-shaped step driven through ONE multi-device handle (abo_mgpu_*), the
    shape a Julia host uses.  The candidate grid is generated on the devices (Latin hypercube, shard by shard) and stays
    resident; a step = replicated full refit on every device + posterior/EI over every shard + per-device top-100 +
    ONE RCCL all-gather of 100 x (score, index) per device + merge (acq_utils.jl:50-52 reproduced globally).  --share-device
    puts all shards on GPU 0 (rehearsal on a one-GPU box: RCCL refuses a device listed twice, the exchange then goes through
    the host, and the JSON line says so)."""
    import torch                                        # device count only (does not initialise the GPU)

    fam_name, d, N, M_per, ell, sf2, noise, acq_name, p0 = CONFIGS[args.config]
    G = args.gpus
    have = torch.cuda.device_count()
    if not args.share_device and have < G:
        raise SystemExit(f"bench.py --gpus {G}: this machine shows {have} GPU(s); --share-device rehearses {G} shards on GPU 0")
    if args.share_device:
        # all shards on one device share ITS buffer pool (32 GiB by default: sized for one shard per device); eight shards recycle
        # 8 × 19 GB of factor + int8 scratch per step — beyond the limit every step would pay hipFree / hipMalloc for the excess
        os.environ.setdefault("ABO_POOL_LIMIT_MB", str(230 * 1024))
    if args.config == "c5":
        return run_single_process_c5(args)
    # a resident grid would otherwise also keep its fp64 K_ZX (64 GiB per device at C3: the down-date path of config 5);
    # this step re-evaluates the posterior from scratch, as abo_acq does
    os.environ["ABO_CAND_KZX_GIB"] = "0"
    if args.contraction:
        os.environ["ABO_CONTRACTION"] = args.contraction

    import abstractbayesopt.jl_amd as abo
    from abstractbayesopt.jl_amd import synth

    strong = args.config == "c4"
    M_total = M_per if strong else M_per * G
    M_per = M_total // G
    devices = [0] * G if args.share_device else list(range(G))
    X, y = synth.standardized_problem(N, d, 0.03)
    gp = abo.HipShardedGP(sf2 * abo.with_lengthscale(getattr(abo, fam_name)(), ell), noise, devices=devices, chunk=args.chunk)
    best_y = float(y.min())
    acq = abo.ExpectedImprovement(p0, best_y) if acq_name == "ei" else abo.UpperConfidenceBound(p0)
    model = abo.update(gp, X, y)
    cands = abo.ShardedCandidates(model, lhs=(M_total, np.zeros(d), np.ones(d), 2))
    L = abo._lib.lib()

    def shard_timings(m):
        out = []
        for i in range(G):
            t = abo._lib.AboTimings()
            abo._lib.check(L.abo_get_timings(m.shard(i), t))
            out.append(t.as_dict())
        return out

    step_ms, per_dev, phases = [], [], []
    for step in range(args.warmup + args.steps):
        if step == args.warmup:
            t_all = time.perf_counter()                   # every call below returns with all devices idle: no barrier needed
        t0 = time.perf_counter()
        model = abo.update(gp, X, y)                      # full refit on every device (X, y: 0.5 MiB of host data)
        cands.refresh(model)                              # posterior of every resident shard
        tv, ti = cands.evaluate(model, acq, 100)          # epilogue + per-device top-100 + all-gather + merge
        if step >= args.warmup:
            step_ms.append((time.perf_counter() - t0) * 1e3)
            st = shard_timings(model)
            per_dev.append([t["fit_total_ms"] + t["acq_total_ms"] for t in st])
            phases.append(st[0])
    ms = (time.perf_counter() - t_all) * 1e3 / args.steps
    med = {k: float(np.median([p[k] for p in phases])) for k in phases[0]}
    exchange, why = model.exchange(), model.exchange_note()
    out = {
        "metric": "GP-update+acq-eval ms per BO step at N train pts x M candidates",
        "value": ms, "unit": "ms", "n_gpus": G, "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms,
        "higher_is_better": False, "scaling": "strong" if strong else "weak", "vs_baseline": None, "dtype": "f64",
        "data": "synthetic",
        "config": {"workload": f"{args.config.upper()}: d={d} {fam_name} ell={ell} sigma_f2={sf2} noise={noise}, N={N} train, "
                               f"M={M_per} candidates per GPU ({M_total} total), {acq_name.upper()} p0={p0}, top-100, "
                               f"full refit every step",
                   "N": N, "M_per_gpu": M_per, "M_total": M_total, "d": d, "kernel": fam_name, "acq": acq_name,
                   "host": "one process, library-owned worker thread per device (abo_mgpu_*); grid resident, device-generated "
                           "Latin hypercube",
                   "sharding": f"candidates x{G} contiguous, replicated refit, one all-gather of top-100 (score, index) per device",
                   "devices": devices, "distinct_devices": len(set(devices)),
                   "exchange": exchange, "exchange_note": why if exchange == "host" else "ncclAllGather over the devices' streams",
                   "rccl_ranks": G if exchange == "rccl" else 0,
                   "contraction": contraction_label(abo, med)},
        "candidates_per_s": M_total / (ms * 1e-3),
        "roofline": dominant_kernel_roofline(abo, med, args.config, N, M_per),
        "phases_ms_device0": {k: v for k, v in med.items() if k.endswith("_ms")},
        "per_device_hip_event_ms_per_step": [float(v) for v in np.median(np.asarray(per_dev), axis=0)],
        "median_ms_per_step": float(np.median(step_ms)), "min_ms_per_step": float(np.min(step_ms)),
        "max_ms_per_step": float(np.max(step_ms)),
        "top1": {"score": float(tv[0]), "index": int(ti[0])},
    }
    if len(set(devices)) < G:
        out["rehearsal"] = f"{G} shards share {len(set(devices))} physical device(s): NOT a scaling measurement"
    print(json.dumps(out))


def run_single_process_c5(args):
    """BASELINE config 5 through the multi-device handle: a step = abo_mgpu_cand_qei (q = 8 greedy picks, each: EI + arg-max per
    device, ONE all-gather of the devices' pick records; block form: the covariance columns of the T best candidates from one pass
    over every device's K_ZX, no fantasy appends — or, --qei-block 0, the same fantasy append + O(N·M) down-date on every device per
    pick) + abo_mgpu_append of the real observation with the grid's down-date in the same call (its column from the batch's chain)."""
    import abstractbayesopt.jl_amd as abo
    from abstractbayesopt.jl_amd import multigpu, synth

    fam_name, d, N, M_per, ell, sf2, noise, _, xi = CONFIGS["c5"]
    G, Q = args.gpus, 8
    M_total = M_per * G
    devices = [0] * G if args.share_device else list(range(G))
    if args.qei_block is not None:
        abo._lib.check(abo._lib.lib().abo_set_qei_block(int(args.qei_block)))
    X = synth.points(1, N, d)
    y_raw = synth.objective(X, noise_std=float(np.sqrt(noise)))
    y_mean, y_std = y_raw.mean(), y_raw.std(ddof=1)
    y = (y_raw - y_mean) / y_std
    gp = abo.HipShardedGP(sf2 * abo.with_lengthscale(getattr(abo, fam_name)(), ell), noise, devices=devices, chunk=args.chunk,
                          n_max=N + 64)
    model = abo.update(gp, X, y)
    cands = abo.ShardedCandidates(model, lhs=(M_total, np.zeros(d), np.ones(d), 2))
    refresh_all = []
    for rep in range(3):
        t0 = time.perf_counter()
        model = abo.update(gp, X, y)
        cands.refresh(model)
        refresh_all.append((time.perf_counter() - t0) * 1e3)
    best_y = float(y.min())
    step_ms, picks = [], None
    for step in range(args.warmup + args.steps):
        if step == args.warmup:
            t_all = time.perf_counter()
        t0 = time.perf_counter()
        pts, idxs, vals = cands.greedy_qei(model, Q, xi, best_y)
        qst = cands.qei_stats(model)
        x_new = pts[0]
        y_new = ((np.sin(2 * np.pi * x_new).sum() / np.sqrt(d)) - y_mean) / y_std
        model = multigpu.append(model, x_new, float(y_new), cands)
        best_y = min(best_y, float(y_new))
        picks = (idxs, vals)
        if step >= args.warmup:
            step_ms.append((time.perf_counter() - t0) * 1e3)
    ms = (time.perf_counter() - t_all) * 1e3 / args.steps
    exchange, why = model.exchange(), model.exchange_note()
    out = {
        "metric": "GP-update+acq-eval ms per BO step at N train pts x M candidates",
        "value": ms, "unit": "ms", "n_gpus": G, "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms,
        "higher_is_better": False, "scaling": "weak", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
        "config": {"workload": f"C5: d={d} {fam_name} ell={ell} noise={noise}, N={N}(+1 per step) train, resident grid M={M_per} per "
                               f"GPU ({M_total} total), greedy q-EI q={Q} (abo_mgpu_cand_qei; block form unless qei_last_step.block = 0) + 1 real "
                               f"bordered append per step",
                   "N": N, "M_per_gpu": M_per, "M_total": M_total, "d": d, "q": Q,
                   "host": "one process, library-owned worker thread per device (abo_mgpu_cand_qei / abo_mgpu_append)",
                   "sharding": f"grid x{G}, one all-gather of (score, index, mu, x) per pick",
                   "devices": devices, "distinct_devices": len(set(devices)), "exchange": exchange,
                   "exchange_note": why if exchange == "host" else "ncclAllGather over the devices' streams",
                   "rccl_ranks": G if exchange == "rccl" else 0},
        "refresh_ms": float(np.median(refresh_all[1:])), "refresh_ms_all": refresh_all,
        "median_ms_per_step": float(np.median(step_ms)), "min_ms_per_step": float(np.min(step_ms)),
        "max_ms_per_step": float(np.max(step_ms)),
        "last_batch": {"indices": [int(i) for i in picks[0]], "ei": [float(v) for v in picks[1]]},
        "qei_last_step": qst,
    }
    if len(set(devices)) < G:
        out["rehearsal"] = f"{G} shards share {len(set(devices))} physical device(s): NOT a scaling measurement"
    print(json.dumps(out))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--config", default="c3", choices=sorted(CONFIGS))
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-secondary", action="store_true", help="skip the C2 and C5 entries of the default C3 line (profiling runs)")
    ap.add_argument("--cpu-sample-m", type=int, default=21846,
                    help="candidates PER REPETITION of the CPU-baseline sample (default 3 x 21846 = 65538 in all; BASELINE.md §4 "
                         "allows timing 65536 candidates and scaling linearly in M)")
    ap.add_argument("--cpu-reps", type=int, default=3, help="repetitions of the CPU-baseline sample (median reported)")
    ap.add_argument("--single-process", action="store_true",
                    help="drive --gpus N devices from THIS process through the library's multi-device handle (abo_mgpu_*, "
                         "RCCL all-gather inside the library); the default for a plain `python bench.py --gpus N` with N > 1 — "
                         "under torchrun (WORLD_SIZE set) every rank drives its own GPU instead")
    ap.add_argument("--chunk", type=int, default=0, help="candidate chunk size override (0 = library default)")
    ap.add_argument("--qei-block", type=int, default=None,
                    help="config 5: points per block of the block-form greedy q-EI (default: the library's, 32; 0 = the plain loop "
                         "with one pass over K_ZX per pick, for A/B runs)")
    ap.add_argument("--refresh-every", type=int, default=512,
                    help="config 5: BO steps between full refreshes = spare rows of the factor (n_max = N + this); value_amortized = "
                         "value + refresh_ms / this (+ the q-EI block rebuilds of such a cycle).  512: what profiles/r06_c5_refresh_drift_512.txt supports "
                         "(no drift after 512 appends)")
    ap.add_argument("--contraction", default=None,
                    help="engine of the N^2*M variance contraction: auto (library default), fp64, int8 or int8:<moduli>")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                    help="collective backend; gloo + --share-device rehearses the N>1 path on a one-GPU box")
    ap.add_argument("--share-device", action="store_true", help="all ranks use GPU 0 (rehearsal only)")
    args = ap.parse_args()
    # the roofline figures of this line come from the library's per-kernel HIP events: ABO_PHASE_EVENTS=0 in the caller's environment
    # would blank them, so it is dropped here.  Unset, the library records them for every model of more than one row block (C2 / C3 /
    # C5) and leaves them out for the one-block models of the C1-shaped leg, where they are a fifth of a step (its entry says so).
    if os.environ.get("ABO_PHASE_EVENTS") == "0":
        del os.environ["ABO_PHASE_EVENTS"]

    if plan_launch(args, os.environ) == "library":
        return run_single_process(args)

    import torch
    import torch.distributed as dist

    import abstractbayesopt.jl_amd as abo
    from abstractbayesopt.jl_amd import distributed as D
    from abstractbayesopt.jl_amd import synth

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.share_device:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    # ABO_FORCE_DIST=1 takes the N>1 code path (process group, barriers, all_gather of the top-k) at world size 1:
    # that is how the RCCL calls are exercised on a one-GPU box (tests/test_gpu_distributed.py)
    use_dist = world > 1 or bool(os.environ.get("ABO_FORCE_DIST"))
    if use_dist:
        # a rank that dies or never arrives must end the run with an error, not hold the others in a collective for ever: the
        # process group's watchdog aborts a collective that has not completed after 5 minutes (a step is ≤ a few seconds)
        import datetime
        limit = datetime.timedelta(seconds=int(os.environ.get("ABO_BENCH_COLLECTIVE_TIMEOUT_S", "300")))
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=dev, timeout=limit)
        else:
            dist.init_process_group("gloo", timeout=limit)

    cfg = CONFIGS[args.config]
    if args.config == "c5":
        out = run_c5(args, cfg, world, rank, local_rank, dev, use_dist)
        if rank == 0:
            print(json.dumps(out))
        if use_dist:
            dist.destroy_process_group()
        return
    fam_name, d, N, M_per, ell, sf2, noise, acq_name, p0 = cfg
    strong = args.config == "c4"
    if strong:
        M_total = M_per
        M_per = M_total // world
    else:
        M_total = M_per * world
    lo, hi = D.shard_range(M_total, rank, world)

    # synthetic inputs, regenerated from counters on every rank; resident in HBM before timing
    X, y = synth.standardized_problem(N, d, 0.03)
    Xd, yd = torch.from_numpy(X).to(dev), torch.from_numpy(y).to(dev)
    Zd = torch.from_numpy(synth.points(2, hi - lo, d, first=lo)).to(dev)
    gp = abo.HipStandardGP(sf2 * abo.with_lengthscale(getattr(abo, fam_name)(), ell), noise, device=local_rank, chunk=args.chunk,
                           contraction=args.contraction)
    best_y = float(y.min())
    acq = abo.ExpectedImprovement(p0, best_y) if acq_name == "ei" else abo.UpperConfidenceBound(p0)
    K_TOP = 100

    def sync():
        torch.cuda.synchronize(dev)
        if use_dist:
            dist.barrier()
            torch.cuda.synchronize(dev)

    def reduce_max(v):
        t = torch.tensor([v], dtype=torch.float64, device=dev if args.backend == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item())

    phases = []
    step_ms = []
    model = None
    top = None
    for step in range(args.warmup + args.steps):
        if step == args.warmup:
            sync()
            t0 = time.perf_counter()
        ts = time.perf_counter()
        model = abo.update(gp, Xd, yd)                                          # full refit
        _, tv, ti = abo.evaluate(acq, model, Zd, k=K_TOP, idx_base=lo, return_scores=False)
        top = D.all_gather_topk(tv, ti, K_TOP) if use_dist else (tv, ti)
        if step >= args.warmup:
            phases.append(model.timings())
            step_ms.append((time.perf_counter() - ts) * 1e3)     # the C-ABI calls are synchronous: no extra sync needed
    sync()
    elapsed = time.perf_counter() - t0
    if use_dist:
        elapsed = reduce_max(elapsed)
    ms_per_step = elapsed * 1e3 / args.steps
    xchg = exchange_record(args, use_dist, local_rank)

    # SURVEY 8(d) asks for the metric in two shapes.  `value` above is "arg-max / top-100 only"; the reference's own
    # API shape (acq_utils.jl:50) also hands the M scores back to the host: timed here on two extra steps
    # (untimed for `value`), scores landing in a host array through the C-ABI's D2H copy.
    variants = None
    if world == 1:
        # the caller's arrays live across BO steps (one untimed step first: the first pass over 8 MiB of fresh pageable memory and
        # the library's score buffer are one-time costs — round 3 timed them into two steps and read +14 ms for an 8 MiB copy)
        host_scores, htv, hti = np.empty(M_per), np.empty(K_TOP), np.empty(K_TOP, dtype=np.int64)
        t1 = 0.0
        for it in range(4):
            if it == 1:
                torch.cuda.synchronize(dev)
                t1 = time.perf_counter()
            model = abo.update(gp, Xd, yd)
            st = abo._lib.lib().abo_acq(model._require(), Zd.data_ptr(), M_per, d, abo._lib.DEVICE, acq.kind, acq._p0(),
                                        acq._best(), lo, host_scores.ctypes.data, K_TOP, htv.ctypes.data,
                                        hti.ctypes.data, abo._lib.HOST)
            abo._lib.check(st)
        torch.cuda.synchronize(dev)
        variants = {"topk_only_ms": ms_per_step, "scores_to_host_ms": (time.perf_counter() - t1) * 1e3 / 3,
                    "scores_bytes_d2h": 8 * M_per}
        # The reference's call shape end to end (acq_utils.jl:47-50: `scores = acqf(surrogate, grid_points)` on host arrays; update(model,
        # xs, ys) on host arrays, StandardGP.jl:79-83): X, y and the candidate grid handed over as PAGEABLE host arrays on every step
        # (what a Julia `Matrix` is), the M scores returned to a host array — measured, not estimated.  Never `value`.
        Xh, yh = np.ascontiguousarray(X), np.ascontiguousarray(y)
        Zh = synth.points(2, hi - lo, d, first=lo)
        t3 = 0.0
        for it in range(4):
            if it == 1:
                torch.cuda.synchronize(dev)
                t3 = time.perf_counter()
            model = abo.update(gp, Xh, yh)
            st = abo._lib.lib().abo_acq(model._require(), Zh.ctypes.data, M_per, d, abo._lib.HOST, acq.kind, acq._p0(),
                                        acq._best(), lo, host_scores.ctypes.data, K_TOP, htv.ctypes.data,
                                        hti.ctypes.data, abo._lib.HOST)
            abo._lib.check(st)
        torch.cuda.synchronize(dev)
        variants["host_arrays_ms"] = (time.perf_counter() - t3) * 1e3 / 3
        variants["host_arrays_bytes_h2d"] = 8 * (N * d + N + M_per * d)
        variants["host_arrays_bytes_d2h"] = 8 * M_per + 16 * K_TOP
        variants["host_arrays_note"] = ("X, y, Z pageable host arrays every step, all M scores + top-100 back to host arrays: the "
                                        "reference's call shape (acq_utils.jl:47-50); value keeps the inputs resident in HBM")
        del Zh
        # the same step on the library's other contraction engine (two untimed-for-`value` steps), and how far the two
        # engines' selections and scores are apart on this very workload
        used = int(phases[0]["contraction_engine"])
        other = "fp64" if used == abo._lib.CONTRACT_INT8 else "int8"
        gp2 = abo.HipStandardGP(sf2 * abo.with_lengthscale(getattr(abo, fam_name)(), ell), noise, device=local_rank,
                                chunk=args.chunk, contraction=other)
        torch.cuda.synchronize(dev)
        t2 = time.perf_counter()
        for _ in range(2):
            m2 = abo.update(gp2, Xd, yd)
            _, tv2, ti2 = abo.evaluate(acq, m2, Zd, k=K_TOP, idx_base=lo, return_scores=False)
        torch.cuda.synchronize(dev)
        variants[f"{other}_engine_ms"] = (time.perf_counter() - t2) * 1e3 / 2
        host = lambda v: v.detach().cpu().numpy() if hasattr(v, "detach") else np.asarray(v)
        variants["engines_top100_same_indices"] = bool(np.array_equal(host(ti2), host(top[1])))
        variants["engines_top100_max_abs_score_diff"] = float(np.max(np.abs(host(tv2) - host(top[0]))))
        del m2, gp2

    if rank == 0:
        med = {k: float(np.median([p[k] for p in phases])) for k in phases[0]}
        int8 = int(med["contraction_engine"]) == abo._lib.CONTRACT_INT8
        roofline = dominant_kernel_roofline(abo, med, args.config, N, M_per)
        out = {
            "metric": "GP-update+acq-eval ms per BO step at N train pts x M candidates",
            "value": ms_per_step, "unit": "ms", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": ms_per_step, "higher_is_better": False, "scaling": "strong" if strong else "weak", "vs_baseline": None,
            "dtype": "f64", "data": "synthetic",
            "config": {"workload": f"{args.config.upper()}: d={d} {fam_name} ell={ell} sigma_f2={sf2} noise={noise}, "
                                   f"N={N} train, M={M_per} candidates per GPU ({M_total} total), "
                                   f"{acq_name.upper()} p0={p0}, top-{K_TOP}, full refit every step",
                       "N": N, "M_per_gpu": M_per, "M_total": M_total, "d": d, "kernel": fam_name, "acq": acq_name,
                       "sharding": f"candidates x{world}, all_gather top-{K_TOP}",
                       "exchange": xchg,
                       "contraction": contraction_label(abo, med)},
            "candidates_per_s": M_total / (ms_per_step * 1e-3),
            "roofline": roofline,
            "phases_ms": {k: v for k, v in med.items() if k.endswith("_ms")},
            "top1": {"score": float(top[0][0]), "index": int(top[1][0])},
            "hip_event_ms_per_step": med["fit_total_ms"] + med["acq_total_ms"],
            "median_ms_per_step": float(np.median(step_ms)), "min_ms_per_step": float(np.min(step_ms)),
            "max_ms_per_step": float(np.max(step_ms)),        # this rank's per-step wall clock (median of >= 5)
        }
        if variants:
            out["variants"] = variants
        # the legs below are auxiliary: a failure in one of them (a host without SciPy's LAPACK threads, a device too full for the C5
        # grid, ...) is reported in its place and must not take the measured headline line with it
        if world == 1 and not args.no_cpu_baseline:
            try:
                out["cpu_baseline"] = cpu_baseline(cfg, args.cpu_sample_m, args.cpu_reps)
                if out["cpu_baseline"].get("host_blas", {}).get("under_threaded"):
                    out["cpu_baseline"]["flag"] = ("baseline under-threaded: the host BLAS ran far below a threaded LAPACK's rate on "
                                                   "this box (host_blas) - no speed-up ratio is printed against it")
                else:
                    out["speedup_vs_cpu_port"] = out["cpu_baseline"]["value"] / ms_per_step
                    out["speedup_vs_cpu_blas3_floor"] = out["cpu_baseline"]["blas3_floor"]["value"] / ms_per_step
            except Exception as e:                      # noqa: BLE001 - reported, not swallowed
                out["cpu_baseline"] = {"error": f"{type(e).__name__}: {e}"}
                print(f"bench: cpu_baseline leg failed: {e!r}", file=sys.stderr)
        if world == 1 and args.config == "c3" and not args.no_secondary:
            # the other single-GPU configurations of BASELINE.json, timed by the same run: C2 (small N) and C5
            # (incremental update + greedy q-EI on a resident grid; its own roofline is the HBM-streaming down-date)
            del model, Zd
            abo._lib.lib().abo_pool_trim(local_rank)
            secondary = []

            def leg(name, f):
                try:
                    secondary.append(f())
                except Exception as e:                  # noqa: BLE001 - reported, not swallowed
                    secondary.append({"workload": name, "error": f"{type(e).__name__}: {e}"})
                    print(f"bench: secondary leg {name} failed: {e!r}", file=sys.stderr)

            def c5_leg():
                c5 = run_c5(args, CONFIGS["c5"], 1, 0, local_rank, dev, False, steps=48, warmup=2)
                keep = {k: c5[k] for k in ("value", "unit", "steps", "warmup", "roofline", "block_build_roofline", "secondary_roofline", "qei", "refresh_ms",
                                           "value_amortized", "refresh", "phases_ms", "refresh_phases_ms") if k in c5}
                keep["workload"] = c5["config"]["workload"]
                return keep

            leg("C2", lambda: quick_config(abo, synth, torch, dev, local_rank, "c2", K_TOP))
            leg("C5", c5_leg)
            leg("C1 shape", lambda: quick_c1_shape(abo, synth, torch, dev, local_rank, K_TOP))
            leg("hyper-parameter objective", lambda: quick_nlml_grad(abo, synth, torch, dev, local_rank))
            out["secondary"] = secondary
        print(json.dumps(out))
    if use_dist:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
