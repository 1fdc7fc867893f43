/*
 * abo_hip.h — C-ABI of libabo_hip.so, the MI355X (gfx950) GP-surrogate backend that drops in behind
 * AbstractBayesOpt.jl's AbstractSurrogate / AbstractAcquisition interface.
 *
 * The reference has no FFI of its own (it is pure Julia); each entry point below states the
 * reference function whose arithmetic it replaces (file:line relative to the reference repo) —
 * i.e. what a `HipStandardGP <: AbstractSurrogate` shim would `ccall` from that method
 * (binding shown in INTEGRATION.md).
 *
 * Conventions
 *  - every function returns an int32 status (ABO_OK == 0); on failure abo_last_error() holds text
 *  - points are POINT-MAJOR contiguous fp64: X[i*d + c]  (== a Julia d×N column-major Matrix)
 *  - `*_space` says where a caller buffer lives: ABO_HOST (pageable/pinned host memory) or
 *    ABO_DEVICE (memory of the handle's GPU, e.g. a torch tensor's data_ptr or a hipMalloc)
 *  - calls are synchronous: results are complete (host copies done, device buffers written and
 *    the internal stream idle) when the call returns
 *  - a handle is not re-entrant; different handles may be used from different threads
 *  - indices are 0-based (the Julia shim adds 1)
 */
#ifndef ABO_HIP_H
#define ABO_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define ABO_ABI_VERSION 7   /* 2: abo_mgpu_* (multi-device handles), k and d limits lifted
                               3: abo_set_contraction, abo_timings grew (contraction engine and its phases)
                               4: abo_refine, abo_optimize_acquisition, abo_mgpu_optimize_acquisition, abo_fit_acq, abo_mgpu_create_grad,
                                  abo_mgpu_append_grad; abo_timings grew (refinement stage)
                               5: abo_mgpu_cand_get; abo_acq_lhs (the grid stage on a single handle); weighted-sum objectives and
                                  gradient-enhanced handles in the refinement stage (abo_acq_term, abo_refine_terms,
                                  abo_optimize_acquisition_terms, abo_mgpu_optimize_acquisition_terms, ABO_ACQ_GRADNORM_UCB)
                               6: block form of greedy q-EI: abo_cand_qei (+ _begin / _top / _block / _pick / _end for hosts that
                                  shard the set themselves), abo_qei_stats, abo_set_qei_block, abo_mgpu_cand_qei_stats;
                                  abo_cand_downdate finds the down-date column of a pick appended for real in the batch's chain
                               7: abo_cand_qei_top takes the capacity of the caller's record buffer; abo_cand_qei_eligible (what ranks
                                  that shard a set themselves agree on before they take the block form); abo_cand_qei keeps its
                                  pick loop on the device (one launch per pick, one read-back per batch); abo_fill_distance;
                                  abo_timings grew (abo_nlml_grad phases, the bordered append's mat-vecs); the abo_test_* building
                                  blocks left the shipped library (test build only: ABO_TEST_HOOKS) */

/* status codes */
enum {
    ABO_OK = 0,
    ABO_ENOTPD = 1,   /* K + noise·I not positive definite → LinearAlgebra.PosDefException(info)
                         (caught at src/bayesian_opt.jl:126-141) */
    ABO_EDIM = 2,     /* dimension mismatch → DimensionMismatch (test/test_bayesian_opt.jl:788-817) */
    ABO_EINVAL = 3,   /* bad argument / not fitted */
    ABO_EHIP = 4,     /* HIP runtime error */
    ABO_ENOMEM = 5    /* device allocation failed */
};

/* kernel families: k(x,z) = sigma_f2 · kappa(||x−z|| / ell)  (StandardGP normal form,
 * src/surrogates/StandardGP.jl:41-64; kappas: [upstream KernelFunctions] SqExponentialKernel /
 * Matern52Kernel / Matern32Kernel and src/surrogates/GradientGP.jl:94-101, :320-327) */
enum { ABO_KERNEL_SE = 0, ABO_KERNEL_MATERN52 = 1, ABO_KERNEL_MATERN72 = 2, ABO_KERNEL_MATERN32 = 3 };

/* acquisition epilogues */
enum {
    ABO_ACQ_EI = 0,   /* src/acquisition_functions/ExpectedImprovement.jl:40-66   p0 = xi   */
    ABO_ACQ_UCB = 1,  /* src/acquisition_functions/UpperConfidenceBound.jl:38-45  p0 = beta */
    ABO_ACQ_PI = 2,   /* src/acquisition_functions/ProbabilityImprovement.jl:38-63 p0 = xi  */
    ABO_ACQ_MEAN = 3, /* score = −mu (exploitation only; no reference counterpart) */
    ABO_ACQ_GRADNORM_UCB = 4  /* src/acquisition_functions/gradNormUCB.jl:43-51  p0 = beta; gradient-enhanced handles, and only in
                                 the abo_*_terms entry points (abo_predict_grad_cov scores a batch with it directly) */
};

/* An objective of the acquisition stage: f(x) = Σ_t weight_t · acq_t(x) on ONE posterior evaluation — EnsembleAcquisition
 * (src/acquisition_functions/EnsembleAcq.jl:12-27, :53-55; the host normalises the weights as the reference's constructor does);
 * a plain acquisition function is one term of weight 1.  At most 8 terms. */
typedef struct abo_acq_term {
    int32_t kind;       /* ABO_ACQ_* */
    int32_t reserved;
    double p0;          /* xi (EI, PI) or beta (UCB, GradientNormUCB) */
    double best_y;      /* EI, PI */
    double weight;
} abo_acq_term;

enum { ABO_HOST = 0, ABO_DEVICE = 1 };

/* engine of the N²·M contraction V = L⁻¹K_XZ behind posterior_var (src/surrogates/StandardGP.jl:377-379):
 *   ABO_CONTRACT_FP64  v_mfma_f64_16x16x4_f64 kernels (78.6 TFLOP/s pipe)
 *   ABO_CONTRACT_INT8  exact products of fixed-point images of the two fp64 operands (K_XZ to 2^-52 of sigma_f2, each row of L⁻¹ to ≥ 50 bits
 *                      below its L1 norm) on v_mfma_i32_16x16x64_i8,
 *                      through residues modulo `nmod` coprime moduli ≤ 256 and a Chinese-remainder reconstruction in fp64
 *                      (14 moduli: errors of the size of the fp64 kernels' own rounding; each modulus less ≈ 14× more error,
 *                      7 % less time).  Serves the posteriors (abo_predict, abo_acq, resident grids; abo_predict_grad and
 *                      abo_predict_grad_cov of a gradient-enhanced handle) up to 65536 factor rows.
 *                      Its scratch is 2·nmod bytes per (candidate, factor row) of a chunk plus nmod·rows² bytes of planes; when the
 *                      device cannot give that (or more than ABO_OZ_SCRATCH_LIMIT_MB allows) the chunk is halved down to 4096
 *                      candidates, and below that the call runs on the fp64 kernels — never an error.
 *   ABO_CONTRACT_AUTO  INT8 from 1280 (padded) factor rows on, FP64 below (the measured crossover: profiles/r06_engine_crossover.txt) */
enum { ABO_CONTRACT_AUTO = 0, ABO_CONTRACT_FP64 = 1, ABO_CONTRACT_INT8 = 2 };

typedef struct abo_gp abo_gp;     /* opaque, reference-counted: one (immutable) conditioned model */
typedef struct abo_cand abo_cand; /* opaque: a candidate set resident in HBM together with its posterior */

typedef struct abo_params {
    int32_t family;     /* ABO_KERNEL_* */
    int32_t device;     /* HIP device ordinal this handle lives on */
    double ell;         /* lengthscale  (get_lengthscale, src/surrogates/StandardGP.jl:261) */
    double sigma_f2;    /* kernel scale (get_scale, :274) */
    double noise_var;   /* observation noise variance (StandardGP.noise_var, :13) */
    double mean_c;      /* prior mean: 0 = ZeroMean, c = ConstMean(c) (:42-44, :223-229) */
    double jitter;      /* 0 = reference behaviour (never add jitter); >0 = opt-in retry with
                           noise_var + jitter·10^r, r = 0..3, when the factorisation fails */
    int64_t n_max;      /* capacity hint for appends (0 = size to the fit) */
    int64_t chunk;      /* candidate chunk size for the posterior (0 = auto) */
} abo_params;

/* phase timings of the last fit / acq call, milliseconds, measured with HIP events on the
 * handle's own stream (see abo_get_timings).  The *_total_ms fields are always measured; the phase fields inside a call
 * (kernel matrix / Cholesky / ..., kxz / var_gemm / finalize / topk, the oz_* fields) are instrumentation: recorded for every
 * model of more than one 128-row block, left out (they read 0) for a one-block model — N <= 128, the reference's own loops,
 * where ~40 event records are a fifth of a 0.2 ms step.  ABO_PHASE_EVENTS=1 records them always, =0 never. */
typedef struct abo_timings {
    double fit_kernel_matrix_ms, fit_cholesky_ms, fit_inverse_ms, fit_alpha_ms, fit_total_ms;
    double acq_kxz_ms, acq_var_gemm_ms, acq_finalize_ms, acq_topk_ms, acq_total_ms;
    int64_t var_gemm_launches;   /* number of launches of the dominant kernel in the last acq */
    double var_gemm_flop;        /* algorithmic flop (N²·M, triangular) those launches performed */
    double downdate_ms;          /* last abo_cand_downdate on this handle: its O(N·M) pass (K_ZX mat-vec or kernel sweep) */
    double downdate_bytes;       /* bytes of resident K_ZX that pass streamed (8·N·M); 0 when it re-evaluated the kernel */
    /* ABI 3: which engine ran the contraction of the last posterior call (ABO_CONTRACT_FP64 / _INT8; 0 = none), and for the
     * int8 engine its moduli count, its phases (acq_var_gemm_ms is their sum: quantisation of K_XZ, residue GEMMs,
     * reconstruction) and the ALGORITHMIC int8 operations of the GEMM launches, moduli × N²·M (one launch per chunk covers all
     * moduli; what the kernel issues beyond the triangular product is not credited).  oz_prepare_ms: residue planes of L⁻¹, spent in
     * the last posterior call (0 when that call reused cached planes) */
    int64_t contraction_engine, oz_nmod;
    double oz_prepare_ms, oz_quant_ms, oz_gemm_ms, oz_crt_ms, oz_gemm_ops;
    /* ABI 4: the last abo_refine / abo_optimize_acquisition on this handle: duration of its ONE refinement launch, the starts it
     * refined and the acquisition evaluations (value + analytic gradient each) all starts took together */
    double refine_ms;
    int64_t refine_starts, refine_evals;
    /* ABI 6: 1 when the last abo_cand_downdate on this handle took its column from the chain of the set's last block-form q-EI
     * batch (no pass over K_ZX: downdate_bytes = 0, downdate_ms = the new K_ZX column only) */
    int64_t downdate_from_chain;
    /* ABI 7: the last abo_nlml_grad on this handle: K⁻¹ = L⁻ᵀL⁻¹ on the fp64 MFMA GEMM (N³/3 flop, lower tiles), and the
     * sweep that generates ∂K/∂log ℓ tile by tile and reduces tr((K⁻¹ − ααᵀ)∂K/∂θ) */
    double nlml_kinv_ms, nlml_trace_ms;
    /* ABI 7: the bordered append that made this handle (abo_append): its two triangular mat-vecs l = L⁻¹k, v = L⁻ᵀl together (HIP
     * events; 0 without phase events) and the bytes they stream, 8·N² (one triangle of L⁻¹ and of L⁻ᵀ, once each) */
    double append_trmv_ms, append_trmv_bytes;
} abo_timings;

/* --- lifetime -------------------------------------------------------------------------------
 * StandardGP(kernel, noise_var; mean) (src/surrogates/StandardGP.jl:41-64).  The handle starts
 * un-conditioned (gpx === nothing). */
int32_t abo_create(const abo_params* params, abo_gp** out);
/* GradientGP(kernel, p, noise_var; mean=gradConstMean(c)) (src/surrogates/GradientGP.jl:617-639): gradient-
 * enhanced GP with p = d+1 outputs per point (f and ∂f/∂x_c), multi-output kernel gradKernel (:573-606) with
 * analytic derivatives, rows ordered by outputs (MOInputIsotopicByOutputs).  mean_c: p prior means (NULL = 0).
 * 2 ≤ p ≤ 129 (d ≤ 128 inputs; d ≤ 32: the derivative blocks are generated from register-resident coordinates, beyond that from
 * slabs of 32 coordinates, fp64 output only — ABI 7; rounds 2 - 5 stopped at d = 32).
 * abo_fit then takes y of length p·N ordered by outputs (prep_output, :893-895); abo_predict / abo_acq address
 * the function output.  Inside the library the (d+1)N-row system is kept POINT-MAJOR (row i·p + q: all outputs of point i
 * adjacent), so that a new observation appends p rows at the end of the factor (abo_append_grad); every vector that crosses
 * the ABI (y, alpha) is by outputs, abo_get_factor's L / Linv are in the library's row order. */
int32_t abo_create_grad(const abo_params* params, int32_t p, const double* mean_c, abo_gp** out);
/* Engine of the variance contraction for this handle and the models appended from it (gp == NULL: the process default, which
 * new handles start from; the environment variable ABO_CONTRACTION = auto | fp64 | int8 | int8:<nmod> seeds it).  nmod: 8 … 16
 * moduli, 0 = 14.  No reference counterpart: the reference computes this product in fp64 BLAS ([upstream AbstractGPs]
 * diag_Xt_invA_X); both engines meet the same parity bounds (tests/test_gpu_ozaki.py). */
int32_t abo_set_contraction(abo_gp* gp, int32_t engine, int32_t nmod);
/* Base.copy(::StandardGP) (src/surrogates/StandardGP.jl:26, surrogates_utils.jl:12-14): device
 * state is immutable after fit, so a copy is a shared reference. */
int32_t abo_retain(abo_gp* gp);
/* finaliser; frees device state when the last reference goes */
int32_t abo_destroy(abo_gp* gp);

/* --- update ---------------------------------------------------------------------------------
 * update(model::StandardGP, xs, ys) (src/surrogates/StandardGP.jl:79-83): full refit,
 * K = k(X,X) + noise·I, L = chol(K), delta = y − mean_c, alpha = K⁻¹ delta, W = L⁻¹.
 * *info = 0 on success, k>0 (1-based LAPACK potrf convention) with status ABO_ENOTPD when the
 * leading minor of order k is not positive definite; the handle then stays un-conditioned. */
int32_t abo_fit(abo_gp* gp, const double* X, int64_t N, int32_t d, const double* y, int32_t space,
                int64_t* info);

/* update(model, xs, ys) + scores = acqf(model, grid) + sortperm(scores; rev=true)[1:k] — abo_fit followed by abo_acq — in ONE call
 * with ONE host synchronisation: the acquisition's launches are queued directly behind the fit's, the LAPACK-style `info` and the
 * fit's scalars are read once, at the end (after a failed factorisation the acquisition's launches run on leftovers and their
 * results are discarded: status ABO_ENOTPD, *info as abo_fit, outputs untouched in meaning).  What a BO loop at the reference's own
 * sizes (5 … 100 points, 10 000 grid points, acq_utils.jl:37) spends most of its step on is the host round trip between the two
 * calls.  The handle ends up exactly as after abo_fit.  best_y is the caller's (EI / PI take min(ys), ExpectedImprovement.jl:81-83).
 * With an opt-in jitter (abo_params.jitter > 0) or a gradient-enhanced handle the two calls run one after the other. */
int32_t abo_fit_acq(abo_gp* gp, const double* X, int64_t N, int32_t d, const double* y, int32_t space, int64_t* info, const double* Z,
                    int64_t M, int32_t z_space, int32_t kind, double p0, double best_y, int64_t idx_base, double* scores, int32_t k,
                    double* top_val, int64_t* top_idx, int32_t out_space);

/* --- incremental update (BASELINE config 5; no counterpart in the reference, which always refits —
 * nearest code: update(), src/surrogates/StandardGP.jl:79-83) -------------------------------------
 * Bordered ("rank-1 append") Cholesky update: returns in *out a NEW model conditioned on the N+1
 * points that shares the factor storage of `gp`; `gp` itself stays valid and unchanged (rows ≤ N of
 * the factor are never touched), which is what the driver's rollback needs (src/bayesian_opt.jl:116-141).
 *   k = k(X,x), l = L⁻¹k, l_nn² = k(x,x) + noise − ‖l‖² (ABO_ENOTPD, *info = N+1 if ≤ 0),
 *   L' = [[L,0],[lᵀ,l_nn]], α' = [α − βv ; β] with v = K⁻¹k, β = (y − m − kᵀα)/l_nn².
 * O(N²) flops, two passes over L⁻¹.  Falls back to a full refit (with doubled capacity) when the
 * storage is full (abo_params.n_max) or while a LARGER model on the same storage is still alive
 * (destroying the fantasy models of a q-EI exploration makes their parent appendable in place again).
 * x: d host doubles. */
int32_t abo_append(abo_gp* gp, const double* x, int32_t d, double y, int64_t* info, abo_gp** out);
/* the same for a gradient-enhanced model (abo_create_grad): y holds the p observed values {f(x), ∂f/∂x_1 … ∂f/∂x_d}; the
 * p rows are appended one after the other at the end of the point-major factor (each a bordered update whose kernel row
 * comes from the analytic derivative blocks of gradKernel, src/surrogates/GradientGP.jl:573-606), O(p·N²p²) instead of the
 * O(N³p³) refit the reference does per step (update(::GradientGP), :659-668).  params.n_max counts POINTS.  On failure
 * *info is the order of the failing leading minor in the library's row order (N·p + q + 1).  abo_cand_* work on such a
 * handle too (function-value grid; a down-date after abo_append_grad is p rank-1 down-dates). */
int32_t abo_append_grad(abo_gp* gp, const double* x, int32_t d, const double* y, int64_t* info, abo_gp** out);

/* --- posterior ------------------------------------------------------------------------------
 * posterior_mean / posterior_var (src/surrogates/StandardGP.jl:361-363, :377-379), fused as in
 * unstandardized_mean_and_var's mean_and_var (:395-404):
 *   mu = mean_c + K_ZX·alpha,   var = sigma_f2 − colsum((L⁻¹K_XZ)²) + 1e-18  (latent variance).
 * mu / var may each be NULL. */
int32_t abo_predict(abo_gp* gp, const double* Z, int64_t M, int32_t d, int32_t z_space, double* mu,
                    double* var, int32_t out_space);

/* posterior_grad_mean / posterior_grad_var (src/surrogates/GradientGP.jl:936-956): all p outputs of M points,
 * ordered by outputs (p·M values each; mu / var may be NULL). */
int32_t abo_predict_grad(abo_gp* gp, const double* Z, int64_t M, int32_t d, int32_t z_space, double* mu,
                         double* var, int32_t out_space);
/* posterior_grad_cov(model, [x]) per point (GradientGP.jl:966-971): mu [M][p], cov [M][p][p] (point-major) and,
 * if score != NULL, the GradientNormUCB value −(mᵀm + trΣ) + β·sqrt(max(4mᵀΣm + 2‖Σ‖_F², 1e-12)) on the gradient
 * block (src/acquisition_functions/gradNormUCB.jl:43-51).  Any output may be NULL. */
int32_t abo_predict_grad_cov(abo_gp* gp, const double* Z, int64_t M, int32_t d, int32_t z_space, double beta,
                             double* mu, double* cov, double* score, int32_t out_space);

/* --- acquisition over a candidate batch -------------------------------------------------------
 * scores = acqf(surrogate, grid_points); sortperm(scores; rev=true)[1:k]
 * (src/acquisition_functions/acq_utils.jl:50-52) with the EI/UCB/PI epilogue fused behind the
 * posterior.  scores (length M) is optional; top_val/top_idx (length k) are optional when k == 0.
 * Ordering is Julia's stable reverse sort: descending score, ties → lowest index, NaN first.
 * top_idx are global indices idx_base + j (idx_base = this rank's shard offset).  k is free (n_local is a plain Int in
 * the reference, acq_utils.jl:33-38): k ≤ 1024 is one selection round, larger k takes ⌈k/1024⌉ rounds; when M < k the
 * tail is filled with (NaN, −1).  scores/top_* live in out_space. */
int32_t abo_acq(abo_gp* gp, const double* Z, int64_t M, int32_t d, int32_t z_space, int32_t kind,
                double p0, double best_y, int64_t idx_base, double* scores, int32_t k,
                double* top_val, int64_t* top_idx, int32_t out_space);

/* abo_acq for a weighted-sum objective (EnsembleAcq.jl:53-55): the posterior is evaluated once, every member's epilogue runs on
 * it; with a GRADNORM_UCB term (gradient-enhanced handles) the per-point mean and covariance block of all outputs are evaluated
 * instead.  One term of weight 1 is abo_acq, bit for bit. */
int32_t abo_acq_terms(abo_gp* gp, const double* Z, int64_t M, int32_t d, int32_t z_space, const abo_acq_term* terms,
                      int32_t nterms, int64_t idx_base, double* scores, int32_t k, double* top_val, int64_t* top_idx,
                      int32_t out_space);
/* the grid stage of optimize_acquisition (acq_utils.jl:44-52) on ONE handle, the grid never crossing PCIe: an n-point Latin
 * hypercube generated on the device (abo_lhs), scored, the k best returned — top_val / top_idx (k), top_x (k × d, optional: their
 * coordinates).  What abo_mgpu_acq_lhs is for a group.  Host outputs. */
int32_t abo_acq_lhs(abo_gp* gp, int64_t n, int32_t d, const double* lower, const double* upper, uint64_t seed, int32_t kind,
                    double p0, double best_y, int32_t k, double* top_val, int64_t* top_idx, double* top_x);

/* --- optimize_acquisition (src/acquisition_functions/acq_utils.jl:33-73) on the device -------------------------------
 * The reference: Latin-hypercube grid of n_grid points → scores → the n_local best as starts (:44-52) → for EVERY start one
 * box-constrained L-BFGS run, Fminbox(LBFGS(HagerZhang(linesearchmax = 20))) with Optim.Options(g_tol = 1e-5, f_abstol = 2.2e-9,
 * x_abstol = 1e-4), each objective value an M = 1 posterior call and each gradient a finite-difference stencil of them
 * (:55-71) → the best refined point (:66-72).
 * abo_refine is the second stage for S given starts: ONE launch, one workgroup per start running that start's whole projected
 * L-BFGS on the device — analytic ∇μ, ∇σ² from the kernel's derivative (∂k/∂x, L⁻¹, L⁻ᵀ), closed-form ∂EI/∂(μ,σ²) (UCB, PI
 * likewise), Armijo backtracking with at most linesearch_max trials, the reference's three stopping rules.  Maximises the
 * acquisition function inside the box [lower, upper].  x_out S × d, f_out S (the acquisition value at x_out: what abo_acq returns
 * for that point up to the rounding of a differently ordered sum), iters_out (optional) S × 2 = {iterations, evaluations}.  A start whose value is not finite
 * is returned unchanged.  All buffers HOST memory.
 * PARITY UNPINNED: the result is a local maximiser of the acquisition function in the box at the reference's tolerances — NOT Optim's
 * iterate.  The reference runs Fminbox (a log-barrier outer loop that keeps iterates strictly interior) around L-BFGS with HagerZhang
 * line searches and Optim's convergence bookkeeping; this is a projected L-BFGS with Armijo backtracking that may return a maximiser
 * ON the bound, and the reference holds no fixture of optimize_acquisition outputs.  What is tested: never below the start, inside
 * the box, at least SciPy L-BFGS-B's optimum on the CPU oracle's acquisition from the same start (tests/test_gpu_refine.py).
 * The option fields are clamped (max_iter ≤ 10000, linesearch_max ≤ 64, history ≤ 64).
 * Gradient-enhanced handles (ABI 5; the reference's tutorials drive optimize_acquisition with GradientGP, gradNormUCB.jl:39-51):
 * the starts advance in lockstep rounds; a round evaluates the all-output posterior of every pending point (the kernels of
 * abo_predict_grad_cov) and takes ∇μ = E[∇f(x)] − m_∇ and ∇σ² = 2·Cov(f(x), ∇f(x)) from it — the analytic gradient of EI / UCB /
 * PI at no extra cost; a GRADNORM_UCB term is differentiated by central differences over a 2d-point stencil evaluated in the same
 * batch (the reference differentiates every objective that way).
 * The *_terms forms take a weighted-sum objective (abo_acq_term: EnsembleAcquisition; ∇ = Σ w_t ∇acq_t on one posterior).
 * abo_optimize_acquisition is the whole function in ONE call: grid generated on the device (abo_lhs, `seed`), scored and
 * reduced to the min(n_local, n_grid) best (abo_acq), those refined, the best point returned: best_x (d), best_val; optional
 * starts_x (k × d) / starts_val (k): the selected grid points and their scores in selection order; refined_x / refined_val: what
 * each became.  The result is the refined point with the largest value (first one on ties, the reference keeps the first under
 * its strict `>`), or the best grid point if no refined value reaches its score. */
typedef struct abo_refine_opts {   /* NULL or zero fields = the reference's settings */
    int32_t max_iter;        /* L-BFGS iterations per start, 0 = 100 */
    int32_t linesearch_max;  /* trials per line search, 0 = 20 (acq_utils.jl:10) */
    int32_t history;         /* L-BFGS pairs kept, 0 = 10 (Optim's LBFGS default) */
    int32_t reserved;
    double g_tol, f_abstol, x_abstol;   /* 0 = 1e-5, 2.2e-9, 1e-4 (acq_utils.jl:62) */
} abo_refine_opts;
int32_t abo_refine(abo_gp* gp, int32_t kind, double p0, double best_y, const double* lower, const double* upper, int32_t d,
                   const double* starts, int32_t S, const abo_refine_opts* opts, double* x_out, double* f_out, int32_t* iters_out);
int32_t abo_optimize_acquisition(abo_gp* gp, int32_t kind, double p0, double best_y, const double* lower, const double* upper,
                                 int32_t d, int64_t n_grid, int32_t n_local, uint64_t seed, const abo_refine_opts* opts,
                                 double* best_x, double* best_val, double* starts_x, double* starts_val, double* refined_x,
                                 double* refined_val);
int32_t abo_refine_terms(abo_gp* gp, const abo_acq_term* terms, int32_t nterms, const double* lower, const double* upper, int32_t d,
                         const double* starts, int32_t S, const abo_refine_opts* opts, double* x_out, double* f_out,
                         int32_t* iters_out);
int32_t abo_optimize_acquisition_terms(abo_gp* gp, const abo_acq_term* terms, int32_t nterms, const double* lower,
                                       const double* upper, int32_t d, int64_t n_grid, int32_t n_local, uint64_t seed,
                                       const abo_refine_opts* opts, double* best_x, double* best_val, double* starts_x,
                                       double* starts_val, double* refined_x, double* refined_val);

/* --- resident candidate sets (BASELINE config 5: greedy q-EI over a fixed grid) --------------------
 * abo_cand_create copies M candidates to the device and evaluates their posterior with `gp`
 * (same arithmetic as abo_predict).  After `gp2 = abo_append(gp, x*, y*)`, abo_cand_downdate(gp2, c)
 * updates the stored posterior in O(N·M) instead of O(N²·M) — as one streaming pass over the K_ZX the set keeps
 * resident in HBM when M·N_max·8 bytes fit the budget (environment ABO_CAND_KZX_GIB, default 64; 0 = never), else
 * by re-evaluating the kernel:
 *   c(z) = k(z,x*) − k_zᵀK⁻¹k_*,   σ²(z) −= c(z)²/l_nn²,   μ(z) += c(z)·(y* − μ(x*))/l_nn²
 * (ABO_EINVAL if gp2 is not the one-point append of the model the set was last synced with).
 * abo_cand_acq runs the EI/UCB/PI epilogue + top-k of abo_acq on the stored posterior;
 * abo_cand_point returns one candidate's coordinates and posterior (host outputs, any may be NULL);
 * abo_cand_refresh re-evaluates from scratch (after a refit or a hyper-parameter change);
 * abo_cand_save / abo_cand_restore snapshot and roll back the stored posterior (greedy q-EI explores
 * fantasy appends, then the real observation is appended to the un-fantasised model). */
int32_t abo_cand_create(abo_gp* gp, const double* Z, int64_t M, int32_t d, int32_t z_space, abo_cand** out);
int32_t abo_cand_destroy(abo_cand* c);
int32_t abo_cand_refresh(abo_gp* gp, abo_cand* c);
int32_t abo_cand_downdate(abo_gp* gp, abo_cand* c);
int32_t abo_cand_save(abo_gp* gp, abo_cand* c);
int32_t abo_cand_restore(abo_gp* gp, abo_cand* c);
int32_t abo_cand_acq(abo_gp* gp, abo_cand* c, int32_t kind, double p0, double best_y, int64_t idx_base,
                     double* scores, int32_t k, double* top_val, int64_t* top_idx, int32_t out_space);
int32_t abo_cand_get(abo_gp* gp, abo_cand* c, double* mu, double* var, int32_t out_space);
int32_t abo_cand_point(abo_gp* gp, abo_cand* c, int64_t idx, double* x, double* mu, double* var);
/* takes candidate idx out of the running until the next refresh / restore: its stored posterior becomes (μ = +Inf,
 * σ² = 0), so EI = PI = 0 and UCB = −Inf there.  Greedy q-EI uses it (opt-in) to return q distinct points: with
 * observation noise the Kriging-believer fantasy does not collapse the variance at a picked point. */
int32_t abo_cand_exclude(abo_gp* gp, abo_cand* c, int64_t idx);

/* --- greedy (Kriging-believer) q-EI on a resident set, block form (ABI 6) -----------------------------------------
 * No reference counterpart (the reference's EI is single-point, src/acquisition_functions/ExpectedImprovement.jl:40-66, and its
 * update always refits, src/surrogates/StandardGP.jl:79-83): q × [EI over the grid → arg-max → condition the grid's posterior on
 * the fantasy observation (x_j, μ(x_j))] — each sub-step exactly `update` + EI semantics.  What a pick needs of
 * the resident K_ZX is one column of posterior covariances c_j(z) = Cov_{j−1}(z, x_j):
 *     Cov₀(z, x) = k(z, x) − k_zᵀK⁻¹k_x                                     under the model the set is synced with (the base),
 *     c_j(z) = Cov₀(z, x_j) − Σ_{i<j} c_i(z)·c_i(x_j)/s_i,   s_i = σ²_{i−1}(x_i) + σ²_n,   σ²_j(z) = σ²_{j−1}(z) − c_j(z)²/s_j
 * (μ is not touched: the fantasy value is the mean).  Cov₀ for a BLOCK of T points is ONE pass over K_ZX (a T-column product
 * on the fp64 matrix pipe under the 8·N·M-byte stream) where the plain loop (abo_append + abo_cand_downdate per pick) streams
 * K_ZX once per pick.  The block is the T best candidates of the current scores; a pick outside every block builds a new block
 * from the scores of that moment.  Nothing is appended to the model: on return model and set are as before (the batch is
 * rolled back), and the chain c_1 … c_{q−1} stays with the set — appending the picks for real, in order
 * (abo_append(gp, x_1, y_1) → abo_cand_downdate, …), finds each down-date column there instead of streaming K_ZX again
 * (c_j does not depend on the observed value).  Blocks and chain FOLLOW the model through real appends (any abo_cand_downdate puts
 * its column into the chain): the next batch, on the appended model, still finds the columns of the earlier blocks — a BO step's
 * picks are mostly the previous step's runners-up — corrected by the chain entries made since each block was built; a refresh,
 * another lineage of the model or 64 chain entries start the state afresh.
 *   abo_cand_qei: the whole batch on one handle.  x_out q × d, idx_out / ei_out q (idx = idx_base + local index);
 *     distinct != 0 excludes every picked candidate for the rest of the call; block = T (16 … 64, rounded up to a multiple of
 *     16; 0 = the process default, abo_set_qei_block / ABO_QEI_BLOCK, initially 32; < 0 = the plain loop).  The plain loop
 *     is also what runs for a gradient-enhanced model, for a set whose K_ZX is not resident, and for q > 64.  stats may be NULL.
 *   For a host that shards the set itself (one process per GPU: abstractbayesopt.jl_amd/incremental.py over torch.distributed)
 *   the same batch in steps; every shard makes the same calls with the same exchanged numbers, so a sharded set repeats the
 *   single set's arithmetic bit for bit:
 *     _begin(gp, c, q, block)                         snapshot σ², μ; empty chain and blocks
 *     _top(gp, c, xi, best_y, idx_base, k, rec)       EI over the shard and its k best as records of 4 + d + n doubles
 *                                                     {EI, global index (−1: none), μ, σ², x[0..d), c_1(x) … c_n(x)}, n = entries of the
 *                                                     chain (abo_cand_qei_has: real appends carried over + the batch's picks so far)
 *     _block(gp, c, pts, gidx, T)                     Cov₀ columns of T points (the merged best T of all shards) in one pass
 *     _pick(gp, c, gidx, var_x, cx, n, excl, info)    condition on the pick (its point must be in a block): var_x = σ²(x) and
 *                                                     cx[i] = c_{i+1}(x), i < n, both from the winner's record; excl ≥ 0: local index
 *                                                     to exclude; σ²(x) + noise ≤ 0 → ABO_ENOTPD, *info = N + n + 1 (the failed pivot
 *                                                     of the plain loop's bordered append)
 *     _end(gp, c)                                     roll σ², μ back; the chain stays for abo_cand_downdate */
typedef struct abo_qei_stats {
    int32_t picks, block, block_builds, block_hits;   /* block = T (0: the plain loop ran); hits: picks found in an existing block */
    double total_ms;                /* host wall clock of the call */
    double block_ms;                /* HIP events: all block builds of the batch (K⁻¹K_XT + the pass over K_ZX + kernel values) */
    double pass_ms;                 /* HIP events: the product over the resident K_ZX of the LAST block build */
    double pass_bytes, pass_flop;   /* algorithmic: 8·N·M bytes of K_ZX streamed once, 2·N·M·T flop */
} abo_qei_stats;
int32_t abo_set_qei_block(int32_t block);
int32_t abo_cand_qei(abo_gp* gp, abo_cand* c, int32_t q, double xi, double best_y, int32_t distinct, int64_t idx_base, int32_t block,
                     double* x_out, int64_t* idx_out, double* ei_out, abo_qei_stats* stats);
int32_t abo_cand_qei_begin(abo_gp* gp, abo_cand* c, int32_t q, int32_t block);
/* *ok = 1 when _begin(gp, c, q, block) would open a block-form batch on this shard (model fitted and in sync with the set, K_ZX
 * resident, q ≤ 64, block size > 0, …), 0 otherwise — the reason is then in abo_last_error.  Changes nothing.  Ranks that shard a set
 * themselves exchange this flag and take the block form only if EVERY rank can (a rank that cannot would leave the others'
 * all-gathers without a partner). */
int32_t abo_cand_qei_eligible(abo_gp* gp, abo_cand* c, int32_t q, int32_t block, int32_t* ok);
/* rec: k records of 4 + d + n doubles each, n = the chain's entries at the moment of the call (abo_cand_qei_has; it grows with
 * every pick and with the real appends carried over from earlier batches); cap_words = doubles the caller's buffer holds — ABO_EINVAL
 * (nothing written) when k·(4 + d + n) exceeds it.  A set's block slots are keyed by global index = idx_base + local index: every call
 * of one set passes the same idx_base (another value drops the blocks; the next pick builds a new one). */
int32_t abo_cand_qei_top(abo_gp* gp, abo_cand* c, double xi, double best_y, int64_t idx_base, int32_t k, double* rec, int64_t cap_words);
int32_t abo_cand_qei_block(abo_gp* gp, abo_cand* c, const double* pts, const int64_t* gidx, int32_t T);
int32_t abo_cand_qei_pick(abo_gp* gp, abo_cand* c, int64_t gidx, double var_x, const double* cx, int32_t n, int64_t excl,
                          int64_t* info);
int32_t abo_cand_qei_end(abo_gp* gp, abo_cand* c);
/* *has = 1 when candidate gidx is a point of one of the set's blocks; *nchain = entries of the set's chain (a record of _top carries
 * that many values behind x).  Either output may be NULL. */
int32_t abo_cand_qei_has(abo_gp* gp, abo_cand* c, int64_t gidx, int32_t* has, int32_t* nchain);
/* statistics of the set's current / last block-form batch (abo_cand_qei fills its own `stats` from the same numbers) */
int32_t abo_cand_qei_stats(abo_gp* gp, abo_cand* c, abo_qei_stats* out);

/* --- grid generation and stand-alone epilogue (DEVICE buffers) --------------------------------------
 * abo_lhs: points j0 .. j0+count−1 of an n-point Latin-hypercube design in the box [lower, upper]
 * (QuasiMonteCarlo.sample(n, lower, upper, LatinHypercubeSample()), src/acquisition_functions/acq_utils.jl:44-47)
 * written point-major to Z_dev; counter-based (keyed Feistel permutation per coordinate), so each rank
 * generates its own shard and the candidate grid never crosses PCIe.  lower/upper: d host doubles.
 * abo_score: scores[j] = acq(mu[j], var[j]) — the EI / UCB / PI epilogue on an existing posterior
 * (several acquisition functions on one posterior pass: EnsembleAcquisition, EnsembleAcq.jl:53-55). */
int32_t abo_lhs(int32_t device, int64_t n, int32_t d, const double* lower, const double* upper, uint64_t seed,
                int64_t j0, int64_t count, double* Z_dev);
int32_t abo_score(int32_t device, const double* mu, const double* var, int64_t M, int32_t kind, double p0,
                  double best_y, double* scores);
/* monte_carlo_fill_distance (src/BO_utils.jl:140-159; the lower length-scale bound of optimize_hyperparameters,
 * src/BO_utils.jl:87-125): *out = max over the n_samples points S of the distance to the nearest of the N training points X
 * (both point-major, d coordinates; HOST or DEVICE memory each; out: one host double).  The caller draws the sample points — the
 * reference draws them with its own RNG in the box — so the result is the reference's for the same samples (ABI 7). */
int32_t abo_fill_distance(int32_t device, const double* X, int64_t N, int32_t d, int32_t x_space, const double* S, int64_t n_samples,
                          int32_t s_space, double* out);

/* --- scalars ----------------------------------------------------------------------------------
 * nlml (src/surrogates/StandardGP.jl:99-114) of the fitted state:
 * ½(N log 2π + logdet(K+noise I) + deltaᵀ alpha). */
int32_t abo_nlml(abo_gp* gp, double* out);

/* NLML and its analytic gradient with respect to (log ell, log sigma_f2) — what Optim's
 * `autodiff=:forward` computes with ForwardDiff duals in optimize_hyperparameters
 * (src/bayesian_opt.jl:253-285), which cannot cross a C-ABI:
 *   dNLML/dθ = ½ tr((K⁻¹ − ααᵀ) ∂K/∂θ),  K⁻¹ = L⁻ᵀL⁻¹ formed on the MFMA GEMM, ∂K/∂log ell generated on
 *   the fly (for a gradient-enhanced handle: generated as a matrix from the analytic derivative blocks of
 *   gradKernel, src/surrogates/GradientGP.jl:573-606, whose nlml is :684-698).  Needs a freshly fitted handle (not an
 *   appended view).  Any output may be NULL. */
int32_t abo_nlml_grad(abo_gp* gp, double* nlml, double* d_log_ell, double* d_log_sigma_f2);

/* --- introspection (tests) ----------------------------------------------------------------------
 * L (N×N row-major, strictly-upper part zero), alpha (N), Linv = L⁻¹ (N×N row-major); any may be
 * NULL.  Host buffers. */
int32_t abo_get_factor(abo_gp* gp, double* L, double* alpha, double* Linv);
int32_t abo_get_n(abo_gp* gp, int64_t* N, int32_t* d);
/* the training data the model is conditioned on, back to host buffers: X (points × d, point-major) and y (factor rows:
 * one per point, or p per point ordered by outputs for a gradient-enhanced model).  The reference keeps no
 * serialisation code; its BOStruct is rebuilt from (xs, ys, hyper-parameters) (src/bayesian_opt.jl:81,:163-165) —
 * this is the device-side half of that: checkpoint = hyper-parameters + these arrays, resume = abo_fit. */
int32_t abo_get_data(abo_gp* gp, double* X, double* y);
int32_t abo_get_timings(abo_gp* gp, abo_timings* out);

/* --- multi-device handles (BASELINE configs 4 and 5: candidates sharded over the GPUs of one node) -----------------
 * One host process drives all devices: the Julia host has no process launcher, so the sharding of
 * `scores = acqf(surrogate, grid)` + `sortperm(scores; rev=true)[1:n_local]` (src/acquisition_functions/acq_utils.jl:50-52)
 * happens inside the library.  An abo_mgpu owns one abo_gp per listed device (a device may be listed more than once: each
 * entry is one shard).  Every device fits the same model redundantly (deterministic kernels → bit-identical factors; the
 * refit is ≈1 % of a config-3 step), candidates are cut into contiguous shards (sizes differ by at most one, earlier
 * shards take the larger ones), one host thread per shard drives its device, and the only exchange is the selection: each
 * device's k × (score, global index) go through ONE ncclAllGather (RCCL over xGMI; RCCL has no MAXLOC) and are merged in
 * the reference's order — descending score, ties → lowest index, NaN first — so the result equals the single-device
 * abo_acq on the whole batch bit for bit.  RCCL is loaded at run time (librccl.so.1, whichever copy the process already
 * has); when it is missing, fails to initialise, or the list names one device twice (RCCL refuses that), the pairs are
 * copied to the host per device instead — same merge, same result (abo_mgpu_info tells which).  Environment
 * ABO_MGPU_EXCHANGE = host | rccl overrides the choice (rccl on a one-entry list exercises the RCCL calls at world size 1).
 * All buffers of these entry points are HOST memory.  A handle is not re-entrant. */
typedef struct abo_mgpu abo_mgpu;     /* opaque: one model replicated on ndev devices */
typedef struct abo_mcand abo_mcand;   /* opaque: a candidate set sharded over those devices, resident with its posterior */
enum { ABO_XCHG_HOST = 0, ABO_XCHG_RCCL = 1 };

/* HipStandardGP(kernel, noise_var; mean, devices = [...]) — params->device is ignored, dev[0..ndev) are the HIP ordinals,
 * 1 ≤ ndev ≤ 16 */
int32_t abo_mgpu_create(const abo_params* params, int32_t ndev, const int32_t* dev, abo_mgpu** out);
/* the same for the gradient-enhanced model (abo_create_grad on every device: GradientGP(kernel, p, noise_var; mean), GradientGP.jl:
 * 617-639).  abo_mgpu_fit then takes y of length p·N ordered by outputs, abo_mgpu_predict / _acq / _acq_lhs / _cand_* address the
 * function output, abo_mgpu_append_grad appends one observation {f, ∂f/∂x_1 …} on every device (abo_append_grad), and
 * abo_mgpu_cand_qei conditions each pick on the posterior mean of all p outputs at the picked point (the Kriging-believer fantasy of a
 * model that observes gradients). */
int32_t abo_mgpu_create_grad(const abo_params* params, int32_t p, const double* mean_c, int32_t ndev, const int32_t* dev, abo_mgpu** out);
/* Base.copy (StandardGP.jl:26): a new group sharing every per-device state (abo_retain) */
int32_t abo_mgpu_clone(abo_mgpu* mg, abo_mgpu** out);
int32_t abo_mgpu_destroy(abo_mgpu* mg);
/* ndev, the device list (16 entries, may be NULL) and the exchange transport in use (ABO_XCHG_*) */
int32_t abo_mgpu_info(abo_mgpu* mg, int32_t* ndev, int32_t* dev, int32_t* exchange);
/* the per-device handle of shard i (borrowed: valid until the next abo_mgpu_fit / abo_mgpu_append / abo_mgpu_destroy):
 * abo_get_factor, abo_get_timings, abo_nlml, abo_nlml_grad … apply to it */
int32_t abo_mgpu_get(abo_mgpu* mg, int32_t i, abo_gp** out);
/* update(model, xs, ys): the same full refit on every device, concurrently (abo_fit) */
int32_t abo_mgpu_fit(abo_mgpu* mg, const double* X, int64_t N, int32_t d, const double* y, int64_t* info);
/* posterior_mean / posterior_var over M candidates, shard i computed on device i (abo_predict) */
int32_t abo_mgpu_predict(abo_mgpu* mg, const double* Z, int64_t M, int32_t d, double* mu, double* var);
/* abo_acq over M candidates sharded across the devices; scores (M, optional) land in the caller's array shard by shard,
 * top_val / top_idx (k) are the merged global selection with 0-based indices into Z */
int32_t abo_mgpu_acq(abo_mgpu* mg, const double* Z, int64_t M, int32_t d, int32_t kind, double p0, double best_y,
                     double* scores, int32_t k, double* top_val, int64_t* top_idx);
/* the grid stage of optimize_acquisition (acq_utils.jl:44-52) without the candidates ever crossing PCIe: device i generates
 * its shard of the n-point Latin-hypercube design (abo_lhs), scores it and selects; top_x (k × d, optional) receives the
 * coordinates of the selected points */
int32_t abo_mgpu_acq_lhs(abo_mgpu* mg, int64_t n, int32_t d, const double* lower, const double* upper, uint64_t seed,
                         int32_t kind, double p0, double best_y, int32_t k, double* top_val, int64_t* top_idx, double* top_x);
/* BASELINE config 5 across devices.  abo_mgpu_append: every device applies the same bordered append (abo_append) and the
 * group moves on to the N+1-point model (clone first to keep the old one); a candidate set, if given, is down-dated in the
 * same call (abo_cand_downdate).  abo_mgpu_cand_create / _lhs: shard a candidate grid over the devices and evaluate its
 * posterior; _refresh after a refit.  abo_mgpu_cand_acq: epilogue + merged top-k on the stored posterior.
 * abo_mgpu_cand_qei: greedy (Kriging-believer) q-EI — q × [EI + arg-max per device, ONE all-gather of the devices' pick
 * records] with the grid conditioned on each pick (y = μ(x): variances only) before the next; on return model and set are as
 * before.  ABI 6: the block form of abo_cand_qei run across the shards whenever every shard qualifies — per batch the shards'
 * top-T are gathered, every shard forms the covariance columns of the SAME T block points in one pass over its part of K_ZX
 * and conditions on a pick by a rank-1 correction from the chain of earlier picks' columns (records {EI, index, μ, σ², x,
 * chain values}); no fantasy append, no pass per pick; blocks and chain stay with the set from call to call
 * (abo_mgpu_cand_qei_stats).  Otherwise (a shard without room for the block buffers, a gradient-enhanced model, q > 64): the
 * plain loop — the same fantasy append (y = μ(x)) and O(N·M) down-date on every device between picks (q − 1 of them: the last
 * pick conditions nothing), rolled back at the end.  Both give the same picks as one handle over the whole set.  x_out q × d,
 * idx_out / ei_out q.  distinct != 0 excludes every picked candidate for the rest of the call. */
int32_t abo_mgpu_append(abo_mgpu* mg, const double* x, int32_t d, double y, int64_t* info, abo_mcand* cands);
int32_t abo_mgpu_append_grad(abo_mgpu* mg, const double* x, int32_t d, const double* y, int64_t* info, abo_mcand* cands);
/* abo_optimize_acquisition across the group's devices: the grid stage as abo_mgpu_acq_lhs (shards generated, scored and reduced
 * on their devices, ONE all-gather of the selections), then the selected starts are dealt out contiguously and every device
 * refines its share with abo_refine's launch; a start's refinement does not depend on which device runs it, so the result equals
 * the single-device call with the same seed bit for bit. */
int32_t abo_mgpu_optimize_acquisition(abo_mgpu* mg, int32_t kind, double p0, double best_y, const double* lower, const double* upper,
                                      int32_t d, int64_t n_grid, int32_t n_local, uint64_t seed, const abo_refine_opts* opts,
                                      double* best_x, double* best_val, double* starts_x, double* starts_val, double* refined_x,
                                      double* refined_val);
int32_t abo_mgpu_optimize_acquisition_terms(abo_mgpu* mg, const abo_acq_term* terms, int32_t nterms, const double* lower,
                                            const double* upper, int32_t d, int64_t n_grid, int32_t n_local, uint64_t seed,
                                            const abo_refine_opts* opts, double* best_x, double* best_val, double* starts_x,
                                            double* starts_val, double* refined_x, double* refined_val);
int32_t abo_mgpu_cand_create(abo_mgpu* mg, const double* Z, int64_t M, int32_t d, abo_mcand** out);
int32_t abo_mgpu_cand_create_lhs(abo_mgpu* mg, int64_t n, int32_t d, const double* lower, const double* upper, uint64_t seed,
                                 abo_mcand** out);
int32_t abo_mgpu_cand_refresh(abo_mgpu* mg, abo_mcand* mc);
int32_t abo_mgpu_cand_destroy(abo_mcand* mc);
int32_t abo_mgpu_cand_acq(abo_mgpu* mg, abo_mcand* mc, int32_t kind, double p0, double best_y, int32_t k, double* top_val,
                          int64_t* top_idx);
/* the stored posterior of the whole sharded set in the candidates' global order (abo_cand_get shard by shard; mu / var: M host
 * doubles each, either may be NULL) */
int32_t abo_mgpu_cand_get(abo_mgpu* mg, abo_mcand* mc, double* mu, double* var);
int32_t abo_mgpu_cand_qei(abo_mgpu* mg, abo_mcand* mc, int32_t q, double xi, double best_y, int32_t distinct, double* x_out,
                          int64_t* idx_out, double* ei_out);
/* statistics of the last abo_mgpu_cand_qei on this set (shard 0's; block = 0 when the plain loop ran).  ABI 6: abo_mgpu_cand_qei
 * runs the block form (abo_cand_qei's; block size = the process default) whenever every shard qualifies, the plain loop else. */
int32_t abo_mgpu_cand_qei_stats(abo_mgpu* mg, abo_mcand* mc, abo_qei_stats* out);

/* --- memory -----------------------------------------------------------------------------------
 * Device buffers of destroyed handles are cached per device (update() makes a new model every BO
 * step and drops the old one, src/surrogates/StandardGP.jl:82 / src/bayesian_opt.jl:116-125) and
 * reused by the next handle; this returns the cached blocks of one device to the driver.  The cache
 * is bounded by the environment variable ABO_POOL_LIMIT_MB (default 32768). */
int32_t abo_pool_trim(int32_t device);

/* --- errors / version ---------------------------------------------------------------------------*/
/* copies the calling thread's last error text (NUL-terminated, truncated to cap) */
int32_t abo_last_error(char* buf, size_t cap);
int32_t abo_abi_version(void);

/* --- building blocks exposed for tests and profiling (all buffers DEVICE memory) ------------------
 * NOT part of the shipped ABI: compiled only into the test build of the library (-DABO_TEST_HOOKS:
 * abstractbayesopt.jl_amd/lib/libabo_hip_test.so, what the GPU suite loads); libabo_hip.so exports none of them. */
#ifdef ABO_TEST_HOOKS
/* The int8-residue engine's host constants for n moduli (no GPU needed): p[16] moduli, tables[4][16] = {1/p, 2^26 mod p
 * (symmetric), head and tail of (P/p)·((P/p)⁻¹ mod p)}, scal[3] = {head of P, tail of P, 1/P}, *eP with 2^eP ≤ P/4. */
int32_t abo_test_oz_plan(int32_t n, int32_t* p, double* tables, double* scal, int32_t* eP);
/* partial[tb][j] = Σ_{i in 128-row block tb, i < nvalid} (Σ_{k ≤ i} W[i][k]·Kxz[j][k])² through the int8-residue engine's own
 * quantisers, GEMM and reconstruction, for ANY lower-triangular W [Np][ldw] and any Kxz [Mc][ldk] with |Kxz| ≤ kmax (device
 * buffers; Np, Mc multiples of 128; partial [Np/128][ldp]). */
int32_t abo_test_oz_contract(int32_t device, const double* W, int64_t ldw, int32_t Np, int32_t nvalid, const double* Kxz, int64_t ldk,
                             int32_t Mc, double kmax, int32_t nmod, double* partial, int64_t ldp);
/* f[j], grad[j][0..d) = value and analytic gradient of the acquisition function at Z[j] (host buffers, M × d): the evaluation
 * the refinement stage is built on, one workgroup per point */
int32_t abo_test_acq_grad(abo_gp* gp, int32_t kind, double p0, double best_y, const double* Z, int64_t M, int32_t d, double* f,
                          double* grad);
int32_t abo_test_acq_grad_terms(abo_gp* gp, const abo_acq_term* terms, int32_t nterms, const double* Z, int64_t M, int32_t d,
                                double* f, double* grad);
/* out[i] = kappa(family, d2[i]) evaluated with the device math of the kernel-matrix generator */
int32_t abo_test_kappa(int32_t device, int32_t family, const double* d2, double* out, int64_t n);
/* C[i][j] = alpha·Σ_k A[i][k]·B[j][k] + beta·C[i][j]; M, N multiples of 128, K multiple of 16,
 * leading dimensions even.  Exercises the fp64 MFMA tile core every solver stage is built on. */
int32_t abo_test_gemm_nt(int32_t device, const double* A, const double* B, double* C, int32_t M,
                         int32_t N, int32_t K, int64_t lda, int64_t ldb, int64_t ldc, double alpha,
                         double beta);
#endif /* ABO_TEST_HOOKS */

#ifdef __cplusplus
}
#endif
#endif /* ABO_HIP_H */
