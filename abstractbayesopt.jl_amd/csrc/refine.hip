// Local refinement stage of optimize_acquisition on the device (reference: src/acquisition_functions/acq_utils.jl:55-73 —
// for every one of the n_local best grid points one box-constrained L-BFGS run, Fminbox(LBFGS(HagerZhang(linesearchmax = 20))),
// Optim.Options(g_tol = 1e-5, f_abstol = 2.2e-9, x_abstol = 1e-4), each objective evaluation an M = 1 posterior call and each
// gradient a finite-difference stencil of such calls).
//
// Here ONE launch refines all starts: one 512-thread workgroup per start runs the whole projected L-BFGS of that start — value
// and ANALYTIC gradient of the acquisition function per evaluation, Armijo backtracking (≤ linesearch_max trials), the
// reference's three stopping rules — with no host round trip in between.  Every loop is bounded (max_iter × linesearch_max
// evaluations), so the grid always drains.
//
// One evaluation at a point x (all threads of the workgroup; k, v, u live in a per-start scratch of 4·Np doubles):
//   k_i  = σ_f² κ(‖x/ℓ − X_i/ℓ‖²)          g_i = σ_f² κ'(·)   (derivative with respect to the squared scaled distance)
//   μ    = m + Σ k_i α_i
//   v    = L⁻¹ k      (row i of W against k: one wave per row, 16-byte loads, xor-tree reduction — fixed order)
//   σ²   = σ_f² − ‖v‖² + 1e-18              (the same latent variance as abo_predict: StandardGP.jl:377-379)
//   u    = L⁻ᵀ v = K⁻¹ k                    (row i of WT against v)
//   ∇μ   = Σ α_i ∂k_i/∂x,   ∇σ² = −2 Σ u_i ∂k_i/∂x,   ∂k_i/∂x_c = g_i · 2 (x_c − X_ic)/ℓ²
//   f, ∂f/∂μ, ∂f/∂σ² of EI / UCB / PI in closed form (ExpectedImprovement.jl:40-66, UpperConfidenceBound.jl:38-45,
//   ProbabilityImprovement.jl:38-63 — including their σ² ≤ 1e-12 branch) → ∇f = ∂f/∂μ ∇μ + ∂f/∂σ² ∇σ².
// The reference differentiates the same f by central differences; the analytic gradient is checked against central differences
// of the library's own acquisition values in tests/test_gpu_refine.py.
//
// Cost per evaluation and start: N² flop and 8·N² bytes of L⁻¹ (both triangles once), served from L2 / Infinity Cache when the
// starts walk in step: microseconds at the sizes the reference's loop lives at (N < 10³).  From 1024 factor rows on the lockstep
// variant at the end of this file takes over (a round's evaluations batched on the MFMA tile core: L⁻¹ read once per round).
#include "abo_kappa.h"
#include "abo_kernels.h"
#include "../../include/abo_hip.h"
#include "abo_acq_dev.h"

namespace abo {

typedef double d2_t __attribute__((ext_vector_type(2)));

// κ(u) and dκ/du, u = squared scaled distance; κ with exactly the operations of kappa_eval (abo_kappa.h), so a refined point's
// value equals what abo_acq returns for it
template <int FAM>
__device__ __forceinline__ void kappa_and_deriv(double u, double& k, double& dk) {
    if constexpr (FAM == ABO_KERNEL_SE) {
        const double e = exp_nonpos(-0.5 * u);
        k = e; dk = -0.5 * e;
    } else if constexpr (FAM == ABO_KERNEL_MATERN52) {
        const double a = 2.23606797749978969640917366873128;
        const double r = sqrt_pos(u), e = exp_nonpos(-a * r), t = fma(a, r, 1.0);
        k = fma(u, 5.0 / 3.0, t) * e; dk = (-5.0 / 6.0) * t * e;
    } else if constexpr (FAM == ABO_KERNEL_MATERN72) {
        const double a = 2.64575131106459059050161575363926;
        const double r = sqrt_pos(u), e = exp_nonpos(-a * r), t = fma(a, r, 1.0);
        k = fma(u * r, 7.0 * a / 15.0, fma(u, 14.0 / 5.0, t)) * e;
        dk = (-7.0 / 10.0) * fma(u, 7.0 / 3.0, t) * e;
    } else {
        const double a = 1.73205080756887729352744634150587;
        const double r = sqrt_pos(u), e = exp_nonpos(-a * r);
        k = fma(a, r, 1.0) * e; dk = -1.5 * e;
    }
}

constexpr int RCH = 8;          // gradient components reduced per pass
constexpr int RT = 512;         // threads per workgroup (= per start): 8 waves keep 8 × 4 rows of L⁻¹ in flight
constexpr int RW = RT / 64;

// sum of nv ≤ 2·RCH per-thread values over the workgroup, fixed order (lanes: xor tree, then the waves in order);
// results in out[0..nv) for every thread after the call
__device__ __forceinline__ void block_sum(double* vals, int nv, double* red, double* out) {
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    for (int q = 0; q < nv; ++q) {
        double v = vals[q];
#pragma unroll
        for (int o = 32; o >= 1; o >>= 1) v += __shfl_xor(v, o);
        if (lane == 0) red[wave * 2 * RCH + q] = v;
    }
    __syncthreads();
    if (t < nv) {
        double r = red[t];
#pragma unroll
        for (int w = 1; w < RW; ++w) r += red[w * 2 * RCH + t];
        out[t] = r;
    }
    __syncthreads();
}

// value f and gradient grad[0..d) of the acquisition function at x (LDS).  xs: LDS [dp]; scr: this start's 4·Np doubles;
// red: LDS [RW·2·RCH]; out: LDS [2·RCH]; fres: LDS [4] (f, μ, σ²).  All threads call it; all see the result after the final barrier.
template <int FAM>
__device__ void eval_point(const RefineArgs& a, const double* x, double* xs, double* scr, double* red, double* out, double* fres,
                           double* grad) {
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int N = a.N, Np = a.Np, d = a.d, dp = a.dp;
    double* kv = scr;
    double* gv = scr + Np;
    double* vv = scr + 2 * (int64_t)Np;
    double* uv = scr + 3 * (int64_t)Np;
    for (int c = t; c < dp; c += RT) xs[c] = c < d ? x[c] * a.s : 0.0;
    __syncthreads();
    // kernel values, their derivatives, the mean
    double acc[2 * RCH];
    acc[0] = 0.0;
    for (int i = t; i < Np; i += RT) {
        double k = 0.0, dk = 0.0;
        if (i < N) {
            const double* xi = a.Xs + (int64_t)i * dp;
            double u = 0.0;
            for (int c = 0; c < dp; ++c) { const double e = xi[c] - xs[c]; u = fma(e, e, u); }
            kappa_and_deriv<FAM>(u, k, dk);
            k *= a.sigma_f2; dk *= a.sigma_f2;
            acc[0] = fma(k, a.alpha[i], acc[0]);
        }
        kv[i] = k; gv[i] = dk;
    }
    block_sum(acc, 1, red, out);                  // (its barriers also publish kv / gv to the workgroup)
    const double mu = a.mean_c + out[0];
    __syncthreads();
    // v = W k: four rows per wave at a time (they share the loads of k), the pair loop unrolled twice — eight 16-byte loads of
    // L⁻¹ in flight per lane.  Row i of W holds exact zeros beyond column i, so the rows of a group run to the group's longest
    // row without masks.  Rows ≥ N (identity padding, or the remains of a discarded appended branch) are not read: v = 0 there.
    acc[0] = 0.0;
    for (int i0 = 4 * wave; i0 < Np; i0 += 4 * RW) {
        double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;
        if (i0 < N) {
            const double* w0 = a.W + (int64_t)i0 * a.ld;
            const double* w1 = w0 + (i0 + 1 < N ? a.ld : 0);
            const double* w2 = w0 + (i0 + 2 < N ? 2 * a.ld : 0);
            const double* w3 = w0 + (i0 + 3 < N ? 3 * a.ld : 0);
            const int npair = (i0 + 3) / 2 + 1;                      // columns 0 … i0+3 (+1 of zeros)
            int q = lane;
            for (; q + 64 < npair; q += 128) {
                const d2_t ka = *reinterpret_cast<const d2_t*>(kv + 2 * q), kb = *reinterpret_cast<const d2_t*>(kv + 2 * q + 128);
                const d2_t a0 = *reinterpret_cast<const d2_t*>(w0 + 2 * q), b0 = *reinterpret_cast<const d2_t*>(w0 + 2 * q + 128);
                const d2_t a1 = *reinterpret_cast<const d2_t*>(w1 + 2 * q), b1 = *reinterpret_cast<const d2_t*>(w1 + 2 * q + 128);
                const d2_t a2 = *reinterpret_cast<const d2_t*>(w2 + 2 * q), b2 = *reinterpret_cast<const d2_t*>(w2 + 2 * q + 128);
                const d2_t a3 = *reinterpret_cast<const d2_t*>(w3 + 2 * q), b3 = *reinterpret_cast<const d2_t*>(w3 + 2 * q + 128);
                s0 = fma(b0[1], kb[1], fma(b0[0], kb[0], fma(a0[1], ka[1], fma(a0[0], ka[0], s0))));
                s1 = fma(b1[1], kb[1], fma(b1[0], kb[0], fma(a1[1], ka[1], fma(a1[0], ka[0], s1))));
                s2 = fma(b2[1], kb[1], fma(b2[0], kb[0], fma(a2[1], ka[1], fma(a2[0], ka[0], s2))));
                s3 = fma(b3[1], kb[1], fma(b3[0], kb[0], fma(a3[1], ka[1], fma(a3[0], ka[0], s3))));
            }
            for (; q < npair; q += 64) {
                const d2_t ka = *reinterpret_cast<const d2_t*>(kv + 2 * q);
                const d2_t a0 = *reinterpret_cast<const d2_t*>(w0 + 2 * q), a1 = *reinterpret_cast<const d2_t*>(w1 + 2 * q);
                const d2_t a2 = *reinterpret_cast<const d2_t*>(w2 + 2 * q), a3 = *reinterpret_cast<const d2_t*>(w3 + 2 * q);
                s0 = fma(a0[1], ka[1], fma(a0[0], ka[0], s0));
                s1 = fma(a1[1], ka[1], fma(a1[0], ka[0], s1));
                s2 = fma(a2[1], ka[1], fma(a2[0], ka[0], s2));
                s3 = fma(a3[1], ka[1], fma(a3[0], ka[0], s3));
            }
#pragma unroll
            for (int o = 32; o >= 1; o >>= 1) {
                s0 += __shfl_xor(s0, o); s1 += __shfl_xor(s1, o); s2 += __shfl_xor(s2, o); s3 += __shfl_xor(s3, o);
            }
            if (i0 + 1 >= N) s1 = 0.0;
            if (i0 + 2 >= N) s2 = 0.0;
            if (i0 + 3 >= N) s3 = 0.0;
        }
        if (lane == 0) {
            vv[i0] = s0; vv[i0 + 1] = s1; vv[i0 + 2] = s2; vv[i0 + 3] = s3;
            acc[0] = fma(s3, s3, fma(s2, s2, fma(s1, s1, fma(s0, s0, acc[0]))));
        }
    }
    if (lane != 0) acc[0] = 0.0;
    block_sum(acc, 1, red, out);
    const double var = a.sigma_f2 - out[0] + 1e-18;
    __syncthreads();
    // u = Wᵀ v: rows of WT (upper: exact zeros in front of the diagonal), columns up to N (v is zero from N on, so a stale column
    // of a discarded appended branch — at most the one that completes the last pair — meets a zero)
    for (int i0 = 4 * wave; i0 < N; i0 += 4 * RW) {
        const double* w0 = a.WT + (int64_t)i0 * a.ld;
        const double* w1 = w0 + (i0 + 1 < N ? a.ld : 0);
        const double* w2 = w0 + (i0 + 2 < N ? 2 * a.ld : 0);
        const double* w3 = w0 + (i0 + 3 < N ? 3 * a.ld : 0);
        const int q1 = (N + 1) >> 1;
        double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;
        int q = (i0 >> 1) + lane;
        for (; q + 64 < q1; q += 128) {
            const d2_t ka = *reinterpret_cast<const d2_t*>(vv + 2 * q), kb = *reinterpret_cast<const d2_t*>(vv + 2 * q + 128);
            const d2_t a0 = *reinterpret_cast<const d2_t*>(w0 + 2 * q), b0 = *reinterpret_cast<const d2_t*>(w0 + 2 * q + 128);
            const d2_t a1 = *reinterpret_cast<const d2_t*>(w1 + 2 * q), b1 = *reinterpret_cast<const d2_t*>(w1 + 2 * q + 128);
            const d2_t a2 = *reinterpret_cast<const d2_t*>(w2 + 2 * q), b2 = *reinterpret_cast<const d2_t*>(w2 + 2 * q + 128);
            const d2_t a3 = *reinterpret_cast<const d2_t*>(w3 + 2 * q), b3 = *reinterpret_cast<const d2_t*>(w3 + 2 * q + 128);
            s0 = fma(b0[1], kb[1], fma(b0[0], kb[0], fma(a0[1], ka[1], fma(a0[0], ka[0], s0))));
            s1 = fma(b1[1], kb[1], fma(b1[0], kb[0], fma(a1[1], ka[1], fma(a1[0], ka[0], s1))));
            s2 = fma(b2[1], kb[1], fma(b2[0], kb[0], fma(a2[1], ka[1], fma(a2[0], ka[0], s2))));
            s3 = fma(b3[1], kb[1], fma(b3[0], kb[0], fma(a3[1], ka[1], fma(a3[0], ka[0], s3))));
        }
        for (; q < q1; q += 64) {
            const d2_t ka = *reinterpret_cast<const d2_t*>(vv + 2 * q);
            const d2_t a0 = *reinterpret_cast<const d2_t*>(w0 + 2 * q), a1 = *reinterpret_cast<const d2_t*>(w1 + 2 * q);
            const d2_t a2 = *reinterpret_cast<const d2_t*>(w2 + 2 * q), a3 = *reinterpret_cast<const d2_t*>(w3 + 2 * q);
            s0 = fma(a0[1], ka[1], fma(a0[0], ka[0], s0));
            s1 = fma(a1[1], ka[1], fma(a1[0], ka[0], s1));
            s2 = fma(a2[1], ka[1], fma(a2[0], ka[0], s2));
            s3 = fma(a3[1], ka[1], fma(a3[0], ka[0], s3));
        }
#pragma unroll
        for (int o = 32; o >= 1; o >>= 1) {
            s0 += __shfl_xor(s0, o); s1 += __shfl_xor(s1, o); s2 += __shfl_xor(s2, o); s3 += __shfl_xor(s3, o);
        }
        if (lane == 0) {
            uv[i0] = s0;
            if (i0 + 1 < N) uv[i0 + 1] = s1;
            if (i0 + 2 < N) uv[i0 + 2] = s2;
            if (i0 + 3 < N) uv[i0 + 3] = s3;
        }
    }
    __syncthreads();
    double dmu, dvar;
    const double f = terms_value_and_partials(a.terms, mu, var, dmu, dvar);
    // gradient, RCH components per pass: ∂k_i/∂x_c = g_i · 2 (xs_c − Xs_ic) · s
    for (int c0 = 0; c0 < d; c0 += RCH) {
        const int nc = (d - c0) < RCH ? (d - c0) : RCH;
#pragma unroll
        for (int q = 0; q < 2 * RCH; ++q) acc[q] = 0.0;
        for (int i = t; i < N; i += RT) {
            const double* xi = a.Xs + (int64_t)i * dp;
            const double gi = 2.0 * a.s * gv[i], ai = a.alpha[i] * gi, ui = uv[i] * gi;
#pragma unroll
            for (int q = 0; q < RCH; ++q) {
                if (q < nc) {
                    const double e = xs[c0 + q] - xi[c0 + q];
                    acc[q] = fma(ai, e, acc[q]);
                    acc[RCH + q] = fma(ui, e, acc[RCH + q]);
                }
            }
        }
        block_sum(acc, 2 * RCH, red, out);
        if (t < nc) grad[c0 + t] = dmu * out[t] + dvar * (-2.0 * out[RCH + t]);
        __syncthreads();
    }
    if (t == 0) { fres[0] = f; fres[1] = mu; fres[2] = var; }
    __syncthreads();
}

// One iteration's head of the projected L-BFGS at the iterate x with gradient g (thread-serial: O(m·d)): projected gradient pg, the
// curvature pair of the step just accepted (have_prev), the stop test, the two-loop ascent direction p and the first step length.
// Returns 1 to stop (projected gradient within g_tol, or not a number).  Used by the one-launch kernel (state in LDS) and by the
// lockstep variant (state in global memory): the same arithmetic in the same order.
__device__ __forceinline__ int lbfgs_direction(int d, int m, const double* x, const double* g, double* pg, double* p, const double* xprev,
                                               const double* pgprev, const double* lo, const double* up, double* Sh, double* Yh, double* rho,
                                               double* al, bool have_prev, int& nh, double g_tol, double& tstep) {
    // projected gradient: components that push against an active bound are dropped (maximisation)
    double gmax = 0.0;
    bool bad = false;
    for (int c = 0; c < d; ++c) {
        double v = g[c];
        if ((x[c] <= lo[c] && v < 0.0) || (x[c] >= up[c] && v > 0.0)) v = 0.0;
        bad = bad || !(v == v);
        gmax = fmax(gmax, fabs(v));
        if (have_prev && m > 0 && nh == m && c == 0) {                   // make room for the new pair before pg is overwritten
            for (int h = 1; h < m; ++h)
                for (int cc = 0; cc < d; ++cc) { Sh[(h - 1) * d + cc] = Sh[h * d + cc]; Yh[(h - 1) * d + cc] = Yh[h * d + cc]; }
            nh = m - 1;
        }
        if (have_prev && m > 0) { Sh[nh * d + c] = x[c] - xprev[c]; Yh[nh * d + c] = pgprev[c] - v; }   // s = Δx, y = −Δ(pg)
        pg[c] = v;
    }
    if (have_prev && m > 0) ++nh;
    if (bad || gmax <= g_tol) return 1;
    // two-loop recursion for an ascent direction; pairs with s·y ≤ 0 are skipped
    for (int c = 0; c < d; ++c) p[c] = pg[c];
    for (int h = nh - 1; h >= 0; --h) {
        double sy = 0.0, sq = 0.0;
        for (int c = 0; c < d; ++c) { sy = fma(Sh[h * d + c], Yh[h * d + c], sy); sq = fma(Sh[h * d + c], p[c], sq); }
        rho[h] = sy > 1e-300 ? 1.0 / sy : 0.0;
        al[h] = rho[h] * sq;
        for (int c = 0; c < d; ++c) p[c] = fma(-al[h], Yh[h * d + c], p[c]);
    }
    if (nh > 0) {
        double sy = 0.0, yy = 0.0;
        for (int c = 0; c < d; ++c) { sy = fma(Sh[(nh - 1) * d + c], Yh[(nh - 1) * d + c], sy); yy = fma(Yh[(nh - 1) * d + c], Yh[(nh - 1) * d + c], yy); }
        const double sc = (sy > 1e-300 && yy > 0.0) ? sy / yy : 1.0;
        for (int c = 0; c < d; ++c) p[c] *= sc;
    }
    for (int h = 0; h < nh; ++h) {
        double yq = 0.0;
        for (int c = 0; c < d; ++c) yq = fma(Yh[h * d + c], p[c], yq);
        const double b = rho[h] * yq;
        for (int c = 0; c < d; ++c) p[c] = fma(al[h] - b, Sh[h * d + c], p[c]);
    }
    double ppg = 0.0, pmax = 0.0, wmin = 1.0e300;
    for (int c = 0; c < d; ++c) ppg = fma(p[c], pg[c], ppg);
    if (!(ppg > 0.0)) for (int c = 0; c < d; ++c) p[c] = pg[c];      // not an ascent direction: steepest ascent
    for (int c = 0; c < d; ++c) {
        pmax = fmax(pmax, fabs(p[c]));
        if (up[c] > lo[c]) wmin = fmin(wmin, up[c] - lo[c]);            // a degenerate side (lower == upper) pins its coordinate, it does not limit the others
    }
    // first step: a tenth of the narrowest (non-degenerate) box side at most
    tstep = nh == 0 ? fmin(1.0, 0.1 * wmin / fmax(pmax, 1e-300)) : 1.0;
    return 0;
}

// LDS layout (doubles): x, g, pg, p, xprev, pgprev, cand, gc, lo, up, xrun [d each]; xs [dp]; Sh, Yh [m·d each]; rho, al [m each];
// red [RW·2·RCH]; out [2·RCH]; fres [4]; ctrl [4 ints → 2 doubles]
size_t refine_lds_bytes(int d, int dp, int m) {
    return sizeof(double) * ((size_t)11 * d + dp + (size_t)2 * m * d + 2 * m + RW * 2 * RCH + 2 * RCH + 4 + 2);
}

// The reference's optimiser is Fminbox(LBFGS) (acq_utils.jl:9-13): an OUTER loop of inner L-BFGS runs.  An inner run ends when one of
// its iterations moves x by ≤ x_abstol or f by ≤ f_abstol (or its line search finds no step); the outer loop then starts a fresh inner
// run (new L-BFGS state) from where that one ended, and stops only when a WHOLE inner run has moved x by ≤ x_abstol or f by ≤ f_abstol
// (or the gradient test holds).  The stage below keeps that structure — the same two tolerances on both levels, one shared budget of
// max_iter accepted steps: a run that ends (rule or failed search) is followed by a fresh one from the same point unless it as a
// whole made no progress beyond the tolerances.  (Round 4 stopped at the end of the FIRST inner run: a quasi-Newton step shortened by
// a curvature pair taken across a change of the active set ended the whole refinement 0.07 short of the maximiser;
// profiles/r05_refine_diag.txt.)
__device__ __forceinline__ int outer_converged(int d, const double* x, const double* xrun, double f, double frun, double x_abstol, double f_abstol) {
    double dx = 0.0;
    for (int c = 0; c < d; ++c) dx = fmax(dx, fabs(x[c] - xrun[c]));
    return (dx <= x_abstol || fabs(f - frun) <= f_abstol) ? 1 : 0;
}

template <int FAM>
__global__ void __launch_bounds__(RT) refine_kernel(RefineArgs a) {
    extern __shared__ double sm[];
    const int t = threadIdx.x, d = a.d, dp = a.dp, m = a.history;
    double* x = sm;
    double* g = x + d;
    double* pg = g + d;
    double* p = pg + d;
    double* xprev = p + d;
    double* pgprev = xprev + d;
    double* cand = pgprev + d;
    double* gc = cand + d;
    double* lo = gc + d;
    double* up = lo + d;
    double* xrun = up + d;
    double* xs = xrun + d;
    double* Sh = xs + dp;
    double* Yh = Sh + (size_t)m * d;
    double* rho = Yh + (size_t)m * d;
    double* al = rho + m;
    double* red = al + m;
    double* out = red + RW * 2 * RCH;
    double* fres = out + 2 * RCH;
    int* ctrl = reinterpret_cast<int*>(fres + 4);
    const int sidx = blockIdx.x;
    double* scr = a.scratch + (int64_t)sidx * 4 * a.Np;

    for (int c = t; c < d; c += RT) {
        lo[c] = a.lower[c]; up[c] = a.upper[c];
        const double v = a.starts[(int64_t)sidx * d + c];
        x[c] = v == v ? fmin(fmax(v, a.lower[c]), a.upper[c]) : v;         // a NaN coordinate stays NaN (→ non-finite value → returned as is)
        xrun[c] = x[c];
    }
    __syncthreads();
    eval_point<FAM>(a, x, xs, scr, red, out, fres, g);
    double f = fres[0];
    double frun = f;                                                       // value at the beginning of the current inner run
    int nev = 1, it = 0, nh = 0;
    bool have_prev = false;
    double tstep = 1.0;
    if (f == f && fabs(f) < 1.0e300) {                                   // a non-finite start value: nothing to refine
        // `it` counts accepted steps.  A pass of this loop ends in one, or in a failed line search — which ends the inner run, and
        // a fresh run's first search failing ends the refinement (nothing has moved since the run began): two failures never
        // follow each other, so 2·max_iter + 2 passes bound the loop
        for (int pass = 0; pass < 2 * a.max_iter + 2 && it < a.max_iter; ++pass) {
            if (t == 0) {
                const int stop = lbfgs_direction(d, m, x, g, pg, p, xprev, pgprev, lo, up, Sh, Yh, rho, al, have_prev, nh, a.g_tol, fres[3]);
                ctrl[0] = stop;
            }
            __syncthreads();
            if (ctrl[0]) break;
            tstep = fres[3];
            bool accepted = false;
            double fc = f;
            for (int ls = 0; ls < a.ls_max; ++ls) {
                for (int c = t; c < d; c += RT) cand[c] = fmin(fmax(fma(tstep, p[c], x[c]), lo[c]), up[c]);
                __syncthreads();
                eval_point<FAM>(a, cand, xs, scr, red, out, fres, gc);
                fc = fres[0];
                ++nev;
                if (t == 0) {
                    double lin = 0.0;
                    for (int c = 0; c < d; ++c) lin = fma(pg[c], cand[c] - x[c], lin);
                    ctrl[1] = (fc == fc && fabs(fc) < 1.0e300 && fc >= f + 1e-4 * lin) ? 1 : 0;      // Armijo on the projected step
                }
                __syncthreads();
                if (ctrl[1]) { accepted = true; break; }
                tstep *= 0.5;
            }
            bool run_ends = !accepted;                                   // no acceptable step: the inner run ends where it stands
            if (accepted) {
                if (t == 0) {
                    double dx = 0.0;
                    for (int c = 0; c < d; ++c) dx = fmax(dx, fabs(cand[c] - x[c]));
                    ctrl[2] = (dx <= a.x_abstol || fabs(fc - f) <= a.f_abstol) ? 1 : 0;
                    for (int c = 0; c < d; ++c) { xprev[c] = x[c]; pgprev[c] = pg[c]; x[c] = cand[c]; g[c] = gc[c]; }
                }
                __syncthreads();
                f = fc;
                have_prev = true;
                ++it;
                run_ends = ctrl[2] != 0;
            }
            if (run_ends) {
                // the outer level (outer_converged): stop when the whole inner run moved x or f by no more than the tolerances,
                // else a fresh inner run — no curvature pairs, cautious first step — from this point
                if (t == 0) {
                    ctrl[3] = outer_converged(d, x, xrun, f, frun, a.x_abstol, a.f_abstol);
                    for (int c = 0; c < d; ++c) xrun[c] = x[c];
                }
                __syncthreads();
                if (ctrl[3]) break;
                frun = f;
                have_prev = false; nh = 0;
            }
        }
    }
    for (int c = t; c < d; c += RT) a.x_out[(int64_t)sidx * d + c] = x[c];
    if (t == 0) {
        a.f_out[sidx] = f;
        if (a.iters_out) { a.iters_out[2 * sidx] = it; a.iters_out[2 * sidx + 1] = nev; }
    }
}

// value and gradient at S points in one launch (one workgroup per point): the test hook behind abo_test_acq_grad
template <int FAM>
__global__ void __launch_bounds__(RT) acq_grad_kernel(RefineArgs a) {
    extern __shared__ double sm[];
    const int t = threadIdx.x, d = a.d, dp = a.dp;
    double* x = sm;
    double* g = x + d;
    double* xs = g + d;
    double* red = xs + dp;
    double* out = red + RW * 2 * RCH;
    double* fres = out + 2 * RCH;
    const int sidx = blockIdx.x;
    for (int c = t; c < d; c += RT) x[c] = a.starts[(int64_t)sidx * d + c];
    __syncthreads();
    eval_point<FAM>(a, x, xs, a.scratch + (int64_t)sidx * 4 * a.Np, red, out, fres, g);
    for (int c = t; c < d; c += RT) a.x_out[(int64_t)sidx * d + c] = g[c];
    if (t == 0) a.f_out[sidx] = fres[0];
}

// ---- lockstep variant for large N ----------------------------------------------------------------------------------------------------------
// One workgroup per start re-streams L⁻¹ (8·N² bytes) per evaluation at what ONE CU can pull (≈ 150 GB/s): microseconds up to N ≈ 10³,
// milliseconds at N = 8192.  Every active start owes exactly one evaluation per round (its start, or its current line-search trial), so
// from 1024 factor rows on the starts advance in LOCKSTEP ROUNDS and a round's evaluations are batched:
//     rl_kgen_kernel      k and κ' of every pending point against the training set                         [Sp][Np] each
//     gemm_nt (fp64 MFMA) V = K·L⁻ᵀ  (A = W lower-triangular k-range, B = the pending points' k rows; transposed copy)  L⁻¹ read ONCE per round
//     rl_zero_tail        V[:, N … Np) = 0 (rows ≥ N of a shared factor may hold a discarded appended branch)
//     gemm_nt             U = V·L⁻¹  (A = WT upper-triangular k-range)
//     rl_reduce_kernel    μ, σ², ∇μ, ∇σ² → f, ∇f per point (the tail of eval_point)
//     rl_step_kernel      one thread per start: consume (f, ∇f), advance the start's L-BFGS state machine (the same lbfgs_direction, the
//                         same Armijo rule and stopping tests as refine_kernel), write its next pending point
// all on the handle's stream, no host round trip inside a batch of rounds; the host looks at a device counter every 8 rounds and stops
// when no start is active.  Same algorithm as refine_kernel; v and u come out of the MFMA tile core in another summation order, so the
// two variants agree to rounding, not bit for bit (tests/test_gpu_refine.py).
struct RlState {                 // per-start scalars
    double f, tstep, frun;
    int phase, it, ls, nh, nev, have_prev;      // phase 0 = waiting for the start's value, 1 = in a line search, 2 = finished
};

// doubles of vector state per start: x, g, pg, p, xprev, pgprev [6·d], Sh, Yh [2·m·d], rho, al [2·m], xrun [d] (x at the beginning of
// the current inner run: refine_kernel's outer level)
__host__ __device__ inline size_t rl_vec_doubles(int d, int m) { return (size_t)7 * d + (size_t)2 * m * d + 2 * m; }

template <int FAM>
// Rows of a round's batch are COMPACT: row c belongs to the c-th still-active start, start_of[c] (ascending start index; built by
// rl_compact_kernel after every step), rows ≥ *nact are zero.  2175 evaluations over 100 starts in 75 rounds means 29 active
// starts per round on average: the products below are sized by the active count, not by the number of starts.
__global__ void __launch_bounds__(256) rl_kgen_kernel(RefineArgs a, const double* __restrict__ P, const int* __restrict__ start_of,
                                                      const int* __restrict__ nact, double* __restrict__ KX, double* __restrict__ GV) {
    __shared__ double xs[1024];
    const int c0 = blockIdx.y, t = threadIdx.x;
    const int i = blockIdx.x * 256 + t;
    const bool on = c0 < *nact;
    const int j = on ? start_of[c0] : 0;
    for (int c = t; c < a.dp; c += 256) xs[c] = (on && c < a.d) ? P[(int64_t)j * a.d + c] * a.s : 0.0;
    __syncthreads();
    if (i >= a.Np) return;
    double k = 0.0, dk = 0.0;
    if (on && i < a.N) {
        const double* xi = a.Xs + (int64_t)i * a.dp;
        double u = 0.0;
        for (int c = 0; c < a.dp; ++c) { const double e = xi[c] - xs[c]; u = fma(e, e, u); }
        kappa_and_deriv<FAM>(u, k, dk);
        k *= a.sigma_f2; dk *= a.sigma_f2;
    }
    KX[(int64_t)c0 * a.Np + i] = k;
    GV[(int64_t)c0 * a.Np + i] = dk;
}

// start_of[0 .. nact) = the active starts in ascending order, *nact = their number (one workgroup; S is a few hundred at most)
__global__ void __launch_bounds__(256) rl_compact_kernel(const int* __restrict__ active, int S, int* __restrict__ start_of,
                                                         int* __restrict__ nact) {
    __shared__ int wsum[4], base;
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    if (t == 0) base = 0;
    __syncthreads();
    for (int j0 = 0; j0 < S; j0 += 256) {
        const int j = j0 + t;
        const int on = j < S && active[j] != 0 ? 1 : 0;
        const unsigned long long m = __ballot(on);
        const int before = __popcll(m & ((1ull << lane) - 1ull));
        if (lane == 0) wsum[wave] = __popcll(m);
        __syncthreads();
        int off = base;
        for (int w = 0; w < wave; ++w) off += wsum[w];
        if (on) start_of[off + before] = j;
        __syncthreads();
        if (t == 0) base += wsum[0] + wsum[1] + wsum[2] + wsum[3];
        __syncthreads();
    }
    if (t == 0) *nact = base;
}

__global__ void rl_zero_tail_kernel(double* V, int Np, int N, int Sp) {
    const int w = Np - N;
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e < w * Sp) V[(int64_t)(e / w) * Np + N + e % w] = 0.0;
}

// EV[j] = {f, ∇f[0..d)} of pending point j
__global__ void __launch_bounds__(RT) rl_reduce_kernel(RefineArgs a, const double* __restrict__ P, const int* __restrict__ start_of,
                                                       const int* __restrict__ nact, const double* __restrict__ KX,
                                                       const double* __restrict__ GV, const double* __restrict__ V,
                                                       const double* __restrict__ U, double* __restrict__ EV) {
    extern __shared__ double sm[];
    double* xs = sm;                         // [dp]
    double* red = xs + a.dp;                 // [RW·2·RCH]
    double* out = red + RW * 2 * RCH;        // [2·RCH]
    const int c = blockIdx.x, t = threadIdx.x, N = a.N, d = a.d, dp = a.dp;
    if (c >= *nact) return;                  // uniform
    const int j = start_of[c];               // compact row c → its start: P and EV are indexed by start
    const double* kv = KX + (int64_t)c * a.Np;
    const double* gv = GV + (int64_t)c * a.Np;
    const double* vv = V + (int64_t)c * a.Np;
    const double* uv = U + (int64_t)c * a.Np;
    for (int q = t; q < dp; q += RT) xs[q] = q < d ? P[(int64_t)j * d + q] * a.s : 0.0;
    __syncthreads();
    double acc[2 * RCH];
    acc[0] = 0.0; acc[1] = 0.0;
    for (int i = t; i < N; i += RT) {
        acc[0] = fma(kv[i], a.alpha[i], acc[0]);
        acc[1] = fma(vv[i], vv[i], acc[1]);
    }
    block_sum(acc, 2, red, out);
    const double mu = a.mean_c + out[0];
    const double var = a.sigma_f2 - out[1] + 1e-18;
    __syncthreads();
    double dmu, dvar;
    const double f = terms_value_and_partials(a.terms, mu, var, dmu, dvar);
    double* ev = EV + (int64_t)j * (d + 1);
    for (int c0 = 0; c0 < d; c0 += RCH) {
        const int nc = (d - c0) < RCH ? (d - c0) : RCH;
#pragma unroll
        for (int q = 0; q < 2 * RCH; ++q) acc[q] = 0.0;
        for (int i = t; i < N; i += RT) {
            const double* xi = a.Xs + (int64_t)i * dp;
            const double gi = 2.0 * a.s * gv[i], ai = a.alpha[i] * gi, ui = uv[i] * gi;
#pragma unroll
            for (int q = 0; q < RCH; ++q) {
                if (q < nc) {
                    const double e = xs[c0 + q] - xi[c0 + q];
                    acc[q] = fma(ai, e, acc[q]);
                    acc[RCH + q] = fma(ui, e, acc[RCH + q]);
                }
            }
        }
        block_sum(acc, 2 * RCH, red, out);
        if (t < nc) ev[1 + c0 + t] = dmu * out[t] + dvar * (-2.0 * out[RCH + t]);
        __syncthreads();
    }
    if (t == 0) ev[0] = f;
}

// one 64-thread workgroup per start: phase −1 (first call) sets the pending point to the clipped start; afterwards consume EV, advance,
// publish the next pending point.  The start's vector state (x, g, pg, p, xprev, pgprev, the curvature pairs) is staged through LDS:
// the L-BFGS algebra is a chain of short dependent loops, and run by one thread straight on global memory every step of it paid a
// memory round trip (46 µs per round at d = 8, m = 10 — 8 % of a round at N = 8192); the lanes move the state in and out, lane 0
// runs the same arithmetic in the same order on the LDS copy.  counters[0] = number of active starts after this round (int64: the
// GEMMs' early-exit word is counters[1] = 1 when 0).
__global__ void __launch_bounds__(64) rl_step_kernel(RefineArgs a, int S, int first, double* __restrict__ P, int* __restrict__ active,
                                                     const double* __restrict__ EV, double* __restrict__ vec, RlState* __restrict__ st,
                                                     long long* __restrict__ counters) {
    extern __shared__ double sv[];
    const int j = blockIdx.x, t = threadIdx.x;
    if (j >= S) return;
    const int d = a.d, m = a.history;
    const int nv = (int)rl_vec_doubles(d, m);
    double* gvec = vec + (size_t)j * nv;
    double* cand = P + (size_t)j * d;              // the pending point doubles as the line search's trial point
    if (first) {
        for (int c = t; c < d; c += 64) {
            const double v = a.starts[(size_t)j * d + c];
            const double x = v == v ? fmin(fmax(v, a.lower[c]), a.upper[c]) : v;
            gvec[c] = x;
            cand[c] = x;
        }
        if (t == 0) {
            RlState& s = st[j];
            s.f = 0.0; s.tstep = 1.0; s.frun = 0.0; s.phase = 0; s.it = 0; s.ls = 0; s.nh = 0; s.nev = 0; s.have_prev = 0;
            active[j] = 1;
            atomicAdd(reinterpret_cast<unsigned long long*>(&counters[0]), 1ull);
        }
        return;
    }
    if (!active[j]) return;                        // uniform over the workgroup
    // stage: state, bounds, the evaluation, the trial point
    double* x = sv;
    double* g = x + d;
    double* pg = g + d;
    double* p = pg + d;
    double* xprev = p + d;
    double* pgprev = xprev + d;
    double* Sh = pgprev + d;
    double* Yh = Sh + (size_t)m * d;
    double* rho = Yh + (size_t)m * d;
    double* al = rho + m;
    double* xrun = al + m;
    double* lo = sv + nv;
    double* up = lo + d;
    double* ev = up + d;                           // [d + 1]
    double* cd = ev + d + 1;                       // [d]
    for (int e = t; e < nv; e += 64) sv[e] = gvec[e];
    for (int c = t; c < d; c += 64) { lo[c] = a.lower[c]; up[c] = a.upper[c]; cd[c] = cand[c]; }
    for (int c = t; c <= d; c += 64) ev[c] = EV[(size_t)j * (d + 1) + c];
    __syncthreads();
    __shared__ int fin;
    if (t == 0) {
        RlState s = st[j];
        const double fc = ev[0];
        ++s.nev;
        bool begin_iteration = false, finished = false, run_ends = false;
        if (s.phase == 0) {                              // the start's own value and gradient
            s.f = fc; s.frun = fc;
            for (int c = 0; c < d; ++c) { g[c] = ev[1 + c]; xrun[c] = x[c]; }
            if (!(fc == fc && fabs(fc) < 1.0e300)) finished = true;      // a non-finite start value: nothing to refine
            else begin_iteration = true;
        } else {                                         // a line-search trial came back
            double lin = 0.0;
            for (int c = 0; c < d; ++c) lin = fma(pg[c], cd[c] - x[c], lin);
            const bool ok = fc == fc && fabs(fc) < 1.0e300 && fc >= s.f + 1e-4 * lin;
            if (ok) {
                double dx = 0.0;
                for (int c = 0; c < d; ++c) dx = fmax(dx, fabs(cd[c] - x[c]));
                run_ends = dx <= a.x_abstol || fabs(fc - s.f) <= a.f_abstol;
                for (int c = 0; c < d; ++c) { xprev[c] = x[c]; pgprev[c] = pg[c]; x[c] = cd[c]; g[c] = ev[1 + c]; }
                s.f = fc; s.have_prev = 1; ++s.it;
                if (!run_ends) { if (s.it >= a.max_iter) finished = true; else begin_iteration = true; }
            } else {
                s.tstep *= 0.5;
                if (++s.ls >= a.ls_max) run_ends = true;                 // no acceptable step: the inner run ends where it stands
            }
        }
        if (run_ends) {
            // refine_kernel's outer level: stop when the whole inner run moved x or f by no more than the tolerances, else a fresh
            // inner run (no curvature pairs, cautious first step) from this point
            const int conv = outer_converged(d, x, xrun, s.f, s.frun, a.x_abstol, a.f_abstol);
            for (int c = 0; c < d; ++c) xrun[c] = x[c];
            if (conv || s.it >= a.max_iter) finished = true;
            else { s.frun = s.f; s.have_prev = 0; s.nh = 0; begin_iteration = true; }
        }
        if (begin_iteration) {
            double t0 = 1.0;
            if (lbfgs_direction(d, m, x, g, pg, p, xprev, pgprev, lo, up, Sh, Yh, rho, al, s.have_prev != 0, s.nh, a.g_tol, t0)) finished = true;
            else { s.tstep = t0; s.ls = 0; s.phase = 1; }
        }
        if (finished) {
            s.phase = 2;
            a.f_out[j] = s.f;
            if (a.iters_out) { a.iters_out[2 * j] = s.it; a.iters_out[2 * j + 1] = s.nev; }
        } else {
            for (int c = 0; c < d; ++c) cd[c] = fmin(fmax(fma(s.tstep, p[c], x[c]), lo[c]), up[c]);
        }
        st[j] = s;
        fin = finished ? 1 : 0;
    }
    __syncthreads();
    for (int e = t; e < nv; e += 64) gvec[e] = sv[e];
    if (fin) {
        for (int c = t; c < d; c += 64) a.x_out[(size_t)j * d + c] = x[c];
        if (t == 0) {
            active[j] = 0;
            const unsigned long long left = atomicAdd(reinterpret_cast<unsigned long long*>(&counters[0]), ~0ull) - 1ull;
            if (left == 0) counters[1] = 1;              // the GEMMs of later rounds exit on this word
        }
    } else {
        for (int c = t; c < d; c += 64) cand[c] = cd[c];
    }
}

// LDS of rl_step_kernel: the start's state + bounds + evaluation + trial point
__host__ inline size_t rl_step_lds(int d, int m) { return sizeof(double) * (rl_vec_doubles(d, m) + 4 * (size_t)d + 1); }

// starts still active when the host stops issuing rounds: their current iterate is the result
__global__ void rl_flush_kernel(RefineArgs a, int S, int* __restrict__ active, const double* __restrict__ vec, RlState* __restrict__ st) {
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= S || !active[j]) return;
    const double* x = vec + (size_t)j * rl_vec_doubles(a.d, a.history);
    for (int c = 0; c < a.d; ++c) a.x_out[(size_t)j * a.d + c] = x[c];
    a.f_out[j] = st[j].phase == 0 ? __longlong_as_double(0x7ff8000000000000ll) : st[j].f;      // never evaluated: NaN
    if (a.iters_out) { a.iters_out[2 * j] = st[j].it; a.iters_out[2 * j + 1] = st[j].nev; }
    active[j] = 0;
    st[j].phase = 2;
}

// k-chunk of the split-k form of a round's two GEMMs (0: the plain launch).  A round's products have ONE row of 128×128 output tiles
// per 128 starts and a k range of up to Np: 64 tiles at Np = 8192 — a quarter of the CUs, each running a 1 ms k loop through the
// small-launch path.  Cut into 16 k-chunks per tile the same product is ≈ 8·Np/128 workgroups of Np/16 k each on the LDS-tiled core.
static int lockstep_ksplit(int Np) {
    const char* e = getenv("ABO_REFINE_KSPLIT");                     // A/B runs: 0 = the plain launch
    if (e) { const int v = atoi(e); return v > 0 ? (v + 127) / 128 * 128 : 0; }
    if (Np < 2048) return 0;
    // the longest chunk for which the (tile, chunk) workgroups still fill the device's 512 slots (2 per CU): a launch that needs a
    // second round of workgroups pays a whole chunk for its last few
    const int Tn = Np / 128;
    for (int ks = 128;; ks += 128) {
        int wg = 0;
        for (int tj = 0; tj < Tn; ++tj) wg += ((tj + 1) * 128 + ks - 1) / ks;
        if (wg <= 512 || ks >= Np) return ks;
    }
}

size_t refine_lockstep_bytes(int S, int Np, int d, int history) {
    const size_t Sp = (size_t)pad_up(S, 128);
    const int ks = lockstep_ksplit(Np);
    const size_t nz = ks ? (size_t)(Np + ks - 1) / ks : 1;           // partial products of the split-k form (else the unused plain output)
    return sizeof(double) * (Sp * d + (4 + nz) * Sp * (size_t)Np + Sp * (d + 1) + (size_t)S * rl_vec_doubles(d, history)) + sizeof(RlState) * S +
           sizeof(int) * 2 * Sp + 64;
}

// work: refine_lockstep_bytes(…) of device memory.  Synchronous with respect to the stream only at the round-counter reads.
hipError_t launch_refine_lockstep(const RefineArgs& a, int S, void* work, hipStream_t s) {
    if (S <= 0) return hipSuccess;
    const int Sp = (int)pad_up(S, 128), Np = a.Np, d = a.d, m = a.history;
    char* w = static_cast<char*>(work);
    long long* counters = reinterpret_cast<long long*>(w); w += 64;
    int* nact = reinterpret_cast<int*>(&counters[2]);
    double* P = reinterpret_cast<double*>(w); w += sizeof(double) * (size_t)Sp * d;
    double* KX = reinterpret_cast<double*>(w); w += sizeof(double) * (size_t)Sp * Np;
    double* GV = reinterpret_cast<double*>(w); w += sizeof(double) * (size_t)Sp * Np;
    double* V = reinterpret_cast<double*>(w); w += sizeof(double) * (size_t)Sp * Np;
    double* U = reinterpret_cast<double*>(w); w += sizeof(double) * (size_t)Sp * Np;
    const int ks = lockstep_ksplit(Np);
    const int nz = ks ? (Np + ks - 1) / ks : 1;
    double* Ct = reinterpret_cast<double*>(w); w += sizeof(double) * (size_t)nz * Sp * Np;  // split-k partials / the plain GEMMs' untransposed outputs
    double* EV = reinterpret_cast<double*>(w); w += sizeof(double) * (size_t)Sp * (d + 1);
    double* vec = reinterpret_cast<double*>(w); w += sizeof(double) * (size_t)S * rl_vec_doubles(d, m);
    RlState* st = reinterpret_cast<RlState*>(w); w += sizeof(RlState) * S;
    int* active = reinterpret_cast<int*>(w); w += sizeof(int) * Sp;
    int* start_of = reinterpret_cast<int*>(w);
    hipError_t e;
    if ((e = hipMemsetAsync(counters, 0, 64, s)) != hipSuccess) return e;
    if ((e = hipMemsetAsync(active, 0, sizeof(int) * Sp, s)) != hipSuccess) return e;
    if ((e = hipMemsetAsync(P, 0, sizeof(double) * (size_t)Sp * d, s)) != hipSuccess) return e;
    hipLaunchKernelGGL(rl_step_kernel, dim3(S), dim3(64), rl_step_lds(d, m), s, a, S, 1, P, active, EV, vec, st, counters);
    hipLaunchKernelGGL(rl_compact_kernel, dim3(1), dim3(256), 0, s, active, S, start_of, nact);
    // every loop of the one-launch kernel is bounded by the same product (64-bit: the caller's limits are clamped in
    // api.hip: refine_defaults, but the product of two ints is not an int)
    // evaluations a start can take: its own, ≤ ls_max per accepted step, ≤ ls_max per failed search — and two failed searches never
    // follow each other (refine_kernel)
    const int64_t max_rounds = 1 + (2 * (int64_t)a.max_iter + 2) * (int64_t)a.ls_max;
    const size_t lds = sizeof(double) * ((size_t)a.dp + RW * 2 * RCH + 2 * RCH);
    long long left = -1;
    // what the host knows of the active count: S at first, then what the last counter read said (it only falls) — an upper bound that
    // sizes the round's launches (rows of the batch, 16-row groups of the split-k products); the kernels themselves go by *nact
    int bound = S;
    for (int64_t r = 0; r < max_rounds; ++r) {
        const int rows = bound < Sp ? bound : Sp;                 // compact rows that can be active this round
        const int R128 = (int)pad_up(rows, 128), R16 = (int)pad_up(rows, 16);
        dim3 kg((Np + 255) / 256, R16);
        switch (a.family) {
            case ABO_KERNEL_SE: hipLaunchKernelGGL((rl_kgen_kernel<ABO_KERNEL_SE>), kg, dim3(256), 0, s, a, P, start_of, nact, KX, GV); break;
            case ABO_KERNEL_MATERN52: hipLaunchKernelGGL((rl_kgen_kernel<ABO_KERNEL_MATERN52>), kg, dim3(256), 0, s, a, P, start_of, nact, KX, GV); break;
            case ABO_KERNEL_MATERN72: hipLaunchKernelGGL((rl_kgen_kernel<ABO_KERNEL_MATERN72>), kg, dim3(256), 0, s, a, P, start_of, nact, KX, GV); break;
            default: hipLaunchKernelGGL((rl_kgen_kernel<ABO_KERNEL_MATERN32>), kg, dim3(256), 0, s, a, P, start_of, nact, KX, GV); break;
        }
        if (ks) {
            // split-k: V[c][i] = Σ_{k ≤ i} KX[c][k]·W[i][k] as k-chunks per tile + a fixed-order sum of the partials; with at most 64
            // active rows the skinny kernel (gemm.hip: GemmArgs::mrows) streams L⁻¹ once instead of multiplying 128 rows
            GemmArgs g1{};
            g1.A = KX; g1.lda = Np; g1.B = a.W; g1.ldb = a.ld; g1.C = Ct; g1.ldc = Np; g1.sC = (int64_t)Sp * Np;
            g1.M = R128; g1.N = Np; g1.K = Np; g1.kmode = K_B_LOWER; g1.lower_only = 0; g1.batch = 1; g1.alpha = 1.0; g1.beta = 0.0;
            g1.ksplit = ks; g1.mrows = R16 <= 64 ? R16 : 0; g1.info = reinterpret_cast<const int64_t*>(&counters[1]);
            if ((e = launch_gemm_nt(g1, s)) != hipSuccess) return e;
            if ((e = launch_splitk_reduce(Ct, Np, g1.sC, nz, R16, Np, Np, ks, K_B_LOWER, V, Np, s)) != hipSuccess) return e;
            if (Np > a.N) hipLaunchKernelGGL(rl_zero_tail_kernel, dim3(((Np - a.N) * R16 + 255) / 256), dim3(256), 0, s, V, Np, a.N, R16);
            GemmArgs g2 = g1;    // U[c][i] = Σ_{k ≥ i} V[c][k]·WT[i][k]
            g2.A = V; g2.B = a.WT; g2.kmode = K_B_UPPER;
            if ((e = launch_gemm_nt(g2, s)) != hipSuccess) return e;
            if ((e = launch_splitk_reduce(Ct, Np, g2.sC, nz, R16, Np, Np, ks, K_B_UPPER, U, Np, s)) != hipSuccess) return e;
        } else {
            GemmArgs g1{};       // Ct1[i][c] = Σ_{k ≤ i} W[i][k]·KX[c][k]; transposed copy V[c][i]
            g1.A = a.W; g1.lda = a.ld; g1.B = KX; g1.ldb = Np; g1.C = Ct; g1.ldc = Sp; g1.Ct = V; g1.ldct = Np;
            g1.M = Np; g1.N = R128; g1.K = Np; g1.kmode = K_A_LOWER; g1.lower_only = 0; g1.batch = 1; g1.alpha = 1.0; g1.beta = 0.0;
            g1.info = reinterpret_cast<const int64_t*>(&counters[1]);
            if ((e = launch_gemm_nt(g1, s)) != hipSuccess) return e;
            if (Np > a.N) hipLaunchKernelGGL(rl_zero_tail_kernel, dim3(((Np - a.N) * R128 + 255) / 256), dim3(256), 0, s, V, Np, a.N, R128);
            GemmArgs g2 = g1;    // U[c][i] = Σ_{k ≥ i} WT[i][k]·V[c][k]
            g2.A = a.WT; g2.B = V; g2.Ct = U; g2.kmode = K_A_UPPER;
            if ((e = launch_gemm_nt(g2, s)) != hipSuccess) return e;
        }
        hipLaunchKernelGGL(rl_reduce_kernel, dim3(rows), dim3(RT), lds, s, a, P, start_of, nact, KX, GV, V, U, EV);
        hipLaunchKernelGGL(rl_step_kernel, dim3(S), dim3(64), rl_step_lds(d, m), s, a, S, 0, P, active, EV, vec, st, counters);
        hipLaunchKernelGGL(rl_compact_kernel, dim3(1), dim3(256), 0, s, active, S, start_of, nact);
        if ((r & 7) == 7 || r + 1 == max_rounds) {
            if ((e = hipMemcpyAsync(&left, counters, sizeof left, hipMemcpyDeviceToHost, s)) != hipSuccess) return e;
            if ((e = hipStreamSynchronize(s)) != hipSuccess) return e;
            if (left == 0) break;
            if (left < bound) bound = (int)left;
        }
    }
    // a start still active here has used up the round budget (cannot happen while every start stops after max_iter accepted steps of
    // ≤ ls_max trials each, but the outputs must never be left unwritten): publish its current iterate
    if (left != 0) hipLaunchKernelGGL(rl_flush_kernel, dim3((S + 127) / 128), dim3(128), 0, s, a, S, active, vec, st);
    return hipGetLastError();
}

// ---- gradient-enhanced models and objectives with a GRADNORM_UCB term: the lockstep state machine over a caller-supplied evaluation ----
// (abo_kernels.h: launch_refine_lockstep_grad).  A round: rl_stencil_kernel (the pending points, plus their 2d central-difference
// neighbours when the objective has a GRADNORM_UCB term) → ev.eval (all-output posterior mean[p] and covariance[p][p] of every
// point: the kernels behind abo_predict_grad_cov) → rl_grad_ev_kernel (f, ∇f per start) → rl_step_kernel (the same state machine
// as the StandardGP variants).
//   function-value terms:  μ = m₀, σ² = Σ₀₀, ∇μ = E[∇f] − m_∇ (the posterior mean of the gradient outputs IS the gradient of the
//                          posterior mean), ∇σ² = 2·Cov(f, ∇f) (row 0 of the block: k(x,x) is constant and the prior covariance
//                          of f(x) with ∇f(x) at the same point vanishes for a stationary kernel)
//   GRADNORM_UCB terms:    central differences of the term's value over x ± h·e_c, h = ∛ε·max(1, |x_c|) — the reference
//                          differentiates every objective by finite differences of M = 1 calls (acq_utils.jl:55-71); an analytic
//                          form would need third derivatives of the kernel
__global__ void rl_stencil_kernel(const double* __restrict__ P, int S, int d, int npp, double* __restrict__ PTS, double* __restrict__ H) {
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= S * d) return;
    const int j = e / d, c = e % d;
    const double* x = P + (size_t)j * d;
    const double h = 6.0554544523933429e-06 * fmax(1.0, fabs(x[c]));          // cbrt(2^-52)
    const double xp = x[c] + h, xm = x[c] - h;
    H[e] = xp - xm;
    double* base = PTS + (size_t)j * npp * d;
    base[c] = x[c];
    for (int q = 0; q < d; ++q) {                       // thread (j, c) writes column c of every stencil point of start j
        base[(size_t)(1 + 2 * q) * d + c] = q == c ? xp : x[c];
        base[(size_t)(2 + 2 * q) * d + c] = q == c ? xm : x[c];
    }
}

// one thread per start: EV[j] = {f, ∇f[0..d)}; f_out / g_out (optional): the same for the test hook
__global__ void rl_grad_ev_kernel(RefineArgs a, const double* __restrict__ mean_g, const double* __restrict__ MU,
                                  const double* __restrict__ COV, const double* __restrict__ H, int S, int npp,
                                  const int* __restrict__ active, double* __restrict__ EV, double* __restrict__ f_out,
                                  double* __restrict__ g_out) {
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= S || (active && !active[j])) return;
    const int d = a.d, p = d + 1;
    const double* m = MU + (size_t)j * npp * p;
    const double* C = COV + (size_t)j * npp * p * p;
    double dmu, dvar;
    double f = terms_value_and_partials(a.terms, m[0], C[0], dmu, dvar);
    const bool st = npp > 1;
    if (st) f += terms_gradnorm(a.terms, m, C, p);
    double* ev = EV ? EV + (size_t)j * (d + 1) : nullptr;
    for (int c = 0; c < d; ++c) {
        double g = dmu * (m[1 + c] - mean_g[c]) + dvar * (2.0 * C[1 + c]);
        if (st) {
            const double* mp = m + (size_t)(1 + 2 * c) * p;
            const double* Cp = C + (size_t)(1 + 2 * c) * p * p;
            const double fp = terms_gradnorm(a.terms, mp, Cp, p), fm = terms_gradnorm(a.terms, mp + p, Cp + p * p, p);
            g += (fp - fm) / H[(size_t)j * d + c];
        }
        if (ev) ev[1 + c] = g;
        if (g_out) g_out[(size_t)j * d + c] = g;
    }
    if (ev) ev[0] = f;
    if (f_out) f_out[j] = f;
}

size_t refine_lockstep_grad_bytes(int S, int d, int history, bool stencil) {
    const size_t npp = stencil ? 2 * (size_t)d + 1 : 1, p = (size_t)d + 1;
    return sizeof(double) * ((size_t)S * d * 2 + (size_t)S * npp * (d + p + p * p) + (size_t)S * (d + 1) +
                             (size_t)S * rl_vec_doubles(d, history)) + sizeof(RlState) * S + (sizeof(int) * S + 63) / 64 * 64 + 64 + 64;
    // (the int array is the last carve: rounded up to 64 bytes, so that whatever the caller places behind this block — api.hip puts
    // the prior means of the gradient outputs there, read with 8-byte loads — is aligned for any number of starts)
}

namespace {
struct GradWork {
    long long* counters;
    double *P, *PTS, *H, *MU, *COV, *EV, *vec;
    RlState* st;
    int* active;
    int npp;
};
GradWork carve_grad_work(void* work, int S, int d, int m, bool stencil) {
    GradWork w{};
    const size_t npp = stencil ? 2 * (size_t)d + 1 : 1, p = (size_t)d + 1;
    char* c = static_cast<char*>(work);
    w.counters = reinterpret_cast<long long*>(c); c += 64;
    w.P = reinterpret_cast<double*>(c); c += sizeof(double) * (size_t)S * d;
    w.H = reinterpret_cast<double*>(c); c += sizeof(double) * (size_t)S * d;
    w.PTS = reinterpret_cast<double*>(c); c += sizeof(double) * (size_t)S * npp * d;
    w.MU = reinterpret_cast<double*>(c); c += sizeof(double) * (size_t)S * npp * p;
    w.COV = reinterpret_cast<double*>(c); c += sizeof(double) * (size_t)S * npp * p * p;
    w.EV = reinterpret_cast<double*>(c); c += sizeof(double) * (size_t)S * (d + 1);
    w.vec = reinterpret_cast<double*>(c); c += sizeof(double) * (size_t)S * rl_vec_doubles(d, m);
    w.st = reinterpret_cast<RlState*>(c); c += sizeof(RlState) * S;
    w.active = reinterpret_cast<int*>(c);
    w.npp = (int)npp;
    if (!stencil) w.PTS = w.P;                           // one point per start: evaluated in place
    return w;
}
}  // namespace

hipError_t launch_refine_lockstep_grad(const RefineArgs& a, int S, const double* mean_g, void* work, const GradEval& ev, hipStream_t s) {
    if (S <= 0) return hipSuccess;
    const int d = a.d, m = a.history;
    const bool stencil = terms_have_gradnorm(a.terms);
    GradWork w = carve_grad_work(work, S, d, m, stencil);
    hipError_t e;
    if ((e = hipMemsetAsync(w.counters, 0, 64, s)) != hipSuccess) return e;
    if ((e = hipMemsetAsync(w.active, 0, sizeof(int) * S, s)) != hipSuccess) return e;
    if ((e = hipMemsetAsync(w.P, 0, sizeof(double) * (size_t)S * d, s)) != hipSuccess) return e;
    hipLaunchKernelGGL(rl_step_kernel, dim3(S), dim3(64), rl_step_lds(d, m), s, a, S, 1, w.P, w.active, w.EV, w.vec, w.st, w.counters);
    // evaluations a start can take: its own, ≤ ls_max per accepted step, ≤ ls_max per failed search — and two failed searches never
    // follow each other (refine_kernel)
    const int64_t max_rounds = 1 + (2 * (int64_t)a.max_iter + 2) * (int64_t)a.ls_max;
    long long left = -1;
    for (int64_t r = 0; r < max_rounds; ++r) {
        if (stencil) hipLaunchKernelGGL(rl_stencil_kernel, dim3((S * d + 127) / 128), dim3(128), 0, s, w.P, S, d, w.npp, w.PTS, w.H);
        if ((e = ev.eval(ev.ctx, w.PTS, S * w.npp, w.MU, w.COV, s)) != hipSuccess) return e;
        hipLaunchKernelGGL(rl_grad_ev_kernel, dim3((S + 63) / 64), dim3(64), 0, s, a, mean_g, w.MU, w.COV, w.H, S, w.npp, w.active, w.EV,
                           (double*)nullptr, (double*)nullptr);
        hipLaunchKernelGGL(rl_step_kernel, dim3(S), dim3(64), rl_step_lds(d, m), s, a, S, 0, w.P, w.active, w.EV, w.vec, w.st, w.counters);
        if ((r & 7) == 7 || r + 1 == max_rounds) {
            if ((e = hipMemcpyAsync(&left, w.counters, sizeof left, hipMemcpyDeviceToHost, s)) != hipSuccess) return e;
            if ((e = hipStreamSynchronize(s)) != hipSuccess) return e;
            if (left == 0) break;
        }
    }
    if (left != 0) hipLaunchKernelGGL(rl_flush_kernel, dim3((S + 127) / 128), dim3(128), 0, s, a, S, w.active, w.vec, w.st);
    return hipGetLastError();
}

hipError_t launch_acq_grad_via_eval(const RefineArgs& a, int S, const double* mean_g, void* work, const GradEval& ev, hipStream_t s) {
    if (S <= 0) return hipSuccess;
    const int d = a.d;
    const bool stencil = terms_have_gradnorm(a.terms);
    GradWork w = carve_grad_work(work, S, d, a.history, stencil);
    hipError_t e;
    if ((e = hipMemcpyAsync(w.P, a.starts, sizeof(double) * (size_t)S * d, hipMemcpyDeviceToDevice, s)) != hipSuccess) return e;
    if (stencil) hipLaunchKernelGGL(rl_stencil_kernel, dim3((S * d + 127) / 128), dim3(128), 0, s, w.P, S, d, w.npp, w.PTS, w.H);
    if ((e = ev.eval(ev.ctx, w.PTS, S * w.npp, w.MU, w.COV, s)) != hipSuccess) return e;
    hipLaunchKernelGGL(rl_grad_ev_kernel, dim3((S + 63) / 64), dim3(64), 0, s, a, mean_g, w.MU, w.COV, w.H, S, w.npp, (const int*)nullptr,
                       (double*)nullptr, a.f_out, a.x_out);
    return hipGetLastError();
}

template <int FAM>
static hipError_t launch_refine_fam(const RefineArgs& a, int S, bool grad_only, hipStream_t s) {
    if (grad_only) {
        const size_t lds = sizeof(double) * ((size_t)2 * a.d + a.dp + RW * 2 * RCH + 2 * RCH + 4);
        hipLaunchKernelGGL((acq_grad_kernel<FAM>), dim3(S), dim3(RT), lds, s, a);
    } else {
        hipLaunchKernelGGL((refine_kernel<FAM>), dim3(S), dim3(RT), refine_lds_bytes(a.d, a.dp, a.history), s, a);
    }
    return hipGetLastError();
}

hipError_t launch_refine(const RefineArgs& a, int S, int grad_only, hipStream_t s) {
    if (S <= 0) return hipSuccess;
    switch (a.family) {
        case ABO_KERNEL_SE: return launch_refine_fam<ABO_KERNEL_SE>(a, S, grad_only != 0, s);
        case ABO_KERNEL_MATERN52: return launch_refine_fam<ABO_KERNEL_MATERN52>(a, S, grad_only != 0, s);
        case ABO_KERNEL_MATERN72: return launch_refine_fam<ABO_KERNEL_MATERN72>(a, S, grad_only != 0, s);
        case ABO_KERNEL_MATERN32: return launch_refine_fam<ABO_KERNEL_MATERN32>(a, S, grad_only != 0, s);
        default: return hipErrorInvalidValue;
    }
}

}  // namespace abo
