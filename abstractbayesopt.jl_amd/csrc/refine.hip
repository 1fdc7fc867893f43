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
// starts walk in step: microseconds at the sizes the reference's loop lives at (N ≤ 10³), ≈ 10 ms at N = 8192 where 100 starts
// re-stream 512 MB each (a batched-GEMM evaluation would amortise that; not built — DESIGN.md §7).
#include "abo_kappa.h"
#include "abo_kernels.h"
#include "../../include/abo_hip.h"

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

__device__ __forceinline__ double r_norm_cdf(double z) { return 0.5 * erfc(-z * 0.70710678118654752440084436210485); }
__device__ __forceinline__ double r_norm_pdf(double z) { return exp(-0.5 * z * z) * 0.39894228040143267793994605993438; }

// acquisition value (the arithmetic of misc.hip: acq_score) and its partial derivatives with respect to μ and σ²
__device__ __forceinline__ double acq_value_and_partials(int kind, double mu, double var, double p0, double best_y, double& dmu,
                                                         double& dvar) {
    if (kind == ABO_ACQ_UCB) {
        const double sg = sqrt(fmax(var, 0.0));
        dmu = -1.0; dvar = var > 0.0 ? 0.5 * p0 / sg : 0.0;
        return -mu + p0 * sg;
    }
    if (kind == ABO_ACQ_MEAN) { dmu = -1.0; dvar = 0.0; return -mu; }
    const double delta = (best_y - p0) - mu;
    if (var <= 1e-12) { dmu = delta > 0.0 ? -1.0 : 0.0; dvar = 0.0; return fmax(delta, 0.0); }
    const double sg = sqrt(var), z = delta / sg, cdf = r_norm_cdf(z), pdf = r_norm_pdf(z);
    if (kind == ABO_ACQ_EI) { dmu = -cdf; dvar = 0.5 * pdf / sg; return delta * cdf + sg * pdf; }
    dmu = -pdf / sg; dvar = -0.5 * pdf * z / var;                 // PI = Φ(z)
    return cdf;
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
    const double f = acq_value_and_partials(a.kind, mu, var, a.p0, a.best_y, dmu, dvar);
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

// LDS layout (doubles): x, g, pg, p, xprev, pgprev, cand, gc, lo, up [d each]; xs [dp]; Sh, Yh [m·d each]; rho, al [m each];
// red [RW·2·RCH]; out [2·RCH]; fres [4]; ctrl [4 ints → 2 doubles]
size_t refine_lds_bytes(int d, int dp, int m) {
    return sizeof(double) * ((size_t)10 * d + dp + (size_t)2 * m * d + 2 * m + RW * 2 * RCH + 2 * RCH + 4 + 2);
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
    double* xs = up + d;
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
    }
    __syncthreads();
    eval_point<FAM>(a, x, xs, scr, red, out, fres, g);
    double f = fres[0];
    int nev = 1, it = 0, nh = 0;
    bool have_prev = false;
    double tstep = 1.0;
    if (f == f && fabs(f) < 1.0e300) {                                   // a non-finite start value: nothing to refine
        for (it = 0; it < a.max_iter; ++it) {
            if (t == 0) {
                // projected gradient: components that push against an active bound are dropped (maximisation)
                double gmax = 0.0;
                bool bad = false;
                for (int c = 0; c < d; ++c) {
                    double v = g[c];
                    if ((x[c] <= lo[c] && v < 0.0) || (x[c] >= up[c] && v > 0.0)) v = 0.0;
                    pg[c] = v;
                    bad = bad || !(v == v);
                    gmax = fmax(gmax, fabs(v));
                }
                if (have_prev && m > 0) {                                // curvature pair of the accepted step: s = Δx, y = −Δ(pg)
                    if (nh == m) {
                        for (int h = 1; h < m; ++h)
                            for (int c = 0; c < d; ++c) { Sh[(h - 1) * d + c] = Sh[h * d + c]; Yh[(h - 1) * d + c] = Yh[h * d + c]; }
                        nh = m - 1;
                    }
                    for (int c = 0; c < d; ++c) { Sh[nh * d + c] = x[c] - xprev[c]; Yh[nh * d + c] = pgprev[c] - pg[c]; }
                    ++nh;
                }
                const int stop = (bad || gmax <= a.g_tol) ? 1 : 0;
                if (!stop) {
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
                    fres[3] = nh == 0 ? fmin(1.0, 0.1 * wmin / fmax(pmax, 1e-300)) : 1.0;
                }
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
            if (!accepted) break;
            if (t == 0) {
                double dx = 0.0;
                for (int c = 0; c < d; ++c) dx = fmax(dx, fabs(cand[c] - x[c]));
                ctrl[2] = (dx <= a.x_abstol || fabs(fc - f) <= a.f_abstol) ? 1 : 0;
                for (int c = 0; c < d; ++c) { xprev[c] = x[c]; pgprev[c] = pg[c]; x[c] = cand[c]; g[c] = gc[c]; }
            }
            __syncthreads();
            f = fc;
            have_prev = true;
            if (ctrl[2]) { ++it; break; }
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
