// Serial pieces of the factorisation: the 128×128 diagonal-block Cholesky + triangular inverse,
// the triangular mat-vecs for alpha, and the NLML reductions.
//
// Reference arithmetic being replaced: [upstream AbstractGPs] posterior(FiniteGP, y) —
// C = cholesky(K + σ²I) (LAPACK potrf), α = C \ (y − m)  — called from update(),
// src/surrogates/StandardGP.jl:79-83; −logpdf at :99-114.  The blocked driver lives in api.hip:
// right-looking, 128-wide panels; this kernel is its potf2 + trtri step, the MFMA GEMM core
// (gemm.hip) does the panel solve, the trailing SYRK and the blocked L⁻¹.
#include "abo_kernels.h"
#include "abo_kappa.h"
#include <cstdlib>

// tools/chol_diag_probe.hip defines ABO_CHOL_PROBE to time the phases of chol_diag_kernel with the 100 MHz
// constant clock; in the library build the macro is empty.
#ifdef ABO_CHOL_PROBE
__device__ long long abo_probe_clk[16];
#define PROBE(i) do { if (threadIdx.x == 0) abo_probe_clk[i] = wall_clock64(); } while (0)
#else
#define PROBE(i) do { } while (0)
#endif

namespace abo {

typedef double d4_t __attribute__((ext_vector_type(4)));

constexpr int NB = 128;
constexpr int LDA = NB + 1;   // odd stride: column walks spread over the banks
constexpr int SB = 16;        // sub-block edge = MFMA tile edge
constexpr int NSB = NB / SB;  // 8

// in-wave ordering point for LDS traffic between lanes of ONE wave: the LDS queue of a wave is
// in-order, so a drained counter plus a compiler fence is all that is needed (no s_barrier)
__device__ __forceinline__ void wave_lds_sync() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_s_waitcnt(0xc07f);   // lgkmcnt(0)
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// wave-uniform broadcast of lane l's double (l a compile-time constant after unrolling): two v_readlane_b32
__device__ __forceinline__ double readlane_f64(double v, int l) {
    const int lo = __builtin_amdgcn_readlane(__double2loint(v), l);
    const int hi = __builtin_amdgcn_readlane(__double2hiint(v), l);
    return __hiloint2double(hi, lo);
}

// The register step of a 16-column sub-step (phase 1, (1)+(2) below), shared by every build of the diagonal-block kernel: lanes 0-15
// of the wave hold the 16 rows of the diagonal sub-block, lanes 16-63 rows below it, one row (16 doubles) per lane, in x[].
// The step is a LATENCY chain, not an issue-slot problem (round 6 measured it: a third fewer VALU instructions per column — pivot
// reciprocal only, no selects — left the step at its 2.1 µs; tools/chol_diag_probe): per column the dependent path is
// v_readlane → v_rsq_f64 → seven dependent fp64 operations → scale → v_readlane → fma → v_readlane, ≈ 275 cycles.  So — parity is the
// oracle's tolerance since round 6, no longer round 4's bits — the columns are taken TWO at a time, in closed form:
//     d0 = A_jj,  b = A_{j+1,j},  a = A_{j+1,j+1}:     det = a·d0 − b²      (the second pivot is d1 = det/d0)
//     r0 = 1/√d0 and rD = 1/√det from v_rsq_f64 seeds, two coupled Goldschmidt steps each, INTERLEAVED (one chain's latency)
//     1/L_jj = r0,   L_{j+1,j} = b·r0,   1/L_{j+1,j+1} = √d0/√det = (d0·r0)·rD
//     column j ← column j · r0;   column j+1 ← (column j+1 − column j · L_{j+1,j}) · (d0·r0)·rD
// — one dependent chain per PAIR of columns (≈ 14 operations, one v_rsq latency, three v_readlane round trips) where two columns took
// two (≈ 22, two, four).  det = a·d0 − b² cancels exactly as a − (b/√d0)² does (relative error ε·a·d0/det against ε·a/d1: the
// same).  The diagonal entries are L_jj = d0·r0 and L_{j+1,j+1} = d1·r1 like every other entry of their columns (≤ 2 ulp from the
// square roots; everything downstream divides by multiplying with the stored reciprocals, so factor and inverse stay consistent).
// What a finished pair owes the columns beyond the next pair is issued as filler behind the steps of the NEXT pair's chain (the
// scheduling fences pin that order), with the pair's entries L[k][j], L[k][j+1] read back as broadcasts from a 2 × 64-double LDS
// buffer of the wave (the LDS queue of a wave is in order: no synchronisation); only the updates of the next pair's two columns —
// the next chain waits for them — take their multipliers by v_readlane.  No per-column selects: the reciprocals go to LDS from one
// lane, the pivot tests are compares on wave-uniform arguments accumulated on the scalar unit.
// cb: 2 × 64 doubles of LDS per wave; dinv_out: 16 doubles of LDS that receive 1/L_jj (the wave that owns the block's dinv passes it,
// the others a scratch row).  Returns the mask of columns whose pivot was not a positive normal number (bit j; wave-uniform; NaN
// included; LAPACK's info is its lowest set bit).
typedef __attribute__((address_space(3))) double lds_f64;
__device__ __forceinline__ unsigned register_potf2_step_lds(double (&x)[16], int lane, lds_f64* cb, lds_f64* dinv_out) {
    constexpr int SBX = 16;
    unsigned bad = 0;
#ifdef ABO_STEP_EMPTY                                   /* tools/chol_diag_probe: what a sub-step costs WITHOUT its register step */
    return 0u;
#endif
    const unsigned dinv_addr = (unsigned)(size_t)dinv_out;         // LDS byte address (wave-uniform)
    double d0 = readlane_f64(x[0], 0), b = readlane_f64(x[0], 1), a = readlane_f64(x[1], 1);
    double w0[SBX], w1[SBX];                           // L[k][j−2], L[k][j−1]: read from the buffer at the END of the pair before (below)
#pragma unroll
    for (int j = 0; j < SBX; j += 2) {
        const int j2 = j >= 2 ? j - 2 : 0, j1 = j >= 2 ? j - 1 : 0;         // the finished pair whose owed updates fill this chain
        int ka = j + 2, kb = j + 2;
#define ABO_FILL() do { if (j > 0 && ka < SBX) { x[ka] = fma(-x[j2], w0[ka], x[ka]); ++ka; } \
                        if (j > 0 && kb < SBX) { x[kb] = fma(-x[j1], w1[kb], x[kb]); ++kb; } \
                        __builtin_amdgcn_sched_barrier(0); } while (0)
        const double t = b * b;
        bad |= __ballot(d0 >= 2.3e-308) == 0ull ? (1u << j) : 0u;         // one v_cmp on the uniform argument; the rest is scalar
        const double y0 = __builtin_amdgcn_rsq(d0);
        __builtin_amdgcn_sched_barrier(0);
        const double det = fma(a, d0, -t);
        double g0 = d0 * y0, h0 = 0.5 * y0;
        ABO_FILL();
        bad |= __ballot(det >= 2.3e-308) == 0ull ? (2u << j) : 0u;
        const double yD = __builtin_amdgcn_rsq(det);
        double r0 = fma(-h0, g0, 0.5);
        ABO_FILL();                                    // (the LDS reads are back about here)
        double gD = det * yD, hD = 0.5 * yD;
        g0 = fma(g0, r0, g0);
        h0 = fma(h0, r0, h0);
        ABO_FILL();
        double rD = fma(-hD, gD, 0.5);
        r0 = fma(-h0, g0, 0.5);
        ABO_FILL();
        gD = fma(gD, rD, gD);
        hD = fma(hD, rD, hD);
        h0 = fma(h0, r0, h0);
        ABO_FILL();
        rD = fma(-hD, gD, 0.5);
        const double rp0 = h0 + h0;
        ABO_FILL();
        hD = fma(hD, rD, hD);
        const double m = d0 * rp0, l10 = b * rp0;
        x[j] = x[j] * rp0;                             // column j, its diagonal entry included
        ABO_FILL();
        const double rdet = hD + hD;
        double tmp = fma(-x[j], l10, x[j + 1]);
        ABO_FILL();
        const double rp1 = m * rdet;
        ABO_FILL();
        x[j + 1] = tmp * rp1;                          // column j + 1
        // the two reciprocals to LDS from lane 0 alone, without a branch (a conditional store puts a control-flow region into every
        // chain and the scheduler's order with it: measured 2.4 → 5.4 µs a step): the wave runs this function with all lanes enabled
        asm volatile("s_mov_b64 exec, 1\n\tds_write_b64 %0, %1\n\tds_write_b64 %0, %2 offset:8\n\ts_mov_b64 exec, -1"
                     :: "v"(dinv_addr + 8u * j), "v"(rp0), "v"(rp1) : "memory");
        ABO_FILL();
#pragma unroll
        for (int rest = 0; rest < SBX; ++rest) ABO_FILL();             // whatever the previous pair still owes
        if (j + 2 < SBX) {
            if (j + 4 < SBX) {
                // the pair goes to the wave's buffer, and the broadcasts the NEXT chain's fillers multiply with are read back at once:
                // they have a whole chain prologue to arrive in (read at the top of the chain that uses them, the first filler's
                // s_waitcnt stalled the wave — and the pivot chain with it — for the LDS round trip, once per pair)
                cb[lane] = x[j];
                cb[64 + lane] = x[j + 1];
#pragma unroll
                for (int k = j + 4; k < SBX; ++k) { w0[k] = cb[k]; w1[k] = cb[64 + k]; }
            }
            __builtin_amdgcn_sched_barrier(0);
            const double l20 = readlane_f64(x[j], j + 2), l30 = readlane_f64(x[j], j + 3);
            const double l21 = readlane_f64(x[j + 1], j + 2), l31 = readlane_f64(x[j + 1], j + 3);
            __builtin_amdgcn_sched_barrier(0);
            x[j + 2] = fma(-x[j], l20, x[j + 2]);
            x[j + 3] = fma(-x[j], l30, x[j + 3]);
            __builtin_amdgcn_sched_barrier(0);
            x[j + 2] = fma(-x[j + 1], l21, x[j + 2]);
            x[j + 3] = fma(-x[j + 1], l31, x[j + 3]);
            __builtin_amdgcn_sched_barrier(0);
            d0 = readlane_f64(x[j + 2], j + 2);
            b = readlane_f64(x[j + 2], j + 3);
            a = readlane_f64(x[j + 3], j + 3);
        }
#undef ABO_FILL
        __builtin_amdgcn_sched_barrier(0);
    }
    return bad;
}

// One workgroup (16 waves), the 128×128 block resident in LDS (129 KB of the CU's 160 KB), worked on
// as an 8×8 grid of 16×16 sub-blocks so that everything but the 16×16 diagonal factorisations runs
// on the fp64 MFMA (v_mfma_f64_16x16x4_f64) with 16 + 8 workgroup barriers in total:
//   phase 1  right-looking Cholesky over sub-block columns p = 0..7:
//            (1)+(2) the 16×16 diagonal sub-block is factored and the rows below it are solved (x·L_ppᵀ = a) in
//                registers, one row per lane, column values broadcast with v_readlane (pivot check on the way)
//            (3) trailing sub-blocks C_ij −= L_ip·L_jpᵀ, one MFMA chain per sub-block, spread over the waves
//   phase 2  X = L⁻¹ in place: column n of X_cc is written into ROW n of the (free) strict upper triangle,
//            i.e. the upper triangle ends up holding Xᵀ, reciprocals of the diagonal in dinv[]:
//            (a) 16×16 diagonal inverses, one thread per column;
//            (b) off-diagonal sub-blocks by distance δ = i−c: X_ic = −X_ii·(Σ_{c≤k<i} L_ik·X_kc); the inner sum
//                comes out of the MFMA in C/D layout, which is exactly the B-operand layout of the next
//                MFMA when its k-steps are taken as {g, g+4, g+8, g+12} — no data movement in between
//   phase 3  write L back to K (lower, zeros above), X to W (lower) and Xᵀ to WT (upper).
// The kernel comes in three builds of the same phases (MODE):
//   0  everything (factor + full 128×128 inverse) of ONE block: a model of a single 128-row block that mode 3 does not serve
//   2  trtri, batched — one workgroup per diagonal block (blockIdx.x): reads the finished L block back, runs phase 2 and
//      writes W / WT.  ONE launch of N/128 workgroups after the factorisation instead of N/128 serial 12 µs phases (the panel
//      chain's own diagonal-block kernel is potf2_pipe_kernel below: the 128×128 inverse is not needed on the chain).
//   3  the WHOLE fit of a model with N ≤ 128 points (the reference's own regime: its examples and tests run 5 … 50
//      points, src/acquisition_functions/acq_utils.jl:37 scores 10 000 candidates per step) in this one launch: scaled
//      inputs, K + σ²I generated straight into LDS (same arithmetic as kgen_kernel, same bits), factor, inverse,
//      δ = y − m, α = L⁻ᵀ(L⁻¹δ), log-determinant and δᵀα.  Replaces thirteen launches (two memsets, point scaling, centring,
//      kernel matrix, diagonal fix, this kernel in mode 0, two triangular mat-vecs, the NLML terms) whose launch
//      boundaries were most of a 90 µs fit.
#define AA(r, c) a[(r) * LDA + (c)]
constexpr int DT = 1024;      // threads of the diagonal-block kernel (16 waves: latency hiding for the LDS phases)
template <int FAM>
__device__ __forceinline__ double small_kernel_entry(const double* xi, const double* xj, int dp, double sigma_f2) {
    double r = 0.0;
    for (int c = 0; c < dp; ++c) {
        const double e = xj[c] - xi[c];
        r = fma(e, e, r);
    }
    return sigma_f2 * kappa_eval<FAM>(r);
}

template <int MODE>
__global__ void __launch_bounds__(DT) chol_diag_kernel(double* K, double* W, double* WT, int64_t ld, int r0,
                                                        int64_t* info, FitSmallArgs fs) {
    __shared__ double a[NB * LDA];
    __shared__ double dinv[NB];
    __shared__ __attribute__((aligned(16))) double colbuf[3][128];    // per register-step wave: the pair of columns just finished
    __shared__ int fail;
    __shared__ double xs[MODE == 3 ? NB * 16 : 1];          // mode 3: scaled inputs, δ and the intermediate of the two solves
    __shared__ double dl[MODE == 3 ? NB : 1];
    __shared__ double tv[MODE == 3 ? NB : 1];
    const int t = threadIdx.x;
    const int lane = t & 63, wave = t >> 6;
    const int r16 = lane & 15, g = lane >> 4;
    if constexpr (MODE == 3) {
        if (t == 0) *info = 0;       // the one launch of the fit: nothing ran before it that could have failed (a failing pivot writes behind a barrier)
    } else {
        if (*info != 0) return;
    }
    if (t == 0) fail = 0;
    if constexpr (MODE == 2) r0 = blockIdx.x * NB;
    // mode 3 (the whole fit of an N ≤ 128 model): the block beyond the last sub-block that holds a training point is the identity —
    // factor, inverse and every product with it are known in advance — so the sub-block loops stop at nsb = ⌈N/16⌉ (the
    // reference's loops run 5 … 50 points: 1 … 4 of the 8 sub-blocks).  Same bits as the full sweep: the skipped steps only ever
    // multiplied by exact zeros and ones.
    const int nsb = MODE == 3 ? (fs.N + SB - 1) / SB : NSB;
    const int NBe = SB * nsb;                              // rows / columns that take part
    PROBE(0);
    double* Kb = K + (int64_t)r0 * ld + r0;
    if constexpr (MODE == 3) {
        // scaled, zero-padded inputs (also the model's Xs) and centred targets
        for (int idx = t; idx < NB * fs.dp; idx += DT) {
            const int i = idx / fs.dp, c = idx % fs.dp;
            const bool real = i < fs.N && c < fs.d;
            const double raw = real ? fs.Xraw[(int64_t)i * fs.d + c] : 0.0;
            if (real && fs.Xkeep) fs.Xkeep[(int64_t)i * fs.d + c] = raw;      // the model's own copy of the caller's inputs
            const double v = raw * fs.s;
            xs[i * 16 + c] = v;
            fs.Xs[idx] = v;
        }
        if (t < NB) {
            const double yv = t < fs.N ? fs.y[t] : 0.0;
            if (t < fs.N && fs.ykeep) fs.ykeep[t] = yv;
            const double v = t < fs.N ? yv - fs.mean_c : 0.0;
            dl[t] = v;
            fs.delta[t] = v;
        }
        __syncthreads();
        // K + σ²I with identity padding: element (i, j) exactly as kgen_kernel + diag_fix_kernel give it
        for (int idx = t; idx < NB * NB; idx += DT) {
            const int i = idx >> 7, j = idx & 127;
            double v = i == j ? 1.0 : 0.0;
            if (i < fs.N && j < fs.N) {
                const double* xi = xs + i * 16;
                const double* xj = xs + j * 16;
                switch (fs.family) {
                    case ABO_KERNEL_SE: v = small_kernel_entry<ABO_KERNEL_SE>(xi, xj, fs.dp, fs.sigma_f2); break;
                    case ABO_KERNEL_MATERN52: v = small_kernel_entry<ABO_KERNEL_MATERN52>(xi, xj, fs.dp, fs.sigma_f2); break;
                    case ABO_KERNEL_MATERN72: v = small_kernel_entry<ABO_KERNEL_MATERN72>(xi, xj, fs.dp, fs.sigma_f2); break;
                    default: v = small_kernel_entry<ABO_KERNEL_MATERN32>(xi, xj, fs.dp, fs.sigma_f2); break;
                }
                if (i == j) v += fs.noise;
            }
            AA(i, j) = v;
        }
    } else {
        for (int idx = t; idx < NB * NB; idx += DT) {
            const int i = idx >> 7, j = idx & 127;
            AA(i, j) = Kb[(int64_t)i * ld + j];
        }
    }
    if constexpr (MODE == 2) {
        __syncthreads();
        if (t < NB) dinv[t] = 1.0 / AA(t, t);              // the factor's diagonal is positive (a failed fit never gets here)
    }
    __syncthreads();
    PROBE(1);

    // ---------------- phase 1: Cholesky ----------------
    for (int p = 0; p < (MODE == 2 ? 0 : nsb); ++p) {
        const int o = SB * p;
        const int below = NBe - o - SB;                    // rows under the diagonal sub-block
        // (1)+(2) fused, in registers: lanes 0-15 of a wave hold the 16 rows of the diagonal sub-block, lanes 16-63
        // hold 48 of the rows below it, one row (16 doubles) per lane.  Column j: the pivot and the scaled column
        // entries L[k][j] are wave-uniform v_readlane broadcasts from lanes 0-15, so factoring the diagonal block and
        // solving the rows below it (x·L_ppᵀ = a) are the same instruction stream.  Each wave that owns rows below
        // redoes the diagonal block itself (identical arithmetic), so there is no LDS traffic and no cross-wave
        // synchronisation inside the 16 columns.
        if (wave * 48 < below || wave == 0) {
            const int row = lane < SB ? o + lane : o + SB + wave * 48 + (lane - SB);
            const bool valid = row < NBe;
            double x[SB];
#pragma unroll
            for (int c = 0; c < SB; ++c) x[c] = valid ? AA(row, o + c) : 0.0;
            const unsigned badc = register_potf2_step_lds(x, lane, (lds_f64*)colbuf[wave], wave == 0 ? (lds_f64*)&dinv[o] : (lds_f64*)&colbuf[wave][32]);
            if (badc) {
                if (t == 0) { fail = 1; *info = (int64_t)r0 + o + __builtin_ctz(badc) + 1; }
            } else if (valid) {
                if (lane >= SB) {
#pragma unroll
                    for (int c = 0; c < SB; ++c) AA(row, o + c) = x[c];
                } else if (wave == 0) {
#pragma unroll
                    for (int c = 0; c < SB; ++c)
                        if (c <= lane) AA(row, o + c) = x[c];      // strict upper part of the registers is scratch
                }
            }
        }
        __syncthreads();
        if (p < 4) PROBE(6 + 2 * p);                       // probe build only: end of the register potf2 + row solve of sub-step p
        if (fail) return;                                  // uniform (LDS flag after the barrier)
        {                                                  // (3) trailing update on the lower sub-blocks
            const int nb = nsb - 1 - p;                    // sub-block rows/cols left
            const int total = nb * (nb + 1) / 2;
            for (int e = wave; e < total; e += DT / 64) {
                int bi = 0, rem = e;                       // e -> (bi ≥ bj) in row-major lower order
                while (rem > bi) { rem -= bi + 1; ++bi; }
                const int bj = rem;
                const int ri = SB * (p + 1 + bi), rj = SB * (p + 1 + bj);
                d4_t c;
#pragma unroll
                for (int r = 0; r < 4; ++r) c[r] = AA(ri + g + 4 * r, rj + r16);
#pragma unroll
                for (int s4 = 0; s4 < 4; ++s4)
                    c = __builtin_amdgcn_mfma_f64_16x16x4f64(-AA(ri + r16, o + g + 4 * s4), AA(rj + r16, o + g + 4 * s4), c, 0, 0, 0);
#pragma unroll
                for (int r = 0; r < 4; ++r) AA(ri + g + 4 * r, rj + r16) = c[r];
            }
        }
        __syncthreads();
        if (p < 4) PROBE(7 + 2 * p);                       // probe build only: end of the trailing update of sub-step p
    }

    PROBE(2);
    if constexpr (MODE == 3) {
        if (t >= NBe && t < NB) dinv[t] = 1.0;             // identity part: never factored above
        __syncthreads();
    }
    // ---------------- phase 2: X = L⁻¹, Xᵀ into the strict upper triangle ----------------
    if (t < NBe) {                                         // (a) diagonal sub-block inverses (identity sub-blocks: nothing to do)
        const int o = SB * (t >> 4), n = t & 15;
        double x[SB];
#pragma unroll
        for (int m = 0; m < SB; ++m) {
            double sacc = 0.0;
#pragma unroll
            for (int k = 0; k < m; ++k) sacc = fma(AA(o + m, o + k), x[k], sacc);
            x[m] = (m < n) ? 0.0 : (m == n ? dinv[o + m] : -sacc * dinv[o + m]);
        }
#pragma unroll
        for (int m = 0; m < SB; ++m)
            if (m > n) AA(o + n, o + m) = x[m];
    }
    __syncthreads();
    PROBE(3);
    for (int dl = 1; dl < nsb; ++dl) {                     // (b) sub-blocks at distance dl below the diagonal
        for (int c = wave; c + dl < nsb; c += DT / 64) {
            const int i = c + dl;
            const int oc = SB * c, oi = SB * i;
            d4_t tt = {0.0, 0.0, 0.0, 0.0};
            for (int k = c; k < i; ++k) {
                const int ok = SB * k;
#pragma unroll
                for (int s4 = 0; s4 < 4; ++s4) {
                    const int kk = g + 4 * s4;
                    const double av = AA(oi + r16, ok + kk);                       // L_ik[m = r16][kk]
                    double bv = AA(oc + r16, ok + kk);                             // X_kc[kk][n = r16] (stored transposed)
                    if (k == c) bv = (kk > r16) ? bv : (kk == r16 ? dinv[oc + r16] : 0.0);
                    tt = __builtin_amdgcn_mfma_f64_16x16x4f64(av, bv, tt, 0, 0, 0);
                }
            }
            d4_t xr = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
            for (int s4 = 0; s4 < 4; ++s4) {
                const int kk = g + 4 * s4;
                const double up = AA(oi + kk, oi + r16);                           // X_ii[m = r16][kk], m > kk
                const double av = (r16 > kk) ? up : (r16 == kk ? dinv[oi + r16] : 0.0);
                xr = __builtin_amdgcn_mfma_f64_16x16x4f64(av, tt[s4], xr, 0, 0, 0);   // B[kk = g+4·s4][n] = tt[s4]
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) AA(oc + r16, oi + g + 4 * r) = -xr[r];   // X_ic[m][n] → a[c·16+n][i·16+m]
        }
        __syncthreads();
    }

    PROBE(4);
    // ---------------- phase 3: write back ----------------
    double* Wb = W + (int64_t)r0 * ld + r0;
    double* WTb = WT + (int64_t)r0 * ld + r0;
    for (int idx = t; idx < NB * NB; idx += DT) {
        const int i = idx >> 7, j = idx & 127;
        double l, w, wt;
        if (i > j) { l = AA(i, j); w = AA(j, i); wt = 0.0; }
        else if (i == j) { l = AA(i, i); w = dinv[i]; wt = dinv[i]; }
        else { l = 0.0; w = 0.0; wt = AA(i, j); }
        if constexpr (MODE == 0 || MODE == 3) Kb[(int64_t)i * ld + j] = l;
        Wb[(int64_t)i * ld + j] = w;
        WTb[(int64_t)i * ld + j] = wt;
    }
    PROBE(5);
    if constexpr (MODE == 3) {
        // α = L⁻ᵀ(L⁻¹δ) from the block in LDS: X = L⁻¹ has its diagonal in dinv and its strict lower part transposed in the
        // strict upper triangle (a[k][i] = X[i][k], k < i).  One thread per row, sums in index order (deterministic).
        if (t < NB) {
            double acc = dinv[t] * dl[t];
            for (int k = 0; k < t; ++k) acc = fma(AA(k, t), dl[k], acc);
            tv[t] = acc;
        }
        __syncthreads();
        if (t < NB) {
            double acc = dinv[t] * tv[t];
            for (int i = t + 1; i < NB; ++i) acc = fma(AA(t, i), tv[i], acc);
            fs.alpha[t] = acc;
            dl[t] = dl[t] * acc;                                     // δ_i α_i (zero on padding rows); dl[t] is this thread's own
        }
        __syncthreads();                                             // every tv[i] has been read
        if (t < NB) tv[t] = t < fs.N ? 2.0 * log(AA(t, t)) : 0.0;    // log-determinant terms
        __syncthreads();
        if (t == 0) {
            double ld2 = 0.0, q = 0.0;
            for (int i = 0; i < NB; ++i) { ld2 += tv[i]; q += dl[i]; }
            fs.scal[0] = ld2;
            fs.scal[1] = q;
        }
    }
}
// Order in which the inverse of a 16×16 lower-triangular sub-block (potf2_pipe_kernel: inverse) uses the factor's strictly-lower entries,
// two rows at a time: for the pair (m, m + 1) the entries (m, k), (m + 1, k) for k < m alternate, then (m + 1, m).  Entry = row·16 + column.
struct InvSeq { int e[120]; };
constexpr InvSeq make_inv_seq() {
    InvSeq t{};
    int c = 0;
    for (int m = 1; m < 16; m += 2) {
        for (int k = 0; k < m; ++k) {
            t.e[c++] = m * 16 + k;
            if (m + 1 < 16) t.e[c++] = (m + 1) * 16 + k;
        }
        if (m + 1 < 16) t.e[c++] = (m + 1) * 16 + m;
    }
    return t;
}

// Operand stream of trsm_stream_kernel, written by potf2_pipe_kernel: stage k (sub-block column k of the diagonal block) holds, per
// lane (m = lane & 15, g = lane >> 4), the A operands of the solve's MFMAs in the order they are issued — X_kk[m][4·s4 + g] for
// s4 = 0 … 3, then −L_jk[m][4·s4 + g] for s4 = 0 … 3, j = k+1 … 7 — 4 + 4·(7 − k) doubles, 144 in all; entry i of lane l lies at
// (i / 2)·128 + 2·l + (i & 1): a lane's 16-byte load takes two consecutive operands, a wave's load is one 1 KB run.
constexpr int TRSM_OPS = 144;
__host__ __device__ constexpr int trsm_stage_base(int k) { return 32 * k - 2 * k * (k - 1); }
#define TRSM_OP(i, l) ((((i) >> 1) << 7) + 2 * (l) + ((i) & 1))

// ---- potf2 with its side work taken off the panel chain (round 5) ----------------------------------------------------------------------
// Round 4's diagonal-block kernel ran its steps one after the other: load the block (4.0 µs), eight times [register step 2.6 µs, trailing update of
// ALL remaining sub-blocks 1.5 … 0.2 µs], the eight 16×16 inverses (2.0 µs), write-back (2.6 µs) — tools/chol_diag_probe.  Only the register
// steps and the update of the NEXT sub-block column are a dependent chain; here everything else runs beside a register step, on waves
// of the SIMDs the spine waves do not use (a wave that shares a SIMD with a spine wave takes issue slots from it: the register step
// is issue-bound):
//   before step 0       sub-block column 0 is loaded (2 of the 16 loads a thread made) — the loads of everything else are in flight
//   beside step 0       the rest of the block's lower sub-blocks arrives and goes to LDS
//   after step p        F(p): panel p's update of sub-block column p + 1 — all the next register step waits for
//   beside step p ≥ 1   D(p−1): panel p−1's update of the sub-block columns beyond p;  column block p−1 of L (final) is written back
//   beside steps 4 … 7  (one spine wave left) the inverses of diagonal sub-blocks 0 … 6 are formed and written to W / WT, two a step;
//                       the last one follows the last step
// (F / D are the MFMA chain of chol_diag_kernel's step (3), panel by panel, barrier-ordered.)
// LDS of the diagonal-block kernel
#define POTF2_PIPE_LDS() \
    __shared__ __attribute__((aligned(16))) double a[NB * LDA]; \
    __shared__ double dinv[NB]; \
    __shared__ __attribute__((aligned(16))) double colbuf[3][128];         /* per spine wave: the pair of columns just finished (register_potf2_step_lds) */ \
    __shared__ int fail

__device__ __forceinline__ void potf2_pipe_body(double* a, double* dinv, double (*colbuf)[128], int& fail,
                                                double* K, double* W, double* WT, int64_t ld, int r0, int64_t* info, double* P) {
    typedef double d2_t __attribute__((ext_vector_type(2)));
    const int t = threadIdx.x;
    const int lane = t & 63, wave = t >> 6;
    const int r16 = lane & 15, g = lane >> 4;
    if (t == 0) fail = 0;
    double* Kb = K + (int64_t)r0 * ld + r0;
    double* Wd = W + (int64_t)r0 * ld + r0;
    double* WTd = WT + (int64_t)r0 * ld + r0;
    PROBE(0);
    // every load of the block is issued before anything is waited for (the status word included): sub-block column 0 first — the counter
    // of outstanding loads is in order, the first register step waits for these two only — then, on the 13 waves without a spine row at
    // step 0, the 28 lower sub-blocks beyond it, whole sub-blocks per wave (at most three each: six 16-byte loads a lane)
    const double c0a = Kb[(int64_t)(t >> 4) * ld + (t & 15)];
    const double c0b = Kb[(int64_t)(64 + (t >> 4)) * ld + (t & 15)];
    const int hw = wave - 3, rr = lane >> 3, c2 = lane & 7;
    d2_t v[3][2];
    if (wave >= 3) {
#pragma unroll
        for (int u = 0; u < 3; ++u) {
            const int sb = hw + 13 * u;
            if (sb < 28) {
                int bi = 0, rem = sb;
                while (rem > bi) { rem -= bi + 1; ++bi; }
                const double* src = Kb + (int64_t)(SB * (bi + 1) + rr) * ld + SB * (rem + 1) + 2 * c2;
                v[u][0] = *reinterpret_cast<const d2_t*>(src);
                v[u][1] = *reinterpret_cast<const d2_t*>(src + 8 * ld);
            }
        }
    }
    const int64_t failed_before = *info;
    AA(t >> 4, t & 15) = c0a;
    AA(64 + (t >> 4), t & 15) = c0b;
    if (failed_before != 0) return;                                        // uniform
    __syncthreads();
    if (wave >= 3) {                                                       // beside register step 0
#pragma unroll
        for (int u = 0; u < 3; ++u) {
            const int sb = hw + 13 * u;
            if (sb < 28) {
                int bi = 0, rem = sb;
                while (rem > bi) { rem -= bi + 1; ++bi; }
                const int i = SB * (bi + 1) + rr, j = SB * (rem + 1) + 2 * c2;
                AA(i, j) = v[u][0][0];
                AA(i, j + 1) = v[u][0][1];
                AA(i + 8, j) = v[u][1][0];
                AA(i + 8, j + 1) = v[u][1][1];
            }
        }
    }
    PROBE(1);
    // panel q's MFMA update of sub-block (bi, bj) of the block that remains behind it (chol_diag_kernel's step (3)); three LDS addresses
    // per lane, every read and write at a constant offset from them (all twelve reads are issued before the first MFMA)
    typedef lds_f64 lds_double;
    auto update = [&](int q, int bi, int bj) {
        const int o = SB * q;
        const int ri = SB * (q + 1 + bi), rj = SB * (q + 1 + bj);
        lds_double* cp = (lds_double*)&AA(ri + g, rj + r16);
        lds_double* ap = (lds_double*)&AA(ri + r16, o + g);
        lds_double* bp = (lds_double*)&AA(rj + r16, o + g);
        d4_t c;
        double av[4], bv[4];
#pragma unroll
        for (int s4 = 0; s4 < 4; ++s4) { av[s4] = ap[4 * s4]; bv[s4] = bp[4 * s4]; }
#pragma unroll
        for (int r = 0; r < 4; ++r) c[r] = cp[4 * r * LDA];
#pragma unroll
        for (int s4 = 0; s4 < 4; ++s4) c = __builtin_amdgcn_mfma_f64_16x16x4f64(-av[s4], bv[s4], c, 0, 0, 0);
#pragma unroll
        for (int r = 0; r < 4; ++r) cp[4 * r * LDA] = c[r];
    };
    // inverse of diagonal sub-block q by lanes 0-15 of ONE wave (column n of X_qq into ROW n of the sub-block's strict upper
    // triangle: phase 2(a) of chol_diag_kernel), then that sub-block's entries of W and WT by the whole wave
    auto inverse = [&](int q) {
        const int o = SB * q;
        if (lane < SB) {
            const int n = lane;
            constexpr int D = 12;                                          // reads in flight ahead of the chains that use them
            constexpr InvSeq seq = make_inv_seq();
            lds_double* ab = (lds_double*)&AA(o, o);                       // the sub-block's corner: every read below is `ab + constant`
            double x[SB], win[D];
#pragma unroll
            for (int i = 0; i < D; ++i) win[i] = ab[(seq.e[i] >> 4) * LDA + (seq.e[i] & 15)];
            x[0] = (0 < n) ? 0.0 : dinv[o];
            // Rows m and m + 1 run as two interleaved chains (row m + 1 needs x[m] only in its last step); each row's own sum keeps its
            // order k = 0, 1, …: same bits as the one-row-at-a-time loop.  The read that refills a window slot is made to depend on the
            // sum just formed: left alone, instruction selection puts all 120 reads in front of the first multiply-add and spills them
            // (a scheduling fence does not hold pure arithmetic back).
            int i = 0;
#define ABO_REFILL(acc) do { if (i + D < 120) { asm volatile("" : "+v"(ab), "+v"(acc)); \
                             win[i % D] = ab[(seq.e[i + D] >> 4) * LDA + (seq.e[i + D] & 15)]; } ++i; } while (0)
#pragma unroll
            for (int m = 1; m < SB; m += 2) {
                const double da = dinv[o + m], db = m + 1 < SB ? dinv[o + m + 1] : 0.0;
                double sa = 0.0, sb = 0.0;
#pragma unroll
                for (int k = 0; k < m; ++k) {
                    sa = fma(win[i % D], x[k], sa);
                    ABO_REFILL(sa);
                    if (m + 1 < SB) {
                        sb = fma(win[i % D], x[k], sb);
                        ABO_REFILL(sb);
                    }
                }
                x[m] = (m < n) ? 0.0 : (m == n ? da : -sa * da);
                if (m + 1 < SB) {
                    sb = fma(win[i % D], x[m], sb);
                    ABO_REFILL(sb);
                    x[m + 1] = (m + 1 < n) ? 0.0 : (m + 1 == n ? db : -sb * db);
                }
            }
#undef ABO_REFILL
#pragma unroll
            for (int m = 0; m < SB; ++m)
                if (m > n) AA(o + n, o + m) = x[m];
        }
        wave_lds_sync();
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int i = (lane + 64 * e) >> 4, j = lane & 15;
            const double w = i > j ? AA(o + j, o + i) : (i == j ? dinv[o + i] : 0.0);          // X[i][j] lives at a[j][i]
            Wd[(int64_t)(o + i) * ld + o + j] = w;
            WTd[(int64_t)(o + j) * ld + o + i] = w;
        }
        if (P) {                                                           // stage q of trsm_stream_kernel's operand stream: X_qq[m = r16][kk = 4·s4 + g]
#pragma unroll
            for (int s4 = 0; s4 < 4; ++s4) {
                const int c = 4 * s4 + g;
                const double w = r16 > c ? AA(o + c, o + r16) : (r16 == c ? dinv[o + r16] : 0.0);
                P[TRSM_OP(trsm_stage_base(q) + s4, lane)] = w;
            }
        }
    };
    for (int p = 0; p < NSB; ++p) {
        const int o = SB * p;
        const int below = NB - o - SB;
        const int nspine = below > 96 ? 3 : (below > 48 ? 2 : 1);          // the waves `wave * 48 < below || wave == 0`
        if (wave < nspine) {
            const int row = lane < SB ? o + lane : o + SB + wave * 48 + (lane - SB);
            const bool valid = row < NB;
            double x[SB];
#pragma unroll
            for (int c = 0; c < SB; ++c) x[c] = valid ? AA(row, o + c) : 0.0;
            const unsigned badc = register_potf2_step_lds(x, lane, (lds_f64*)colbuf[wave], wave == 0 ? (lds_f64*)&dinv[o] : (lds_f64*)&colbuf[wave][32]);
            if (badc) {
                if (t == 0) { fail = 1; *info = (int64_t)r0 + o + __builtin_ctz(badc) + 1; }
            } else if (valid) {
                if (lane >= SB) {
#pragma unroll
                    for (int c = 0; c < SB; ++c) AA(row, o + c) = x[c];
                } else if (wave == 0) {
#pragma unroll
                    for (int c = 0; c < SB; ++c)
                        if (c <= lane) AA(row, o + c) = x[c];
                }
            }
        } else if (p >= 4 && (wave & 3) >= 2) {
            // steps 4 … 7 have one spine wave: SIMDs 2 and 3 take the seven inverses that are due, one wave alone on its SIMD each
            // (beside a wave that issues MFMA updates an inverse takes 3.4 µs instead of 2.1 — longer than the register step)
            const int qi = 2 * (p - 4) + (wave == 15 ? 0 : 1);
            if ((wave == 15 || wave == 14) && qi < NSB - 1) inverse(qi);
        } else if (p > 0 && (p < 4 || (wave & 3) >= nspine)) {
            // D(p−1) and the write-back of column block p−1.  Steps 1 – 3 (21, 15, 10 sub-block updates: four dependent MFMAs each, 256
            // cycles of ONE SIMD's matrix pipe): over ALL the waves that are not spine waves — round 5 kept them off the spine waves'
            // SIMDs, and round 6 measured these sub-steps with the register step EMPTY at the same 3.4 / 2.9 / 2.2 µs: the side work
            // on two SIMDs was the critical path there, not the spine (tools/chol_diag_probe -DABO_STEP_EMPTY); spread over all
            // SIMDs it and the spine waves slow each other down to 2.8 / 2.4 / 2.3 µs.  Steps 4 – 7 (≤ 6 updates): SIMD 1 only, as
            // before — SIMDs 2 and 3 belong to the inverses.
            const int lo = p < 4 ? 0 : nspine, hi = p < 4 ? 4 : 2;          // SIMDs lo … hi−1 (p < 4: all, minus the spine waves themselves)
            const int per = hi - lo;
            const int hid = p < 4 ? wave - nspine : (wave >> 2) * per + ((wave & 3) - lo);
            const int Hd = p < 4 ? DT / 64 - nspine : 4 * per;
            const int q = p - 1;
            const int nb = NSB - 1 - q;                                    // sub-block rows / columns behind panel q; column 0 is done (F)
            const int total = (nb - 1) * nb / 2;
            for (int e = hid; e < total; e += Hd) {
                int bi = 0, rem = e;
                while (rem > bi) { rem -= bi + 1; ++bi; }
                update(q, bi + 1, rem + 1);
            }
            for (int idx = hid * 64 + lane; idx < NB * SB; idx += Hd * 64) {            // column block q of L: final
                const int i = idx >> 4, j = SB * q + (idx & 15);
                Kb[(int64_t)i * ld + j] = i >= j ? AA(i, j) : 0.0;
            }
            if (P) {                                                       // … and, negated, as stage q of the panel solve's operand stream
                const int nj = NSB - 1 - q;
                for (int e = hid; e < 4 * nj; e += Hd) {
                    const int s4 = e / nj, jj = e - s4 * nj;
                    P[TRSM_OP(trsm_stage_base(q) + 4 + e, lane)] = -AA(SB * (q + 1 + jj) + r16, SB * q + 4 * s4 + g);
                }
            }
        }
        __syncthreads();
        if (p < 4) PROBE(6 + 2 * p);
        if (fail) return;                                                  // uniform (LDS flag after the barrier)
        if (p + 1 < NSB) {
            for (int e = wave; e < NSB - 1 - p; e += DT / 64) update(p, e, 0);          // F(p)
            __syncthreads();
        }
        if (p < 4) PROBE(7 + 2 * p);
    }
    PROBE(2);
    for (int idx = t; idx < NB * SB; idx += DT) {                          // the last column block; D(7) is empty
        const int i = idx >> 4, j = SB * (NSB - 1) + (idx & 15);
        Kb[(int64_t)i * ld + j] = i >= j ? AA(i, j) : 0.0;
    }
    PROBE(3);
    if (wave == 0) inverse(NSB - 1);
    PROBE(4);
    PROBE(5);
}

// Workgroup 0 is the diagonal-block kernel.  The chain leaves 255 compute units idle while it runs, so the launch's other workgroups
// (nz of them, when the caller asks) do what two whole-matrix memsets in front of the fit used to: they zero rows r0 … r0+127 of W and
// WT over all ld columns EXCEPT the 128 × 128 diagonal block (workgroup 0 puts the 16 × 16 inverses there, the batched trtri behind the
// factorisation rewrites the whole block) — 16 MB per launch at N = 8192, done long before the block is factored.
__global__ void __launch_bounds__(DT) potf2_pipe_kernel(double* K, double* W, double* WT, int64_t ld, int r0, int64_t* info, double* P) {
    if (blockIdx.x > 0) {
        typedef double d2_t __attribute__((ext_vector_type(2)));
        const int64_t per_row = ld / 2;                                  // 16-byte stores: ld is a multiple of 128
        const int64_t total = 2 * (int64_t)NB * per_row;                 // both matrices
        const d2_t z = {0.0, 0.0};
        for (int64_t idx = (int64_t)(blockIdx.x - 1) * DT + threadIdx.x; idx < total; idx += (int64_t)(gridDim.x - 1) * DT) {
            const int64_t e = idx >= (int64_t)NB * per_row ? idx - (int64_t)NB * per_row : idx;
            const int row = (int)(e / per_row);
            const int64_t col = 2 * (e % per_row);
            if (col >= r0 && col < r0 + NB) continue;
            double* base = idx >= (int64_t)NB * per_row ? WT : W;
            *reinterpret_cast<d2_t*>(base + (int64_t)(r0 + row) * ld + col) = z;
        }
        return;
    }
    POTF2_PIPE_LDS();
    potf2_pipe_body(a, dinv, colbuf, fail, K, W, WT, ld, r0, info, P);
}

#undef AA

// a model of ONE 128-row block that the whole-fit kernel does not serve (d > 16, or spare capacity): factor + inverse in one launch
hipError_t launch_chol_diag(double* K, double* W, double* WT, int64_t ld, int r0, int64_t* info, hipStream_t s) {
    hipLaunchKernelGGL(chol_diag_kernel<0>, dim3(1), dim3(DT), 0, s, K, W, WT, ld, r0, info, FitSmallArgs{});
    return hipGetLastError();
}

// the diagonal block of a panel; P: where the packed operands of the panel solve go (TRSM_OPS × 64 doubles)
// zero_rows: the launch also zeroes rows r0 … r0+127 of W / WT outside the diagonal block (see the kernel)
hipError_t launch_potf2_diag(double* K, double* W, double* WT, int64_t ld, int r0, int64_t* info, hipStream_t s, double* P, bool zero_rows) {
    int nz = 0;
    if (zero_rows) { nz = (int)(ld / 32); nz = nz < 1 ? 1 : (nz > 255 ? 255 : nz); }     // ≥ 4 stores a thread
    hipLaunchKernelGGL(potf2_pipe_kernel, dim3(1 + nz), dim3(DT), 0, s, K, W, WT, ld, r0, info, P);
    return hipGetLastError();
}

hipError_t launch_trtri_diag_batched(double* K, double* W, double* WT, int64_t ld, int nblocks, int64_t* info, hipStream_t s) {
    if (nblocks <= 0) return hipSuccess;
    hipLaunchKernelGGL(chol_diag_kernel<2>, dim3(nblocks), dim3(DT), 0, s, K, W, WT, ld, 0, info, FitSmallArgs{});
    return hipGetLastError();
}

hipError_t launch_fit_small(double* K, double* W, double* WT, int64_t* info, const FitSmallArgs& fs, hipStream_t s) {
    if (fs.N < 1 || fs.N > NB || fs.dp < 1 || fs.dp > 16) return hipErrorInvalidValue;
    hipLaunchKernelGGL(chol_diag_kernel<3>, dim3(1), dim3(DT), 0, s, K, W, WT, (int64_t)NB, 0, info, fs);
    return hipGetLastError();
}

// Panel solve as a blocked triangular solve:  X = A·L⁻ᵀ  for the rows below a freshly factored 128×128 diagonal block
// (rows r0+128 … r0+128+nrows, columns r0 … r0+127; in place).  One wave per 16 rows, in transposed space so that every
// product chains through the matrix pipe without data movement:  with Y_j = X_jᵀ (16 columns of block j × 16 rows),
//     Y_j = L_jj⁻¹ · (A_jᵀ − Σ_{k<j} L_jk · Y_k)
// L_jk / L_jj⁻¹ are A operands, the Y_k are previous MFMA results, whose C/D register r is exactly the B operand of k-step r.
// 176 MFMAs per wave (64 of them on the dependent path); no 128×128 inverse is needed — only the eight 16×16 diagonal inverses.
// Right-looking order: as soon as Y_k is final every later block takes its update, k-step by k-step over DIFFERENT accumulators —
// independent MFMAs that issue back to back; the dependent path is 8 × (4 + 4) MFMAs.
// The A operands are streamed from the packed copy potf2_pipe_kernel leaves: no staging of the 128×128 block in LDS (64 loads a thread,
// a transposing fill and a barrier in front of the first MFMA were a third of round 4's 16.5 µs), no LDS at all — a wave is on its own: 32 loads of its right-hand sides, 72 16-byte loads of operands (one 1 KB run per wave and
// load, L2 hits), all issued before the first MFMA, then the 144 MFMAs (the negation of L_jk is in the stream).  One wave per workgroup, so that a launch over few rows still spreads over
// as many SIMDs as it has waves (N = 1024: 56 instead of 14 workgroups).
__global__ void __launch_bounds__(64) trsm_stream_kernel(double* K, const double* __restrict__ P, int64_t ld, int r0, int nrows,
                                                         const int64_t* info) {
    typedef double d2_t __attribute__((ext_vector_type(2)));
    const int lane = threadIdx.x, n = lane & 15, g = lane >> 4;
    const int rb = blockIdx.x;
    if (rb * 16 >= nrows) return;
    double* Arow = K + (int64_t)(r0 + NB + rb * 16 + n) * ld + r0;    // this lane's row, the panel's 128 columns
    d4_t Y[8];
#pragma unroll
    for (int j = 0; j < 8; ++j)
#pragma unroll
        for (int r = 0; r < 4; ++r) Y[j][r] = Arow[16 * j + 4 * r + g];
    d2_t op[TRSM_OPS / 2];
#pragma unroll
    for (int i = 0; i < TRSM_OPS / 2; ++i) op[i] = reinterpret_cast<const d2_t*>(P)[i * 64 + lane];
    if (*info != 0) return;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        const int b = trsm_stage_base(k);
        d4_t y = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
        for (int s4 = 0; s4 < 4; ++s4) y = __builtin_amdgcn_mfma_f64_16x16x4f64(op[(b + s4) >> 1][(b + s4) & 1], Y[k][s4], y, 0, 0, 0);
        Y[k] = y;
#pragma unroll
        for (int s4 = 0; s4 < 4; ++s4)
#pragma unroll
            for (int j = k + 1; j < 8; ++j) {
                const int i = b + 4 + s4 * (7 - k) + (j - k - 1);
                Y[j] = __builtin_amdgcn_mfma_f64_16x16x4f64(op[i >> 1][i & 1], Y[k][s4], Y[j], 0, 0, 0);
            }
    }
#pragma unroll
    for (int j = 0; j < 8; ++j)
#pragma unroll
        for (int r = 0; r < 4; ++r) Arow[16 * j + 4 * r + g] = Y[j][r];
}

hipError_t launch_trsm_stream(double* K, const double* P, int64_t ld, int r0, int nrows, const int64_t* info, hipStream_t s) {
    if (nrows <= 0) return hipSuccess;
    hipLaunchKernelGGL(trsm_stream_kernel, dim3((nrows + 15) / 16), dim3(64), 0, s, K, P, ld, r0, nrows, info);
    return hipGetLastError();
}

// out[r] = Σ_k Wm[r][k]·v[k] over the triangular part of rows r < Np (lower: k ≤ r; upper: r ≤ k < Np).
// A wave takes TWO groups of 4 consecutive rows — group q and group nq−1−q, a short and a long one, so that every wave of the
// launch streams the same number of bytes (with one group per wave, handed out top to bottom, the launch ended in a tail of
// the longest rows alone, latency-bound: 5.3 TB/s at N = 16384; balanced and with four steps of loads in flight per lane it is
// HBM-bound).  Within a group the four rows share every 16-byte load of v, lanes stride k by 128, partial sums per lane in
// increasing k and then the fixed xor tree — the order of round 1, same bits.  The k range is that of the longest of the four
// rows — the entries it adds for the shorter ones are structural zeros of the triangular factor — and v is masked at the
// range end, so stale values beyond Np (the remains of a discarded append) never enter.  HBM-bound: 4·Np² bytes.
constexpr int TRMV_U = 4;     // k steps of 128 whose loads are issued together
__device__ __forceinline__ void trmv_rows4(const double* __restrict__ Wm, int64_t ld, const double* __restrict__ v,
                                           double* __restrict__ out, int Np, int lower, int row0, int lane) {
    typedef double d2_t __attribute__((ext_vector_type(2)));
    const int kb = lower ? 0 : (row0 & ~1);
    const int ke = lower ? min(row0 + 4, Np) : Np;
    const double* r0 = Wm + (int64_t)row0 * ld;
    const double* r1 = r0 + (row0 + 1 < Np ? ld : 0);
    const double* r2 = r0 + (row0 + 2 < Np ? 2 * ld : 0);
    const double* r3 = r0 + (row0 + 3 < Np ? 3 * ld : 0);
    double a0 = 0.0, a1 = 0.0, a2 = 0.0, a3 = 0.0;
    int k = kb + 2 * lane;                                   // ld is even and > k + 1: the pair load stays inside the row
    for (; k + 128 * (TRMV_U - 1) < ke; k += 128 * TRMV_U) {
        d2_t vv[TRMV_U], x0[TRMV_U], x1[TRMV_U], x2[TRMV_U], x3[TRMV_U];
#pragma unroll
        for (int u = 0; u < TRMV_U; ++u) {
            const int ku = k + 128 * u;
            vv[u] = *reinterpret_cast<const d2_t*>(v + ku);
            if (ku + 1 >= ke) vv[u][1] = 0.0;
            x0[u] = __builtin_nontemporal_load(reinterpret_cast<const d2_t*>(r0 + ku));   // read once: stream
            x1[u] = __builtin_nontemporal_load(reinterpret_cast<const d2_t*>(r1 + ku));
            x2[u] = __builtin_nontemporal_load(reinterpret_cast<const d2_t*>(r2 + ku));
            x3[u] = __builtin_nontemporal_load(reinterpret_cast<const d2_t*>(r3 + ku));
        }
#pragma unroll
        for (int u = 0; u < TRMV_U; ++u) {
            a0 = fma(x0[u][1], vv[u][1], fma(x0[u][0], vv[u][0], a0));
            a1 = fma(x1[u][1], vv[u][1], fma(x1[u][0], vv[u][0], a1));
            a2 = fma(x2[u][1], vv[u][1], fma(x2[u][0], vv[u][0], a2));
            a3 = fma(x3[u][1], vv[u][1], fma(x3[u][0], vv[u][0], a3));
        }
    }
    for (; k < ke; k += 128) {
        d2_t vv = *reinterpret_cast<const d2_t*>(v + k);
        if (k + 1 >= ke) vv[1] = 0.0;
        const d2_t x0 = __builtin_nontemporal_load(reinterpret_cast<const d2_t*>(r0 + k));
        const d2_t x1 = __builtin_nontemporal_load(reinterpret_cast<const d2_t*>(r1 + k));
        const d2_t x2 = __builtin_nontemporal_load(reinterpret_cast<const d2_t*>(r2 + k));
        const d2_t x3 = __builtin_nontemporal_load(reinterpret_cast<const d2_t*>(r3 + k));
        a0 = fma(x0[1], vv[1], fma(x0[0], vv[0], a0));
        a1 = fma(x1[1], vv[1], fma(x1[0], vv[0], a1));
        a2 = fma(x2[1], vv[1], fma(x2[0], vv[0], a2));
        a3 = fma(x3[1], vv[1], fma(x3[0], vv[0], a3));
    }
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) {
        a0 += __shfl_xor(a0, o);
        a1 += __shfl_xor(a1, o);
        a2 += __shfl_xor(a2, o);
        a3 += __shfl_xor(a3, o);
    }
    if (lane == 0) {
        out[row0] = a0;
        if (row0 + 1 < Np) out[row0 + 1] = a1;
        if (row0 + 2 < Np) out[row0 + 2] = a2;
        if (row0 + 3 < Np) out[row0 + 3] = a3;
    }
}

__global__ void __launch_bounds__(256) trmv_kernel(const double* __restrict__ Wm, int64_t ld, const double* __restrict__ v,
                                                   double* __restrict__ out, int Np, int lower) {
    const int lane = threadIdx.x & 63;
    const int nq = (Np + 3) / 4;                             // groups of 4 rows
    const int q1 = blockIdx.x * 4 + (threadIdx.x >> 6);      // wave-uniform
    const int q2 = nq - 1 - q1;
    if (q1 > q2) return;
    trmv_rows4(Wm, ld, v, out, Np, lower, 4 * q1, lane);
    if (q2 != q1) trmv_rows4(Wm, ld, v, out, Np, lower, 4 * q2, lane);
}

hipError_t launch_trmv(const double* Wm, int64_t ld, const double* v, double* out, int Np, int lower, hipStream_t s) {
    if (Np <= 0) return hipSuccess;
    const int nq = (Np + 3) / 4, nw = (nq + 1) / 2;          // waves: one per pair of row groups
    hipLaunchKernelGGL(trmv_kernel, dim3((nw + 3) / 4), dim3(256), 0, s, Wm, ld, v, out, Np, lower);
    return hipGetLastError();
}

__global__ void __launch_bounds__(256) nlml_terms_kernel(const double* L, int64_t ld, const double* delta,
                                                         const double* alpha, int N, double* out) {
    __shared__ double r0[256], r1[256];
    double s0 = 0.0, s1 = 0.0;
    for (int i = (int)threadIdx.x; i < N; i += 256) {
        s0 += 2.0 * log(L[(int64_t)i * ld + i]);
        s1 = fma(delta[i], alpha[i], s1);
    }
    r0[threadIdx.x] = s0;
    r1[threadIdx.x] = s1;
    __syncthreads();
    for (int o = 128; o >= 1; o >>= 1) {
        if (threadIdx.x < o) { r0[threadIdx.x] += r0[threadIdx.x + o]; r1[threadIdx.x] += r1[threadIdx.x + o]; }
        __syncthreads();
    }
    if (threadIdx.x == 0) { out[0] = r0[0]; out[1] = r1[0]; }
}

hipError_t launch_nlml_terms(const double* L, int64_t ld, const double* delta, const double* alpha, int N, double* out,
                             hipStream_t s) {
    hipLaunchKernelGGL(nlml_terms_kernel, dim3(1), dim3(256), 0, s, L, ld, delta, alpha, N, out);
    return hipGetLastError();
}

// Bordered append, stage 1 (one workgroup): s2 = kss − ‖l‖², kα = kᵀα_old, pivot check.
__global__ void __launch_bounds__(1024) append_reduce_kernel(AppendArgs p) {
    __shared__ double r0[1024], r1[1024];
    double a0 = 0.0, a1 = 0.0;
    for (int i = (int)threadIdx.x; i < p.N; i += 1024) {
        a0 = fma(p.lvec[i], p.lvec[i], a0);
        a1 = fma(p.krow[i], p.alpha_old[i], a1);
    }
    r0[threadIdx.x] = a0;
    r1[threadIdx.x] = a1;
    __syncthreads();
    for (int o = 512; o >= 1; o >>= 1) {
        if (threadIdx.x < o) { r0[threadIdx.x] += r0[threadIdx.x + o]; r1[threadIdx.x] += r1[threadIdx.x + o]; }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        const double s2 = p.kss - r0[0];
        if (!(s2 > 0.0)) { if (*p.info == 0) *p.info = (int64_t)p.N + 1; p.scal[0] = s2; return; }   // the FIRST failing row is reported
        const double lnn = sqrt(s2);
        const double beta = (p.delta[p.N] - r1[0]) / s2;
        p.scal[0] = s2; p.scal[1] = beta; p.scal[2] = lnn; p.scal[3] = r1[0];
    }
}

// stage 2: rows/columns N of the factors, new alpha, down-date vector
__global__ void __launch_bounds__(256) append_write_kernel(AppendArgs p) {
    if (*p.info != 0) return;
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= p.cap) return;
    const double lnn = p.scal[2], beta = p.scal[1];
    const double rl = 1.0 / lnn;
    const int64_t N = p.N;
    if (j < N) {
        const double v = p.vvec[j];
        const double w = -v * rl;
        p.L[N * p.ld + j] = p.lvec[j];
        p.W[N * p.ld + j] = w;
        p.WT[(int64_t)j * p.ld + N] = w;
        p.alpha_new[j] = fma(-beta, v, p.alpha_old[j]);
        p.vext[j] = -v;
    } else if (j == N) {
        p.L[N * p.ld + N] = lnn;
        p.W[N * p.ld + N] = rl;
        p.WT[N * p.ld + N] = rl;
        p.alpha_new[N] = beta;
        p.vext[N] = 1.0;
    } else {
        p.alpha_new[j] = 0.0;
        p.vext[j] = 0.0;
    }
}

hipError_t launch_append(const AppendArgs& a, hipStream_t s) {
    hipLaunchKernelGGL(append_reduce_kernel, dim3(1), dim3(1024), 0, s, a);
    hipLaunchKernelGGL(append_write_kernel, dim3((a.cap + 255) / 256), dim3(256), 0, s, a);
    return hipGetLastError();
}

__global__ void center_kernel(const double* y, double* delta, int N, int Np, double c) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < Np) delta[i] = (i < N) ? y[i] - c : 0.0;
}

__global__ void center_grad_kernel(const double* y_in, double* ybuf, double* delta, int N, int p, int Np, MeanVec mean, int y_point_major) {
    const int r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= Np) return;
    if (r < N * p) {
        const int i = r / p, q = r % p;
        const double v = y_point_major ? y_in[r] : y_in[(int64_t)q * N + i];
        ybuf[r] = v;
        delta[r] = v - mean.c[q];
    } else {
        delta[r] = 0.0;
    }
}

hipError_t launch_center_grad(const double* y_abi, double* ybuf, double* delta, int N, int p, int Np, MeanVec mean, int y_point_major,
                              hipStream_t s) {
    hipLaunchKernelGGL(center_grad_kernel, dim3((Np + 255) / 256), dim3(256), 0, s, y_abi, ybuf, delta, N, p, Np, mean, y_point_major);
    return hipGetLastError();
}

hipError_t launch_center(const double* y, double* delta, int N, int Np, double c, hipStream_t s) {
    hipLaunchKernelGGL(center_kernel, dim3((Np + 255) / 256), dim3(256), 0, s, y, delta, N, Np, c);
    return hipGetLastError();
}

}  // namespace abo
