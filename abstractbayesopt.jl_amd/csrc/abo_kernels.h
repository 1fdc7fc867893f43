// Internal launcher interface between the C-ABI (api.hip) and the kernel translation units.
// Everything here is device-pointer based; no torch types, no host math.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace abo {

constexpr int TB = 128;  // block (tile) edge used by every blocked stage; all padded sizes are multiples of it
constexpr int MAX_P = 129; // outputs per point of a gradient-enhanced GP (f + d ≤ 128 partial derivatives; d ≤ 32: register-resident generator)

inline int64_t pad_up(int64_t n, int64_t m) { return (n + m - 1) / m * m; }

// ---- fp64 MFMA GEMM family (gemm.hip) -------------------------------------------------------
enum KMode { K_FULL = 0, K_A_LOWER = 1, K_A_UPPER = 2, K_B_LOWER = 3, K_B_UPPER = 4 };

struct GemmArgs {
    const double* A;      // [M][lda]   row-major, k contiguous
    const double* B;      // [N][ldb]   row-major, k contiguous  (C = A·Bᵀ)
    double* C;            // [M][ldc]
    double* Ct;           // optional transposed copy  Ct[j][i] = C[i][j]
    int64_t lda, ldb, ldc, ldct;
    int64_t sA, sB, sC, sCt;   // batch strides (elements)
    int M, N, K;          // M, N multiples of 128; K multiple of 16
    int kmode;            // K_FULL; K_A_LOWER: k < (ti+1)·128; K_A_UPPER: k ≥ ti·128; K_B_LOWER: k < (tj+1)·128 (B lower-triangular);
                          // K_B_UPPER: k ≥ tj·128 (B upper-triangular)
    int lower_only;       // 1: skip tiles with tj > ti (SYRK on the lower triangle)
    int batch;
    double alpha, beta;
    const int64_t* info;  // optional: if *info != 0 the kernel exits at once (failed factorisation)
    // split-k (batch == 1, beta == 0): the k range is cut into chunks of `ksplit` (multiple of 128), chunk z of a tile is computed by
    // its own workgroup into the partial product C + z·sC (a chunk outside the tile's triangular k range writes nothing);
    // launch_splitk_reduce sums the partials of every tile in chunk order (fixed order: deterministic).  For products with few output
    // tiles and a long k — the 128-row batches of the lockstep refinement against L⁻¹.
    int ksplit = 0;
    // split-k only: when 16 ≤ mrows ≤ 64 only the first mrows rows of A (a multiple of 16) matter and M == 128: the product is taken by
    // a kernel without LDS whose waves stream their 32 rows of B once and hold mrows/16 row groups of A — HBM-bound on B instead of
    // MFMA-bound on 128 rows (the lockstep refinement once most starts have finished)
    int mrows = 0;
    // square SYRK on the lower triangle (kmode K_FULL, batch 1), set by launch_gemm_nt: number of lower tiles T(T+1)/2 of a 1-D launch
    // whose workgroup → tile map is XCD-aware (gemm_nt_kernel); 0 = the plain 2-D launch
    int swz = 0;
    int swz_g = 4;        // tile rows per super-row of that map
};
hipError_t launch_gemm_nt(const GemmArgs& a, hipStream_t s);
// out[r][c] = Σ_z P[z][r][c] over the chunks z that intersect the k range of column tile c/128 under `kmode` (K_FULL, K_B_LOWER,
// K_B_UPPER); P: nz partial products of rows × cols (leading dimension ldp, stride sP), K the full k length
hipError_t launch_splitk_reduce(const double* P, int64_t ldp, int64_t sP, int nz, int rows, int cols, int K, int ksplit, int kmode,
                                double* out, int64_t ldo, hipStream_t s);

// V = W·K_XZ restricted to k ≤ i (W lower-triangular), reduced on the fly to per-row-block column
// sums of squares: partial[ti][j] = Σ_{i in block ti} (Σ_k W[i][k]·Kxz[j][k])²
struct VarGemmArgs {
    const double* W;      // [Np][ldw]
    const double* Kxz;    // [Mc][ldk]   candidate-major chunk, k contiguous
    double* partial;      // [Np/128][ldp]
    int64_t ldw, ldk, ldp;
    int Np, Mc;           // multiples of 128
    int nvalid;           // rows ≥ nvalid (the view's N) are excluded from the norm
};
hipError_t launch_var_gemm(const VarGemmArgs& a, hipStream_t s);

// ---- the same contraction on the int8 matrix pipe (ozaki.hip): exact products of fixed-point images of W and K_XZ through
// residues modulo n coprime moduli ≤ 256, Chinese-remainder reconstruction in fp64
constexpr int OZ_MAXMOD = 16;
struct OzPlan {
    int n;                       // number of moduli (8 … 16)
    int p[OZ_MAXMOD];            // 256, 255, 253, 251, 247, 241, 239, 233, 229, 227, 223, 217, 211, 199, 197, 193
    double invp[OZ_MAXMOD];
    double c26[OZ_MAXMOD];       // 2^26 mod p (symmetric): the quantiser reduces x = xh·2^26 + xl as xh·c26 + xl
    double s1[OZ_MAXMOD], s2[OZ_MAXMOD];   // (P/p)·((P/p)⁻¹ mod p) split into a 41-bit head on a common grid and the rest
    double P1, P2, invP;         // P = Π p split the same way; 1/P
    int eP;                      // 2^eP ≤ P/4: the bound every exact integer dot product is kept under
};
bool oz_make_plan(int n, OzPlan* out);
size_t oz_w_bytes(int n, int Np);            // residue planes of W: n × pad256(Np)²
size_t oz_k_bytes(int n, int Np, int Mc);    // residue planes of a candidate chunk (and of its U): n × pad256(Mc) × pad256(Np)
int oz_k_scale(double kmax);                 // sK with rint(K·2^sK) < 2^53 for 0 ≤ K ≤ kmax
// W (lower-triangular, [Np][ldw]) → WR [n][Np256][Np256] int8, sexp[Np256] (row scales s_i), bad_row[Np256] (non-finite rows)
// rows ≥ nvalid (identity padding, or the remains of a discarded appended branch) become zero planes
hipError_t oz_prepare_w(const OzPlan& pl, const double* W, int64_t ldw, int Np, int nvalid, int8_t* WR, int* sexp, int* bad_row, hipStream_t s,
                        int kper = 1, int ktg = 0);
constexpr int OZ_CTR_INTS = 16;
struct OzVarArgs {
    const OzPlan* plan;
    const double* Kxz;     // [Mc][ldk] candidate-major chunk (fp64, as launch_var_gemm takes it)
    int64_t ldk;
    const int8_t* WR; const int* sexp; const int* bad_row;
    int8_t* KR;            // scratch: oz_k_bytes
    int8_t* U;             // scratch: oz_k_bytes
    int* bad_col;          // scratch: pad256(Mc) + OZ_CTR_INTS ints (the persistent GEMM's tile counters sit behind the flags)
    double* partial;       // [Np/128][ldp]
    int64_t ldp;
    int Np, Mc, nvalid, sK;
    int kper = 1, ktg = 0; // gradient-enhanced model: training rows k with k % kper != 0 are scaled by 2^−ktg in the chunk's image
    int planes_ready = 0;  // 1: KR / bad_col were written by the generator (KgenArgs::res), skip the quantisation pass
    // all outputs of a gradient-enhanced model's candidates: chunk row r is output (r0 + r) % rper of its point (rmode 1, point-major
    // rows) or output (r0 + r) / rpts (rmode 2, rows by outputs); derivative rows (output != 0) are scaled by 2^-ktg in the chunk's
    // image as well, and their column sums by 2^(2 ktg) on the way out.  rmode 0: every row is a function value.
    int rmode = 0, rper = 1;
    int64_t r0 = 0, rpts = 1;
    double* Vout = nullptr;     // when given: V = W.K_XZ itself, [Mc][ldv] fp64 (candidate-major), instead of the column sums of squares
    int64_t ldv = 0;
    hipEvent_t ev_quant = nullptr, ev_gemm = nullptr;   // optional: recorded after the quantisation / after the GEMM
    // optional, HOST: where the owner remembers which tile-counter block is known to hold zeros.  The reconstruction kernel that ends a
    // call zeroes the counters again, so the next call on the same scratch needs no memset in front of its GEMM; the slot is cleared
    // while a call is between its GEMM and its reconstruction (an error in there leaves the counters dirty: the next call memsets).
    int** ctr_clean = nullptr;
};
hipError_t launch_var_ozaki(const OzVarArgs& a, hipStream_t s);

// ---- kernel-matrix generation (kgen.hip) ----------------------------------------------------
struct KgenArgs {
    const double* Xs;     // [Np][dp]   training points, pre-scaled by 1/ell, zero padded
    const double* Z;      // [M][d]     raw candidate points (caller layout), scaled on load
    const double* alpha;  // [Np] zero padded, or nullptr (no mean)
    double* Kout;         // [Mc][ldk]  Kout[j][k] = sigma_f2·kappa(||Xs_k − s·z_j||²), 0 for k ≥ N
    double* mu;           // [Mc] mean_c + Σ_k Kout[j][k]·alpha[k], or nullptr
    int64_t ldk;
    int64_t M;            // number of valid candidates overall
    int64_t j0;           // first candidate of this chunk
    int Mc;               // padded chunk rows (multiple of 16)
    int N, Np, d, dp, family;
    double s, sigma_f2, mean_c;
    // gradient-enhanced GP (pt > 1): training rows are (output q', point i) by outputs, pt = d+1 outputs per
    // point, Np pads pt·N; candidate rows carry pc outputs each (1 = function value only, pt = all outputs)
    int pt = 1, pc = 1, point_major = 0;
    int dlogell = 0;      // gradient-enhanced GP only: write dK/dlog(ell) instead of K (hyper-parameter gradient)
    int rvalid = 0;       // gradient-enhanced GP only: valid training rows when not N·pt (a partly appended point)
    double mean_vec[MAX_P] = {0};   // prior mean per output (gradConstMean)
    // int8-residue engine: write the residue planes of the kernel values in the same pass (StandardGP, dp ≤ 32):
    // res[l][j][k] = sym_residue(rint(K[j][k]·2^res_sK), p_l), res_bad[j] = 1 for a candidate with a non-finite kernel value
    // (zeroed by the launcher); Kout may then be nullptr
    int8_t* res = nullptr;
    int64_t res_ld = 0, res_plane = 0;
    int* res_bad = nullptr;
    int res_n = 0, res_sK = 0;
    int res_ktg = 0;      // gradient-enhanced GP: derivative training rows and derivative candidate outputs each carry 2^-res_ktg on top
};
// true when launch_kgen honours KgenArgs::res for this shape and n moduli (the fused output is instantiated for the default plan)
inline bool kgen_writes_residues(const KgenArgs& a, int n) { return a.dp <= 32 && n == 14 && !a.dlogell; }
hipError_t launch_kgen(const KgenArgs& a, hipStream_t s);
// NLML gradient reduction: Σ_ij (Kinv − ααᵀ)_ij ∂K_ij/∂log ℓ over the lower tiles, plus tr(Kinv), αᵀα, αᵀδ
struct NlmlGradArgs {
    const double* Xs;      // [Np][dp]
    const double* Kinv;    // [Np][ld]  lower 128-tiles valid
    const double* alpha;   // [Np]
    const double* delta;   // [Np]
    double* partial;       // [Np/16]
    double* out;           // [4]
    int64_t ld;
    int N, Np, dp, family;
    double sigma_f2;
};
hipError_t launch_nlml_grad(const NlmlGradArgs& a, hipStream_t s);
// same reductions with dK/dlog(ell) given as a matrix D (gradient-enhanced GP: D comes from kgen with dlogell = 1):
// out[0] = sum_{i,k<N} (Kinv[i][k] - alpha_i alpha_k) D[i][k]  (lower 128-tiles, strictly-lower ones counted twice), out[1..3] as above
hipError_t launch_nlml_grad_matrix(const NlmlGradArgs& a, const double* D, int64_t ldd, hipStream_t s);
// test hook: out[i] = kappa(family, d2[i]) with the device math the generator uses
hipError_t launch_kappa_test(int family, const double* d2, double* out, int64_t n, hipStream_t s);
// K[i][i] += noise for i < N; K[i][i] = 1 for N ≤ i < Np (identity padding keeps the factor PD)
hipError_t launch_diag_fix(double* K, int64_t ld, int N, int Np, double noise, hipStream_t s, int64_t* info_reset = nullptr);
// A[i][i] = v for lo ≤ i < hi
hipError_t launch_set_diag(double* A, int64_t ld, int lo, int hi, double v, hipStream_t s);
// Xs[i][c] = X[i][c]·s (zero padded to [Np][dp])
hipError_t launch_scale_points(const double* X, double* Xs, int N, int Np, int d, int dp, double s, hipStream_t st);
// a bordered append's new observation (x on the HOST, d ≤ APPEND_POINT_MAXD) → rows of Xraw / Xs, y, δ = y − m, *info = 0: one launch
constexpr int APPEND_POINT_MAXD = 64;
hipError_t launch_append_point(const double* x_host, int d, int dp, double s, double y, double mean_c, double* Xraw_row, double* Xs_row,
                               double* y_at, double* delta_at, int64_t* info, hipStream_t st);
// everything in front of a fit's kernel matrix in one launch: the model's copies of the caller's X / y (when Xsrc / ysrc are not
// those copies already), scaled zero-padded points, centred targets, alpha = 0
hipError_t launch_fit_prep(const double* Xsrc, const double* ysrc, double* Xraw, double* ybuf, double* Xs, double* delta, double* alpha,
                           int N, int Np, int d, int dp, double s, double mean_c, hipStream_t st);

// ---- factorisation pieces (chol.hip) --------------------------------------------------------
// factor the 128×128 diagonal block at (r0,r0) of K in place (lower), write its inverse into the
// matching diagonal block of W (lower) and WT (upper = transposed); on a non-positive pivot set
// *info = r0 + j + 1 (if still 0) and leave.
hipError_t launch_chol_diag(double* K, double* W, double* WT, int64_t ld, int r0, int64_t* info, hipStream_t s);
// The whole fit of an N ≤ 128, d ≤ 16 model in ONE launch of the diagonal-block kernel (mode 3): writes Xs [128][dp], delta,
// L (→ K), L⁻¹ (→ W, WT), alpha, scal = {log det, δᵀα}; K / W / WT have leading dimension 128.  *info as launch_chol_diag.
struct FitSmallArgs {
    const double* Xraw;   // [N][d] raw inputs (device): the model's copy, or the caller's array when Xkeep is set
    const double* y;      // [N]
    double* Xkeep;        // non-null: the kernel also writes the model's own copies of the inputs it read (the caller's device
    double* ykeep;        //           arrays go straight into the one launch: no staging copies in front of it)
    double* Xs;           // [128][dp]
    double* delta;        // [128]
    double* alpha;        // [128]
    double* scal;         // [2]
    int N, d, dp, family;
    double s, sigma_f2, noise, mean_c;
};
hipError_t launch_fit_small(double* K, double* W, double* WT, int64_t* info, const FitSmallArgs& fs, hipStream_t s);
// the same split for the panel chain: potf2 (factor + the eight 16×16 diagonal sub-block inverses, which go to their final
// places in W / WT), the panel solve below the block as a blocked triangular solve on L (rows r0+128 … r0+128+nrows), and —
// once, after the factorisation — the 128×128 inverses of ALL diagonal blocks in one batched launch
hipError_t launch_potf2_diag(double* K, double* W, double* WT, int64_t ld, int r0, int64_t* info, hipStream_t s, double* P, bool zero_rows = false);
constexpr size_t TRSM_STREAM_BYTES = 144 * 64 * sizeof(double);      // the packed operands potf2_pipe_kernel leaves for trsm_stream_kernel
hipError_t launch_trsm_stream(double* K, const double* P, int64_t ld, int r0, int nrows, const int64_t* info, hipStream_t s);
hipError_t launch_trtri_diag_batched(double* K, double* W, double* WT, int64_t ld, int nblocks, int64_t* info, hipStream_t s);
// out[i] = Σ_{k ≤ i} Wm[i][k]·v[k]   (lower == 1)   or   Σ_{k ≥ i} Wm[i][k]·v[k]   (lower == 0)
hipError_t launch_trmv(const double* Wm, int64_t ld, const double* v, double* out, int Np, int lower, hipStream_t s);
// out[0] = 2·Σ_{i<N} log L[i][i];  out[1] = Σ_{i<N} delta[i]·alpha[i]
hipError_t launch_nlml_terms(const double* L, int64_t ld, const double* delta, const double* alpha, int N, double* out, hipStream_t s);
// delta[i] = y[i] − c (i < N), 0 for padding
hipError_t launch_center(const double* y, double* delta, int N, int Np, double c, hipStream_t s);
// gradient-enhanced GP: targets arrive by outputs (y_abi[q·N + i], prep_output of the reference) and are kept point-major
// (ybuf[i·p + q]); delta[r] = ybuf[r] − mean[r % p] for r < N·p, 0 up to Np
struct MeanVec { double c[MAX_P]; };
hipError_t launch_center_grad(const double* y_abi, double* ybuf, double* delta, int N, int p, int Np, MeanVec mean, int y_point_major,
                              hipStream_t s);

// bordered append (chol.hip): given k = k(X,x*), l = W·k, v = Wᵀ·l, write row N of L and W, column N of WT,
// the new alpha and the down-date vector vext = [−v ; 1]; scal = {l_nn², β, l_nn, kᵀα}; *info = N+1 if l_nn² ≤ 0
struct AppendArgs {
    double* L; double* W; double* WT;
    int64_t ld;
    const double* krow; const double* lvec; const double* vvec;
    const double* alpha_old; double* alpha_new; double* vext;
    const double* delta;
    int N, cap;
    double kss;              // k(x*,x*) + noise
    double* scal;
    int64_t* info;
};
hipError_t launch_append(const AppendArgs& a, hipStream_t s);

// ---- posterior epilogue + selection (misc.hip) ----------------------------------------------
struct FinalizeArgs {
    const double* partial;   // [T][ldp]
    const double* mu_in;     // [Mc]
    double* mu_out;          // [M] or nullptr (global arrays, indexed j0 + j)
    double* var_out;         // [M] or nullptr
    double* score_out;       // [M] or nullptr
    int64_t ldp, j0, M;
    int T, Mc;
    int kind;                // ABO_ACQ_*, or −1 for none
    double sigma_f2, p0, best_y;
    // gradient-enhanced GP: rows carry pc outputs of Mpts points; prior variance of a gradient output
    double prior_grad = 0.0;
    int pc = 1, point_major = 0;
    int64_t Mpts = 0;
};
hipError_t launch_finalize(const FinalizeArgs& a, hipStream_t s);

// gradient-enhanced GP: per-point p×p posterior covariance (+ mean, + GradientNormUCB score) from V = L⁻¹K_XZ
struct GradCovArgs {
    const double* V;          // [points·p][ldv], point-major rows
    const double* mu_rows;    // [points·p]
    int64_t ldv;
    int R, p;
    int64_t pt0;              // global index of the chunk's first point
    double prior0, prior_g, beta;
    double* cov_out;          // [M][p][p] or nullptr
    double* mu_out;           // [M][p] or nullptr
    double* score_out;        // [M] or nullptr
};
hipError_t launch_grad_cov(const GradCovArgs& a, int npoints, hipStream_t s);

// resident-candidate kernels (C5)
// mu[j] += c[j]·beta ; var[j] −= c[j]²/s2        (posterior down-date after a bordered append)
hipError_t launch_downdate(double* mu, double* var, const double* c, int64_t M, double beta, double s2, hipStream_t s);
// resident K_ZX (candidate-major, ld doubles per candidate) of a candidate set:
//   newcol: Kzx[j][col] = sigma_f2·kappa(‖s·z_j − Xs[col]‖²) for all M candidates (the column of a just-appended point)
//   gemv:   c[j] = Σ_{k<n} Kzx[j][k]·v[k]   (the O(N·M) down-date as one streaming pass, no kernel evaluations)
hipError_t launch_cand_newcol(const double* Xs, const double* Z, double* Kzx, int64_t ld, int64_t M, int col, int d, int dp,
                              int family, double s, double sigma_f2, hipStream_t st);
// gradient-enhanced model: column `col` = training row (point pt, output q) against the function value of every candidate
hipError_t launch_cand_newcol_grad(const double* Xs, const double* Z, double* Kzx, int64_t ld, int64_t M, int col, int pt, int q,
                                   int d, int dp, int family, double s, double sigma_f2, hipStream_t st);
hipError_t launch_cand_gemv(const double* Kzx, int64_t ld, const double* v, int n, int64_t M, double* c, hipStream_t s);
// score[j] = acq(mu[j], var[j])
hipError_t launch_score(const double* mu, const double* var, double* score, int64_t M, int kind, double p0, double best_y,
                        hipStream_t s);

// rec = {tv[0], (double)ti[0], mu[ti[0] − idx_base], Z[ti[0] − idx_base][0..d)} (zeros for ti[0] < 0): a device's pick
// record of one greedy q-EI sub-step, 3 + d doubles
hipError_t launch_pick_record(const double* tv, const int64_t* ti, int64_t idx_base, const double* Z, const double* mu, int d,
                              double* rec, hipStream_t s);

// ---- block form of greedy q-EI (qei.hip): covariance columns of T points from ONE pass over the resident K_ZX ----------
constexpr int QEI_MAXQ = 64;       // picks of one batch (chain vectors kept; their coefficients travel as kernel arguments)
constexpr int QEI_MAXT = 64;       // points of one block (the skinny product holds at most four 16-row groups)
// KXT[t][i] = sigma_f2·kappa(‖Xs_i − s·P_t‖²) for i < N, t < T; zeros for N ≤ i < Np and for T ≤ t < rows.  P: raw points [T][d]
hipError_t launch_qei_kxt(const double* Xs, int dp, int N, int Np, const double* P, int d, int T, int rows, int family, double s,
                          double sigma_f2, double* KXT, hipStream_t st);
// V[r][N … Np) = 0 for r < rows (V: [rows][ld])
hipError_t launch_qei_zero_tail(double* V, int64_t ld, int N, int Np, int rows, hipStream_t st);
// C[t][j] += sigma_f2·kappa(‖Ps_t − s·z_j‖²) for t < T, j < M   (Ps: the block points pre-scaled, [T][dp]; C: [T][Mp])
hipError_t launch_qei_cov(const double* Ps, int dp, const double* Z, int64_t M, int64_t Mp, int d, int T, int family, double s,
                          double sigma_f2, double* C, hipStream_t st);
// C[r][j] = alpha·Σ_{k<K} A[r][k]·B[j][k], r < rows16 (16 … 64, a multiple of 16), j < nB (a multiple of 128), K a multiple of 128:
// the one pass over a long B (gemm.hip: qei_pass_kernel; same bits as the skinny split-k kernel with one chunk)
// kmode K_B_LOWER / K_B_UPPER with ksplit (a multiple of 128): the split-k form against a triangular B — chunk z of the k range of B's row
// block tj → the partial product C + z·sC (launch_splitk_reduce sums them)
hipError_t launch_qei_pass(const double* A, int64_t lda, int rows16, const double* B, int64_t ldb, int64_t nB, int K, double alpha,
                           double* C, int64_t ldc, hipStream_t s, int kmode = K_FULL, int ksplit = 0, int64_t sC = 0);
// out[j] = blk[j] − Σ_{first ≤ i < nchain} gam[i]·chain[i][j] ;  var[j] −= out[j]²/s  (var == nullptr: the column only)
struct QeiPickArgs {
    const double* blk;      // [M]   base covariances of the picked point (a row of the block)
    const double* chain;    // [nchain][Mp]
    double* out;            // [M]   usually chain + nchain·Mp
    double* var;            // [M] or nullptr
    int64_t M, Mp;
    int first, nchain;      // chain entries first … nchain − 1 correct the block's column
    double s;
    double gam[QEI_MAXQ];
};
hipError_t launch_qei_pick(const QeiPickArgs& a, hipStream_t st);
// rec[e] = {tv[e], (double)ti[e], mu, var, Z[0..d), chain_0 … chain_{nchain−1}} of the candidate ti[e] − idx_base (zeros behind a
// negative index), e < k; 4 + d + nchain doubles per record
hipError_t launch_qei_record(const double* tv, const int64_t* ti, int k, int64_t idx_base, const double* Z, const double* mu,
                             const double* var, const double* chain, int64_t Mp, int nchain, int d, double* rec, hipStream_t st);

// ---- the pick loop of a block-form batch on ONE handle, without the host (qei.hip: qei_step_kernel; ABI 7) -------------------------
// Launch k of a batch of q picks (k = 0 … q) finishes pick k − 1 and selects pick k:
//   (B) every workgroup reduces the ≤ QEI_STEP_MAXWG partial arg-maxima launch k − 1 left (strict total order: any reduction tree
//       gives the same winner), workgroup 0 writes the pick's record, every workgroup looks the winner up in the slot table, forms
//       s = σ²(x) + σ²_n and γ_i = c_i(x)/s_i and applies c = C₀[slot] − Σ γ_i c_i, σ² −= c²/s to its candidates (the arithmetic of
//       qei_pick_kernel), then
//   (C) scores its candidates (EI) and leaves its partial arg-max {key, index, μ, σ²} for launch k + 1.
// Launch 0 also snapshots (μ, σ²) — it reads them anyway —, the tail launch (k = q) writes the last record and rolls them back: the
// batch needs no copy of its own before or behind it.
// No workgroup waits for another inside a launch (stream order is the only synchronisation): no atomics, no fences — bit-reproducible.
// A pick outside every block, or s ≤ 0, raises st->stop: the remaining launches return at once, the host builds the block (or reports
// the failed pivot) and resumes from launch k.
constexpr int QEI_STEP_MAXWG = 1024;
struct QeiStepState {                    // device memory, one per candidate set
    int32_t stop;                        // 0: running; 1: pick stop_at is in no block; 2: s ≤ 0 at pick stop_at
    int32_t stop_at;
    double s_batch[QEI_MAXQ];            // s of the picks this batch conditioned on (entry t: the batch's pick t)
};
struct QeiStepPartial { uint64_t key; int64_t idx; double mu, var; };
struct QeiStepArgs {
    double* mu; double* var;             // [M] stored posterior of the set (σ² is conditioned in place; the batch rolls it back)
    double* snap_mu; double* snap_var;   // [M] launch 0 copies (μ, σ²) here, the tail launch (k = q) copies them back
    const double* Z;                     // [M][d]
    const double* blk;                   // [slots][Mp] block columns
    double* chain;                       // [rows][Mp]
    QeiStepState* st;
    QeiStepPartial* part;                // [2][QEI_STEP_MAXWG]: launch k writes half k & 1, reads half (k − 1) & 1
    double* rec;                         // [q][wmax] records {EI, global index, μ, σ², x[d], c_1(x) … c_n(x)}
    int64_t M, Mp, idx_base;
    int d, T16, nslots, wmax;
    int k, q;                            // this launch; picks of the batch
    int n0;                              // chain entries when the batch's pick 0 was selected (real entries carried over)
    int nwg_prev;                        // workgroups of launch k − 1 (partials to reduce)
    int distinct;
    double xi, best_y, noise;
    double chain_s0[QEI_MAXQ];           // s_i of the chain entries i < n0 (host knows them)
    int blk_base[4];                     // per block: chain entries already in its columns
    int64_t slot_gidx[4 * QEI_MAXT];     // global candidate index per block row (−1: empty)
};
hipError_t launch_qei_step(const QeiStepArgs& a, int nwg, hipStream_t st);

// out[0] = max_s min_i ‖S_s − X_i‖ (out: one double in device memory)
hipError_t launch_fill_distance(const double* X, int64_t N, int d, const double* S, int64_t ns, double* out, hipStream_t s);
// Z[(j−j0)·d + c] for j in [j0, j0+count): Latin-hypercube points of an n-point design (device lower/upper)
hipError_t launch_lhs(double* Z, int64_t n, int d, const double* lower, const double* upper, uint64_t seed, int64_t j0,
                      int64_t count, hipStream_t s);

// ---- objectives of the acquisition stage: a weighted sum of epilogues on ONE posterior evaluation ------------------
// f(x) = Σ_t w_t · acq_t(x)  (EnsembleAcquisition, EnsembleAcq.jl:53-55; a plain acquisition function is one term of weight 1).
// acq_t ∈ {EI, UCB, PI, MEAN} on the function-value posterior (μ, σ²), or GRADNORM_UCB on the posterior of the gradient outputs
// of a gradient-enhanced model (gradNormUCB.jl:43-51).
constexpr int MAX_TERMS = 8;
constexpr int ACQ_GRADNORM_UCB = 4;            // == ABO_ACQ_GRADNORM_UCB (include/abo_hip.h)
struct AcqTerms {
    int n;
    int kind[MAX_TERMS];
    double p0[MAX_TERMS], best_y[MAX_TERMS], w[MAX_TERMS];
};
inline bool terms_have_gradnorm(const AcqTerms& t) {
    for (int i = 0; i < t.n; ++i) if (t.kind[i] == ACQ_GRADNORM_UCB) return true;
    return false;
}
inline bool terms_plain(const AcqTerms& t) { return t.n == 1 && t.w[0] == 1.0 && t.kind[0] != ACQ_GRADNORM_UCB; }
// score[j] = Σ_t w_t·acq_t(mu[j], var[j]) (function-value terms only)
hipError_t launch_score_terms(const double* mu, const double* var, double* score, int64_t M, const AcqTerms& t, hipStream_t s);
// gradient-enhanced model: score[j] from the per-point mean mu[j][p] and covariance block cov[j][p][p]
hipError_t launch_score_terms_grad(const double* mu, const double* cov, double* score, int64_t M, int p, const AcqTerms& t, hipStream_t s);

// ---- local refinement of optimize_acquisition on the device (refine.hip) ---------------------------------------
struct RefineArgs {
    const double* Xs;      // [Np][dp] scaled training points
    const double* W;       // L⁻¹ (lower), WT = its transpose (upper); leading dimension ld
    const double* WT;
    const double* alpha;   // [Np]
    int64_t ld;
    int N, Np, d, dp, family;
    double s, sigma_f2, mean_c;            // 1/ℓ, σ_f², prior mean
    AcqTerms terms;                        // the objective
    const double* lower;   // device [d]
    const double* upper;   // device [d]
    const double* starts;  // device [S][d]
    double* x_out;         // device [S][d]   (grad_only: the gradient)
    double* f_out;         // device [S]
    int* iters_out;        // device [S][2] = {iterations, evaluations} or nullptr
    double* scratch;       // device [S][4][Np]
    int max_iter, ls_max, history;
    double g_tol, f_abstol, x_abstol;
};
size_t refine_lds_bytes(int d, int dp, int history);
// one workgroup per start: the whole projected L-BFGS of that start in one launch (grad_only = 1: one evaluation per point,
// x_out receives the gradient)
hipError_t launch_refine(const RefineArgs& a, int S, int grad_only, hipStream_t s);
// the same refinement in lockstep rounds with the evaluations of a round batched on the fp64 MFMA tile core (large N: L⁻¹ is read
// once per round instead of once per start); work = refine_lockstep_bytes(S, Np, d, history) of device memory
size_t refine_lockstep_bytes(int S, int Np, int d, int history);
hipError_t launch_refine_lockstep(const RefineArgs& a, int S, void* work, hipStream_t s);
// Gradient-enhanced models (and any objective with a GRADNORM_UCB term): the same lockstep state machine, a round's evaluations
// delivered by the caller.  eval(points [npts][d] (device), npts, mu [npts][p], cov [npts][p][p]) queues the all-output posterior of
// the round's points on the stream (api.hip: the kernels behind abo_predict_grad_cov); the driver derives value and gradient of the
// objective from it — function-value terms analytically (∇μ = E[∇f] − m_∇, ∇σ² = 2·Cov(f, ∇f)), GRADNORM_UCB terms by central
// differences over a 2d-point stencil evaluated in the same batch (the reference differentiates every objective that way,
// acq_utils.jl:55-71).  p = d + 1 outputs; mean_g[d]: prior means of the gradient outputs.
size_t refine_lockstep_grad_bytes(int S, int d, int history, bool stencil);
struct GradEval {
    void* ctx;
    hipError_t (*eval)(void* ctx, const double* pts, int npts, double* mu, double* cov, hipStream_t s);
};
hipError_t launch_refine_lockstep_grad(const RefineArgs& a, int S, const double* mean_g, void* work, const GradEval& ev, hipStream_t s);
// value and gradient of the objective at S points through the same evaluation (the test hook behind abo_test_acq_grad)
hipError_t launch_acq_grad_via_eval(const RefineArgs& a, int S, const double* mean_g, void* work, const GradEval& ev, hipStream_t s);
// out[j][0..d) = Z[idx[j] − idx_base][0..d) for j < k (zeros for idx[j] < 0): the coordinates of selected candidates
hipError_t launch_gather_points(const double* Z, const int64_t* idx, int64_t idx_base, int k, int d, double* out, hipStream_t s);

struct TopkWork {            // scratch sized by topk_workspace_entries()
    uint64_t* keys[2];
    int64_t* idx[2];
};
int64_t topk_workspace_entries(int64_t M, int k);
// top-k of scores[0..M) in Julia's `sortperm(scores; rev=true)` order; writes k (val, idx) pairs
// (idx + idx_base; tail (NaN, −1) when M < k) to device arrays top_val/top_idx.
hipError_t launch_topk(const double* scores, int64_t M, int k, int64_t idx_base, TopkWork w,
                       double* top_val, int64_t* top_idx, hipStream_t s);

}  // namespace abo
