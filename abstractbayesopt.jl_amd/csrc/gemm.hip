// fp64 MFMA tile core for gfx950 and the kernels built on it:
//   gemm_nt_kernel      C = alpha·A·Bᵀ + beta·C   (TRSM panel, SYRK trailing update, blocked L⁻¹, K⁻¹, V = L⁻¹K_XZ)
//   var_gemm256s_kernel partial[ti][j] = Σ_{i∈ti} (W·K_XZ)[i][j]²  — the N²·M contraction that is
//   var_gemm256_kernel  >99 % of the flops of posterior_var (reference: src/surrogates/StandardGP.jl:377-379,
//   var_gemm_kernel     [upstream AbstractGPs] diag_Xt_invA_X(C, K_XZ)); 256×128 tile per 8-wave workgroup with the
//                       zero sub-tiles of the diagonal blocks skipped (production), the same tile without the
//                       skipping (A/B reference), and a 128×128 tile per 4-wave workgroup (Np an odd number of blocks).
//
// Design (CDNA4):
//  * v_mfma_f64_16x16x4_f64 issues every 64 cycles per SIMD (measured: profiles/r01_mfma_f64_probe.txt,
//    77.8 TFLOP/s chip-wide) — the fp64 matrix pipe is the roofline; everything else has to stay
//    out of its way.  The accumulators must be VGPR-form (-mllvm -amdgpu-mfma-vgpr-form /
//    launch_bounds(256,2)); the AGPR form makes hipcc copy 64 registers in and out per k-step.
//  * 128×128 output tile per 256-thread workgroup (4 waves as 2×2, 64×64 per wave = 4×4 MFMA tiles,
//    128 accumulator VGPRs), BK = 16 per LDS stage.  Both operands are k-contiguous ("NT" form), so
//    one 16-byte LDS read per lane feeds two MFMA k-steps: lane l reads [row l&15][k = 2(l>>4), +1].
//    The k-permutation this implies is identical for A and B, so the products pair up correctly.
//  * LDS rows are padded 16 → 20 doubles (160 B): with that stride the four 16-lane groups of a
//    ds_read_b128 each touch 16 distinct 16-byte slots (conflict-free, SQ_LDS_BANK_CONFLICT = 0),
//    and rows stay 16-B aligned for ds_write_b128.
//  * Software pipeline with ONE barrier per stage: LDS holds two stages (80 KB per workgroup, two
//    workgroups per CU), registers hold two half-stage fragment sets and the staging registers of the
//    global loads that run two stages ahead.
//  * Every LDS/VMEM instruction is issued directly behind one of the wave's own MFMAs
//    (sched_group_barrier, steady-state iteration = one basic block).  Measured
//    (profiles/r01_var_gemm_ablation.txt): the same instructions issued as a burst cost 6 % of the
//    kernel — a wave that is not issuing MFMAs advances about one instruction per MFMA slot of its
//    SIMD partner — while dealt out one per MFMA they cost nothing; a strict ping-pong of MFMA and
//    memory roles between the two waves of a SIMD is slower still (the memory phase becomes as long
//    as the MFMA phase).  Static s_setprio between the co-resident workgroups changes nothing.
#include <type_traits>
#include "abo_kernels.h"
#include <cstdlib>

namespace abo {

typedef double d2_t __attribute__((ext_vector_type(2)));
typedef double d4_t __attribute__((ext_vector_type(4)));

constexpr int BM = 128, BN = 128, BK = 16;
constexpr int LDT = 20;                    // LDS row stride in doubles (16 + 4 pad)
constexpr int TILE = BM * LDT;             // doubles per operand stage

__device__ __forceinline__ void tile_gload(const double* __restrict__ g, int64_t ld, int k0, d2_t (&r)[4]) {
    const int t = threadIdx.x;
    const int row = t >> 3, kk = (t & 7) * 2;
    const double* p = g + (int64_t)row * ld + k0 + kk;
#pragma unroll
    for (int q = 0; q < 4; ++q) r[q] = *reinterpret_cast<const d2_t*>(p + (int64_t)(32 * q) * ld);
}

__device__ __forceinline__ void tile_lstore(double* s, const d2_t (&r)[4]) {
    const int t = threadIdx.x;
    const int row = t >> 3, kk = (t & 7) * 2;
#pragma unroll
    for (int q = 0; q < 4; ++q) *reinterpret_cast<d2_t*>(s + (row + 32 * q) * LDT + kk) = r[q];
}

// fragments of one half stage (8 of the 16 k): 4 A rows-of-16 and 4 B rows-of-16, two k each
struct Frag {
    d2_t a[4], b[4];
};

__device__ __forceinline__ void frag_read(const double* ap, const double* bp, int half, Frag& f) {
#pragma unroll
    for (int i = 0; i < 4; ++i) f.a[i] = *reinterpret_cast<const d2_t*>(ap + i * 16 * LDT + half * 8);
#pragma unroll
    for (int i = 0; i < 4; ++i) f.b[i] = *reinterpret_cast<const d2_t*>(bp + i * 16 * LDT + half * 8);
}

// 16 MFMAs: one k-step (kk = 0 or 1 of the fragment pair) × 4×4 tiles
template <int KK>
__device__ __forceinline__ void frag_mma(const Frag& f, d4_t (&acc)[4][4]) {
#pragma unroll
    for (int mi = 0; mi < 4; ++mi)
#pragma unroll
        for (int ni = 0; ni < 4; ++ni)
            acc[mi][ni] = __builtin_amdgcn_mfma_f64_16x16x4f64(f.a[mi][KK], f.b[ni][KK], acc[mi][ni], 0, 0, 0);
}

// acc += A[0:128][kbeg:kend] · B[0:128][kbeg:kend]ᵀ ; Ag/Bg point at the tile's first row.
//
// iteration t (stage = 16 k):
//     32 MFMA on F0 = fragments(t, half 0), and behind one MFMA each:
//         8 ds_write   S(stage t+1) → LDS[(t+1)&1]
//         8 ds_read    fragments(t, half 1) → F1
//         8 global_load stage t+2 → S
//     barrier          (stage t+1 complete in LDS; nobody still reads stage t)
//     32 MFMA on F1, and behind the first eight: 8 ds_read fragments(t+1, half 0) → F0
// The last two stages of a tile (nothing left to prefetch) run a plain version of the same steps.
__device__ __forceinline__ void tile_loop(const double* __restrict__ Ag, int64_t lda, const double* __restrict__ Bg,
                                          int64_t ldb, int kbeg, int kend, double* smem, d4_t (&acc)[4][4]) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int r16 = lane & 15, g = lane >> 4;
    const int aoff = (wm * 64 + r16) * LDT + g * 2;
    const int boff = TILE + (wn * 64 + r16) * LDT + g * 2;
    const int nk = (kend - kbeg) / BK;
    if (nk <= 0) return;
    d2_t sa[4], sb[4];
    Frag f0, f1;
    tile_gload(Ag, lda, kbeg, sa);
    tile_gload(Bg, ldb, kbeg, sb);
    tile_lstore(smem, sa);
    tile_lstore(smem + TILE, sb);
    if (nk > 1) {
        tile_gload(Ag, lda, kbeg + BK, sa);
        tile_gload(Bg, ldb, kbeg + BK, sb);
    }
    __syncthreads();
    frag_read(smem + aoff, smem + boff, 0, f0);
    int t = 0;
    {
        // steady state (stages t+1 and t+2 exist): one basic block per iteration
        for (; t + 2 < nk; ++t) {
            double* cur = smem + (t & 1) * (2 * TILE);
            double* nxt = smem + ((t + 1) & 1) * (2 * TILE);
            tile_lstore(nxt, sa);
            tile_lstore(nxt + TILE, sb);
            frag_read(cur + aoff, cur + boff, 1, f1);
            tile_gload(Ag, lda, kbeg + (t + 2) * BK, sa);
            tile_gload(Bg, ldb, kbeg + (t + 2) * BK, sb);
            frag_mma<0>(f0, acc);
            frag_mma<1>(f0, acc);
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);   // MFMA
                __builtin_amdgcn_sched_group_barrier(0x200, 1, 0);   // DS write
            }
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);   // DS read
            }
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);   // VMEM read
            }
            __builtin_amdgcn_sched_group_barrier(0x008, 8, 0);
            __builtin_amdgcn_sched_barrier(0);
            __syncthreads();
            frag_read(nxt + aoff, nxt + boff, 0, f0);
            frag_mma<0>(f1, acc);
            frag_mma<1>(f1, acc);
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 1);
                __builtin_amdgcn_sched_group_barrier(0x100, 1, 1);
            }
            __builtin_amdgcn_sched_group_barrier(0x008, 24, 1);
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    for (; t < nk; ++t) {
        double* cur = smem + (t & 1) * (2 * TILE);
        double* nxt = smem + ((t + 1) & 1) * (2 * TILE);
        frag_mma<0>(f0, acc);                                      // (c) first 16
        // The sched_barriers pin what hipcc otherwise undoes: every LDS/global operation of the
        // iteration is issued behind 16 MFMAs (issued at the top they put an lgkmcnt wait on
        // themselves in front of the first MFMA: the counter is in-order and 4 bits wide), and the
        // barrier stays behind all 32 MFMAs of F0 (hoisted, every wave sits out the LDS round trip
        // with an empty matrix pipe).
        __builtin_amdgcn_sched_barrier(0);
        frag_read(cur + aoff, cur + boff, 1, f1);                  // (a)
        if (t + 1 < nk) {                          // (b) stage t+1 → LDS, stage t+2 → S
            tile_lstore(nxt, sa);
            tile_lstore(nxt + TILE, sb);
            if (t + 2 < nk) {
                tile_gload(Ag, lda, kbeg + (t + 2) * BK, sa);
                tile_gload(Bg, ldb, kbeg + (t + 2) * BK, sb);
            }
        }
        __builtin_amdgcn_sched_barrier(0);
        frag_mma<1>(f0, acc);                                      // (c) second 16
        __builtin_amdgcn_sched_barrier(0);
        __syncthreads();                                           // (d)
        if (t + 1 < nk) frag_read(nxt + aoff, nxt + boff, 0, f0);  // (e)
        __builtin_amdgcn_sched_barrier(0);
        frag_mma<0>(f1, acc);                                      // (f)
        frag_mma<1>(f1, acc);
    }
    __syncthreads();   // callers reuse the LDS
}

__device__ __forceinline__ void acc_zero(d4_t (&acc)[4][4]) {
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = d4_t{0.0, 0.0, 0.0, 0.0};
}

// ------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256, 2) gemm_nt_kernel(GemmArgs p) {
    __shared__ __attribute__((aligned(16))) double smem[4 * TILE];
    if (p.info != nullptr && *p.info != 0) return;
    // heaviest tiles first: with K_A_LOWER the k range of a tile grows with its row block, so the row blocks are
    // dealt out from the bottom up (the light tiles then fill the tail of the launch instead of the heavy ones forming it);
    // K_A_UPPER already starts with its longest rows
    // (K_B_LOWER: B is the lower-triangular operand — the column blocks are dealt out from the right for the same reason)
    int tj = p.kmode == K_B_LOWER ? (int)gridDim.x - 1 - (int)blockIdx.x : (int)blockIdx.x;
    int ti = p.kmode == K_A_LOWER ? (int)gridDim.y - 1 - (int)blockIdx.y : (int)blockIdx.y;
    const int bz = blockIdx.z;
    if (p.swz) {
        // Square SYRK, 1-D launch over its lower tiles only.  Workgroups go round-robin to the 8 XCDs, each with its own L2: XCD x
        // takes the x-th eighth of the tile sequence, and the sequence runs through super-rows of G tile rows column by column — the
        // 64 workgroups an XCD holds at a time are a patch of G rows × 64/G columns (G = 4: 20 operand panels for 64 tiles),
        // where the row-major 2-D launch had every XCD touch every panel of 8 tile rows (its L2 held none of them until reuse).
        const int L = blockIdx.x;
        const int per = (p.swz + 7) >> 3;
        const int Q = (L & 7) * per + (L >> 3);
        if (Q >= p.swz) return;
        const int T = p.M / BM, G = p.swz_g;
        int g = (int)((sqrt(8.0 * (double)Q + 1.0) - 1.0) / (2.0 * G));        // super-row: G·g·(G·g+1)/2 tiles lie before it
        while ((G * (g + 1)) * (G * (g + 1) + 1) / 2 <= Q) ++g;
        while ((G * g) * (G * g + 1) / 2 > Q) --g;
        int w = Q - (G * g) * (G * g + 1) / 2;
        const int r0 = G * g;
        const int R = min(G, T - r0);
        if (w < R * (r0 + 1)) {                                              // full columns 0 … r0: R tiles each
            tj = w / R;
            ti = r0 + w - tj * R;
        } else {                                                             // the triangle at the right end
            w -= R * (r0 + 1);
            int c = 1;
            while (w >= R - c) { w -= R - c; ++c; }
            tj = r0 + c;
            ti = tj + w;
        }
    }
    if (p.lower_only && tj > ti) return;
    int kbeg = 0, kend = p.K;
    if (p.kmode == K_A_LOWER) kend = min(p.K, (ti + 1) * BM);
    if (p.kmode == K_A_UPPER) kbeg = min(p.K, ti * BM);
    if (p.kmode == K_B_LOWER) kend = min(p.K, (tj + 1) * BN);
    if (p.kmode == K_B_UPPER) kbeg = min(p.K, tj * BN);
    const double* Ag = p.A + (p.ksplit ? 0 : (int64_t)bz * p.sA) + (int64_t)ti * BM * p.lda;
    const double* Bg = p.B + (p.ksplit ? 0 : (int64_t)bz * p.sB) + (int64_t)tj * BN * p.ldb;
    if (p.ksplit) {                                  // chunk bz of this tile's k range; an empty chunk writes nothing
        kbeg = max(kbeg, bz * p.ksplit);
        kend = min(kend, (bz + 1) * p.ksplit);
        if (kbeg >= kend) return;
    }
    d4_t acc[4][4];
    acc_zero(acc);
    tile_loop(Ag, p.lda, Bg, p.ldb, kbeg, kend, smem, acc);

    // C/D map of v_mfma_f64_16x16x4_f64: lane l, reg r -> row (l>>4) + 4r, col l&15
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int r16 = lane & 15, g = lane >> 4;
    double* Cg = p.C + (int64_t)bz * p.sC;
    double* Ctg = p.Ct ? p.Ct + (int64_t)bz * p.sCt : nullptr;
#pragma unroll
    for (int mi = 0; mi < 4; ++mi)
#pragma unroll
        for (int ni = 0; ni < 4; ++ni)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int64_t row = (int64_t)ti * BM + wm * 64 + mi * 16 + g + 4 * r;
                const int64_t col = (int64_t)tj * BN + wn * 64 + ni * 16 + r16;
                double v = p.alpha * acc[mi][ni][r];
                if (p.beta != 0.0) v += p.beta * Cg[row * p.ldc + col];
                Cg[row * p.ldc + col] = v;
                if (Ctg) Ctg[col * p.ldct + row] = v;
            }
}

// Same product for launches with few 128×128 tiles (late Cholesky panels, the whole factorisation at N ≲ 2048,
// the first levels of the blocked L⁻¹).  One CU delivers 0.3 TFLOP/s of fp64 MFMA, so a 128×128×128 tile is
// 14 µs of matrix-pipe time wherever it runs; when the launch cannot give a quarter of the CUs a tile, each tile is
// cut into sixteen 32×32 workgroups (4 waves, one MFMA tile each) instead.  Nothing to share → no LDS and no
// barriers: a lane loads its own fragments (16 B per MFMA pair) straight from L2.  k runs through the MFMAs in
// the same order and with the same lane ↔ k map as tile_loop, so the result is bit-identical to gemm_nt_kernel;
// kmode / lower_only keep their 128-tile granularity.
__global__ void __launch_bounds__(256) gemm_nt_small_kernel(GemmArgs p) {
    if (p.info != nullptr && *p.info != 0) return;
    const int sj = blockIdx.x, si = blockIdx.y, bz = blockIdx.z;
    const int ti = si >> 2, tj = sj >> 2;
    if (p.lower_only && tj > ti) return;
    int kbeg = 0, kend = p.K;
    if (p.kmode == K_A_LOWER) kend = min(p.K, (ti + 1) * BM);
    if (p.kmode == K_A_UPPER) kbeg = min(p.K, ti * BM);
    if (p.kmode == K_B_LOWER) kend = min(p.K, (tj + 1) * BN);
    if (p.kmode == K_B_UPPER) kbeg = min(p.K, tj * BN);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int r16 = lane & 15, g = lane >> 4;
    const double* Ap = p.A + (int64_t)bz * p.sA + (int64_t)(si * 32 + wm * 16 + r16) * p.lda + 2 * g;
    const double* Bp = p.B + (int64_t)bz * p.sB + (int64_t)(sj * 32 + wn * 16 + r16) * p.ldb + 2 * g;
    double* Cg = p.C + (int64_t)bz * p.sC;
    double* Ctg = p.Ct ? p.Ct + (int64_t)bz * p.sCt : nullptr;
    // C goes first (an update's C is final before the launch): behind the MFMA chain its latency was a tenth of a K = 128 launch
    double cin[4] = {0.0, 0.0, 0.0, 0.0};
    if (p.beta != 0.0) {
#pragma unroll
        for (int r = 0; r < 4; ++r)
            cin[r] = Cg[((int64_t)si * 32 + wm * 16 + g + 4 * r) * p.ldc + (int64_t)sj * 32 + wn * 16 + r16];
    }
    d4_t acc = {0.0, 0.0, 0.0, 0.0};
    for (int k = kbeg; k < kend; k += 128) {               // K, kbeg, kend are multiples of 128
        d2_t a[16], b[16];
#pragma unroll
        for (int q = 0; q < 16; ++q) {
            a[q] = *reinterpret_cast<const d2_t*>(Ap + k + 8 * q);
            b[q] = *reinterpret_cast<const d2_t*>(Bp + k + 8 * q);
        }
#pragma unroll
        for (int q = 0; q < 16; ++q) {
            acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a[q][0], b[q][0], acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a[q][1], b[q][1], acc, 0, 0, 0);
        }
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int64_t row = (int64_t)si * 32 + wm * 16 + g + 4 * r;
        const int64_t col = (int64_t)sj * 32 + wn * 16 + r16;
        double v = p.alpha * acc[r];
        if (p.beta != 0.0) v += p.beta * cin[r];
        Cg[row * p.ldc + col] = v;
        if (Ctg) Ctg[col * p.ldct + row] = v;
    }
}

// Split-k chunk of a product with FEW rows of A (16·RG ≤ 64) against a 128-row block of B: no LDS, no barriers.  Wave w owns B rows
// 32w … 32w+31 of the block (two 16-row MFMA groups) and streams them exactly once; the RG row groups of A are re-read by every wave
// (they are small and sit in L2).  Lane ↔ k map and k order of gemm_nt_small_kernel.  P[z][r][i] = Σ_{k in chunk z} A[r][k]·B[i][k].
template <int RG>
__global__ void __launch_bounds__(256) gemm_skinny_kernel(GemmArgs p) {
    if (p.info != nullptr && *p.info != 0) return;
    const int tj = p.kmode == K_B_LOWER ? (int)gridDim.x - 1 - (int)blockIdx.x : (int)blockIdx.x, bz = blockIdx.z;
    int kbeg = 0, kend = p.K;
    if (p.kmode == K_B_LOWER) kend = min(p.K, (tj + 1) * BN);
    if (p.kmode == K_B_UPPER) kbeg = min(p.K, tj * BN);
    kbeg = max(kbeg, bz * p.ksplit);
    kend = min(kend, (bz + 1) * p.ksplit);
    if (kbeg >= kend) return;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int r16 = lane & 15, g = lane >> 4;
    const double* Bp = p.B + (int64_t)(tj * BN + wave * 32 + r16) * p.ldb + 2 * g;
    const double* Ap = p.A + (int64_t)r16 * p.lda + 2 * g;
    d4_t acc[RG][2];
#pragma unroll
    for (int q = 0; q < RG; ++q) { acc[q][0] = d4_t{0.0, 0.0, 0.0, 0.0}; acc[q][1] = d4_t{0.0, 0.0, 0.0, 0.0}; }
    for (int k = kbeg; k < kend; k += 64) {                // kbeg, kend are multiples of 128
        d2_t b0[8], b1[8], a[RG][8];
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            b0[q] = *reinterpret_cast<const d2_t*>(Bp + k + 8 * q);          // (a lane group covers 64 bytes of its row per load: the two halves of a
            b1[q] = *reinterpret_cast<const d2_t*>(Bp + 16 * p.ldb + k + 8 * q);    //  128-byte line arrive with consecutive q — through L2, not past it)
        }
#pragma unroll
        for (int r = 0; r < RG; ++r)
#pragma unroll
            for (int q = 0; q < 8; ++q) a[r][q] = *reinterpret_cast<const d2_t*>(Ap + (int64_t)(16 * r) * p.lda + k + 8 * q);
#pragma unroll
        for (int q = 0; q < 8; ++q)
#pragma unroll
            for (int r = 0; r < RG; ++r) {
                acc[r][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[r][q][0], b0[q][0], acc[r][0], 0, 0, 0);
                acc[r][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[r][q][0], b1[q][0], acc[r][1], 0, 0, 0);
                acc[r][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[r][q][1], b0[q][1], acc[r][0], 0, 0, 0);
                acc[r][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[r][q][1], b1[q][1], acc[r][1], 0, 0, 0);
            }
    }
    double* Cg = p.C + (int64_t)bz * p.sC;
#pragma unroll
    for (int r = 0; r < RG; ++r)
#pragma unroll
        for (int h = 0; h < 2; ++h)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int64_t row = 16 * r + g + 4 * e;                                   // C/D map: lane l, reg e → row (l>>4) + 4e, col l&15
                const int64_t col = (int64_t)tj * BN + wave * 32 + 16 * h + r16;
                Cg[row * p.ldc + col] = p.alpha * acc[r][h][e];
            }
}

// C[r][j] = alpha·Σ_{k<K} A[r][k]·B[j][k] for FEW rows of A (16·RG ≤ 64) against a LONG B that is read exactly once — the block form of
// greedy q-EI (qei.hip): A = K⁻¹K_XT (T rows, a few MB: L2 / MALL resident), B = the resident K_ZX (M rows of N doubles, 17 GB at
// config 5).  HBM-bound on B: 8·K·N bytes against 2·K·N·16RG flop.  The arithmetic of gemm_skinny_kernel — the same lane ↔ k map, the
// same k order per accumulator, hence the same bits (tests/test_gpu_incremental.py) — with BOTH operands going global → LDS by DMA
// (global_load_lds_dwordx4, no staging registers), in stages of 32 k.  What a register-staged kernel cannot have (round 5 built one:
// profiles/r05_qei_pass_variants_ab.txt, removed in round 6) is a memory access shaped for the memory system — the MFMA fragment map makes a
// load instruction touch 16 rows × 64 bytes (half a 128-byte line per row), and a 17 GB stream read that way tops out at 4.9 – 5.5
// TB/s whatever the prefetch depth (profiles/r05_qei_pass_variants_ab.txt).  Here a DMA wave-instruction reads 4 rows × 256
// CONTIGUOUS bytes (whole lines), lane-linearly into LDS; the 16-byte chunks of a row are XOR-swizzled through the per-lane SOURCE
// address (chunk c of row r sits at position c ^ (r & 15)), which makes the fragment reads — lane (r16, g) wants chunk 4q + g of row
// r16 — conflict-free: over each of ds_read_b128's four 16-lane groups (4q + g) ^ r16 takes 16 distinct values.
// Workgroup = 4 waves × 16 rows of B (a wave fetches and reads ITS OWN rows: no barrier for B), A shared (one barrier per stage);
// ring of three stages, two in flight; same lane ↔ k map and k order per accumulator as the kernels above: same bits.
typedef __attribute__((address_space(3))) void* qp_lds_ptr;
template <int RG, int BAUX>
__global__ void __launch_bounds__(256, 2) qei_passd_kernel(const double* __restrict__ A, int64_t lda, const double* __restrict__ B, int64_t ldb,
                                                            double* __restrict__ Cbase, int64_t ldc, int K, double alpha, int kmode, int ksplit,
                                                            int64_t sC) {
    constexpr int SB_ = 64 * 256;                          // bytes of B per stage: 64 rows × 32 k
    constexpr int SA_ = 16 * RG * 256;                     // bytes of A per stage
    constexpr int SLOT = SB_ + SA_;
    __shared__ __attribute__((aligned(1024))) char lds[3 * SLOT];       // ONE LDS object (ozaki.hip: a second one makes the compiler
                                                                        // order every fragment read behind every DMA piece in flight)
    const int tj = kmode == K_B_LOWER ? (int)gridDim.x - 1 - (int)blockIdx.x : (int)blockIdx.x, bz = blockIdx.z;
    int kbeg = 0, kend = K;
    if (kmode == K_B_LOWER) kend = min(K, (tj / 2 + 1) * BN);          // row blocks of 64: block tj lies in the 128-row block tj/2
    if (kmode == K_B_UPPER) kbeg = min(K, (tj / 2) * BN);
    kbeg = max(kbeg, bz * ksplit);
    kend = min(kend, (bz + 1) * ksplit);
    if (kbeg >= kend) return;                              // uniform over the workgroup
    double* C = Cbase + (int64_t)bz * sC;
    const int t = threadIdx.x, lane = t & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int r16 = lane & 15, g = lane >> 4;
    const int64_t row0 = (int64_t)tj * 64 + wave * 16;
    // DMA source addresses: lane L of a piece fills position L % 16 of sub-row L / 16 (4 rows × 256 bytes per piece)
    const int sub = lane >> 4, pos = lane & 15;
    const char* bsrc[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int rw = 4 * j + sub;                        // row inside the wave's 16
        bsrc[j] = reinterpret_cast<const char*>(B + (row0 + rw) * ldb) + 16 * (pos ^ (rw & 15));
    }
    const char* asrc[RG];
#pragma unroll
    for (int i = 0; i < RG; ++i) {
        const int ra = 4 * (wave + 4 * i) + sub;           // A rows 4a … 4a+3 of piece a = wave + 4i
        asrc[i] = reinterpret_cast<const char*>(A + (int64_t)ra * lda) + 16 * (pos ^ (ra & 15));
    }
    const int ns = (kend - kbeg) / 32;                     // stages (a multiple of 4)
    auto issue = [&](int st) {                             // stage st (clamped: past the end a harmless re-fetch keeps the counts uniform)
        const int sc = st < ns ? st : ns - 1;
        const int64_t kb = ((int64_t)kbeg + 32 * sc) * 8;
        char* slot = lds + (st % 3) * SLOT;
#pragma unroll
        for (int j = 0; j < 4; ++j)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(bsrc[j] + kb),
                                             (qp_lds_ptr)(slot + (16 * wave + 4 * j) * 256), 16, 0, BAUX);
#pragma unroll
        for (int i = 0; i < RG; ++i)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(asrc[i] + kb),
                                             (qp_lds_ptr)(slot + SB_ + 4 * (wave + 4 * i) * 256), 16, 0, 0);
    };
    d4_t acc[RG];
#pragma unroll
    for (int r = 0; r < RG; ++r) acc[r] = d4_t{0.0, 0.0, 0.0, 0.0};
    issue(0);
    issue(1);
    for (int st = 0; st < ns; ++st) {
        // this wave's pieces of stage st have landed (those of stage st + 1 may still fly); its fragment reads of stage st − 1 are done
        if constexpr (RG == 1) asm volatile("s_waitcnt vmcnt(5)" ::: "memory");
        else if constexpr (RG == 2) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
        else if constexpr (RG == 3) asm volatile("s_waitcnt vmcnt(7)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();                      // A of stage st complete for all waves; nobody still reads the slot of stage st + 2
        __builtin_amdgcn_sched_barrier(0);
        issue(st + 2);
        const char* slot = lds + (st % 3) * SLOT;
        const char* bp = slot + (16 * wave + r16) * 256;
        const char* ap = slot + SB_ + r16 * 256;
#pragma unroll
        for (int q = 0; q < 4; ++q) {                     // (fragments pinned one k-group ahead of their MFMAs measured 3 % slower at T = 32)
            const int off = 16 * ((4 * q + g) ^ r16);
            const d2_t b = *reinterpret_cast<const d2_t*>(bp + off);
            d2_t a[RG];
#pragma unroll
            for (int r = 0; r < RG; ++r) a[r] = *reinterpret_cast<const d2_t*>(ap + r * 16 * 256 + off);
#pragma unroll
            for (int r = 0; r < RG; ++r) acc[r] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[r][0], b[0], acc[r], 0, 0, 0);
#pragma unroll
            for (int r = 0; r < RG; ++r) acc[r] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[r][1], b[1], acc[r], 0, 0, 0);
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // the tail's re-fetches have landed before the workgroup's LDS goes away
#pragma unroll
    for (int r = 0; r < RG; ++r)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int64_t row = 16 * r + g + 4 * e;
            C[row * ldc + row0 + r16] = alpha * acc[r][e];
        }
}

hipError_t launch_qei_pass(const double* A, int64_t lda, int rows16, const double* B, int64_t ldb, int64_t nB, int K, double alpha,
                           double* C, int64_t ldc, hipStream_t s, int kmode, int ksplit, int64_t sC) {
    if (nB <= 0) return hipSuccess;
    if (rows16 < 16 || rows16 > 64 || rows16 % 16 || nB % 128 || K <= 0 || K % 128 || (lda & 1) || (ldb & 1)) return hipErrorInvalidValue;
    if (kmode != K_FULL && kmode != K_B_LOWER && kmode != K_B_UPPER) return hipErrorInvalidValue;
    if (ksplit <= 0) ksplit = K;
    if (ksplit % 128) return hipErrorInvalidValue;
    dim3 gridd((unsigned)(nB / 64), 1, (unsigned)((K + ksplit - 1) / ksplit));
    // the B stream is read exactly once: its DMA carries the non-temporal hint (aux bit 1) — 5.7 → 6.3 TB/s at T = 16, 5.4 → 5.8 at
    // T = 32 (profiles/r05_qei_pass_variants_ab.txt); A stays cacheable (every workgroup re-reads it)
#define QP_LAUNCH(RG_) hipLaunchKernelGGL((qei_passd_kernel<RG_, 2>), gridd, dim3(256), 0, s, A, lda, B, ldb, C, ldc, K, alpha, kmode, ksplit, sC)
    switch (rows16 / 16) {
        case 1: QP_LAUNCH(1); break;
        case 2: QP_LAUNCH(2); break;
        case 3: QP_LAUNCH(3); break;
        default: QP_LAUNCH(4); break;
    }
#undef QP_LAUNCH
    return hipGetLastError();
}

hipError_t launch_gemm_nt(const GemmArgs& a, hipStream_t s) {
    if (a.M <= 0 || a.N <= 0 || a.batch <= 0) return hipSuccess;
    const int64_t Tm = a.M / BM, Tn = a.N / BN;
    int64_t tiles = Tm * Tn;
    if (a.lower_only) {                                              // tiles (ti, tj ≤ ti), tj < Tn
        const int64_t q = Tm < Tn ? Tm : Tn;
        tiles = q * (q + 1) / 2 + (Tm - q) * Tn;
    }
    tiles *= a.batch;
    const char* fe = getenv("ABO_GEMM_SMALL");                      // A/B runs and tests: 0 never, 1 always
    const int force = fe ? atoi(fe) : -1;
    const bool small = a.ksplit ? false : (force >= 0 ? force == 1 : tiles <= 64);      // measured crossover: small ≈ tiles·K·1.7 ns, tiled ≈ 6 µs + K·0.11 µs
    const bool in_place = a.C == a.A || a.C == a.B;
    if (a.ksplit) {
        if (a.batch != 1 || a.beta != 0.0 || a.ksplit % 128 != 0 || a.Ct) return hipErrorInvalidValue;
        dim3 grid(a.N / BN, a.M / BM, (a.K + a.ksplit - 1) / a.ksplit);
        if (a.mrows >= 16 && a.mrows <= 64 && a.mrows % 16 == 0 && a.M == BM) {
            switch (a.mrows / 16) {
                case 1: hipLaunchKernelGGL((gemm_skinny_kernel<1>), grid, dim3(256), 0, s, a); break;
                case 2: hipLaunchKernelGGL((gemm_skinny_kernel<2>), grid, dim3(256), 0, s, a); break;
                case 3: hipLaunchKernelGGL((gemm_skinny_kernel<3>), grid, dim3(256), 0, s, a); break;
                default: hipLaunchKernelGGL((gemm_skinny_kernel<4>), grid, dim3(256), 0, s, a); break;
            }
            return hipGetLastError();
        }
        hipLaunchKernelGGL(gemm_nt_kernel, grid, dim3(256), 0, s, a);
        return hipGetLastError();
    }
    if (small && !in_place && a.K % 128 == 0) {
        dim3 grid(a.N / 32, a.M / 32, a.batch);
        hipLaunchKernelGGL(gemm_nt_small_kernel, grid, dim3(256), 0, s, a);
    } else if (a.lower_only && a.kmode == K_FULL && a.M == a.N && a.batch == 1 && Tm >= 16 && !getenv("ABO_GEMM_NO_SWIZZLE")) {
        GemmArgs b = a;
        b.swz = (int)tiles;
        b.swz_g = 4;
        hipLaunchKernelGGL(gemm_nt_kernel, dim3((unsigned)(((tiles + 7) / 8) * 8)), dim3(256), 0, s, b);
    } else {
        dim3 grid(a.N / BN, a.M / BM, a.batch);
        hipLaunchKernelGGL(gemm_nt_kernel, grid, dim3(256), 0, s, a);
    }
    return hipGetLastError();
}

// sum of the split-k partial products, chunk order (the same chunks gemm_nt_kernel computed: the others hold nothing)
__global__ void __launch_bounds__(256) splitk_reduce_kernel(const double* __restrict__ P, int64_t ldp, int64_t sP, int nz, int rows, int cols,
                                                             int K, int ksplit, int kmode, double* __restrict__ out, int64_t ldo) {
    const int c2 = (blockIdx.x * 256 + threadIdx.x) * 2, r = blockIdx.y;
    if (c2 >= cols) return;
    const int tj = c2 / BN;
    int kbeg = 0, kend = K;
    if (kmode == K_B_LOWER) kend = min(K, (tj + 1) * BN);
    if (kmode == K_B_UPPER) kbeg = min(K, tj * BN);
    d2_t acc = {0.0, 0.0};
    const double* p = P + (int64_t)r * ldp + c2;
    for (int z = 0; z < nz; ++z) {
        if (max(kbeg, z * ksplit) >= min(kend, (z + 1) * ksplit)) continue;
        const d2_t v = *reinterpret_cast<const d2_t*>(p + (int64_t)z * sP);
        acc[0] += v[0]; acc[1] += v[1];
    }
    *reinterpret_cast<d2_t*>(out + (int64_t)r * ldo + c2) = acc;
}

hipError_t launch_splitk_reduce(const double* P, int64_t ldp, int64_t sP, int nz, int rows, int cols, int K, int ksplit, int kmode,
                                double* out, int64_t ldo, hipStream_t s) {
    if (rows <= 0 || cols <= 0) return hipSuccess;
    hipLaunchKernelGGL(splitk_reduce_kernel, dim3((cols / 2 + 255) / 256, rows), dim3(256), 0, s, P, ldp, sP, nz, rows, cols, K, ksplit,
                       kmode, out, ldo);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------------
// Tile order: heaviest row blocks first (tile (ti, ·) runs ti+1 k-blocks), so the light tiles fill
// the tail of the launch.  blockIdx.x → (ti, tj) with tj fastest: consecutive workgroups share the
// same W row panel and stream different candidate panels.
__global__ void __launch_bounds__(256, 2) var_gemm_kernel(VarGemmArgs p) {
    __shared__ __attribute__((aligned(16))) double smem[4 * TILE];
    const int Tj = p.Mc / BN;
    const int Ti = p.Np / BM;
    const int b = blockIdx.x;
    const int ti = Ti - 1 - b / Tj;
    const int tj = b % Tj;
    const double* Ag = p.W + (int64_t)ti * BM * p.ldw;
    const double* Bg = p.Kxz + (int64_t)tj * BN * p.ldk;
    d4_t acc[4][4];
    acc_zero(acc);
    tile_loop(Ag, p.ldw, Bg, p.ldk, 0, (ti + 1) * BM, smem, acc);

    // column sums of squares over this tile's 128 rows, fixed order (deterministic):
    // registers (mi, r) → lanes sharing a column (xor 16, 32) → the two waves stacked in m (LDS)
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    double* red = smem;  // [2][128]; the tile loop ended with a barrier, LDS is free
    // rows ≥ nvalid of the last row block are padding *for this view*: a later bordered append may
    // have written real factor rows there (shared storage), so they must not enter the norm
    const int row0 = ti * BM + wm * 64 + (lane >> 4);
    const bool edge = (ti + 1) * BM > p.nvalid;
#pragma unroll
    for (int ni = 0; ni < 4; ++ni) {
        double s = 0.0;
#pragma unroll
        for (int mi = 0; mi < 4; ++mi)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                double v = acc[mi][ni][r];
                if (edge && row0 + mi * 16 + 4 * r >= p.nvalid) v = 0.0;
                s = fma(v, v, s);
            }
        s += __shfl_xor(s, 16);
        s += __shfl_xor(s, 32);
        if (lane < 16) red[wm * 128 + wn * 64 + ni * 16 + lane] = s;
    }
    __syncthreads();
    if (threadIdx.x < 128)
        p.partial[(int64_t)ti * p.ldp + (int64_t)tj * BN + threadIdx.x] = red[threadIdx.x] + red[128 + threadIdx.x];
}

// ------------------------------------------------------------------------------------------------
// 256×128 tile: one 512-thread workgroup (8 waves as 4(m)×2(n), 64×64 of outputs per wave, one-barrier interleaved pipeline)
// computes the two vertically adjacent 128×128 tiles that var_gemm_kernel gives to two workgroups, sharing ONE candidate (B) tile
// in LDS: 25 % fewer operand bytes requested from L2 and 25 % fewer VMEM / LDS-store instructions per MFMA.  Used when Np is a
// multiple of 256 (var_gemm_kernel serves an odd number of 128-row blocks); same summation order of the partial rows.
constexpr int BM2 = 256;
constexpr int STAGE2 = (BM2 + BN) * LDT;       // doubles per LDS stage: A rows 0..255, B rows 256..383

// The structural zeros of the diagonal blocks are skipped at 16-row granularity: wave (wm, wn) owns the 16-row sub-tiles {wm, wm+4, wm+8, wm+12} of the
// 256 rows instead of 64 consecutive rows, so that every wave — and with it every SIMD — loses the same share of
// MFMAs when the k loop runs through the two triangular diagonal blocks.  The 128-granular tiling spends a factor
// 1 + 128/N of the credited flops (1.6 % at N = 8192, 12.5 % at N = 1024); this one 1 + 16/N.  Measured
// (tools/skip_experiment.sh): C3 contraction 962 → 955 ms, C2 1.17 → 1.12 ms.
__global__ void __launch_bounds__(512, 2) var_gemm256s_kernel(VarGemmArgs p) {
    __shared__ __attribute__((aligned(16))) double smem[2 * STAGE2];
    const int t = threadIdx.x, lane = t & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    const int r16 = lane & 15, g = lane >> 4;
    const int Tj = p.Mc / BN;
    const int Ti2 = p.Np / BM2;
    const int b = blockIdx.x;
    const int ti2 = Ti2 - 1 - b / Tj;
    const int tj = b % Tj;
    const double* __restrict__ Ag = p.W + (int64_t)ti2 * BM2 * p.ldw;
    const double* __restrict__ Bg = p.Kxz + (int64_t)tj * BN * p.ldk;
    const int nk = (ti2 + 1) * (BM2 / BK);
    const int srow = t >> 3, skk = (t & 7) * 2;
    const int aoff = (wm * 16 + r16) * LDT + g * 2;           // A fragment i: 16-row sub-tile wm + 4·i of the 256 rows
    const int boff = (BM2 + wn * 64 + r16) * LDT + g * 2;
    d2_t sa[4], sb[2];
    Frag f0, f1;
    d4_t acc[4][4];
    acc_zero(acc);

#define G_LOAD(k0)                                                                                              \
    do {                                                                                                        \
        _Pragma("unroll") for (int q = 0; q < 4; ++q)                                                           \
            sa[q] = *reinterpret_cast<const d2_t*>(Ag + (int64_t)(srow + 64 * q) * p.ldw + (k0) + skk);         \
        _Pragma("unroll") for (int q = 0; q < 2; ++q)                                                           \
            sb[q] = *reinterpret_cast<const d2_t*>(Bg + (int64_t)(srow + 64 * q) * p.ldk + (k0) + skk);         \
    } while (0)
#define L_STORE(buf)                                                                                            \
    do {                                                                                                        \
        _Pragma("unroll") for (int q = 0; q < 4; ++q)                                                           \
            *reinterpret_cast<d2_t*>((buf) + (srow + 64 * q) * LDT + skk) = sa[q];                              \
        _Pragma("unroll") for (int q = 0; q < 2; ++q)                                                           \
            *reinterpret_cast<d2_t*>((buf) + (BM2 + srow + 64 * q) * LDT + skk) = sb[q];                        \
    } while (0)

#define FRAG_READ(buf, half, f, a0)                                                                              \
    do {                                                                                                        \
        _Pragma("unroll") for (int i = (a0); i < 4; ++i)                                                        \
            (f).a[i] = *reinterpret_cast<const d2_t*>((buf) + aoff + i * 64 * LDT + (half) * 8);                \
        if ((a0) < 4) {                                                                                         \
            _Pragma("unroll") for (int i = 0; i < 4; ++i)                                                       \
                (f).b[i] = *reinterpret_cast<const d2_t*>((buf) + boff + i * 16 * LDT + (half) * 8);            \
        }                                                                                                       \
    } while (0)
#define MMA_FROM(KK, f, a0)                                                                                     \
    do {                                                                                                        \
        _Pragma("unroll") for (int mi = (a0); mi < 4; ++mi) {                                                   \
            _Pragma("unroll") for (int ni = 0; ni < 4; ++ni)                                                    \
                acc[mi][ni] = __builtin_amdgcn_mfma_f64_16x16x4f64((f).a[mi][KK], (f).b[ni][KK], acc[mi][ni], 0, 0, 0); \
        }                                                                                                       \
    } while (0)

    // One stage = 16 k: 32 MFMAs on F0 (fragments of the stage's first half) with the next stage's LDS stores, the second half's fragment
    // reads and the loads of the stage after next dealt out between them, a barrier, 32 MFMAs on F1 with the next stage's first fragment
    // reads.  A0 = the first live A fragment of this wave (below): the stage runs the same schedule over fragments A0 … 3.
    // Loads past the tile's k range (the last two stages) re-read its last 16 columns: stored to LDS, never used.
    const int klast = (nk - 1) * BK;
    auto stage = [&](int st, auto a0c) {
        constexpr int A0 = decltype(a0c)::value;
        double* cur = smem + (st & 1) * STAGE2;
        double* nxt = smem + ((st + 1) & 1) * STAGE2;
        L_STORE(nxt);
        FRAG_READ(cur, 1, f1, A0);
        const int kn = (st + 2) * BK;
        G_LOAD(kn < klast ? kn : klast);
        MMA_FROM(0, f0, A0);
        MMA_FROM(1, f0, A0);
        // 16·(4 − A0) MFMAs; behind one MFMA each: 6 LDS stores, 8 − A0 fragment reads, 6 loads (what does not fit behind an MFMA follows)
        constexpr int NM = 8 * (4 - A0);                              // MFMAs of this half stage
        if constexpr (NM > 0) {
            constexpr int NR = 8 - A0;
#pragma unroll
            for (int i = 0; i < 6; ++i) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);   // MFMA
                __builtin_amdgcn_sched_group_barrier(0x200, 1, 0);   // DS write
            }
#pragma unroll
            for (int i = 0; i < (NM - 6 < NR ? NM - 6 : NR); ++i) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);   // DS read
            }
            constexpr int left = NM - 6 - (NM - 6 < NR ? NM - 6 : NR);
#pragma unroll
            for (int i = 0; i < (left < 6 ? left : 6); ++i) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);   // VMEM read
            }
            if constexpr (left > 6) __builtin_amdgcn_sched_group_barrier(0x008, left - 6, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
        __syncthreads();
        FRAG_READ(nxt, 0, f0, A0);
        MMA_FROM(0, f1, A0);
        MMA_FROM(1, f1, A0);
        if constexpr (NM > 0) {
            constexpr int NR = 8 - A0;
#pragma unroll
            for (int i = 0; i < NR; ++i) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 1);
                __builtin_amdgcn_sched_group_barrier(0x100, 1, 1);
            }
            __builtin_amdgcn_sched_group_barrier(0x008, NM - NR, 1);
        }
        __builtin_amdgcn_sched_barrier(0);
    };

    G_LOAD(0);
    L_STORE(smem);
    G_LOAD(BK);                                             // nk ≥ 16
    __syncthreads();
    FRAG_READ(smem, 0, f0, 0);
    // The last 16 stages run through the two diagonal 128-blocks of W (upper block: stages nk−16 … nk−9, lower block: the last 8).  In
    // stage s of a diagonal block the 16-row sub-tile m of that block is non-zero only for m ≥ s, and the upper block is zero throughout
    // the last 8 stages.  This wave's fragments are sub-tiles wm, wm + 4 of the upper block, then wm, wm + 4 of the lower block (the
    // interleaved row ↔ wave map keeps the skipped work spread over the four SIMDs), so with s16 = stage − (nk − 16):
    //     fragment 0 is zero from s16 = wm + 1 on, fragment 1 from wm + 5, fragment 2 from wm + 9, fragment 3 from wm + 13:
    // the live set is always {A0 … 3}, and the stages of one A0 are consecutive — five copies of ONE dealt-out schedule, run one after the
    // other.  (Round 5 skipped behind run-time branches on A0: four scheduling regions a stage with nothing dealt out between them.)
    int st = 0;
    const int d0 = nk - 16 + wm;
    for (; st <= d0; ++st) stage(st, std::integral_constant<int, 0>{});
    for (; st <= d0 + 4; ++st) stage(st, std::integral_constant<int, 1>{});
    for (; st <= d0 + 8; ++st) stage(st, std::integral_constant<int, 2>{});
    for (; st <= d0 + 12 && st < nk; ++st) stage(st, std::integral_constant<int, 3>{});
    for (; st < nk; ++st) stage(st, std::integral_constant<int, 4>{});
    __syncthreads();
#undef G_LOAD
#undef L_STORE
#undef FRAG_READ
#undef MMA_FROM

    // column sums of squares per 128-row block: fragments 0,1 belong to the upper block, 2,3 to the lower one;
    // registers → lanes sharing a column (xor 16, 32) → the four wm waves in order (LDS): fixed order, deterministic
    double* red = smem;                                     // [2 halves][4 wm][128]
    const int row0 = ti2 * BM2 + wm * 16 + g;
    const bool edge = (ti2 + 1) * BM2 > p.nvalid;
#pragma unroll
    for (int ni = 0; ni < 4; ++ni) {
#pragma unroll
        for (int half = 0; half < 2; ++half) {
            double sq = 0.0;
#pragma unroll
            for (int mi = 2 * half; mi < 2 * half + 2; ++mi)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    double v = acc[mi][ni][r];
                    if (edge && row0 + mi * 64 + 4 * r >= p.nvalid) v = 0.0;
                    sq = fma(v, v, sq);
                }
            sq += __shfl_xor(sq, 16);
            sq += __shfl_xor(sq, 32);
            if (lane < 16) red[(half * 4 + wm) * 128 + wn * 64 + ni * 16 + lane] = sq;
        }
    }
    __syncthreads();
    if (t < 256) {
        const int half = t >> 7, c = t & 127;
        const double* rh = red + half * 4 * 128 + c;
        p.partial[(int64_t)(2 * ti2 + half) * p.ldp + (int64_t)tj * BN + c] = ((rh[0] + rh[128]) + rh[256]) + rh[384];
    }
}

hipError_t launch_var_gemm(const VarGemmArgs& a, hipStream_t s) {
    if (a.Np % BM2 == 0) {
        const int tiles2 = (a.Np / BM2) * (a.Mc / BN);
        if (tiles2 <= 0) return hipSuccess;
        hipLaunchKernelGGL(var_gemm256s_kernel, dim3(tiles2), dim3(512), 0, s, a);
        return hipGetLastError();
    }
    const int tiles = (a.Np / BM) * (a.Mc / BN);
    if (tiles <= 0) return hipSuccess;
    hipLaunchKernelGGL(var_gemm_kernel, dim3(tiles), dim3(256), 0, s, a);
    return hipGetLastError();
}

}  // namespace abo
