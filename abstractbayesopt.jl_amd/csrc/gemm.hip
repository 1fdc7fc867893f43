// fp64 MFMA tile core for gfx950 and the two kernels built on it:
//   gemm_nt_kernel   C = alpha·A·Bᵀ + beta·C   (TRSM panel, SYRK trailing update, blocked L⁻¹)
//   var_gemm_kernel  partial[ti][j] = Σ_{i∈ti} (W·K_XZ)[i][j]²  — the N²·M contraction that is
//                    >99 % of the flops of posterior_var (reference: src/surrogates/StandardGP.jl:377-379,
//                    [upstream AbstractGPs] diag_Xt_invA_X(C, K_XZ)).
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
//    ds_read_b128 each touch 16 distinct 16-byte slots (conflict-free), and rows stay 16-B aligned
//    for ds_write_b128.
//  * global → register prefetch of stage t+1 is issued before the MFMAs of stage t; registers are
//    written to LDS after the barrier (issue-early / write-late).  Two workgroups per CU (40 KB LDS,
//    ≤256 VGPRs) keep the matrix pipe busy across each other's barriers.
#include "abo_kernels.h"

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

// one BK=16 stage: 2 × (8 ds_read_b128 + 32 MFMA) per wave
__device__ __forceinline__ void tile_mma(const double* As, const double* Bs, d4_t (&acc)[4][4], int wm, int wn,
                                         int lane) {
    const int r16 = lane & 15, g = lane >> 4;
    const double* ap = As + (wm * 64 + r16) * LDT + g * 2;
    const double* bp = Bs + (wn * 64 + r16) * LDT + g * 2;
#pragma unroll
    for (int k8 = 0; k8 < 2; ++k8) {
        d2_t a[4], b[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) a[i] = *reinterpret_cast<const d2_t*>(ap + i * 16 * LDT + k8 * 8);
#pragma unroll
        for (int i = 0; i < 4; ++i) b[i] = *reinterpret_cast<const d2_t*>(bp + i * 16 * LDT + k8 * 8);
#pragma unroll
        for (int kk = 0; kk < 2; ++kk)
#pragma unroll
            for (int mi = 0; mi < 4; ++mi)
#pragma unroll
                for (int ni = 0; ni < 4; ++ni)
                    acc[mi][ni] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[mi][kk], b[ni][kk], acc[mi][ni], 0, 0, 0);
    }
}

// acc += A[0:128][kbeg:kend] · B[0:128][kbeg:kend]ᵀ ; Ag/Bg point at the tile's first row.
__device__ __forceinline__ void tile_loop(const double* __restrict__ Ag, int64_t lda, const double* __restrict__ Bg,
                                          int64_t ldb, int kbeg, int kend, double* smem, d4_t (&acc)[4][4]) {
    double* As = smem;
    double* Bs = smem + TILE;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    d2_t ra[4], rb[4];
    if (kbeg < kend) {
        tile_gload(Ag, lda, kbeg, ra);
        tile_gload(Bg, ldb, kbeg, rb);
        tile_lstore(As, ra);
        tile_lstore(Bs, rb);
    }
    __syncthreads();
    for (int k0 = kbeg; k0 < kend; k0 += BK) {
        const bool more = (k0 + BK) < kend;
        if (more) {
            tile_gload(Ag, lda, k0 + BK, ra);
            tile_gload(Bg, ldb, k0 + BK, rb);
        }
        tile_mma(As, Bs, acc, wm, wn, lane);
        __syncthreads();
        if (more) {
            tile_lstore(As, ra);
            tile_lstore(Bs, rb);
        }
        __syncthreads();
    }
}

__device__ __forceinline__ void acc_zero(d4_t (&acc)[4][4]) {
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = d4_t{0.0, 0.0, 0.0, 0.0};
}

// ------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256, 2) gemm_nt_kernel(GemmArgs p) {
    __shared__ __attribute__((aligned(16))) double smem[2 * TILE];
    if (p.info != nullptr && *p.info != 0) return;
    const int tj = blockIdx.x, ti = blockIdx.y, bz = blockIdx.z;
    if (p.lower_only && tj > ti) return;
    int kbeg = 0, kend = p.K;
    if (p.kmode == K_A_LOWER) kend = min(p.K, (ti + 1) * BM);
    if (p.kmode == K_A_UPPER) kbeg = min(p.K, ti * BM);
    const double* Ag = p.A + (int64_t)bz * p.sA + (int64_t)ti * BM * p.lda;
    const double* Bg = p.B + (int64_t)bz * p.sB + (int64_t)tj * BN * p.ldb;
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

hipError_t launch_gemm_nt(const GemmArgs& a, hipStream_t s) {
    if (a.M <= 0 || a.N <= 0 || a.batch <= 0) return hipSuccess;
    dim3 grid(a.N / BN, a.M / BM, a.batch);
    hipLaunchKernelGGL(gemm_nt_kernel, grid, dim3(256), 0, s, a);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------------
// Tile order: heaviest row blocks first (tile (ti, ·) runs ti+1 k-blocks), so the light tiles fill
// the tail of the launch.  blockIdx.x → (ti, tj) with tj fastest: consecutive workgroups share the
// same W row panel and stream different candidate panels.
__global__ void __launch_bounds__(256, 2) var_gemm_kernel(VarGemmArgs p) {
    __shared__ __attribute__((aligned(16))) double smem[2 * TILE];
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
#pragma unroll
    for (int ni = 0; ni < 4; ++ni) {
        double s = 0.0;
#pragma unroll
        for (int mi = 0; mi < 4; ++mi)
#pragma unroll
            for (int r = 0; r < 4; ++r) s = fma(acc[mi][ni][r], acc[mi][ni][r], s);
        s += __shfl_xor(s, 16);
        s += __shfl_xor(s, 32);
        if (lane < 16) red[wm * 128 + wn * 64 + ni * 16 + lane] = s;
    }
    __syncthreads();
    if (threadIdx.x < 128)
        p.partial[(int64_t)ti * p.ldp + (int64_t)tj * BN + threadIdx.x] = red[threadIdx.x] + red[128 + threadIdx.x];
}

hipError_t launch_var_gemm(const VarGemmArgs& a, hipStream_t s) {
    const int tiles = (a.Np / BM) * (a.Mc / BN);
    if (tiles <= 0) return hipSuccess;
    hipLaunchKernelGGL(var_gemm_kernel, dim3(tiles), dim3(256), 0, s, a);
    return hipGetLastError();
}

}  // namespace abo
