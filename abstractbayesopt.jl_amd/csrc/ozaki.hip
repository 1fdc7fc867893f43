// The N²·M variance contraction on the int8 matrix pipe: V = W·K_XZ computed EXACTLY on fixed-point images of the two fp64
// operands, through residues modulo n pairwise-coprime moduli ≤ 256 (one int8 GEMM per modulus, int32 accumulation, no
// rounding anywhere in the products) and a Chinese-remainder reconstruction in fp64.  Reference arithmetic replaced:
// posterior_var, src/surrogates/StandardGP.jl:377-379 → [upstream AbstractGPs] diag_Xt_invA_X(C, K_XZ) — the same
// partial[ti][j] = Σ_{i∈ti} (Σ_k W[i][k]·K_XZ[j][k])² the fp64 kernels of gemm.hip produce.
//
// Why: the fp64 matrix pipe of gfx950 peaks at 78.6 TFLOP/s, `v_mfma_i32_32x32x32_i8` at ≈ 4.9 POP/s (tools/mfma_i8_probe.hip).
// Scheme (Ozaki, Uchino, Imamura: "Ozaki scheme II", 2025 — restated here from the published algorithm, no code of theirs):
//   1. W'[i][k] = rint(W[i][k]·2^s_i), K'[j][k] = rint(K[j][k]·2^sK): integers below 2^52 / 2^53.  s_i is chosen per row from
//      the row's L1 norm so that |Σ_k W'[i][k]·K'[j][k]| < P/4 for P = Π p_l; sK from the kernel's upper bound σ_f².
//      This is the ONLY approximation: every W entry keeps ≥ 50 bits below its row's L1 norm, every K entry 52–53 bits below
//      σ_f² — the entries that carry a product's weight are represented exactly.
//   2. per modulus p_l: residues in [−128, 127] (one byte per entry), C_l = W_l·K_lᵀ in int32 (|C_l| ≤ k·2^14, exact for
//      k ≤ 2^17), U_l = C_l mod p_l (symmetric) — one byte per output and modulus.
//   3. C' = Σ_l U_l·s_l mod P with s_l = (P/p_l)·((P/p_l)⁻¹ mod p_l): the constants are split s_l = s1_l + s2_l with s1_l on a
//      2^t grid of 41 bits, so that Σ U_l·s1_l is exact in fp64; Q = rint((C1 + C2)/P); C' = (C1 − Q·P1) + (C2 − Q·P2) with
//      P = P1 + P2 split the same way (C1 − Q·P1 exact under fma).  V = C'·2^−(s_i+sK); Σ_i V² in fp64.
// With n = 14 (P ≈ 2^110) the result differs from the exactly rounded product by about what the fp64 MFMA kernel's own
// accumulation error is (tests/test_gpu_ozaki.py records both against a long-double product).
//
// Kernels: oz_rowscale_kernel (L1 / max per row of W → s_i), oz_quant_kernel (fp64 → n residue planes: W once per model; K_XZ per
// chunk only when the generator cannot write the planes itself — kgen.hip does for a StandardGP with d ≤ 32), the residue GEMM
// oz_gemm16p_kernel (256×256 tile per 8-wave workgroup, v_mfma_i32_16x16x64_i8 fed by an LDS-DMA ring, persistent, triangular
// k-range, epilogue = symmetric mod + byte pack + LDS transpose; the register-staged 32×32×32 / 16×16×64 kernels and the
// one-tile-per-workgroup kernel it grew out of are in the history up to round 5 and in profiles/r02_int8_ablation.txt,
// r03_int8_ablation.txt), and oz_crt_kernel (reconstruction + squares + per-row-block column sums).  Residue planes are stored in
// 16 KB blocks of 256 rows × 64 k-bytes (abo_oz_dev.h: oz_plane_off); U in 64 KB blocks of one 256 × 256 tile and modulus (oz_u_off).
#include "abo_kernels.h"
#include "abo_oz_dev.h"
#include <cmath>
#include <cstdlib>
#include <cstring>

namespace abo {

typedef int v4i_t __attribute__((ext_vector_type(4)));
typedef int v16i_t __attribute__((ext_vector_type(16)));

// ---- host: moduli and reconstruction constants --------------------------------------------------------------------------------
namespace {
typedef unsigned __int128 u128;

int gcd_i(int a, int b) { while (b) { int t = a % b; a = b; b = t; } return a; }

int bitlen(u128 x) { int n = 0; while (x) { ++n; x >>= 1; } return n; }

double u128_to_double(u128 x) {          // correctly rounded for x < 2^64·2^53 is not needed: two exact halves, one rounding
    const uint64_t hi = (uint64_t)(x >> 64), lo = (uint64_t)x;
    return std::ldexp((double)hi, 64) + (double)lo;
}
}  // namespace

bool oz_make_plan(int n, OzPlan* out) {
    if (n < 8 || n > OZ_MAXMOD) return false;       // below 8 moduli P has fewer than the 41 bits the split heads take
    OzPlan pl{};
    pl.n = n;
    int c = 256, m = 0;
    while (m < n) {                                   // 256, 255, 253, 251, 247, 241, 239, 233, 229, 227, 223, 217, 211, 199, 197, 193
        bool ok = true;
        for (int q = 0; q < m; ++q) ok = ok && gcd_i(c, pl.p[q]) == 1;
        if (ok) pl.p[m++] = c;
        --c;
    }
    for (int l = 0; l < n; ++l)
        if (pl.p[l] != oz_mod_p(l)) return false;     // the compile-time table of abo_oz_dev.h is this very sequence
    u128 P = 1;
    for (int l = 0; l < n; ++l) P *= (u128)pl.p[l];   // < 2^125.4 for n = 16
    const int t = bitlen(P) - 41;                     // grid of the high parts: 41 significant bits
    for (int l = 0; l < n; ++l) {
        const u128 Mi = P / (u128)pl.p[l];
        const int r = (int)(Mi % (u128)pl.p[l]);
        int q = 1;
        while ((r * q) % pl.p[l] != 1) ++q;           // (P/p)⁻¹ mod p by search (p ≤ 256)
        const u128 s = Mi * (u128)q;                  // < P
        const u128 hi = (s >> t) << t;
        pl.s1[l] = std::ldexp((double)(uint64_t)(s >> t), t);
        pl.s2[l] = u128_to_double(s - hi);
        pl.invp[l] = 1.0 / (double)pl.p[l];
        int c26 = (int)((1u << 26) % (unsigned)pl.p[l]);
        if (2 * c26 > pl.p[l]) c26 -= pl.p[l];
        pl.c26[l] = (double)c26;                      // 2^26 mod p, symmetric
    }
    const u128 P1 = (P >> t) << t;
    pl.P1 = std::ldexp((double)(uint64_t)(P >> t), t);
    pl.P2 = u128_to_double(P - P1);
    pl.invP = 1.0 / u128_to_double(P);
    pl.eP = bitlen(P) - 3;                            // 2^eP ≤ P/4
    *out = pl;
    return true;
}

// ---- device helpers --------------------------------------------------------------------------------------------------------------
// ---- row scales of W ---------------------------------------------------------------------------------------------------------------
// one wave per row i < Np: L1 = Σ_{k≤i} |W[i][k]|, mx = max; s_i = min(eP − 53 − e(L1), 52 − e(mx)) with e(x) the frexp exponent
// (x < 2^e), so that 2^s_i·L1·2^53 ≤ P/4 and |W'| < 2^52.  sexp[i] = s_i; rows ≥ Np (padding to 256) get 0.
// kper / ktg: columns k with k % kper != 0 carry an extra factor 2^ktg (the gradient outputs of a gradient-enhanced model, see
// oz_prepare_w); kper = 1 → none.
__global__ void __launch_bounds__(256) oz_rowscale_kernel(const double* __restrict__ W, int64_t ldw, int Np, int Np256, int eP,
                                                          int* __restrict__ sexp, int kper, int ktg) {
    const double cw = __builtin_ldexp(1.0, ktg);
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (row >= Np256) return;
    double l1 = 0.0, mx = 0.0;
    if (row < Np) {
        const double* w = W + (int64_t)row * ldw;
        for (int k = lane; k <= row; k += 64) {
            const double a = __builtin_fabs(w[k]) * ((kper > 1 && k % kper) ? cw : 1.0);
            l1 += a;
            mx = a > mx ? a : mx;
        }
    }
    for (int o = 32; o; o >>= 1) {
        l1 += __shfl_xor(l1, o);
        const double m2 = __shfl_xor(mx, o);
        mx = m2 > mx ? m2 : mx;
    }
    if (lane == 0) {
        int s = 0;
        if (mx > 0.0 && l1 < 1.0e300) {                // a NaN/Inf row keeps s = 0: its residues are garbage, its V is flagged
            int e1, e2;
            (void)frexp(l1, &e1);
            (void)frexp(mx, &e2);
            const int sa = eP - 53 - e1, sb = 52 - e2;
            s = sa < sb ? sa : sb;
        }
        sexp[row] = s;
    }
}

// output index of chunk row r of a gradient-enhanced model's all-output posterior (OzVarArgs::rmode): 0 = function value
__device__ __forceinline__ int oz_row_output(int rmode, int rper, int64_t r0, int64_t rpts, int r) {
    if (rmode == 1) return (int)((r0 + r) % rper);
    if (rmode == 2) return (int)((r0 + r) / rpts);
    return 0;
}

// ---- fp64 → residue planes -----------------------------------------------------------------------------------------------------------
// out[l][r][k] (int8, ld bytes per row, plane stride `plane`) = sym_residue(rint(in[r][k]·2^s), p_l) for r < rows_in, k < cols_in;
// zeros elsewhere up to rows_out × cols_out.  s = srow[r] when given, else sconst.  A thread converts 16 consecutive k of one row.
// bad[r] is set when a row holds a non-finite value (its residues mean nothing; the reconstruction writes NaN for that row).
struct OzQuantArgs {
    const double* in;
    int64_t ldin;
    int rows_in, cols_in, rows_out, cols_out;
    int lower;                 // 1: entries with k > r are zero (W = L⁻¹; the stored zeros are not even read)
    const int* srow;
    int sconst;
    int kper, ktg;             // columns k with k % kper != 0 get 2^ktg on top (kper ≤ 1: none)
    int rmode, rper, rtg;      // rows whose output index (oz_row_output) is not 0 get 2^rtg on top (rmode 0: none)
    int64_t r0, rpts;
    int8_t* out;
    int64_t ld, plane;
    int* bad;                  // [rows_out] or nullptr
    OzPlan pl;
};

__global__ void __launch_bounds__(256) oz_quant_kernel(OzQuantArgs a) {
    const int64_t gid = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int cpr = a.cols_out >> 4;                   // 16-byte groups per output row
    const int r = (int)(gid / cpr), kc = (int)(gid % cpr) * 16;
    if (r >= a.rows_out) return;
    double xh[16], xl[16];
    const bool live = r < a.rows_in && kc < a.cols_in && !(a.lower && kc > r);
    bool bad = false;
    if (live) {
        const int rt = oz_row_output(a.rmode, a.rper, a.r0, a.rpts, r) != 0 ? a.rtg : 0;
        const double sc = __builtin_ldexp(1.0, (a.srow ? a.srow[r] : a.sconst) + rt);
        const double cw = __builtin_ldexp(1.0, a.ktg);
        const double* src = a.in + (int64_t)r * a.ldin + kc;
#pragma unroll
        for (int q = 0; q < 16; ++q) {
            const bool in = kc + q < a.cols_in && !(a.lower && kc + q > r);
            const double v = in ? src[q] : 0.0;
            bad = bad || !(__builtin_fabs(v) < 1.0e300);
            oz_split(v, (a.kper > 1 && (kc + q) % a.kper) ? sc * cw : sc, xh[q], xl[q]);
        }
    }
    if (bad && a.bad) a.bad[r] = 1;
    int8_t* dst = a.out + oz_plane_off(r, kc, (int)(a.ld >> 6));      // 16 consecutive k never cross a 64-byte block row
    for (int l = 0; l < a.pl.n; ++l) {
        v4i_t w = {0, 0, 0, 0};
        if (live) {
            const double invp = a.pl.invp[l], pd = (double)a.pl.p[l], c26 = a.pl.c26[l];
#pragma unroll
            for (int q = 0; q < 16; ++q) {
                const int rr = sym_residue(xh[q], xl[q], c26, invp, pd) & 0xff;
                w[q >> 2] |= rr << (8 * (q & 3));
            }
        }
        *reinterpret_cast<v4i_t*>(dst + (int64_t)l * a.plane) = w;
    }
}

// ---- int8 NT GEMM with the symmetric-mod epilogue --------------------------------------------------------------------------------------
// U[l][i][j] = (Σ_{k < 256(ti+1)} WR[l][i][k]·KR[l][j][k]) mod p_l for the 256×256 tile (ti, tj) of modulus l.
// MFMA roles: D[m = j][n = i] — the A operand is the candidate tile, the B operand the W tile, so that a lane's four
// consecutive accumulator registers are four consecutive candidates of one row i (one packed dword of U).
// Workgroup = 8 waves as 4 (j) × 2 (i); wave tile 64 (j) × 128 (i) (the kernel and its LDS ring: "16×16×64 GEMM fed by LDS-DMA" below).
constexpr int OZ_T = 256;                  // tile edge

// U, the residues of the products, lives in 64 KB blocks — one per (256 × 256 tile, modulus): [256 rows i][256 candidates j], the n
// blocks of a tile back to back, tiles of one candidate block tj back to back over ti.  Round 6 (the plane-major [l][i][j] layout of
// rounds 2 - 5 made the reconstruction read 1 KB pieces 65 KB apart — 5.2 TB/s; a block is written by ONE epilogue as a contiguous
// 64 KB run and read by the reconstruction as contiguous 32 KB runs per modulus).
__host__ __device__ __forceinline__ int64_t oz_u_block(int ti, int tj, int l, int Ti, int n) {
    return (((int64_t)tj * Ti + ti) * n + l) * (int64_t)(OZ_T * OZ_T);
}

struct OzGemmArgs {
    const int8_t* KR;      // [n][Mc256][ldk]
    const int8_t* WR;      // [n][Np256][ldw]
    int8_t* U;             // [Tj][Ti][n] blocks of 64 KB: the residues of ONE 256 × 256 tile and modulus, [256 rows i][256 candidates j] (oz_u_block)
    int64_t sK, sW;
    int nhs;               // 64-byte half-stages per row of a residue plane (Np256 / 64): plane block (R, H) at (R·nhs + H)·16 KB
    int Ti, Tj, n;
    int tjg;               // column blocks per group (multiple of 8; 64 unless Tj is smaller)
    double invp[OZ_MAXMOD];
    int p[OZ_MAXMOD];
};

// Tile order (oz16p_decode_ticket): groups of (4 row blocks × tjg column blocks) of ONE modulus, heaviest row blocks first; inside a
// group the tiles that land on one XCD form a 4 × (tjg/8) patch that shares its panels in L2.  Row-block offsets inside a group
// alternate direction from group to group: the CUs that ran a group's lightest tiles are free first and take the next group's first
// tiles — which are then its heaviest, so the four row blocks of a patch do not drift apart in k (the candidate panels they share stay
// in L2 only while they walk k together; L2 hit rate 60 % → 78 %).  Row groups innermost: the residue planes of one modulus for 64
// column blocks (128 MB at N = 8192) are swept by all row groups back to back and stay in the Infinity Cache meanwhile.

// ---- epilogue pieces shared by the GEMM kernels -------------------------------------------------------------------------------------------
// four accumulators (four consecutive candidates of one W row) → their symmetric residues mod p, one byte each: q = rint(x/p) is exact
// (|x| < 2^31, so x/p in fp64 is far closer to the true quotient than 1/(2p)) and r = x − q·p lands in [−p/2, p/2].
// Three VALU operations per accumulator (the persistent kernel's epilogue: its residue
// arithmetic is issue-bound, two waves per SIMD taking turns — 4.7 of a tile's 52 µs with the matrix pipe idle): the product x·(1/p)
// is rounded to the nearest integer by adding 1.5·2^52 inside the fma (one rounding instead of two, the same integer: x/p is at
// least 1/(2p) away from a half-integer for odd p), the quotient is then the low dword of the sum's mantissa as it stands — no
// v_rndne, no conversion back —, and r = x − q·p is one v_mad_i32_i24 (|q| ≤ 2^30/199 < 2^23 for every row count the engine takes).
// p = 256: ties round to even.
__device__ __forceinline__ int oz_mod_pack4_mad(int a0, int a1, int a2, int a3, double invp, int p) {
    const int v[4] = {a0, a1, a2, a3};
    int r[4];
#pragma unroll
    for (int b = 0; b < 4; ++b) {
        const double t = __builtin_fma((double)v[b], invp, 6755399441055744.0);
        const int q = __double2loint(t);
        // (as asm: from C the compiler picks the quarter-rate v_mul_lo_u32)
        asm("v_mad_i32_i24 %0, %1, %2, %3" : "=v"(r[b]) : "v"(q), "s"(-p), "v"(v[b]));
    }
    // the four low bytes into one dword: three v_perm_b32 (selector bytes 0-3: second source, 4-7: first source)
    const unsigned lo = __builtin_amdgcn_perm((unsigned)r[1], (unsigned)r[0], 0x0c0c0400u);
    const unsigned hi = __builtin_amdgcn_perm((unsigned)r[3], (unsigned)r[2], 0x0c0c0400u);
    return (int)__builtin_amdgcn_perm(hi, lo, 0x05040100u);
}

struct OzFragA { v4i_t a[4]; };
struct OzFragB { v4i_t b[4]; };

// a unit: 4 A fragments × 4 B fragments = 16 MFMAs on columns 4·half … 4·half+3 of the wave tile
__device__ __forceinline__ void oz16_mma(const OzFragA& fa, const OzFragB& fb, int half, v4i_t (&acc)[4][8]) {
#pragma unroll
    for (int m = 0; m < 4; ++m)
#pragma unroll
        for (int nn = 0; nn < 4; ++nn)
            acc[m][4 * half + nn] = __builtin_amdgcn_mfma_i32_16x16x64_i8(fa.a[m], fb.b[nn], acc[m][4 * half + nn], 0, 0, 0);
}

// ---- 16×16×64 GEMM fed by LDS-DMA --------------------------------------------------------------------------------------------------------
// Wave tile 64 candidates × 128 W rows = 4 × 8 MFMA tiles of 16×16 (D layout of v_mfma_i32_16x16x64_i8: lane → column n = lane % 16,
// register r → row m = 4(lane/16) + r).  On random operands the chip holds a higher clock with the 16×16×64 shape than with 32×32×32
// (tools/mfma_i8_power_probe.hip: 3.41 against 3.11 POP/s sustained).  A register-staged kernel can keep only half a stage of global
// loads in flight (246–253 VGPRs); an L2 miss (22 % of the requests) is then waited for with the matrix pipe idle (SQ_WAIT_ANY 33 % of
// wave time).  Here the tiles go global → LDS directly
// (global_load_lds_dwordx4: no staging registers, no ds_write), in HALF-stages of 64 k-bytes through a ring of four 32 KB slots,
// three half-stages ahead of the one being multiplied.
// LDS image of a slot: [A tile: 256 rows × 64 B][B tile: 256 rows × 64 B], the four 16-byte chunks of a row XOR-swizzled,
// chunk c at position c ^ ((row >> 1) & 3).  A DMA wave-instruction writes 1 KiB lane-linearly (16 rows), so the swizzle is applied to
// the per-lane SOURCE address: lane L fills row L/4, position L%4, and fetches chunk (L%4) ^ ((L>>3)&3) of that row.
// Iteration h (half-stage h in slot h%4; a unit = 4 A fragments × 4 B fragments = 16 MFMAs):
//     issue 4 DMA pieces of half-stage h+3 → slot (h+3)%4        (last read in iteration h−1, behind the barrier just passed)
//     MFMA unit (h−1, B4-7)   | 8 ds_read: A(h), B0-3(h)
//     MFMA unit (h,   B0-3)   | 4 ds_read: B4-7(h)
//     s_waitcnt vmcnt(8) — this wave's pieces of half-stage h+1 have landed — and lgkmcnt(0); raw s_barrier
// (a DMA'd buffer is read only behind the counted vmcnt of the issuing waves and a barrier the reader has passed.)
constexpr int OZ_HS = 64;                       // k-bytes per half-stage
constexpr int OZ_SLOT = 2 * OZ_T * OZ_HS;       // 32 768 bytes: A tile + B tile
typedef __attribute__((address_space(3))) void* oz_lds_ptr;

struct OzDmaCtx {
    const int8_t* ab; const int8_t* bb;         // the wave's first piece of each operand (uniform)
    int64_t astep, bstep;                       // 128 rows further down each operand: the wave's second piece (uniform)
    unsigned ao, bo;                            // per-lane source offset inside a piece's 16 rows: (L/4)·ld + 16·chunk
    int wave;
};

template <int SLOT>
__device__ __forceinline__ void oz_dma_issue(char* lds, const OzDmaCtx& c, int h) {
    const int k = h * 16384;
#pragma unroll
    for (int q = 0; q < 2; ++q) {
        char* da = lds + SLOT * OZ_SLOT + (c.wave + 8 * q) * 1024;
        char* db = da + OZ_T * OZ_HS;
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(c.ab + k + q * c.astep + c.ao), (oz_lds_ptr)da, 16, 0, 0);
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(c.bb + k + q * c.bstep + c.bo), (oz_lds_ptr)db, 16, 0, 0);
    }
}

// one DMA piece: q = 0/1 the wave's first / second 16-row group, op = 0 candidates (A), 1 W rows (B)
template <int SLOT>
__device__ __forceinline__ void oz_dma_piece(char* lds, const OzDmaCtx& c, int k, int q, int op) {
    char* d = lds + SLOT * OZ_SLOT + op * (OZ_T * OZ_HS) + (c.wave + 8 * q) * 1024;
    const int8_t* g = op ? c.bb + k + q * c.bstep + c.bo : c.ab + k + q * c.astep + c.ao;
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g, (oz_lds_ptr)d, 16, 0, 0);
}

__device__ __forceinline__ void oz16_mma_row(const OzFragA& fa, const OzFragB& fb, int half, int m, v4i_t (&acc)[4][8]) {
#pragma unroll
    for (int nn = 0; nn < 4; ++nn)
        acc[m][4 * half + nn] = __builtin_amdgcn_mfma_i32_16x16x64_i8(fa.a[m], fb.b[nn], acc[m][4 * half + nn], 0, 0, 0);
}

#define OZ_FENCE() __builtin_amdgcn_sched_barrier(0)

// every step issues (past the end of the tile: a harmless re-fetch of the last half-stage into a slot nobody reads any more), so
// the in-flight count is the same in every iteration and one loop body serves the whole tile.  The step is cut into groups of
// four MFMAs with the memory instructions dealt out between them (scheduling fences pin the order): four DMA pieces back to back
// hold the wave's issue for several hundred cycles with both waves of the SIMD at the same point of their streams.
// DIAG: the step belongs to the tile's diagonal block (its last four half-stages, hs = 0 … 3): W is lower-triangular, so the residues
// of rows r of the block vanish for k > r, and a unit whose 64 rows all lie above the half-stage's first k-byte multiplies zeros —
// unit 0 of wave row wi (rows 128wi … +63) from hs = 2wi + 1 on, unit 1 (rows 128wi+64 … +127) from hs = 2wi + 2 on: 6 of the 16
// units of a diagonal block are skipped (2.3 % of a C3 launch's MFMAs; measured −1 %).
template <int SLOT, bool DIAG>
__device__ __forceinline__ void oz16d_step_k(char* lds, const OzDmaCtx& c, int k, int ra, int rb, OzFragA (&A)[2], OzFragB& Bx,
                                             OzFragB& By, v4i_t (&acc)[4][8], int hs = 0, int wi = 0) {
    const bool sk_prev = DIAG && (hs - 1 >= 2 * wi + 2);      // the held-back unit is unit 1 of half-stage hs − 1
    const bool sk_u0 = DIAG && (hs >= 2 * wi + 1);
    const char* slot = lds + SLOT * OZ_SLOT;
    const char* pa = slot + ra;
    const char* pb = slot + rb;
    OzFragA& An = A[SLOT & 1];
    const OzFragA& Ao = A[(SLOT & 1) ^ 1];
    constexpr int NS = (SLOT + 3) & 3;
    // the previous half-stage's held-back unit (Ao × By, columns 4-7) with this half-stage's A and B0-3 fragments arriving
    An.a[0] = *reinterpret_cast<const v4i_t*>(pa);
    An.a[1] = *reinterpret_cast<const v4i_t*>(pa + 16 * OZ_HS);
    if (!sk_prev) oz16_mma_row(Ao, By, 1, 0, acc);
    OZ_FENCE();
    An.a[2] = *reinterpret_cast<const v4i_t*>(pa + 32 * OZ_HS);
    An.a[3] = *reinterpret_cast<const v4i_t*>(pa + 48 * OZ_HS);
    if (!sk_prev) oz16_mma_row(Ao, By, 1, 1, acc);
    OZ_FENCE();
    oz_dma_piece<NS>(lds, c, k, 0, 0);
    Bx.b[0] = *reinterpret_cast<const v4i_t*>(pb);
    Bx.b[1] = *reinterpret_cast<const v4i_t*>(pb + 16 * OZ_HS);
    if (!sk_prev) oz16_mma_row(Ao, By, 1, 2, acc);
    OZ_FENCE();
    oz_dma_piece<NS>(lds, c, k, 0, 1);
    Bx.b[2] = *reinterpret_cast<const v4i_t*>(pb + 32 * OZ_HS);
    Bx.b[3] = *reinterpret_cast<const v4i_t*>(pb + 48 * OZ_HS);
    if (!sk_prev) oz16_mma_row(Ao, By, 1, 3, acc);
    OZ_FENCE();
    // this half-stage's first unit (An × Bx, columns 0-3) with the B4-7 fragments arriving
    oz_dma_piece<NS>(lds, c, k, 1, 0);
    By.b[0] = *reinterpret_cast<const v4i_t*>(pb + 64 * OZ_HS);
    By.b[1] = *reinterpret_cast<const v4i_t*>(pb + 80 * OZ_HS);
    if (!sk_u0) oz16_mma_row(An, Bx, 0, 0, acc);
    OZ_FENCE();
    oz_dma_piece<NS>(lds, c, k, 1, 1);
    By.b[2] = *reinterpret_cast<const v4i_t*>(pb + 96 * OZ_HS);
    By.b[3] = *reinterpret_cast<const v4i_t*>(pb + 112 * OZ_HS);
    if (!sk_u0) oz16_mma_row(An, Bx, 0, 1, acc);
    OZ_FENCE();
    if (!sk_u0) oz16_mma_row(An, Bx, 0, 2, acc);
    if (!sk_u0) oz16_mma_row(An, Bx, 0, 3, acc);
    OZ_FENCE();
    asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    OZ_FENCE();
}

// ---- the GEMM: a PERSISTENT kernel ---------------------------------------------------------------------------------------------------------
// One workgroup per CU stays resident and works through tiles of the launch's tile list (dealt out dynamically, below).  What a
// tile cost beyond its MFMAs in a one-tile-per-workgroup kernel (round 2) — the dispatch of a 128 KB-LDS
// workgroup onto the CU that just drained, three half-stages of DMA latency before the first MFMA, the wait for the tail's
// re-fetches before the epilogue may overlay the ring — was about 7 of 54 µs at N = 8192 and half of the tile at N = 1024, with
// nothing else resident on the CU to hide it.  Here the last three steps of a tile fetch the first three half-stages of the NEXT
// tile into ring slots 0-2 (a tile is a multiple of four half-stages: the ring phase carries over), and the epilogue keeps out of
// their way: the residues leave through slot 3 alone, 64 W rows at a time in two 16 KB buffers (rows of 256 bytes, 16-byte chunks
// rotated by the row index: conflict-free for the dword writes and the 16-byte reads).
__device__ __forceinline__ void oz16p_ctx(const OzGemmArgs& a, int ti, int tj, int l, int wave, OzDmaCtx& c) {
    c.ab = a.KR + (int64_t)l * a.sK + ((int64_t)tj * a.nhs) * 16384 + 1024 * wave;
    c.bb = a.WR + (int64_t)l * a.sW + ((int64_t)ti * a.nhs) * 16384 + 1024 * wave;
}

// tools/oz_dev.hip defines OZ_PROBE: workgroup 0 sums the 100 MHz clock over its k loops and its epilogues
#ifdef OZ_PROBE
__device__ long long oz_probe_acc[16];
#define OZ_PROBE_T(v) const long long v = wall_clock64()
#define OZ_PROBE_ADD(i, d) do { if (blockIdx.x == 0 && threadIdx.x == 0) oz_probe_acc[i] += (d); } while (0)
#else
#define OZ_PROBE_T(v) do { } while (0)
#define OZ_PROBE_ADD(i, d) do { } while (0)
#endif

__device__ __forceinline__ void oz16p_epilogue(const OzGemmArgs& a, char* slot3, v4i_t (&acc)[4][8], int l, int ti, int tj, int wi, int wj,
                                               int lane, int tid) {
    const double invp = a.invp[l];
    const int pm = a.p[l];
    // the epilogue's lane constants are recomputed per tile from the lane id (v_mbcnt) and the wave's scalar coordinates: kept across
    // the tile loop they would sit in registers the k-loop has none to spare for (and were spilled to scratch)
    lane = (int)__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));
    tid = 64 * (4 * wi + wj) + lane;            // wi, wj are wave-uniform (scalar registers)
#pragma unroll
    for (int m = 0; m < 4; ++m)          // the four residues of an accumulator quad, packed, take the place of its first register
#pragma unroll
        for (int nn = 0; nn < 8; ++nn) acc[m][nn][0] = oz_mod_pack4_mad(acc[m][nn][0], acc[m][nn][1], acc[m][nn][2], acc[m][nn][3], invp, pm);
    OZ_PROBE_T(pe1);
    OZ_PROBE_ADD(4, pe1);                              // (probe build: Σ of the clock behind the residue arithmetic)
    int8_t* up = a.U + oz_u_block(ti, tj, l, a.Ti, a.n);      // the tile's 64 KB block: rows of 256 bytes, contiguous
    const int r15 = lane & 15;
    // a thread's two 16-byte pieces of a pass: rows tid/16 and tid/16 + 32 (same rotation: the rows are 32 apart), chunk tid%16
    const int sr = tid >> 4, sc = tid & 15;
    const unsigned lds_off = (unsigned)(sr * 256 + ((16 * sc + 16 * (sr & 15)) & 255));
    const unsigned g_off = (unsigned)(sr * OZ_T + 16 * sc);               // 32-bit lane offset on a uniform row base
    const unsigned g_step = (unsigned)(32 * OZ_T);
#pragma unroll
    for (int P = 0; P < 5; ++P) {
        if (P < 4 && wi == (P >> 1)) {              // pass P: W rows 64P … 64P+63 of the tile = row groups nn = 4(P&1) … +3 of wave row P/2
            char* buf = slot3 + (P & 1) * 16384;
#pragma unroll
            for (int m = 0; m < 4; ++m)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int r = 16 * q + r15;
                    const int jl = 64 * wj + 16 * m + 4 * (lane >> 4);
                    *reinterpret_cast<int*>(buf + r * 256 + ((jl + 16 * r15) & 255)) = acc[m][4 * (P & 1) + q][0];
                }
        }
        if (P > 0) {                                // rows of pass P−1 leave as ONE contiguous 16 KB run of the tile's block
            const char* buf = slot3 + ((P - 1) & 1) * 16384;
            int8_t* rows = up + (64 * (P - 1)) * OZ_T;                    // uniform
            const v4i_t w0 = *reinterpret_cast<const v4i_t*>(buf + lds_off);
            const v4i_t w1 = *reinterpret_cast<const v4i_t*>(buf + lds_off + 32 * 256);
            *reinterpret_cast<v4i_t*>(rows + g_off) = w0;
            *reinterpret_cast<v4i_t*>(rows + g_off + g_step) = w1;
        }
        // workgroup barrier for LDS traffic only: __syncthreads() would also wait for the stores just issued (vmcnt(0), a round trip
        // to L2 per pass) and for the next tile's half-stages still in flight
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
#ifdef OZ_PROBE
        { OZ_PROBE_T(pp); OZ_PROBE_ADD(8 + P, pp); }
#endif
    }
}

// The tile list is dealt out dynamically, one list per XCD: list x holds the blocks with blockIdx % 8 == x of the one-tile kernel's
// order (what the round-robin dispatch gives XCD x there), and a workgroup takes the next entry of the list of the XCD it runs on
// (HW_REG_XCC_ID) with one atomic increment — the 32 CUs of an XCD so work through consecutive entries, i.e. through a 4 × (tjg/8)
// patch at a time, whatever the launch's shape (row blocks not a multiple of four, ragged last column group: a static split
// leaves CUs idle there).  An exhausted list sends its workgroups to the next XCD's list, so every tile is taken whatever the
// XCD numbering of the partition mode.  The next tile's ticket is drawn inside the k loop, late (see the kernel), decoded by the
// drawing lane and handed to the other waves through one LDS word.

// x / d and x % d for 0 ≤ x < 2^23, 0 < d < 2^23 without the integer-division expansion (one thread decodes a ticket per tile: the
// seven divisions of the plain decode were 0.7 µs in front of a workgroup barrier): fp32 estimate, one correction step each way
__device__ __forceinline__ void oz_divmod(int x, int d, float rd, int& q, int& r) {
    q = (int)((float)x * rd);
    r = x - q * d;
    if (r < 0) { --q; r += d; }
    if (r >= d) { ++q; r -= d; }
}

// ticket q of list y → the tile (ti, tj, l) packed as ti | tj << 9 | l << 20, or −1 for a padding block: block
// (q / cpx)·per_group + (q % cpx)·8 + y of the order described at OzGemmArgs
__device__ __forceinline__ int oz16p_decode_ticket(const OzGemmArgs& a, int q, int y) {
    const int cpx = a.tjg >> 1, cols_x = a.tjg >> 3;           // per_group / 8, column blocks per XCD patch
    const int ngi = (a.Ti + 3) / 4;
    int grp, c, t, gg, gh, gl, ro, cc;
    oz_divmod(q, cpx, 1.0f / (float)cpx, grp, c);
    oz_divmod(grp, ngi, 1.0f / (float)ngi, t, gg);
    oz_divmod(t, a.n, 1.0f / (float)a.n, gh, gl);
    oz_divmod(c, cols_x, 1.0f / (float)cols_x, ro, cc);
    const int ti = 4 * (ngi - 1 - gg) + ((grp & 1) ? 3 - ro : ro);
    const int tj = gh * a.tjg + y * cols_x + cc;
    return (ti < a.Ti && tj < a.Tj) ? (ti | (tj << 9) | (gl << 20)) : -1;
}

__device__ __forceinline__ int oz16p_take(const OzGemmArgs& a, int* ctr, int x, int per_list, int q) {
    // q: a ticket already drawn from list x (or −1: none)
    for (int t = 0; t < 8; ++t) {
        const int y = (x + t) & 7;
        for (;;) {
            if (t > 0 || q < 0) q = __hip_atomic_fetch_add(ctr + y, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (q >= per_list) break;
            const int tile = oz16p_decode_ticket(a, q, y);
            if (tile >= 0) return tile;
            q = -1;                                   // a padding block of the list: draw again
        }
        q = -1;
    }
    return -1;
}

__global__ void __launch_bounds__(512) oz_gemm16p_kernel(OzGemmArgs a, int total, int* ctr) {
    // ONE LDS object (the ring, and a word behind it for the next tile): with a second __shared__ variable the compiler
    // starts to order every ds_read behind every LDS-DMA piece in flight (s_waitcnt vmcnt(0) before each fragment read: measured
    // 2× slower) — with a single object it leaves that ordering to the counted waits of the steps
    __shared__ __attribute__((aligned(1024))) char oz_lds[4 * OZ_SLOT + 16];
    int* oz_pick = reinterpret_cast<int*>(oz_lds + 4 * OZ_SLOT);
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wj = wave & 3, wi = wave >> 2;
    const int per_list = total >> 3;
    const int x = (int)__builtin_amdgcn_s_getreg((3 << 11) | (0 << 6) | 20) & 7;      // hwreg(HW_REG_XCC_ID, 0, 4)
    if (tid == 0) oz_pick[0] = oz16p_take(a, ctr, x, per_list, -1);
    __syncthreads();
    const int cur = __builtin_amdgcn_readfirstlane(oz_pick[0]);
    __syncthreads();
    if (cur < 0) return;                              // uniform over the workgroup
    int ti = cur & 511, tj = (cur >> 9) & 2047, l = cur >> 20;
    OzDmaCtx c, cn;
    c.wave = wave;
    {
        const int chunk = (lane & 3) ^ ((lane >> 3) & 3);
        c.astep = 8192;
        c.bstep = 8192;
        c.ao = (unsigned)((lane >> 2) * 64 + 16 * chunk);
        c.bo = c.ao;
    }
    cn = c;
    oz16p_ctx(a, ti, tj, l, c.wave, c);
    const int ro = (lane & 15) * OZ_HS + (((lane >> 4) ^ ((lane >> 1) & 3)) * 16);
    const int ra = (64 * wj) * OZ_HS + ro;
    const int rb = OZ_T * OZ_HS + (128 * wi) * OZ_HS + ro;
    const bool drawer = tid == 256;                   // lane 0 of wave 4 draws and decodes the tickets

    oz_dma_issue<0>(oz_lds, c, 0);
    oz_dma_issue<1>(oz_lds, c, 1);
    oz_dma_issue<2>(oz_lds, c, 2);
    asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
    for (;;) {
        const int nh = 4 * (ti + 1);
        // The ticket of the NEXT tile is drawn late — three loop iterations (12 half-stages, ~8 µs) before the diagonal steps that need
        // it, or at the tile's start when the tile is shorter —, so that tiles start in the order of their tickets, as they do under
        // the one-tile kernel's dispatch: drawn a whole tile ahead (first version) the tiles of a patch started up to a tile-length
        // difference apart, the CUs of an XCD drifted in k, and the L2 hit rate fell from 75 % to 67 % (+33 % L2-miss traffic,
        // tools/pmc_l2_ab.sh).  The atomic is issued by hand (the compiler's waits for its answer on the spot); nobody waits for it:
        // vector memory operations return in order, so two steps' s_waitcnt vmcnt(8) later it has come back.
        const int hdraw = nh >= 20 ? nh - 20 : 0;    // drawn here …
        const int hcons = nh >= 12 ? nh - 8 : -1;     // … decoded and published at the top of the last loop iteration (≥ 4 steps later)
        int ticket = per_list;
        const int one = 1;
        if (drawer && hdraw == 0) asm volatile("global_atomic_add %0, %1, %2, off sc0" : "=v"(ticket) : "v"(ctr + x), "v"(one) : "memory");

        OZ_PROBE_T(pt0);
        v4i_t acc[4][8];
#pragma unroll
        for (int m = 0; m < 4; ++m)
#pragma unroll
            for (int nn = 0; nn < 8; ++nn) acc[m][nn] = v4i_t{0, 0, 0, 0};
        OzFragA A[2];
        OzFragB Bx, By;
#pragma unroll
        for (int m = 0; m < 4; ++m) { A[1].a[m] = v4i_t{0, 0, 0, 0}; By.b[m] = v4i_t{0, 0, 0, 0}; }   // the first step's held-back unit adds 0

        int hb = 0;
        for (; hb < nh - 4; hb += 4) {
            if (drawer) {
                if (hb == hdraw && hb != 0) asm volatile("global_atomic_add %0, %1, %2, off sc0" : "=v"(ticket) : "v"(ctr + x), "v"(one) : "memory");
                if (hb == hcons) {                    // the ticket's answer: decoded, and handed to the other waves through the LDS word
                    asm volatile("" : "+v"(ticket));
                    oz_pick[0] = oz16p_take(a, ctr, x, per_list, ticket);
                }
            }
            oz16d_step_k<0, false>(oz_lds, c, (hb + 3) * 16384, ra, rb, A, Bx, By, acc);
            oz16d_step_k<1, false>(oz_lds, c, (hb + 4) * 16384, ra, rb, A, Bx, By, acc);
            oz16d_step_k<2, false>(oz_lds, c, (hb + 5) * 16384, ra, rb, A, Bx, By, acc);
            oz16d_step_k<3, false>(oz_lds, c, (hb + 6) * 16384, ra, rb, A, Bx, By, acc);
        }
        if (hcons < 0) {                              // a one- or two-block tile: nothing (or too little) has been waited for since the draw
            if (drawer) {
                asm volatile("s_waitcnt vmcnt(0)" : "+v"(ticket) :: "memory");
                oz_pick[0] = oz16p_take(a, ctr, x, per_list, ticket);
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
        }
        // the word has been behind at least one of the steps' barriers: read before the first diagonal step, used behind it
        int nbv;
        asm volatile("ds_read_b32 %0, %1" : "=v"(nbv) : "v"((unsigned)(size_t)(oz_lds_ptr)oz_pick) : "memory");
        oz16d_step_k<0, true>(oz_lds, c, (hb + 3) * 16384, ra, rb, A, Bx, By, acc, 0, wi);      // the diagonal block; hb + 3 = nh − 1
        asm volatile("" : "+v"(nbv));                 // (that step's closing lgkmcnt(0) has seen the read come back)
        const int nb = __builtin_amdgcn_readfirstlane(nbv);
        const bool have = nb >= 0;                    // uniform over the workgroup
        const int ti2 = nb & 511, tj2 = (nb >> 9) & 2047, l2 = nb >> 20;
        int k1, k2, k3;
        if (have) { oz16p_ctx(a, ti2, tj2, l2, c.wave, cn); k1 = 0; k2 = 16384; k3 = 32768; }
        else { cn.ab = c.ab; cn.bb = c.bb; k1 = k2 = k3 = (nh - 1) * 16384; }      // no next tile: a harmless re-fetch of the last half-stage
        oz16d_step_k<1, true>(oz_lds, cn, k1, ra, rb, A, Bx, By, acc, 1, wi);                   // … fetching the next tile's first half-stages
        oz16d_step_k<2, true>(oz_lds, cn, k2, ra, rb, A, Bx, By, acc, 2, wi);
        oz16d_step_k<3, true>(oz_lds, cn, k3, ra, rb, A, Bx, By, acc, 3, wi);
        if (wi != 0) oz16_mma(A[1], By, 1, acc);          // the held-back unit of the last half-stage (slot 3 → A[1]); zeros for wave row 0
        // the step's closing barrier: every wave has its fragments of slot 3 in registers — the slot is free for the residues
        OZ_PROBE_T(pt1);
        OZ_PROBE_ADD(5, pt1);
        oz16p_epilogue(a, oz_lds + 3 * OZ_SLOT, acc, l, ti, tj, wi, wj, lane, tid);
        // (the epilogue's closing barrier: nobody still reads slot 3.  Nothing else is waited for here: the last step's vmcnt(8) saw
        // the next tile's half-stage 0 land, and the steps' vmcnt(8) keeps meaning "at most the two newest half-stages are in
        // flight" with this tile's stores in the count — they only make the first waits conservative)
        OZ_PROBE_T(pt2);
        OZ_PROBE_ADD(0, pt1 - pt0); OZ_PROBE_ADD(1, pt2 - pt1); OZ_PROBE_ADD(2, 1); OZ_PROBE_ADD(3, nh);
        if (!have) break;
        ti = ti2; tj = tj2; l = l2;
        c.ab = cn.ab; c.bb = cn.bb;
    }
}

// ---- reconstruction + squares + column sums ------------------------------------------------------------------------------------------
// partial[tb][j] = Σ_{i in row block tb (128 rows), i < nvalid} V[i][j]²,  V = CRT(U[·][i][j])·2^−(s_i + sK).
struct OzCrtArgs {
    const int8_t* U;       // oz_u_block layout
    int Ti;                // 256-row blocks of U (Np256 / 256)
    const int* sexp;       // s_i
    int sK;
    const int* bad_row;    // [Np256] W rows holding non-finite values (or nullptr)
    const int* bad_col;    // [Mc256] candidates holding non-finite kernel values (or nullptr)
    double* partial;       // [Np/128][ldp]
    int64_t ldp;
    int Mc;                // columns to write (multiple of 128)
    int nvalid;
    int rmode, rper, rtg;  // candidate j is a derivative output (oz_row_output != 0): its image was taken at 2^-rtg, sums get 2^(2 rtg)
    int64_t r0, rpts;
    int* ctr_reset;        // the residue GEMM's tile counters (OZ_CTR_INTS): zeroed here, behind the GEMM that used them (or nullptr)
    OzPlan pl;
};

// A workgroup = 4 waves over the 128 rows × 256 candidates that one tile of U holds for row block tb: lane (cg = lane % 16, rq = lane / 16)
// takes the 16 candidates 16·cg … of the rows ≡ rq (mod 4) of its wave's 32 rows — a wave instruction reads 4 rows × 256 bytes = ONE
// contiguous KB of a modulus' block, eight of them back to back per modulus.  The column sums over a wave's rows meet by two xor
// shuffles (rq), the four waves' in LDS, in a fixed order.
// NM > 0: the moduli count is the compile-time NM (the default plan): the residues of a row's NM planes are fetched by NM loads issued
// back to back — each wave keeps NM × 1 KiB in flight instead of one load per loop trip, which is what a kernel that reads 7.5 GB once
// needs to approach the HBM rate — and read past the caches (non-temporal: nothing here is touched twice).
template <int NM>
__global__ void __launch_bounds__(256) oz_crt_kernel(OzCrtArgs a) {
    __shared__ double red[3][16][17];
    const int tb = blockIdx.y, tj = blockIdx.x;
    if (a.ctr_reset && tb == 0 && tj == 0 && threadIdx.x < 8) a.ctr_reset[threadIdx.x] = 0;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int cg = lane & 15, rq = lane >> 4;
    const int j = tj * OZ_T + 16 * cg;
    const bool live = j < a.Mc;                             // Mc is a multiple of 128: a thread's 16 candidates are all in or all out
    double sum[16];
#pragma unroll
    for (int b = 0; b < 16; ++b) sum[b] = 0.0;
    bool bad = false;
    const int n = NM > 0 ? NM : a.pl.n;
    const int rt0 = (tb & 1) * 128 + wave * 32 + rq;        // this lane's first row inside the tile
    const int8_t* ub = a.U + oz_u_block(tb >> 1, tj, 0, a.Ti, n) + (int64_t)rt0 * OZ_T + 16 * cg;
    const int ig0 = (tb >> 1) * OZ_T + rt0;                 // … and in the matrix
    if (live) {
        for (int step = 0; step < 8; ++step) {
            const int i = ig0 + 4 * step;
            if (i >= a.nvalid) break;                       // rows ≥ nvalid of the last block are padding for this view
            const int8_t* u = ub + 4 * step * OZ_T;
            v4i_t wl[NM > 0 ? NM : 1];
            if constexpr (NM > 0) {
#pragma unroll
                for (int l = 0; l < NM; ++l) wl[l] = __builtin_nontemporal_load(reinterpret_cast<const v4i_t*>(u + (int64_t)l * (OZ_T * OZ_T)));
            }
            const double sc = __builtin_ldexp(1.0, -(a.sexp[i] + a.sK));
            if (a.bad_row && a.bad_row[i]) bad = true;
            // four candidates (one dword of every plane) at a time: 8 partial sums live instead of 32 — 4 waves per SIMD instead of 3
#pragma unroll
            for (int g4 = 0; g4 < 4; ++g4) {
                double c1[4], c2[4];
#pragma unroll
                for (int b = 0; b < 4; ++b) { c1[b] = 0.0; c2[b] = 0.0; }
#pragma unroll
                for (int l = 0; l < n; ++l) {
                    const double s1 = a.pl.s1[l], s2 = a.pl.s2[l];
                    unsigned wx;
                    if constexpr (NM > 0) wx = (unsigned)wl[l][g4] ^ 0x80808080u;
                    else wx = (unsigned)*reinterpret_cast<const int*>(u + (int64_t)l * (OZ_T * OZ_T) + 4 * g4) ^ 0x80808080u;
#pragma unroll
                    for (int b = 0; b < 4; ++b) {
                        // signed byte → double without v_cvt_f64_i32 (quarter rate on this chip: with 14 of them per product the kernel
                        // was bound by the conversion, not by its 14 bytes per product): the byte biased by 128 becomes the low
                        // mantissa bits of 2^52, and (2^52 + b + 128) − (2^52 + 128) is the byte's value, exactly
                        const unsigned ub8 = (wx >> (8 * b)) & 0xffu;
                        const double ud = __hiloint2double(0x43300000, (int)ub8) - 4503599627370624.0;
                        c1[b] = __builtin_fma(ud, s1, c1[b]);
                        c2[b] = __builtin_fma(ud, s2, c2[b]);
                    }
                }
#pragma unroll
                for (int b = 0; b < 4; ++b) {
                    const double Q = __builtin_rint((c1[b] + c2[b]) * a.pl.invP);
                    const double cp = __builtin_fma(-Q, a.pl.P1, c1[b]) + __builtin_fma(-Q, a.pl.P2, c2[b]);
                    const double v = cp * sc;
                    sum[4 * g4 + b] = __builtin_fma(v, v, sum[4 * g4 + b]);
                }
            }
        }
    }
    if (bad) sum[0] = __builtin_nan("");                    // a non-finite W row poisons the whole block of rows, as it does in fp64
#pragma unroll
    for (int b = 0; b < 16; ++b) {                          // the four row classes of the wave: lanes cg, cg + 16, cg + 32, cg + 48
        sum[b] += __shfl_xor(sum[b], 16);
        sum[b] += __shfl_xor(sum[b], 32);
    }
    if (wave > 0 && rq == 0) {
#pragma unroll
        for (int b = 0; b < 16; ++b) red[wave - 1][cg][b] = sum[b];
    }
    __syncthreads();
    if (wave == 0 && rq == 0 && live) {
        const bool anybad = sum[0] != sum[0] || red[0][cg][0] != red[0][cg][0] || red[1][cg][0] != red[1][cg][0] ||
                            red[2][cg][0] != red[2][cg][0];
        const double nan = __builtin_nan("");
#pragma unroll
        for (int b = 0; b < 16; ++b) {
            const double t = ((sum[b] + red[0][cg][b]) + red[1][cg][b]) + red[2][cg][b];
            const bool bb = anybad || (a.bad_col && a.bad_col[j + b]);
            const double cf = (a.rmode && oz_row_output(a.rmode, a.rper, a.r0, a.rpts, j + b) != 0) ? __builtin_ldexp(1.0, 2 * a.rtg) : 1.0;
            a.partial[(int64_t)tb * a.ldp + j + b] = bb ? nan : t * cf;
        }
    }
}

// The same reconstruction with V itself as the result (the per-point covariance blocks of a gradient-enhanced model's outputs need
// products of different columns of V, not only the squares): Vout[j][i] = CRT(U[·][i][j])·2^−(s_i + sK)·(2^rtg for derivative candidates),
// candidate-major with ldv doubles per candidate.  A workgroup = 4 waves over 128 rows × 256 candidates; a lane reconstructs 4
// candidates (one dword per residue plane and row) of 8 rows at a time, the wave's 8 × 256 block is transposed through LDS and leaves
// as 64-byte runs of 8 consecutive rows per candidate.
struct OzCrtVArgs {
    const int8_t* U;       // oz_u_block layout
    int Ti;
    const int* sexp;
    int sK;
    const int* bad_row;
    const int* bad_col;
    double* V;
    int64_t ldv;
    int Mc, nvalid;
    int rmode, rper, rtg;
    int64_t r0, rpts;
    int* ctr_reset;        // the residue GEMM's tile counters (OZ_CTR_INTS): zeroed here, behind the GEMM that used them (or nullptr)
    OzPlan pl;
};

__global__ void __launch_bounds__(256) oz_crt_v_kernel(OzCrtVArgs a) {
    __shared__ double tile[4][256][9];
    if (a.ctr_reset && blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x < 8) a.ctr_reset[threadIdx.x] = 0;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int j = blockIdx.x * 256 + 4 * lane;
    const bool live = j < a.Mc;                              // Mc is a multiple of 128, blocks cover 256: whole quads are in or out
    const int n = a.pl.n;
    double cf[4];
    bool cbad[4];
#pragma unroll
    for (int b = 0; b < 4; ++b) {
        cf[b] = (a.rmode && oz_row_output(a.rmode, a.rper, a.r0, a.rpts, j + b) != 0) ? __builtin_ldexp(1.0, a.rtg) : 1.0;
        cbad[b] = live && a.bad_col && a.bad_col[j + b];
    }
    const double nan = __builtin_nan("");
    for (int sub = 0; sub < 4; ++sub) {
        const int i0 = blockIdx.y * 128 + wave * 32 + sub * 8;
        if (i0 >= a.nvalid) break;                           // wave-uniform
#pragma unroll
        for (int ii = 0; ii < 8; ++ii) {
            const int i = i0 + ii;
            double v[4] = {0.0, 0.0, 0.0, 0.0};
            if (live && i < a.nvalid) {
                const int8_t* u = a.U + oz_u_block(i >> 8, (int)blockIdx.x, 0, a.Ti, n) + (i & 255) * OZ_T + 4 * lane;
                double c1[4] = {0.0, 0.0, 0.0, 0.0}, c2[4] = {0.0, 0.0, 0.0, 0.0};
                for (int l = 0; l < n; ++l) {
                    const int w = *reinterpret_cast<const int*>(u + (int64_t)l * (OZ_T * OZ_T));
                    const double s1 = a.pl.s1[l], s2 = a.pl.s2[l];
#pragma unroll
                    for (int b = 0; b < 4; ++b) {
                        const double ud = (double)((w << (24 - 8 * b)) >> 24);
                        c1[b] = __builtin_fma(ud, s1, c1[b]);
                        c2[b] = __builtin_fma(ud, s2, c2[b]);
                    }
                }
                const double sc = __builtin_ldexp(1.0, -(a.sexp[i] + a.sK));
                const bool rbad = a.bad_row && a.bad_row[i];
#pragma unroll
                for (int b = 0; b < 4; ++b) {
                    const double Q = __builtin_rint((c1[b] + c2[b]) * a.pl.invP);
                    const double cp = __builtin_fma(-Q, a.pl.P1, c1[b]) + __builtin_fma(-Q, a.pl.P2, c2[b]);
                    v[b] = (rbad || cbad[b]) ? nan : cp * sc * cf[b];
                }
            }
#pragma unroll
            for (int b = 0; b < 4; ++b) tile[wave][4 * lane + b][ii] = v[b];
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_s_waitcnt(0xc07f);                  // the wave's own LDS writes have landed (in-order queue)
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        const int nrow = a.nvalid - i0 < 8 ? a.nvalid - i0 : 8;
        for (int e = lane; e < 256 * 8; e += 64) {           // 8 consecutive lanes = 64 contiguous bytes of one candidate's row of V
            const int c = e >> 3, ii = e & 7;
            const int jc = blockIdx.x * 256 + c;
            if (jc < a.Mc && ii < nrow) a.V[(int64_t)jc * a.ldv + i0 + ii] = tile[wave][c][ii];
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_s_waitcnt(0xc07f);
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    }
}

// ---- launchers -----------------------------------------------------------------------------------------------------------------------
size_t oz_w_bytes(int n, int Np) { const int64_t q = pad_up(Np, OZ_T); return (size_t)n * q * q; }
size_t oz_k_bytes(int n, int Np, int Mc) { return (size_t)n * pad_up(Mc, OZ_T) * pad_up(Np, OZ_T); }

// Gradient-enhanced models (kper = outputs per point, rows point-major: k % kper = 0 is the function value, the rest its partial
// derivatives): the derivative rows of K_XZ are larger than the function rows by about √c/ℓ, so the product is taken as
// (W·D⁻¹)(D·K_XZ) with D = diag(1 on function rows, 2^−ktg on derivative rows), 2^ktg ≈ √c/ℓ — exact scalings that make every entry
// of D·K_XZ at most ≈ σ_f²√2 (Cauchy–Schwarz on the prior covariance), so that one fixed-point scale serves the whole chunk.
hipError_t oz_prepare_w(const OzPlan& pl, const double* W, int64_t ldw, int Np, int nvalid, int8_t* WR, int* sexp, int* bad_row, hipStream_t s,
                        int kper, int ktg) {
    const int Np256 = (int)pad_up(Np, OZ_T);
    hipError_t e = hipMemsetAsync(bad_row, 0, sizeof(int) * Np256, s);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(oz_rowscale_kernel, dim3((Np256 + 3) / 4), dim3(256), 0, s, W, ldw, nvalid, Np256, pl.eP, sexp, kper, ktg);
    OzQuantArgs q{};
    q.in = W; q.ldin = ldw; q.rows_in = nvalid; q.cols_in = Np; q.rows_out = Np256; q.cols_out = Np256; q.lower = 1;
    q.srow = sexp; q.sconst = 0; q.kper = kper; q.ktg = ktg; q.rmode = 0; q.rper = 1; q.rtg = 0; q.r0 = 0; q.rpts = 1; q.out = WR; q.ld = Np256; q.plane = (int64_t)Np256 * Np256; q.bad = bad_row; q.pl = pl;
    const int64_t threads = (int64_t)Np256 * (Np256 / 16);
    hipLaunchKernelGGL(oz_quant_kernel, dim3((unsigned)((threads + 255) / 256)), dim3(256), 0, s, q);
    return hipGetLastError();
}

// sK: K' = rint(K·2^sK) < 2^53 for K ≤ kmax (kmax = σ_f²: the stationary kernels of this library peak at distance 0)
int oz_k_scale(double kmax) {
    int e;
    (void)std::frexp(kmax * (1.0 + 1e-12), &e);        // kmax < 2^e
    return 52 - e;                                      // K' < 2^52·(1+…) < 2^53
}

hipError_t launch_var_ozaki(const OzVarArgs& v, hipStream_t s) {
    const OzPlan& pl = *v.plan;
    const int Np256 = (int)pad_up(v.Np, OZ_T), Mc256 = (int)pad_up(v.Mc, OZ_T);
    hipError_t e = hipSuccess;
    if (!v.planes_ready && (e = hipMemsetAsync(v.bad_col, 0, sizeof(int) * Mc256, s)) != hipSuccess) return e;
    if (!v.planes_ready) {       // the generator did not write the residue planes itself (d > 32, or a caller-made K_XZ)
        OzQuantArgs q{};
        q.in = v.Kxz; q.ldin = v.ldk; q.rows_in = v.Mc; q.cols_in = v.Np; q.rows_out = Mc256; q.cols_out = Np256; q.lower = 0;
        q.srow = nullptr; q.sconst = v.sK; q.kper = v.kper; q.ktg = -v.ktg; q.rmode = v.rmode; q.rper = v.rper; q.rtg = -v.ktg; q.r0 = v.r0; q.rpts = v.rpts; q.out = v.KR; q.ld = Np256; q.plane = (int64_t)Mc256 * Np256; q.bad = v.bad_col; q.pl = pl;
        const int64_t threads = (int64_t)Mc256 * (Np256 / 16);
        hipLaunchKernelGGL(oz_quant_kernel, dim3((unsigned)((threads + 255) / 256)), dim3(256), 0, s, q);
    }
    if (v.ev_quant && (e = hipEventRecord(v.ev_quant, s)) != hipSuccess) return e;

    OzGemmArgs g{};
    g.KR = v.KR; g.WR = v.WR; g.U = v.U;
    g.nhs = Np256 / 64;
    g.sK = (int64_t)Mc256 * Np256; g.sW = (int64_t)Np256 * Np256;
    g.Ti = Np256 / OZ_T; g.Tj = Mc256 / OZ_T; g.n = pl.n;
    g.tjg = g.Tj >= 64 ? 64 : (int)pad_up(g.Tj, 8);
    for (int l = 0; l < pl.n; ++l) {
        g.invp[l] = pl.invp[l]; g.p[l] = pl.p[l];
    }
    const int ngj = (g.Tj + g.tjg - 1) / g.tjg, ngi = (g.Ti + 3) / 4;
    const unsigned blocks = (unsigned)(ngi * pl.n * ngj * 4 * g.tjg);
    {
        // persistent: one workgroup per CU; the tile counters (one per XCD list) sit behind the chunk's bad_col flags
        static const int cus = [] { int d = 0; hipDeviceProp_t pr; return (hipGetDevice(&d) == hipSuccess && hipGetDeviceProperties(&pr, d) == hipSuccess && pr.multiProcessorCount > 0) ? pr.multiProcessorCount : 256; }();
        int* ctr = v.bad_col + Mc256;
        if (!(v.ctr_clean && *v.ctr_clean == ctr) && (e = hipMemsetAsync(ctr, 0, sizeof(int) * 8, s)) != hipSuccess) return e;
        if (v.ctr_clean) *v.ctr_clean = nullptr;
        const int G = (unsigned)cus < blocks ? cus : (int)blocks;
        hipLaunchKernelGGL(oz_gemm16p_kernel, dim3(G), dim3(512), 0, s, g, (int)blocks, ctr);
    }
    if (v.ev_gemm && (e = hipEventRecord(v.ev_gemm, s)) != hipSuccess) return e;

    if (v.Vout) {
        OzCrtVArgs c{};
        c.U = v.U; c.Ti = g.Ti; c.sexp = v.sexp; c.sK = v.sK; c.bad_row = v.bad_row; c.bad_col = v.bad_col;
        c.V = v.Vout; c.ldv = v.ldv; c.Mc = v.Mc; c.nvalid = v.nvalid; c.pl = pl;
        c.rmode = v.rmode; c.rper = v.rper; c.rtg = v.ktg; c.r0 = v.r0; c.rpts = v.rpts;
        c.ctr_reset = v.ctr_clean ? v.bad_col + Mc256 : nullptr;
        hipLaunchKernelGGL(oz_crt_v_kernel, dim3((v.Mc + 255) / 256, v.Np / 128), dim3(256), 0, s, c);
        e = hipGetLastError();
        if (e == hipSuccess && v.ctr_clean) *v.ctr_clean = v.bad_col + Mc256;
        return e;
    }
    OzCrtArgs c{};
    c.U = v.U; c.Ti = g.Ti; c.sexp = v.sexp; c.sK = v.sK; c.bad_row = v.bad_row; c.bad_col = v.bad_col;
    c.partial = v.partial; c.ldp = v.ldp; c.Mc = v.Mc; c.nvalid = v.nvalid; c.pl = pl;
    c.rmode = v.rmode; c.rper = v.rper; c.rtg = v.ktg; c.r0 = v.r0; c.rpts = v.rpts;
    c.ctr_reset = v.ctr_clean ? v.bad_col + Mc256 : nullptr;
    if (pl.n == 14) hipLaunchKernelGGL(oz_crt_kernel<14>, dim3((v.Mc + 255) / 256, v.Np / 128), dim3(256), 0, s, c);
    else hipLaunchKernelGGL(oz_crt_kernel<0>, dim3((v.Mc + 255) / 256, v.Np / 128), dim3(256), 0, s, c);
    e = hipGetLastError();
    if (e == hipSuccess && v.ctr_clean) *v.ctr_clean = v.bad_col + Mc256;
    return e;
}

}  // namespace abo
