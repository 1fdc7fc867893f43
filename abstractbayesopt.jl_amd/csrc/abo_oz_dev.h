// device helpers shared by the quantisers of the int8-residue engine (ozaki.hip, and kgen.hip's fused residue output)
#pragma once
#include <hip/hip_runtime.h>

namespace abo {

// the moduli (the first n of them for an n-modulus plan: pairwise coprime, descending from 256) and the per-modulus constants
// of the quantiser as compile-time tables — the generator's fused residue output unrolls over them (oz_make_plan checks that its
// own search yields the same sequence)
constexpr int oz_mod_p(int l) {
    constexpr int P[16] = {256, 255, 253, 251, 247, 241, 239, 233, 229, 227, 223, 217, 211, 199, 197, 193};
    return P[l];
}

// Layout of a residue plane ([rows][cols] bytes, rows a multiple of 256, cols a multiple of 64): blocks of 256 rows × 64 k-bytes,
// row-major inside a block, block (R, H) at ((R·nhs) + H)·16384 with nhs = cols / 64.  A DMA piece of the GEMM (16 rows × 64 bytes of
// one half-stage) is then 1 KiB of contiguous memory — eight whole cache lines instead of sixteen half lines.
__host__ __device__ inline long long oz_plane_off(int row, int k, int nhs) {
    return ((long long)(row >> 8) * nhs + (k >> 6)) * 16384 + (row & 255) * 64 + (k & 63);
}

// symmetric residue of an integer-valued double |x| < 2^53 modulo p, from the split x = xh·2^26 + xl (|xh| ≤ 2^27, |xl| ≤ 2^25):
// t = xh·(2^26 mod p) + xl is exact and below 2^35, so rint(t/p) is the exact nearest quotient (t/p is at least 1/(2p) away from
// a half-integer for odd p, the fp64 product errs by < 1e-7) and r = t − q·p lies in [−(p−1)/2, (p−1)/2]; for p = 256 the low
// byte of any representative is the residue.  Five fp64 operations per modulus, no range fix-ups.
__device__ __forceinline__ int sym_residue(double xh, double xl, double c26, double invp, double pd) {
    const double t = __builtin_fma(xh, c26, xl);
    const double q = __builtin_rint(t * invp);
    return (int)__builtin_fma(-q, pd, t);
}

// x = rint(v·sc) split into xh·2^26 + xl
__device__ __forceinline__ void oz_split(double v, double sc, double& xh, double& xl) {
    const double x = __builtin_rint(v * sc);
    xh = __builtin_rint(x * 0x1p-26);
    xl = __builtin_fma(-xh, 0x1p26, x);
}

// ---- the generator's fused residue output, round 6: two-stage reduction ----------------------------------------------------------------
// Rounds 3 – 5 reduced four signed base-2^14 limbs in fp32: 9 VALU operations per value and modulus (three fma for t, multiply, rndne, fma,
// convert, mask, shift-or) plus the limb split.  Here the odd moduli are taken three at a time:
//   stage 1 (fp64, per group g of moduli p_a, p_b, p_c with M = p_a·p_b·p_c < 2^24):  q = rint(x/M), r = x − q·M  (exact under fma; any
//           representative |r| ≤ M/2·(1 + 2^-22) < 2^23 will do: it is an exact fp32 integer rf and an exact int32 ri)
//   stage 2 (per modulus, TWO operations):  t = fma(rf, 1/p, 1.5·2^23) — ONE rounding, to the integer grid of [2^23, 2^24): the low 24 bits of t's
//           image are 2^22 + q with q = RN(rf·fl(1/p)), |q − rf/p| ≤ 1/2 + |rf|/p·2^-24;  D = v_mad_i32_i24(t's image, −p, ri) multiplies the LOW 24
//           BITS: D = ri − q·p − 2^22·p.  2^22·p is a multiple of 256, so D's low byte is the low byte of r' = ri − q·p, and
//           |r'| ≤ p/2 + |rf|·2^-24 ≤ p/2 + M·2^-25 < 128 for every group (largest: 255·253·251 = 16 193 265 → 127.5 + 0.483): r' ≡ x (mod p) lies in
//           [−127, 127], a signed byte.  r' is the NEAREST residue except when rf/p is within 2^-24·|rf|/p of a half-integer, where it may be the
//           representative on the other side (|r'| = (p + 1)/2): the residue GEMM's sums are exact integers reduced mod p, so every
//           representative gives the same U — the engine's results do not change by a bit (the soak hashes of tools/oz_soak.py hold).
//   p = 256: the low byte of x's low 24-bit limb.
// Per pair of values and modulus: 2 fma + 2 mad + 1 v_perm (both bytes into a short) instead of 18 operations; per pair and group 10.
constexpr int oz_group_first(int g) { return 1 + 3 * g; }                         // moduli l = 1 + 3g … (three per group, the last group what is left)
constexpr int oz_group_count(int n) { return (n - 1 + 2) / 3; }
constexpr int oz_group_size(int n, int g) { return (n - oz_group_first(g)) < 3 ? (n - oz_group_first(g)) : 3; }
constexpr double oz_group_M(int n, int g) {
    double m = 1.0;
    for (int i = 0; i < oz_group_size(n, g); ++i) m *= (double)oz_mod_p(oz_group_first(g) + i);
    return m;
}

// x0, x1: integer-valued doubles, |x| < 2^53.  emit(l, s): s = byte(x0 mod p_l) | byte(x1 mod p_l) << 8 for l = 0 … RES−1
template <int RES, typename Emit>
__device__ __forceinline__ void oz_residue_pair(double x0, double x1, Emit&& emit) {
    static_assert(RES >= 2 && RES <= 16, "moduli count");
    static_assert(oz_group_M(RES, 0) < 16777216.0, "a group's product must stay below 2^24");
    {   // p = 256
        const double h0 = __builtin_rint(x0 * 0x1p-24), h1 = __builtin_rint(x1 * 0x1p-24);
        const int i0 = (int)__builtin_fma(-h0, 0x1p24, x0), i1 = (int)__builtin_fma(-h1, 0x1p24, x1);
        emit(0, __builtin_amdgcn_perm((unsigned)i1, (unsigned)i0, 0x0c0c0400u));
    }
#pragma unroll
    for (int g = 0; g < oz_group_count(RES); ++g) {
        const double M = oz_group_M(RES, g), invM = 1.0 / oz_group_M(RES, g);
        const double q0 = __builtin_rint(x0 * invM), q1 = __builtin_rint(x1 * invM);
        const double r0 = __builtin_fma(-q0, M, x0), r1 = __builtin_fma(-q1, M, x1);
        if (oz_group_size(RES, g) == 1) {                       // a group of one: r is the residue
            emit(oz_group_first(g), __builtin_amdgcn_perm((unsigned)(int)r1, (unsigned)(int)r0, 0x0c0c0400u));
            continue;
        }
        const float f0 = (float)r0, f1 = (float)r1;
        const int i0 = (int)f0, i1 = (int)f1;
#pragma unroll
        for (int i = 0; i < oz_group_size(RES, g); ++i) {
            const int l = oz_group_first(g) + i;
            const float invp = 1.0f / (float)oz_mod_p(l);
            const float t0 = __builtin_fmaf(f0, invp, 12582912.0f), t1 = __builtin_fmaf(f1, invp, 12582912.0f);
            int d0, d1;
            // (as asm: from C the compiler picks the quarter-rate v_mul_lo_u32)
            asm("v_mad_i32_i24 %0, %1, %2, %3" : "=v"(d0) : "v"(__builtin_bit_cast(int, t0)), "s"(-oz_mod_p(l)), "v"(i0));
            asm("v_mad_i32_i24 %0, %1, %2, %3" : "=v"(d1) : "v"(__builtin_bit_cast(int, t1)), "s"(-oz_mod_p(l)), "v"(i1));
            emit(l, __builtin_amdgcn_perm((unsigned)d1, (unsigned)d0, 0x0c0c0400u));
        }
    }
}

}  // namespace abo
