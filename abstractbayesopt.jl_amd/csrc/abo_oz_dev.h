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
constexpr double oz_mod_c26(int l) {      // 2^26 mod p, symmetric
    const int p = oz_mod_p(l);
    int c = (int)((1u << 26) % (unsigned)p);
    if (2 * c > p) c -= p;
    return (double)c;
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

// The same residue through four signed base-2^14 limbs in fp32 (the generator's fused output; measured 6 % faster than the fp64
// split there, same bits): x = a3·2^42 + a2·2^28 + a1·2^14 + a0 with |a0|, |a1|, |a2| ≤ 2^13, |a3| ≤ 2^11; t = Σ a_i·(2^(14i) mod p) is an integer
// below 2^22 — exact in fp32 at every partial sum —, t/p is at least 1/(2p) ≥ 1.9e-3 away from a half-integer for odd p while the
// fp32 quotient errs by at most |t|/p·2^-23 ≤ 1.5e-3, so rndne gives the exact nearest quotient and r = t − q·p the symmetric residue.
struct OzLimbs { float a0, a1, a2, a3; };

__device__ __forceinline__ OzLimbs oz_limbs(double v, double sc) {
    const double x = __builtin_rint(v * sc);
    const double a3 = __builtin_rint(x * 0x1p-42);
    const double r1 = __builtin_fma(-a3, 0x1p42, x);
    const double a2 = __builtin_rint(r1 * 0x1p-28);
    const double r2 = __builtin_fma(-a2, 0x1p28, r1);
    const double a1 = __builtin_rint(r2 * 0x1p-14);
    const double a0 = __builtin_fma(-a1, 0x1p14, r2);
    return OzLimbs{(float)a0, (float)a1, (float)a2, (float)a3};
}

constexpr float oz_mod_c14(int l, int i) {      // 2^(14 i) mod p, symmetric
    const long long p = oz_mod_p(l);
    long long c = (1ll << (14 * i)) % p;
    if (2 * c > p) c -= p;
    return (float)c;
}

__device__ __forceinline__ int sym_residue_f32(const OzLimbs& x, float c1, float c2, float c3, float invp, float pf) {
    const float t = __builtin_fmaf(x.a3, c3, __builtin_fmaf(x.a2, c2, __builtin_fmaf(x.a1, c1, x.a0)));
    const float q = __builtin_rintf(t * invp);
    return (int)__builtin_fmaf(-q, pf, t);
}

}  // namespace abo
