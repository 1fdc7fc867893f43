// Device math of the kernel families: k(x, z) = sigma_f2 · kappa(‖x − z‖² / ell²), evaluated with the library's own exp and
// sqrt sequences (no libm calls).  Shared by the kernel-matrix generator (kgen.hip) and the fused small-N fit (chol.hip) so
// that both produce the same bits for the same pair of points.
// Reference: [upstream KernelFunctions] SqExponentialKernel / Matern52Kernel / Matern32Kernel; ApproxMatern52Kernel and
// ApproxMatern72Kernel, src/surrogates/GradientGP.jl:94-101, :320-327.
#pragma once
#include <hip/hip_runtime.h>

#include "../../include/abo_hip.h"

namespace abo {

// exp(x) for x ≤ 0 (all the kernel families need): Cody–Waite reduction x = n·ln2 + r, |r| ≤ ln2/2,
// degree-12 Taylor polynomial (remainder 1.7e-16 relative), ldexp.  18 fp64 instructions with no
// compare/select chain; underflow falls out of v_ldexp_f64 (→ 0 below 2⁻¹⁰⁷⁴).  Measured against
// mpmath in tests/test_gpu_parity.py::test_kappa_device_math (≤ 2 ulp).
__device__ __forceinline__ double exp_nonpos(double x) {
    const double n = rint(x * 1.4426950408889634074);
    double r = fma(n, -6.93147180369123816490e-01, x);     // ln2 hi (32 trailing zero bits)
    r = fma(n, -1.90821492927058770002e-10, r);            // ln2 lo
    double p = 2.08767569878680989792e-09;                 // 1/12!
    p = fma(p, r, 2.50521083854417187751e-08);
    p = fma(p, r, 2.75573192239858906526e-07);
    p = fma(p, r, 2.75573192239858906526e-06);
    p = fma(p, r, 2.48015873015873015873e-05);
    p = fma(p, r, 1.98412698412698412698e-04);
    p = fma(p, r, 1.38888888888888888889e-03);
    p = fma(p, r, 8.33333333333333333333e-03);
    p = fma(p, r, 4.16666666666666666667e-02);
    p = fma(p, r, 1.66666666666666666667e-01);
    p = fma(p, r, 0.5);
    p = fma(p, r, 1.0);
    p = fma(p, r, 1.0);
    return ldexp(p, (int)n);
}

// sqrt(x) for finite x ≥ 0: v_rsq_f64 seed + two coupled Newton steps (Goldschmidt form) + one residual
// correction; x below 1e-290 (kernel value indistinguishable from κ(0)) returns 0 instead of 0·inf.
__device__ __forceinline__ double sqrt_pos(double x) {
    const double y = __builtin_amdgcn_rsq(x);
    double g = x * y, h = 0.5 * y;
    double r = fma(-h, g, 0.5);
    g = fma(g, r, g);
    h = fma(h, r, h);
    r = fma(-h, g, 0.5);
    g = fma(g, r, g);
    h = fma(h, r, h);
    const double e = fma(-g, g, x);
    g = fma(e, h, g);
    return x > 1e-290 ? g : 0.0;
}

template <int FAM>
__device__ __forceinline__ double kappa_eval(double d2) {
    if constexpr (FAM == ABO_KERNEL_SE) {
        return exp_nonpos(-0.5 * d2);
    } else if constexpr (FAM == ABO_KERNEL_MATERN52) {
        // (1 + √5 d + 5 d²/3) e^{−√5 d}; closed form also covers the reference's Taylor branch
        // (src/surrogates/GradientGP.jl:94-101) to 1e-16.  5 d²/3 is a multiplication by the rounded
        // constant 5/3: ≤1 ulp from the reference's division (a 14-instruction fp64 divide otherwise)
        const double s5 = 2.23606797749978969640917366873128;
        const double d = sqrt_pos(d2);
        return fma(d2, 5.0 / 3.0, fma(s5, d, 1.0)) * exp_nonpos(-s5 * d);
    } else if constexpr (FAM == ABO_KERNEL_MATERN72) {
        // src/surrogates/GradientGP.jl:320-327
        const double s7 = 2.64575131106459059050161575363926;
        const double d = sqrt_pos(d2);
        return fma(d2 * d, 7.0 * s7 / 15.0, fma(d2, 14.0 / 5.0, fma(s7, d, 1.0))) * exp_nonpos(-s7 * d);
    } else {
        const double s3 = 1.73205080756887729352744634150587;
        const double d = sqrt_pos(d2);
        return fma(s3, d, 1.0) * exp_nonpos(-s3 * d);
    }
}

}  // namespace abo
