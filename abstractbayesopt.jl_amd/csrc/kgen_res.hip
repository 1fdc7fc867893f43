// The kernel-matrix generator with the int8-residue engine's planes written in the same pass (kgen_kernel<FAM, DP, 14>: the default
// 14-modulus plan, unrolled over the compile-time tables of abo_oz_dev.h).  Core and design notes: kgen_core.h, kgen.hip.
#include "kgen_core.h"

namespace abo {

hipError_t launch_kgen_res14(const KgenArgs& a, hipStream_t s) {
    switch (a.family) {
        case ABO_KERNEL_SE: return launch_kgen_dp<ABO_KERNEL_SE, 14>(a, s);
        case ABO_KERNEL_MATERN52: return launch_kgen_dp<ABO_KERNEL_MATERN52, 14>(a, s);
        case ABO_KERNEL_MATERN72: return launch_kgen_dp<ABO_KERNEL_MATERN72, 14>(a, s);
        case ABO_KERNEL_MATERN32: return launch_kgen_dp<ABO_KERNEL_MATERN32, 14>(a, s);
        default: return hipErrorInvalidValue;
    }
}

}  // namespace abo
