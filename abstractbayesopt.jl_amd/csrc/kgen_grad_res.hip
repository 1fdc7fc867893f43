// The gradient-enhanced generator with the int8-residue engine's planes written in the same pass (kgen_grad_kernel<FAM, DP, false, 14>).
// Core and design notes: kgen_grad_core.h, kgen.hip.
#include "kgen_grad_core.h"

namespace abo {

hipError_t launch_kgen_grad_res14(const KgenArgs& a, hipStream_t s) {
    switch (a.family) {
        case ABO_KERNEL_SE: return launch_grad_kgen_dp<ABO_KERNEL_SE, false, 14>(a, s);
        case ABO_KERNEL_MATERN52: return launch_grad_kgen_dp<ABO_KERNEL_MATERN52, false, 14>(a, s);
        case ABO_KERNEL_MATERN72: return launch_grad_kgen_dp<ABO_KERNEL_MATERN72, false, 14>(a, s);
        default: return hipErrorInvalidValue;
    }
}

}  // namespace abo
