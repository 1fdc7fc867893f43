"""Kernel / mean specifications mirroring the KernelFunctions.jl objects the reference's users
hand to ``StandardGP`` (SqExponentialKernel(), Matern52Kernel(), ``scale * with_lengthscale(k, ℓ)``,
ZeroMean(), ConstMean(c)).  They only *describe* the kernel — every evaluation happens in the HIP
library, which takes the normal form ``sigma_f2 · kappa(||x−z|| / ell)`` the reference's
constructor reduces any kernel to (src/surrogates/StandardGP.jl:41-64,
src/surrogates/surrogates_utils.jl:28-47)."""
from dataclasses import dataclass, replace

SE, MATERN52, MATERN72, MATERN32 = 0, 1, 2, 3
_NAMES = {SE: "Squared Exponential Kernel", MATERN52: "Matern 5/2 Kernel", MATERN72: "Matern 7/2 Kernel",
          MATERN32: "Matern 3/2 Kernel"}


@dataclass(frozen=True)
class Kernel:
    family: int
    lengthscale: object = None   # None = no ScaleTransform attached (defaults to 1.0)
    scale: object = None         # None = not a ScaledKernel (defaults to 1.0)

    def __rmul__(self, s):       # `scale * kernel`  → ScaledKernel(kernel, scale)
        return replace(self, scale=float(s) * (1.0 if self.scale is None else self.scale))

    __mul__ = __rmul__

    def __repr__(self):
        return f"{_NAMES[self.family]} (ℓ = {self.lengthscale or 1.0}, σ² = {self.scale or 1.0})"


def SqExponentialKernel():
    return Kernel(SE)


SEKernel = RBFKernel = GaussianKernel = SqExponentialKernel


def Matern52Kernel():
    return Kernel(MATERN52)


def ApproxMatern52Kernel():
    """src/surrogates/GradientGP.jl:52-101 — same kappa as Matern52Kernel to 1e-12
    (test/test_kernels.jl:42-56), so it maps to the same device family."""
    return Kernel(MATERN52)


def ApproxMatern72Kernel():
    """src/surrogates/GradientGP.jl:278-327."""
    return Kernel(MATERN72)


def Matern32Kernel():
    return Kernel(MATERN32)


def with_lengthscale(k: Kernel, ell: float) -> Kernel:
    """KernelFunctions.with_lengthscale(k, ℓ) = k ∘ ScaleTransform(1/ℓ)."""
    return replace(k, lengthscale=float(ell))


def ScaledKernel(k: Kernel, scale: float) -> Kernel:
    return replace(k, scale=float(scale))


def extract_scale_and_lengthscale(k: Kernel):
    """src/surrogates/surrogates_utils.jl:28-47 → (inner, scale, lengthscale-or-None)."""
    return Kernel(k.family), (1.0 if k.scale is None else k.scale), k.lengthscale


@dataclass(frozen=True)
class ZeroMean:
    c: float = 0.0


@dataclass(frozen=True)
class ConstMean:
    c: float = 0.0
