"""abstractbayesopt.jl_amd — MI355X-native GP surrogate backend behind AbstractBayesOpt.jl's
AbstractSurrogate / AbstractAcquisition interface (hot path only: update → posterior → EI/UCB →
top-k).  Import as ``import abstractbayesopt.jl_amd as abo``."""
from . import _lib, acquisition, distributed, incremental, multigpu, synth
from ._lib import AboError, DimensionMismatch, PosDefException
from .acquisition import (AbstractAcquisition, EnsembleAcquisition, ExpectedImprovement, ProbabilityImprovement,
                          UpperConfidenceBound, device_latin_hypercube, evaluate, latin_hypercube,
                          optimize_acquisition, optimize_acquisition_device, acquisition_value_and_grad, refine_starts,
                          update_and_evaluate)
from .domains import ContinuousDomain
from . import gradient_gp as _g
from .gradient_gp import (GradientNormUCB, HipGradientGP, gradConstMean, posterior_grad_cov, posterior_grad_mean,
                          posterior_grad_var)
from .hyperparams import lengthscale_bounds, monte_carlo_fill_distance, nlml_and_grad, optimize_hyperparameters
from .incremental import ResidentCandidates, append, greedy_qei
from .multigpu import HipShardedGP, HipShardedGradientGP, ShardedCandidates
from .kernels import (ApproxMatern52Kernel, ApproxMatern72Kernel, ConstMean, Kernel, Matern32Kernel, Matern52Kernel,
                      ScaledKernel, SqExponentialKernel, ZeroMean, with_lengthscale)
from . import surrogate as _s
from .surrogate import (AbstractSurrogate, HipStandardGP, _get_minimum, _update_model_parameters,
                        get_factor, get_kernel_constructor, get_lengthscale, get_mean_std, get_scale, mean_and_var,
                        nlml, nlml_fitted, nlml_ls, posterior_mean, posterior_var, prep_input, prep_output,
                        rescale_model, set_default_contraction, std_y, training_data, unstandardized_mean_and_var)

StandardGP = HipStandardGP   # drop-in aliases
GradientGP = HipGradientGP


def update(obj, a, b):
    """`update` is one generic function in the reference, dispatched on its first argument:
    update(model, xs, ys) (StandardGP.jl:79) / update(acq, ys, surrogate) (ExpectedImprovement.jl:81)."""
    if isinstance(obj, GradientNormUCB):
        return obj                                              # gradNormUCB.jl:66-68
    if isinstance(obj, AbstractAcquisition):
        return acquisition.update(obj, a, b)
    if isinstance(obj, HipShardedGP):
        return multigpu.update(obj, a, b)
    if isinstance(obj, HipGradientGP):
        return _g.update(obj, a, b)
    return _s.update(obj, a, b)


def copy(obj):
    """Base.copy for surrogates (StandardGP.jl:26) and acquisition functions."""
    if isinstance(obj, GradientNormUCB):
        return GradientNormUCB(obj.beta)                        # gradNormUCB.jl:24
    if isinstance(obj, AbstractAcquisition):
        return acquisition.copy(obj)
    if isinstance(obj, HipShardedGP):
        return multigpu.copy(obj)
    return _s.copy(obj)


def _dispatch(name):
    """reference methods that exist for both surrogate types dispatch on the model's type"""
    std, grd = getattr(_s, name), getattr(_g, name)

    def f(model, *args, **kw):
        return (grd if hasattr(model, "p") else std)(model, *args, **kw)      # HipGradientGP and HipShardedGradientGP

    f.__name__ = name
    f.__doc__ = std.__doc__
    return f


for _n in ("get_mean_std", "std_y", "rescale_model", "_update_model_parameters", "_get_minimum",
           "unstandardized_mean_and_var", "prep_output", "nlml", "nlml_ls"):
    globals()[_n] = _dispatch(_n)
