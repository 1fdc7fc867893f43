# HipGradientGP.jl — binding of libabo_hip.so (include/abo_hip.h, ABI version 5; checked by HipStandardGP.jl's _ensure_abi) for the gradient-enhanced surrogate of
# AbstractBayesOpt.jl (`GradientGP`, src/surrogates/GradientGP.jl).  `include` it after HipStandardGP.jl (it uses that file's
# AboParams, AboHandle, _check, _pack, _family, LIBABO) and export HipGradientGP.
#
#   reference method (GradientGP.jl line)                         method below                C-ABI entry
#   GradientGP(kernel, p, noise_var; mean) :622                    HipGradientGP(...)          —
#   Base.copy :32                                                  Base.copy                   abo_retain
#   update(model, xs, ys) :659  (refit of the (d+1)N-row system)   update                      abo_create_grad + abo_fit
#   posterior_mean / posterior_var :985, :1001 (function output)   same names                  abo_predict
#   posterior_grad_mean / _var :936, :953 (all p outputs)          same names                  abo_predict_grad
#   posterior_grad_cov :969 (one point: p×p)                       same name                   abo_predict_grad_cov
#   GradientNormUCB functor, gradNormUCB.jl:43-51                  functor method              abo_predict_grad_cov (score)
#   optimize_acquisition(acqf, model, domain), acq_utils.jl:33-73  optimize_acquisition        abo_optimize_acquisition_terms
#   nlml / nlml_ls :684, :719                                      same names                  abo_nlml (+ abo_nlml_grad)
#   get_mean_std / std_y / rescale_model :753, :785, :802          same names                  host only
#   _update_model_parameters :834, get_lengthscale / get_scale / get_kernel_constructor :849-875,
#   prep_input :889, prep_output :919, _get_minimum :1044          same names                  host only
#   unstandardized_mean_and_var :1019                              same name                   abo_predict_grad
#   (none: the reference refits per step)                          append                      abo_append_grad
#
# Layout at the ABI: xs point-major d×N; ys BY OUTPUTS (all f values, then all ∂₁f, … — exactly prep_output's vector);
# predictions of all outputs come back by outputs too, which is the order the reference's posterior_grad_* return.
# Inside the library the factor is point-major (an observation appends its p rows at the end); nothing of that shows here.
# NOTE: not executed in the build image (no Julia toolchain); the same calls run from Python (gradient_gp.py) in the GPU suite.

struct HipGradientGP{T,G<:AbstractGPs.GP} <: AbstractSurrogate
    gp::G                                # prior with the gradKernel-wrapped normal-form kernel, as GradientGP.jl:17-22
    noise_var::T
    p::Int
    gpx::Union{Nothing,AboHandle}
    device::Int32; jitter::Float64; n_max::Int64
    devices::Vector{Int32}               # more than one entry: abo_mgpu_create_grad (the model replicated, candidates sharded)
end

function HipGradientGP(kernel::Kernel, p::Int, noise_var; mean=gradConstMean(zeros(p)), device=0, devices=[device], jitter=0.0, n_max=0)
    s = GradientGP(kernel, p, noise_var; mean=mean)           # reuse the normal-form + gradKernel logic (:622-643)
    HipGradientGP(s.gp, noise_var, p, nothing, Int32(devices[1]), Float64(jitter), Int64(n_max), Int32.(devices))
end
HipGradientGP(gp, noise_var, p, gpx, device, jitter, n_max) = HipGradientGP(gp, noise_var, p, gpx, device, jitter, n_max, Int32[device])
_with(m::HipGradientGP, gpx) = HipGradientGP(m.gp, m.noise_var, m.p, gpx, m.device, m.jitter, m.n_max, m.devices)
_multi(m::HipGradientGP) = length(m.devices) > 1

get_lengthscale(m::HipGradientGP) = 1 ./ m.gp.kernel.base_kernel.kernel.transform.s
get_scale(m::HipGradientGP) = m.gp.kernel.base_kernel.σ²
get_kernel_constructor(m::HipGradientGP) = m.gp.kernel.base_kernel.kernel.kernel
_mean_vec(m::HipGradientGP) = m.gp.mean isa gradConstMean ? collect(Float64, m.gp.mean.c) : zeros(m.p)
prep_input(m::HipGradientGP, xs) = xs                          # the library addresses outputs itself (no (x, output) tuples)
prep_output(::HipGradientGP, y::Vector) = vec(permutedims(reduce(hcat, y)))      # by outputs, :919
_get_minimum(::HipGradientGP, ys::Vector) = minimum(y[1] for y in ys)            # function values only, :1044
_update_model_parameters(m::HipGradientGP, k::Kernel) =
    HipGradientGP(k, m.p, m.noise_var; mean=m.gp.mean, devices=m.devices, jitter=m.jitter, n_max=m.n_max)

# standardisation helpers: host arithmetic only — forwarded to the reference's own methods (GradientGP.jl:753-820) on a prior-only
# GradientGP holding the same gp / noise / p
_ref(m::HipGradientGP) = GradientGP(m.gp, m.noise_var, m.p, nothing)
get_mean_std(m::HipGradientGP, y_train::AbstractVector, choice::String) = get_mean_std(_ref(m), y_train, choice)
std_y(m::HipGradientGP, ys::AbstractVector, μ::AbstractVector, σ::AbstractVector) = std_y(_ref(m), ys, μ, σ)
function rescale_model(m::HipGradientGP, σ::AbstractVector)
    r = rescale_model(_ref(m), σ)
    HipGradientGP(r.gp, r.noise_var, m.p, nothing, m.device, m.jitter, m.n_max, m.devices)
end

function Base.copy(m::HipGradientGP)                                              # :32
    m.gpx === nothing && return m
    if m.gpx.multi
        h = Ref{Ptr{Cvoid}}()
        _check(@ccall LIBABO.abo_mgpu_clone(m.gpx.ptr::Ptr{Cvoid}, h::Ptr{Ptr{Cvoid}})::Int32)
        return _with(m, AboHandle(h[], true))
    end
    _check(@ccall LIBABO.abo_retain(m.gpx.ptr::Ptr{Cvoid})::Int32)
    _with(m, AboHandle(m.gpx.ptr))
end

function _create(m::HipGradientGP)
    prm = Ref(AboParams(_family(get_kernel_constructor(m)), m.device, get_lengthscale(m)[1], get_scale(m)[1], m.noise_var,
                        _mean_vec(m)[1], m.jitter, m.n_max, 0))
    h = Ref{Ptr{Cvoid}}(); mv = _mean_vec(m)
    GC.@preserve mv _check(@ccall LIBABO.abo_create_grad(prm::Ptr{AboParams}, m.p::Int32, mv::Ptr{Float64}, h::Ptr{Ptr{Cvoid}})::Int32)
    AboHandle(h[])
end

function update(m::HipGradientGP, xs::AbstractVector, ys::AbstractVector)         # :659-668
    _ensure_abi()
    X = _pack(xs); d, N = size(X)
    length(ys) == N || throw(DimensionMismatch("xs has $N points, ys $(length(ys)) observations"))
    all(y -> length(y) == m.p, ys) || throw(DimensionMismatch("each observation must hold p = $(m.p) values"))
    y = collect(Float64, prep_output(m, ys)); info = Ref{Int64}(0)
    if _multi(m)                                           # replicated on every listed device (abo_mgpu_create_grad + abo_mgpu_fit)
        prm = Ref(AboParams(_family(get_kernel_constructor(m)), m.device, get_lengthscale(m)[1], get_scale(m)[1], m.noise_var,
                            _mean_vec(m)[1], m.jitter, m.n_max, 0))
        h = Ref{Ptr{Cvoid}}(); mv = _mean_vec(m); devs = m.devices
        GC.@preserve mv devs _check(@ccall LIBABO.abo_mgpu_create_grad(prm::Ptr{AboParams}, m.p::Int32, mv::Ptr{Float64},
                                                                        length(devs)::Int32, devs::Ptr{Int32}, h::Ptr{Ptr{Cvoid}})::Int32)
        g = AboHandle(h[], true)
        GC.@preserve X y _check(@abocall(LIBABO.abo_mgpu_fit(g.ptr::Ptr{Cvoid}, X::Ptr{Float64}, N::Int64, d::Int32,
            y::Ptr{Float64}, info::Ptr{Int64})::Int32), info[])
        return _with(m, g)
    end
    hd = _create(m)
    GC.@preserve X y _check(@abocall(LIBABO.abo_fit(hd.ptr::Ptr{Cvoid}, X::Ptr{Float64}, N::Int64, d::Int32,
        y::Ptr{Float64}, 0::Int32, info::Ptr{Int64})::Int32), info[])
    _with(m, hd)
end

function append(m::HipGradientGP, x::AbstractVector{Float64}, y::AbstractVector{Float64})   # one observation = p rows
    length(y) == m.p || throw(DimensionMismatch("the observation must hold p = $(m.p) values"))
    h = Ref{Ptr{Cvoid}}(); info = Ref{Int64}(0)
    if m.gpx.multi
        n = copy(m)
        GC.@preserve x y _check(@ccall(LIBABO.abo_mgpu_append_grad(n.gpx.ptr::Ptr{Cvoid}, x::Ptr{Float64}, length(x)::Int32,
            y::Ptr{Float64}, info::Ptr{Int64}, C_NULL::Ptr{Cvoid})::Int32), info[])
        return n
    end
    GC.@preserve x y _check(@ccall(LIBABO.abo_append_grad(m.gpx.ptr::Ptr{Cvoid}, x::Ptr{Float64}, length(x)::Int32, y::Ptr{Float64},
                                                           info::Ptr{Int64}, h::Ptr{Ptr{Cvoid}})::Int32), info[])
    _with(m, AboHandle(h[]))
end

function _predict_f(m::HipGradientGP, x, want_mu, want_var)                        # function output only
    Z = _pack(x); d, M = size(Z)
    mu = want_mu ? Vector{Float64}(undef, M) : Float64[]; var = want_var ? Vector{Float64}(undef, M) : Float64[]
    pm = want_mu ? pointer(mu) : Ptr{Float64}(C_NULL); pv = want_var ? pointer(var) : Ptr{Float64}(C_NULL)
    GC.@preserve Z mu var begin
        if m.gpx.multi                                       # function output, candidates sharded over the group's devices
            _check(@abocall LIBABO.abo_mgpu_predict(m.gpx.ptr::Ptr{Cvoid}, Z::Ptr{Float64}, M::Int64, d::Int32,
                                                                 pm::Ptr{Float64}, pv::Ptr{Float64})::Int32)
        else
            _check(@abocall LIBABO.abo_predict(m.gpx.ptr::Ptr{Cvoid}, Z::Ptr{Float64}, M::Int64, d::Int32,
                                                            0::Int32, pm::Ptr{Float64}, pv::Ptr{Float64}, 0::Int32)::Int32)
        end
    end
    mu, var
end
# all-output posteriors and NLML of a replicated model are answered by shard 0's handle (every shard holds the same factor)
function _shard0(m::HipGradientGP)
    m.gpx.multi || return m.gpx.ptr
    h = Ref{Ptr{Cvoid}}()
    _check(@ccall LIBABO.abo_mgpu_get(m.gpx.ptr::Ptr{Cvoid}, 0::Int32, h::Ptr{Ptr{Cvoid}})::Int32)
    h[]
end
posterior_mean(m::HipGradientGP, x::AbstractVector) = _predict_f(m, x, true, false)[1]    # :985
posterior_var(m::HipGradientGP, x::AbstractVector)  = _predict_f(m, x, false, true)[2]    # :1001
posterior_mean(m::HipGradientGP, x::Real) = posterior_mean(m, [x])
posterior_var(m::HipGradientGP, x::Real)  = posterior_var(m, [x])

function _predict_all(m::HipGradientGP, x, want_mu, want_var)                      # all p outputs, by outputs (length p·M)
    Z = _pack(x); d, M = size(Z)
    mu = want_mu ? Vector{Float64}(undef, M * m.p) : Float64[]; var = want_var ? Vector{Float64}(undef, M * m.p) : Float64[]
    pm = want_mu ? pointer(mu) : Ptr{Float64}(C_NULL); pv = want_var ? pointer(var) : Ptr{Float64}(C_NULL)
    GC.@preserve Z mu var _check(@abocall LIBABO.abo_predict_grad(_shard0(m)::Ptr{Cvoid}, Z::Ptr{Float64}, M::Int64,
        d::Int32, 0::Int32, pm::Ptr{Float64}, pv::Ptr{Float64}, 0::Int32)::Int32)
    mu, var
end
posterior_grad_mean(m::HipGradientGP, x) = _predict_all(m, x isa Real ? [x] : x, true, false)[1]   # :936
posterior_grad_var(m::HipGradientGP, x)  = _predict_all(m, x isa Real ? [x] : x, false, true)[2]   # :953

# per-point p×p covariance blocks (+ means, + GradientNormUCB scores) — cross-point covariances are not formed
function _grad_cov(m::HipGradientGP, x, β)
    Z = _pack(x); d, M = size(Z); p = m.p
    mu = Matrix{Float64}(undef, p, M); cov = Array{Float64}(undef, p, p, M); sc = Vector{Float64}(undef, M)
    GC.@preserve Z mu cov sc _check(@abocall LIBABO.abo_predict_grad_cov(_shard0(m)::Ptr{Cvoid}, Z::Ptr{Float64},
        M::Int64, d::Int32, 0::Int32, Float64(β)::Float64, mu::Ptr{Float64}, cov::Ptr{Float64}, sc::Ptr{Float64}, 0::Int32)::Int32)
    mu, cov, sc
end
posterior_grad_cov(m::HipGradientGP, x) = (c = _grad_cov(m, x isa Real ? [x] : x, 0.0)[2]; size(c, 3) == 1 ? c[:, :, 1] : c)   # :969
(a::GradientNormUCB)(m::HipGradientGP, x::AbstractVector) = _grad_cov(m, x, a.β)[3]       # gradNormUCB.jl:43-51, all points at once

# optimize_acquisition (acq_utils.jl:33-73) on a gradient-enhanced model in ONE ccall — what the reference's tutorials run
# (GradientGP + GradientNormUCB / EI; the stock path: up to n_local serial Optim runs of M = 1 ccalls, each refactoring nothing
# but paying O(((d+1)N)²) per objective value and 2d more per finite-difference gradient).  Function-value acquisitions get their
# analytic gradient from the all-output posterior (∇μ = E[∇f] − m_∇, ∇σ² = 2·Cov(f, ∇f)); GradientNormUCB is differentiated by
# central differences on the device, as the reference differentiates everything.
_terms(a::GradientNormUCB, w=1.0) = [AboAcqTerm(Int32(4), Int32(0), Float64(a.β), 0.0, Float64(w))]
function optimize_acquisition(acqf::Union{ExpectedImprovement,UpperConfidenceBound,ProbabilityImprovement,GradientNormUCB,EnsembleAcquisition},
                              m::HipGradientGP, domain::ContinuousDomain; n_grid::Int=10_000, n_local::Int=100, seed::UInt64=rand(UInt64))
    m.gpx === nothing && throw(ArgumentError("surrogate is not conditioned on data yet (gpx === nothing)"))
    _optimize_terms(_terms(acqf), m.gpx, domain, n_grid, n_local, seed)
end
# fused function-value acquisitions on the gradient-enhanced model (more specific than the ::AbstractSurrogate methods)
function _acq_f(m::HipGradientGP, x, kind, p0, best)
    Z = _pack(x); d, M = size(Z); s = Vector{Float64}(undef, M)
    GC.@preserve Z s begin
        if m.gpx.multi
            _check(@abocall LIBABO.abo_mgpu_acq(m.gpx.ptr::Ptr{Cvoid}, Z::Ptr{Float64}, M::Int64, d::Int32, kind::Int32, p0::Float64,
                best::Float64, s::Ptr{Float64}, 0::Int32, C_NULL::Ptr{Float64}, C_NULL::Ptr{Int64})::Int32)
        else
            _check(@abocall LIBABO.abo_acq(m.gpx.ptr::Ptr{Cvoid}, Z::Ptr{Float64}, M::Int64, d::Int32, 0::Int32, kind::Int32,
                p0::Float64, best::Float64, 0::Int64, s::Ptr{Float64}, 0::Int32, C_NULL::Ptr{Float64}, C_NULL::Ptr{Int64}, 0::Int32)::Int32)
        end
    end
    s
end
(a::ExpectedImprovement)(m::HipGradientGP, x::AbstractVector)    = _acq_f(m, x, _acq_args(a)...)
(a::UpperConfidenceBound)(m::HipGradientGP, x::AbstractVector)   = _acq_f(m, x, _acq_args(a)...)
(a::ProbabilityImprovement)(m::HipGradientGP, x::AbstractVector) = _acq_f(m, x, _acq_args(a)...)

function unstandardized_mean_and_var(m::HipGradientGP, X, params::Tuple)           # :1019
    μ, σ = params[1], params[2][1]
    mu, var = _predict_all(m, X, true, true)
    (reshape(mu, :, m.p) .* σ) .+ μ', reshape(var, :, m.p) .* σ^2
end

function _fitted_with(m::HipGradientGP, logℓ, logs, xs, ys)
    g = HipGradientGP(exp(logs) * with_lengthscale(get_kernel_constructor(m), exp(logℓ)), m.p, m.noise_var;
                      mean=m.gp.mean, device=m.device, jitter=m.jitter)
    update(g, xs, ys)
end
function nlml(m::HipGradientGP, params, xs::AbstractVector, ys::AbstractVector)    # :684
    out = Ref{Float64}(); f = _fitted_with(m, params[1], params[2], xs, ys)
    _check(@ccall LIBABO.abo_nlml(f.gpx.ptr::Ptr{Cvoid}, out::Ptr{Float64})::Int32); out[]
end
nlml_ls(m::HipGradientGP, log_ℓ, log_scale, xs::AbstractVector, ys::AbstractVector) = nlml(m, (log_ℓ, log_scale), xs, ys)   # :719
# value and analytic gradient in one refit (∂K/∂log ℓ of the multi-output system from the derivative blocks): for Optim's only_fg!
function nlml_and_grad(m::HipGradientGP, params, xs::AbstractVector, ys::AbstractVector)
    v = Ref{Float64}(); g1 = Ref{Float64}(); g2 = Ref{Float64}(); f = _fitted_with(m, params[1], params[2], xs, ys)
    _check(@ccall LIBABO.abo_nlml_grad(f.gpx.ptr::Ptr{Cvoid}, v::Ptr{Float64}, g1::Ptr{Float64}, g2::Ptr{Float64})::Int32)
    v[], [g1[], g2[]]
end
# Dual-typed parameters (the stock driver's `autodiff=:forward`, bayesian_opt.jl:276-285): value + analytic gradient at the values,
# the caller's partials pushed through the chain rule — see HipStandardGP.jl
function nlml(m::HipGradientGP, params::AbstractVector{<:ForwardDiff.Dual{T}}, xs::AbstractVector, ys::AbstractVector) where {T}
    v, g = nlml_and_grad(m, ForwardDiff.value.(params), xs, ys)
    ForwardDiff.Dual{T}(v, g[1] * ForwardDiff.partials(params[1]) + g[2] * ForwardDiff.partials(params[2]))
end
function nlml_ls(m::HipGradientGP, log_ℓ::ForwardDiff.Dual{T}, log_scale::Real, xs::AbstractVector, ys::AbstractVector) where {T}
    v, g = nlml_and_grad(m, [ForwardDiff.value(log_ℓ), Float64(log_scale)], xs, ys)
    ForwardDiff.Dual{T}(v, g[1] * ForwardDiff.partials(log_ℓ))
end
