# HipStandardGP.jl — binding of libabo_hip.so (include/abo_hip.h, ABI version 1) for AbstractBayesOpt.jl.
# Drop next to src/surrogates/StandardGP.jl, `include("surrogates/HipStandardGP.jl")` from src/AbstractBayesOpt.jl
# (after StandardGP.jl and the acquisition functions) and export HipStandardGP.
# NOTE: written against the reference sources without a Julia toolchain at hand (none in the build image); the
# Python package abstractbayesopt.jl_amd/ is the same binding over ctypes and is what the parity tests exercise.
# This file is the code shown in INTEGRATION.md, verbatim.

# src/surrogates/HipStandardGP.jl   — binds include/abo_hip.h (ABI version 1)
const LIBABO = get(ENV, "ABO_HIP_LIB", "libabo_hip.so")

struct AboParams            # must match `struct abo_params`
    family::Int32; device::Int32
    ell::Float64; sigma_f2::Float64; noise_var::Float64; mean_c::Float64; jitter::Float64
    n_max::Int64; chunk::Int64
end

mutable struct AboHandle    # owns one reference to an `abo_gp`
    ptr::Ptr{Cvoid}
    function AboHandle(p)
        h = new(p); finalizer(h -> (@ccall LIBABO.abo_destroy(h.ptr::Ptr{Cvoid})::Int32), h); h
    end
end

struct HipStandardGP{T} <: AbstractSurrogate
    gp::AbstractGPs.GP                   # prior (mean + normal-form kernel), as StandardGP.jl:11-16
    noise_var::T
    gpx::Union{Nothing,AboHandle}        # device state instead of a PosteriorGP
    device::Int32; jitter::Float64
end

_family(::SqExponentialKernel) = Int32(0); _family(::Matern52Kernel) = Int32(1)
_family(::ApproxMatern52Kernel) = Int32(1); _family(::ApproxMatern72Kernel) = Int32(2)
_family(::Matern32Kernel) = Int32(3)

function HipStandardGP(kernel::Kernel, noise_var; mean=nothing, device=0, jitter=0.0)
    s = StandardGP(kernel, noise_var; mean=mean)           # reuse the normal-form logic (StandardGP.jl:41-64)
    HipStandardGP(s.gp, noise_var, nothing, Int32(device), Float64(jitter))
end

function _check(st::Int32, info::Int64=0)
    st == 0 && return
    buf = Vector{UInt8}(undef, 512)
    @ccall LIBABO.abo_last_error(buf::Ptr{UInt8}, 512::Csize_t)::Int32
    msg = unsafe_string(pointer(buf))
    st == 1 && throw(LinearAlgebra.PosDefException(info))   # caught at bayesian_opt.jl:126-141
    st == 2 && throw(DimensionMismatch(msg))                # test_bayesian_opt.jl:788-817
    st == 3 && throw(ArgumentError(msg))
    error(msg)                                              # ABO_EHIP / ABO_ENOMEM propagate (:127-131)
end

# pack Vector{Float64} (d = 1) or Vector{<:AbstractVector} into a point-major d×M Matrix
_pack(x::AbstractVector{<:Real}) = reshape(collect(Float64, x), 1, :)
function _pack(x::AbstractVector{<:AbstractVector})
    d = length(first(x)); all(v -> length(v) == d, x) || throw(DimensionMismatch("ragged input"))
    reduce(hcat, x)                                         # d×M column-major == point-major
end
_pack(x::AbstractMatrix{Float64}) = x                       # ColVecs-style d×M: zero copy

Base.copy(m::HipStandardGP) = m.gpx === nothing ? m : begin     # StandardGP.jl:26
    _check(@ccall LIBABO.abo_retain(m.gpx.ptr::Ptr{Cvoid})::Int32)
    HipStandardGP(m.gp, m.noise_var, AboHandle(m.gpx.ptr), m.device, m.jitter)
end

function update(m::HipStandardGP, xs::AbstractVector, ys::AbstractVector)      # StandardGP.jl:79-83
    X = _pack(xs); d, N = size(X); length(ys) == N || throw(DimensionMismatch("xs/ys"))
    p = Ref(AboParams(_family(get_kernel_constructor(m)), m.device, get_lengthscale(m)[1], get_scale(m)[1],
                      m.noise_var, m.gp.mean isa ZeroMean ? 0.0 : m.gp.mean.c, m.jitter, 0, 0))
    h = Ref{Ptr{Cvoid}}(); info = Ref{Int64}(0)
    _check(@ccall LIBABO.abo_create(p::Ptr{AboParams}, h::Ptr{Ptr{Cvoid}})::Int32)
    hd = AboHandle(h[]); y = collect(Float64, ys)
    GC.@preserve X y _check(@ccall(gc_safe=true, LIBABO.abo_fit(hd.ptr::Ptr{Cvoid}, X::Ptr{Float64}, N::Int64,
        d::Int32, y::Ptr{Float64}, 0::Int32, info::Ptr{Int64})::Int32), info[])
    HipStandardGP(m.gp, m.noise_var, hd, m.device, m.jitter)
end

function _predict(m::HipStandardGP, x, want_mu, want_var)
    Z = _pack(x); d, M = size(Z)
    mu = want_mu ? Vector{Float64}(undef, M) : Float64[]; var = want_var ? Vector{Float64}(undef, M) : Float64[]
    GC.@preserve Z mu var _check(@ccall gc_safe=true LIBABO.abo_predict(m.gpx.ptr::Ptr{Cvoid}, Z::Ptr{Float64},
        M::Int64, d::Int32, 0::Int32, (want_mu ? pointer(mu) : C_NULL)::Ptr{Float64},
        (want_var ? pointer(var) : C_NULL)::Ptr{Float64}, 0::Int32)::Int32)
    mu, var
end
posterior_mean(m::HipStandardGP, x::AbstractVector) = _predict(m, x, true, false)[1]     # StandardGP.jl:361
posterior_var(m::HipStandardGP, x::AbstractVector)  = _predict(m, x, false, true)[2]     # StandardGP.jl:377
posterior_mean(m::HipStandardGP, x::Real) = posterior_mean(m, [x]); posterior_var(m::HipStandardGP, x::Real) = posterior_var(m, [x])

# fused acquisition: more specific than (EI)(::AbstractSurrogate, x) at ExpectedImprovement.jl:40
function _acq(m::HipStandardGP, x, kind, p0, best; k=0)
    Z = _pack(x); d, M = size(Z); s = Vector{Float64}(undef, M)
    tv = Vector{Float64}(undef, k); ti = Vector{Int64}(undef, k)
    GC.@preserve Z s tv ti _check(@ccall gc_safe=true LIBABO.abo_acq(m.gpx.ptr::Ptr{Cvoid}, Z::Ptr{Float64}, M::Int64,
        d::Int32, 0::Int32, kind::Int32, p0::Float64, best::Float64, 0::Int64, s::Ptr{Float64}, k::Int32,
        tv::Ptr{Float64}, ti::Ptr{Int64}, 0::Int32)::Int32)
    s, tv, ti .+ 1                                            # C indices are 0-based
end
(EI::ExpectedImprovement)(m::HipStandardGP, x::AbstractVector)   = _acq(m, x, Int32(0), EI.ξ, EI.best_y)[1]
(UCB::UpperConfidenceBound)(m::HipStandardGP, x::AbstractVector) = _acq(m, x, Int32(1), UCB.β, 0.0)[1]
(PI::ProbabilityImprovement)(m::HipStandardGP, x::AbstractVector) = _acq(m, x, Int32(2), PI.ξ, PI.best_y)[1]

function nlml(m::HipStandardGP, params, xs, ys)                                   # StandardGP.jl:99-114 (value)
    ℓ, s = exp.(params); g = HipStandardGP(s * with_lengthscale(get_kernel_constructor(m), ℓ), m.noise_var;
                                           mean=m.gp.mean, device=m.device, jitter=m.jitter)
    out = Ref{Float64}(); _check(@ccall LIBABO.abo_nlml(update(g, xs, ys).gpx.ptr::Ptr{Cvoid}, out::Ptr{Float64})::Int32); out[]
end
# get_lengthscale / get_scale / get_kernel_constructor / get_mean_std / std_y / rescale_model /
# _update_model_parameters / prep_input / prep_output / _get_minimum: identical one-liners to
# StandardGP.jl:164-287,:301,:315,:418 with `HipStandardGP` in place of `StandardGP`.


function append(m::HipStandardGP, x::AbstractVector{Float64}, y::Float64)          # O(N²) instead of a refit
    h = Ref{Ptr{Cvoid}}(); info = Ref{Int64}(0)
    GC.@preserve x _check(@ccall(LIBABO.abo_append(m.gpx.ptr::Ptr{Cvoid}, x::Ptr{Float64}, length(x)::Int32, y::Float64,
                                                     info::Ptr{Int64}, h::Ptr{Ptr{Cvoid}})::Int32), info[])
    HipStandardGP(m.gp, m.noise_var, AboHandle(h[]), m.device, m.jitter)
end
# update(BO, x, y, i) (bayesian_opt.jl:113-150) can call `append(BO.model, x, y)` instead of
# `update(BO.model, BO.xs, BO.ys)`; `prev_gp = copy(BO.model)` stays valid because rows ≤ N are never touched.
# Resident grids for q-EI: abo_cand_create / abo_cand_acq / abo_cand_downdate / abo_cand_save / abo_cand_restore / abo_cand_exclude.
