# HipStandardGP.jl — binding of libabo_hip.so (include/abo_hip.h, ABI version 5) for AbstractBayesOpt.jl.
#
# Drop next to src/surrogates/StandardGP.jl, `include("surrogates/HipStandardGP.jl")` from src/AbstractBayesOpt.jl
# (after StandardGP.jl and the acquisition functions) and export HipStandardGP.  Every method the BO driver calls on
# its surrogate is defined below for the new type — the driver itself (`optimize`, `update(BO, …)`,
# `standardize_problem`, `optimize_hyperparameters`), the domains and the acquisition TYPES stay untouched:
#
#   driver call site (reference file:line)                          method below
#   bayesian_opt.jl:81,:116   copy(model)                           Base.copy
#   bayesian_opt.jl:125,:379,:422; BO_utils.jl:60  update           update
#   ExpectedImprovement.jl:41-42 … posterior_mean / posterior_var   posterior_mean / posterior_var (vector and scalar)
#   bayesian_opt.jl:250-259   prep_input, prep_output, nlml, nlml_ls
#   bayesian_opt.jl:319-327   get_scale, get_kernel_constructor, _update_model_parameters
#   bayesian_opt.jl:391       get_lengthscale, get_scale
#   bayesian_opt.jl:428; ExpectedImprovement.jl:82  _get_minimum
#   BO_utils.jl:48,:55,:59    get_mean_std, rescale_model, std_y
#   StandardGP.jl:395-404     unstandardized_mean_and_var
#   bayesian_opt.jl:430       optimize_acquisition(acqf, model, domain)   optimize_acquisition (one ccall: grid + refinement)
#
# NOTE: no Julia toolchain exists in the build image, so this file has not been executed there.  What HAS been
# executed is the same sequence of C-ABI calls from a host with neither Python nor PyTorch in the process
# (tests/c_abi_harness.c, run by tests/test_gpu_c_abi.py on the GPU box), and the Python package
# abstractbayesopt.jl_amd/, which is this binding written with ctypes.

const LIBABO = get(ENV, "ABO_HIP_LIB", "libabo_hip.so")
# Long calls (a config-3 step is half a second) should not hold up Julia's GC: `@ccall gc_safe=true` exists from Julia 1.12 on;
# the reference supports 1.11 too (Project.toml:37), where the plain `@ccall` is used.  Every call that can take more than a few
# microseconds goes through @abocall.
@static if VERSION >= v"1.12.0-"
    macro abocall(ex) esc(:(@ccall gc_safe=true $ex)) end
else
    macro abocall(ex) esc(:(@ccall $ex)) end
end
const ABO_ABI = Int32(7)                 # ABO_ABI_VERSION of the header this file was written against
const _abi_checked = Ref(false)
# a stale libabo_hip.so on the load path would otherwise fail at the first missing symbol, somewhere inside a BO step
function _ensure_abi()
    _abi_checked[] && return
    v = @ccall LIBABO.abo_abi_version()::Int32
    v == ABO_ABI || error("libabo_hip.so reports ABI version $v, HipStandardGP.jl needs $ABO_ABI (rebuild the library: " *
                          "python -c 'import __graft_entry__ as g; g.build()')")
    _abi_checked[] = true
end

struct AboParams            # must match `struct abo_params` (include/abo_hip.h)
    family::Int32; device::Int32
    ell::Float64; sigma_f2::Float64; noise_var::Float64; mean_c::Float64; jitter::Float64
    n_max::Int64; chunk::Int64
end

mutable struct AboHandle    # owns one reference to an `abo_gp` (or, multi = true, to an `abo_mgpu`)
    ptr::Ptr{Cvoid}
    multi::Bool
    function AboHandle(p, multi=false)
        h = new(p, multi)
        finalizer(h) do x
            x.multi ? (@ccall LIBABO.abo_mgpu_destroy(x.ptr::Ptr{Cvoid})::Int32) : (@ccall LIBABO.abo_destroy(x.ptr::Ptr{Cvoid})::Int32)
        end
        h
    end
end

struct HipStandardGP{T} <: AbstractSurrogate
    gp::AbstractGPs.GP                   # prior (mean + normal-form kernel), as StandardGP.jl:11-16
    noise_var::T
    gpx::Union{Nothing,AboHandle}        # device state instead of a PosteriorGP
    devices::Vector{Int32}               # one entry: single-device handle; several: abo_mgpu (sharding inside the library)
    jitter::Float64
    n_max::Int64                         # capacity for `append` (0 = size to the fit)
end

_family(::SqExponentialKernel) = Int32(0); _family(::Matern52Kernel) = Int32(1)
_family(::ApproxMatern52Kernel) = Int32(1); _family(::ApproxMatern72Kernel) = Int32(2)
_family(::Matern32Kernel) = Int32(3)

function HipStandardGP(kernel::Kernel, noise_var; mean=nothing, devices=[0], jitter=0.0, n_max=0)
    s = StandardGP(kernel, noise_var; mean=mean)           # reuse the normal-form logic (StandardGP.jl:41-64)
    HipStandardGP(s.gp, noise_var, nothing, Int32.(devices), Float64(jitter), Int64(n_max))
end
# Engine of the N²·M variance contraction behind posterior_var (include/abo_hip.h: abo_set_contraction), process-wide for the
# handles created from now on — :auto (int8-residue engine from 1536 factor rows, fp64 MFMA below), :fp64 or :int8;
# `moduli` = 8 … 16 (0 = 14).  The environment variable ABO_CONTRACTION = auto | fp64 | int8 | int8:<moduli> seeds the same default.
function set_contraction!(engine::Symbol=:auto; moduli::Integer=0)
    e = Dict(:auto => Int32(0), :fp64 => Int32(1), :int8 => Int32(2))[engine]
    _check(@ccall LIBABO.abo_set_contraction(C_NULL::Ptr{Cvoid}, e::Int32, Int32(moduli)::Int32)::Int32)
end

_with(m::HipStandardGP, gpx) = HipStandardGP(m.gp, m.noise_var, gpx, m.devices, m.jitter, m.n_max)
_multi(m::HipStandardGP) = length(m.devices) > 1

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
    d = length(first(x)); all(v -> length(v) == d, x) || throw(DimensionMismatch("input points differ in length"))
    reduce(hcat, x)                                         # d×M column-major == point-major
end
_pack(x::AbstractMatrix{Float64}) = x                       # ColVecs-style d×M: zero copy

# ---- hyper-parameter accessors: read from the prior's normal-form kernel, as StandardGP.jl:261-287 does ------------
get_lengthscale(m::HipStandardGP) = 1 ./ m.gp.kernel.kernel.transform.s           # 1-element Vector
get_scale(m::HipStandardGP) = m.gp.kernel.σ²                                      # 1-element Vector
get_kernel_constructor(m::HipStandardGP) = m.gp.kernel.kernel.kernel
_mean_c(m::HipStandardGP) = m.gp.mean isa ZeroMean ? 0.0 : Float64(m.gp.mean.c)
prep_input(::HipStandardGP, xs::Vector) = xs                                      # StandardGP.jl:301
prep_output(::HipStandardGP, ys::Vector) = ys                                     # StandardGP.jl:315
_get_minimum(::HipStandardGP, ys::Vector) = minimum(ys)                            # StandardGP.jl:418

# a new un-conditioned model with another kernel, everything else kept (StandardGP.jl:246-248)
_update_model_parameters(m::HipStandardGP, k::Kernel) =
    HipStandardGP(k, m.noise_var; mean=m.gp.mean, devices=m.devices, jitter=m.jitter, n_max=m.n_max)

# ---- standardisation helpers: host scalars only, so they FORWARD to the reference's own methods (StandardGP.jl:164-232) on a
# prior-only StandardGP holding the same gp / noise — nothing of their arithmetic is restated here -----------------------
_ref(m::HipStandardGP) = StandardGP(m.gp, m.noise_var, nothing)
get_mean_std(m::HipStandardGP, y_train::Vector, choice::String) = get_mean_std(_ref(m), y_train, choice)
std_y(m::HipStandardGP, ys::Vector, μ, σ) = std_y(_ref(m), ys, μ, σ)
function rescale_model(m::HipStandardGP, σ)
    r = rescale_model(_ref(m), σ)                            # kernel scale / σ², noise / σ², a ConstMean moves with the data
    HipStandardGP(r.gp, r.noise_var, nothing, m.devices, m.jitter, m.n_max)
end
function unstandardized_mean_and_var(m::HipStandardGP, xs::AbstractVector, params::Tuple)   # StandardGP.jl:395-404
    μ, σ = params[1], params[2]
    mu, var = _predict(m, xs, true, true)
    (mu .* σ) .+ μ, var .* σ^2
end

# ---- copy / update -----------------------------------------------------------------------------------------------
function Base.copy(m::HipStandardGP)                                              # StandardGP.jl:26
    m.gpx === nothing && return m
    if m.gpx.multi
        h = Ref{Ptr{Cvoid}}()
        _check(@ccall LIBABO.abo_mgpu_clone(m.gpx.ptr::Ptr{Cvoid}, h::Ptr{Ptr{Cvoid}})::Int32)
        return _with(m, AboHandle(h[], true))
    end
    _check(@ccall LIBABO.abo_retain(m.gpx.ptr::Ptr{Cvoid})::Int32)
    _with(m, AboHandle(m.gpx.ptr))
end

_params(m::HipStandardGP) = AboParams(_family(get_kernel_constructor(m)), m.devices[1], get_lengthscale(m)[1],
                                      get_scale(m)[1], m.noise_var, _mean_c(m), m.jitter, m.n_max, 0)

function update(m::HipStandardGP, xs::AbstractVector, ys::AbstractVector)          # StandardGP.jl:79-83
    _ensure_abi()
    X = _pack(xs); d, N = size(X); length(ys) == N || throw(DimensionMismatch("xs has $N points, ys $(length(ys)) values"))
    p = Ref(_params(m)); h = Ref{Ptr{Cvoid}}(); info = Ref{Int64}(0); y = collect(Float64, ys)
    if _multi(m)
        devs = m.devices
        GC.@preserve devs _check(@ccall LIBABO.abo_mgpu_create(p::Ptr{AboParams}, length(devs)::Int32, devs::Ptr{Int32},
                                                                h::Ptr{Ptr{Cvoid}})::Int32)
        hd = AboHandle(h[], true)
        GC.@preserve X y _check(@abocall(LIBABO.abo_mgpu_fit(hd.ptr::Ptr{Cvoid}, X::Ptr{Float64}, N::Int64,
            d::Int32, y::Ptr{Float64}, info::Ptr{Int64})::Int32), info[])
        return _with(m, hd)
    end
    _check(@ccall LIBABO.abo_create(p::Ptr{AboParams}, h::Ptr{Ptr{Cvoid}})::Int32)
    hd = AboHandle(h[])
    GC.@preserve X y _check(@abocall(LIBABO.abo_fit(hd.ptr::Ptr{Cvoid}, X::Ptr{Float64}, N::Int64,
        d::Int32, y::Ptr{Float64}, 0::Int32, info::Ptr{Int64})::Int32), info[])
    _with(m, hd)
end

# ---- posterior -----------------------------------------------------------------------------------------------------
function _predict(m::HipStandardGP, x, want_mu, want_var)
    m.gpx === nothing && throw(ArgumentError("surrogate is not conditioned on data yet (gpx === nothing)"))
    Z = _pack(x); d, M = size(Z)
    mu = want_mu ? Vector{Float64}(undef, M) : Float64[]; var = want_var ? Vector{Float64}(undef, M) : Float64[]
    pm = want_mu ? pointer(mu) : Ptr{Float64}(C_NULL); pv = want_var ? pointer(var) : Ptr{Float64}(C_NULL)
    GC.@preserve Z mu var begin
        if m.gpx.multi
            _check(@abocall LIBABO.abo_mgpu_predict(m.gpx.ptr::Ptr{Cvoid}, Z::Ptr{Float64}, M::Int64, d::Int32,
                                                                 pm::Ptr{Float64}, pv::Ptr{Float64})::Int32)
        else
            _check(@abocall LIBABO.abo_predict(m.gpx.ptr::Ptr{Cvoid}, Z::Ptr{Float64}, M::Int64, d::Int32,
                                                            0::Int32, pm::Ptr{Float64}, pv::Ptr{Float64}, 0::Int32)::Int32)
        end
    end
    mu, var
end
posterior_mean(m::HipStandardGP, x::AbstractVector) = _predict(m, x, true, false)[1]     # StandardGP.jl:361
posterior_var(m::HipStandardGP, x::AbstractVector)  = _predict(m, x, false, true)[2]     # StandardGP.jl:377
posterior_mean(m::HipStandardGP, x::Real) = posterior_mean(m, [x])                        # StandardGP.jl:329
posterior_var(m::HipStandardGP, x::Real)  = posterior_var(m, [x])                         # StandardGP.jl:345

# ---- fused acquisition: more specific than (EI)(::AbstractSurrogate, x) at ExpectedImprovement.jl:40 ---------------
function _acq(m::HipStandardGP, x, kind, p0, best; k=0, scores=true)
    Z = _pack(x); d, M = size(Z)
    s = scores ? Vector{Float64}(undef, M) : Float64[]; ps = scores ? pointer(s) : Ptr{Float64}(C_NULL)
    tv = Vector{Float64}(undef, k); ti = Vector{Int64}(undef, k)
    GC.@preserve Z s tv ti begin
        if m.gpx.multi
            _check(@abocall LIBABO.abo_mgpu_acq(m.gpx.ptr::Ptr{Cvoid}, Z::Ptr{Float64}, M::Int64, d::Int32,
                kind::Int32, p0::Float64, best::Float64, ps::Ptr{Float64}, k::Int32, tv::Ptr{Float64}, ti::Ptr{Int64})::Int32)
        else
            _check(@abocall LIBABO.abo_acq(m.gpx.ptr::Ptr{Cvoid}, Z::Ptr{Float64}, M::Int64, d::Int32, 0::Int32,
                kind::Int32, p0::Float64, best::Float64, 0::Int64, ps::Ptr{Float64}, k::Int32, tv::Ptr{Float64},
                ti::Ptr{Int64}, 0::Int32)::Int32)
        end
    end
    s, tv, ti .+ 1                                            # C indices are 0-based
end
_acq_args(a::ExpectedImprovement) = (Int32(0), Float64(a.ξ), Float64(a.best_y))
_acq_args(a::UpperConfidenceBound) = (Int32(1), Float64(a.β), 0.0)
_acq_args(a::ProbabilityImprovement) = (Int32(2), Float64(a.ξ), Float64(a.best_y))
(a::ExpectedImprovement)(m::HipStandardGP, x::AbstractVector)    = _acq(m, x, _acq_args(a)...)[1]
(a::UpperConfidenceBound)(m::HipStandardGP, x::AbstractVector)   = _acq(m, x, _acq_args(a)...)[1]
(a::ProbabilityImprovement)(m::HipStandardGP, x::AbstractVector) = _acq(m, x, _acq_args(a)...)[1]

# The grid stage of optimize_acquisition (acq_utils.jl:44-52) in one call: the Latin-hypercube grid is generated on the
# devices (shard by shard), scored, and only the n_local best starts come back.  Returns (points::Vector{Vector}, scores).
function grid_stage(acqf::Union{ExpectedImprovement,UpperConfidenceBound,ProbabilityImprovement}, m::HipStandardGP,
                    lower::Vector{Float64}, upper::Vector{Float64}; n_grid=10_000, n_local=100, seed=rand(UInt64))
    m.gpx === nothing && throw(ArgumentError("surrogate is not conditioned on data yet (gpx === nothing)"))
    kind, p0, best = _acq_args(acqf); d = length(lower); k = min(n_local, n_grid)
    tv = Vector{Float64}(undef, k); ti = Vector{Int64}(undef, k); tx = Matrix{Float64}(undef, d, k)
    GC.@preserve lower upper tv ti tx begin
        if m.gpx.multi
            _check(@abocall LIBABO.abo_mgpu_acq_lhs(m.gpx.ptr::Ptr{Cvoid}, n_grid::Int64, d::Int32, lower::Ptr{Float64},
                upper::Ptr{Float64}, seed::UInt64, kind::Int32, p0::Float64, best::Float64, k::Int32, tv::Ptr{Float64},
                ti::Ptr{Int64}, tx::Ptr{Float64})::Int32)
        else                                     # the same stage on the model's own handle (abo_acq_lhs): no refit, no group
            _check(@abocall LIBABO.abo_acq_lhs(m.gpx.ptr::Ptr{Cvoid}, n_grid::Int64, d::Int32, lower::Ptr{Float64},
                upper::Ptr{Float64}, seed::UInt64, kind::Int32, p0::Float64, best::Float64, k::Int32, tv::Ptr{Float64},
                ti::Ptr{Int64}, tx::Ptr{Float64})::Int32)
        end
    end
    [tx[:, j] for j in 1:k], tv
end
# optimize_acquisition (acq_utils.jl:33-73) in ONE ccall — more specific than the generic method (concrete acquisition and surrogate
# types), so `optimize(BO)` picks it without any change to the driver (bayesian_opt.jl:430).  The Latin-hypercube grid is generated
# on the device(s), scored, reduced to the n_local best (:44-52); every start is then refined by its own workgroup in one launch:
# the whole projected L-BFGS on the device with the analytic gradient of the acquisition function, the reference's tolerances
# (:62) and linesearchmax (:10); the best refined point comes back (:66-72).  The stock path would run up to n_local serial Optim
# runs of M = 1 ccalls with finite-difference gradients.
struct AboRefineOpts        # must match `struct abo_refine_opts`; zeros = the reference's settings
    max_iter::Int32; linesearch_max::Int32; history::Int32; reserved::Int32
    g_tol::Float64; f_abstol::Float64; x_abstol::Float64
end
function optimize_acquisition(acqf::Union{ExpectedImprovement,UpperConfidenceBound,ProbabilityImprovement}, m::HipStandardGP,
                              domain::ContinuousDomain; n_grid::Int=10_000, n_local::Int=100, seed::UInt64=rand(UInt64))
    m.gpx === nothing && throw(ArgumentError("surrogate is not conditioned on data yet (gpx === nothing)"))
    kind, p0, best = _acq_args(acqf)
    lower = collect(Float64, domain.lower); upper = collect(Float64, domain.upper); d = length(lower)
    bx = Vector{Float64}(undef, d); bv = Ref{Float64}(); opts = Ref(AboRefineOpts(0, 0, 0, 0, 0.0, 0.0, 0.0))
    GC.@preserve lower upper bx begin
        if m.gpx.multi
            _check(@abocall LIBABO.abo_mgpu_optimize_acquisition(m.gpx.ptr::Ptr{Cvoid}, kind::Int32, p0::Float64,
                best::Float64, lower::Ptr{Float64}, upper::Ptr{Float64}, d::Int32, n_grid::Int64, n_local::Int32, seed::UInt64,
                opts::Ptr{AboRefineOpts}, bx::Ptr{Float64}, bv::Ptr{Float64}, C_NULL::Ptr{Float64}, C_NULL::Ptr{Float64},
                C_NULL::Ptr{Float64}, C_NULL::Ptr{Float64})::Int32)
        else
            _check(@abocall LIBABO.abo_optimize_acquisition(m.gpx.ptr::Ptr{Cvoid}, kind::Int32, p0::Float64,
                best::Float64, lower::Ptr{Float64}, upper::Ptr{Float64}, d::Int32, n_grid::Int64, n_local::Int32, seed::UInt64,
                opts::Ptr{AboRefineOpts}, bx::Ptr{Float64}, bv::Ptr{Float64}, C_NULL::Ptr{Float64}, C_NULL::Ptr{Float64},
                C_NULL::Ptr{Float64}, C_NULL::Ptr{Float64})::Int32)
        end
    end
    bx
end
# Weighted-sum objectives: an EnsembleAcquisition (EnsembleAcq.jl:12-27, :53-55; nested ones included) flattens into at most 8
# (kind, p0, best_y, weight) terms — value Σ wᵢ·acqᵢ on ONE posterior evaluation, gradient Σ wᵢ ∇acqᵢ — and the whole
# optimize_acquisition is again one ccall (abo_optimize_acquisition_terms).  HipGradientGP.jl adds the GradientNormUCB term.
struct AboAcqTerm           # must match `struct abo_acq_term`
    kind::Int32; reserved::Int32
    p0::Float64; best_y::Float64; weight::Float64
end
_terms(a::Union{ExpectedImprovement,UpperConfidenceBound,ProbabilityImprovement}, w=1.0) =
    (t = _acq_args(a); [AboAcqTerm(t[1], Int32(0), t[2], t[3], Float64(w))])
_terms(a::EnsembleAcquisition, w=1.0) = reduce(vcat, [_terms(a.acquisitions[i], w * a.weights[i]) for i in eachindex(a.weights)])
function _optimize_terms(terms::Vector{AboAcqTerm}, gpx::AboHandle, domain::ContinuousDomain, n_grid::Int, n_local::Int, seed::UInt64)
    length(terms) <= 8 || throw(ArgumentError("the library takes at most 8 acquisition terms, got $(length(terms))"))
    lower = collect(Float64, domain.lower); upper = collect(Float64, domain.upper); d = length(lower)
    bx = Vector{Float64}(undef, d); bv = Ref{Float64}(); opts = Ref(AboRefineOpts(0, 0, 0, 0, 0.0, 0.0, 0.0)); nt = length(terms)
    GC.@preserve terms lower upper bx begin
        if gpx.multi
            _check(@abocall LIBABO.abo_mgpu_optimize_acquisition_terms(gpx.ptr::Ptr{Cvoid}, terms::Ptr{AboAcqTerm}, nt::Int32,
                lower::Ptr{Float64}, upper::Ptr{Float64}, d::Int32, n_grid::Int64, n_local::Int32, seed::UInt64,
                opts::Ptr{AboRefineOpts}, bx::Ptr{Float64}, bv::Ptr{Float64}, C_NULL::Ptr{Float64}, C_NULL::Ptr{Float64},
                C_NULL::Ptr{Float64}, C_NULL::Ptr{Float64})::Int32)
        else
            _check(@abocall LIBABO.abo_optimize_acquisition_terms(gpx.ptr::Ptr{Cvoid}, terms::Ptr{AboAcqTerm}, nt::Int32,
                lower::Ptr{Float64}, upper::Ptr{Float64}, d::Int32, n_grid::Int64, n_local::Int32, seed::UInt64,
                opts::Ptr{AboRefineOpts}, bx::Ptr{Float64}, bv::Ptr{Float64}, C_NULL::Ptr{Float64}, C_NULL::Ptr{Float64},
                C_NULL::Ptr{Float64}, C_NULL::Ptr{Float64})::Int32)
        end
    end
    bx
end
function optimize_acquisition(acqf::EnsembleAcquisition, m::HipStandardGP, domain::ContinuousDomain; n_grid::Int=10_000,
                              n_local::Int=100, seed::UInt64=rand(UInt64))
    m.gpx === nothing && throw(ArgumentError("surrogate is not conditioned on data yet (gpx === nothing)"))
    _optimize_terms(_terms(acqf), m.gpx, domain, n_grid, n_local, seed)
end
# (EA::EnsembleAcquisition)(m, x): one posterior pass, every member's epilogue on it (abo_acq_terms)
function (a::EnsembleAcquisition)(m::HipStandardGP, x::AbstractVector)
    m.gpx.multi && return sum([a.weights[i] .* a.acquisitions[i](m, x) for i in eachindex(a.weights)])    # (the reference's form)
    terms = _terms(a); Z = _pack(x); d, M = size(Z); s = Vector{Float64}(undef, M); nt = length(terms)
    GC.@preserve terms Z s _check(@abocall LIBABO.abo_acq_terms(m.gpx.ptr::Ptr{Cvoid}, Z::Ptr{Float64}, M::Int64, d::Int32, 0::Int32,
        terms::Ptr{AboAcqTerm}, nt::Int32, 0::Int64, s::Ptr{Float64}, 0::Int32, C_NULL::Ptr{Float64}, C_NULL::Ptr{Int64}, 0::Int32)::Int32)
    s
end

# ---- NLML (value; StandardGP.jl:99-114, :133-149) and its analytic gradient ------------------------------------------
function _fitted_with(m::HipStandardGP, logℓ, logs, xs, ys)
    g = HipStandardGP(exp(logs) * with_lengthscale(get_kernel_constructor(m), exp(logℓ)), m.noise_var;
                      mean=m.gp.mean, devices=m.devices[1:1], jitter=m.jitter)
    update(g, xs, ys)
end
function nlml(m::HipStandardGP, params, xs::AbstractVector, ys::AbstractVector)
    out = Ref{Float64}(); f = _fitted_with(m, params[1], params[2], xs, ys)
    _check(@ccall LIBABO.abo_nlml(f.gpx.ptr::Ptr{Cvoid}, out::Ptr{Float64})::Int32); out[]
end
nlml_ls(m::HipStandardGP, log_ℓ, log_scale, xs::AbstractVector, ys::AbstractVector) = nlml(m, (log_ℓ, log_scale), xs, ys)
# value and analytic gradient w.r.t. (log ℓ, log σ_f²) from ONE refit (abo_nlml_grad)
function nlml_and_grad(m::HipStandardGP, params, xs::AbstractVector, ys::AbstractVector)
    v = Ref{Float64}(); g1 = Ref{Float64}(); g2 = Ref{Float64}(); f = _fitted_with(m, params[1], params[2], xs, ys)
    _check(@ccall LIBABO.abo_nlml_grad(f.gpx.ptr::Ptr{Cvoid}, v::Ptr{Float64}, g1::Ptr{Float64}, g2::Ptr{Float64})::Int32)
    v[], [g1[], g2[]]
end
# The STOCK driver needs nothing else: optimize_hyperparameters (bayesian_opt.jl:253-285) hands `p -> nlml(model, p, xs, ys)` to
# Optim with `autodiff=:forward`, i.e. it evaluates the objective on ForwardDiff.Dual parameters.  Duals cannot cross a C-ABI —
# and need not: the library returns ∂NLML/∂(log ℓ, log σ_f²) analytically, so the Dual-typed methods evaluate at the VALUES and
# push the caller's partials through the chain rule,
#     nlml(p) = Dual(v, g₁·∂p₁ + g₂·∂p₂),
# which is exactly what ForwardDiff would have produced by differentiating through kernel matrix, Cholesky and solve.  More
# specific than the generic methods above, so dispatch picks them for Dual parameters only.  (A PosDefException of the refit
# propagates, as `logpdf` would throw it; the driver's per-restart try/catch, :276-299, skips that restart.)
function nlml(m::HipStandardGP, params::AbstractVector{<:ForwardDiff.Dual{T}}, xs::AbstractVector, ys::AbstractVector) where {T}
    v, g = nlml_and_grad(m, ForwardDiff.value.(params), xs, ys)
    ForwardDiff.Dual{T}(v, g[1] * ForwardDiff.partials(params[1]) + g[2] * ForwardDiff.partials(params[2]))
end
# length_scale_only = true (:253-256): only log ℓ is a Dual, the scale is the clamped start value (:247), a plain Float64
function nlml_ls(m::HipStandardGP, log_ℓ::ForwardDiff.Dual{T}, log_scale::Real, xs::AbstractVector, ys::AbstractVector) where {T}
    v, g = nlml_and_grad(m, [ForwardDiff.value(log_ℓ), Float64(log_scale)], xs, ys)
    ForwardDiff.Dual{T}(v, g[1] * ForwardDiff.partials(log_ℓ))
end

# ---- incremental update (BASELINE config 5; the reference always refits) ---------------------------------------------
function append(m::HipStandardGP, x::AbstractVector{Float64}, y::Float64)          # O(N²) instead of a refit
    info = Ref{Int64}(0)
    if m.gpx.multi
        n = copy(m)                                              # the old model stays valid (rollback is free)
        GC.@preserve x _check(@ccall(LIBABO.abo_mgpu_append(n.gpx.ptr::Ptr{Cvoid}, x::Ptr{Float64}, length(x)::Int32, y::Float64,
                                                             info::Ptr{Int64}, C_NULL::Ptr{Cvoid})::Int32), info[])
        return n
    end
    h = Ref{Ptr{Cvoid}}()
    GC.@preserve x _check(@ccall(LIBABO.abo_append(m.gpx.ptr::Ptr{Cvoid}, x::Ptr{Float64}, length(x)::Int32, y::Float64,
                                                     info::Ptr{Int64}, h::Ptr{Ptr{Cvoid}})::Int32), info[])
    _with(m, AboHandle(h[]))
end
# update(BO, x, y, i) (bayesian_opt.jl:113-150) can call `append(BO.model, x, y)` instead of
# `update(BO.model, BO.xs, BO.ys)`; `prev_gp = copy(BO.model)` stays valid because rows ≤ N are never touched.

# ---- resident candidate grid + greedy q-EI (BASELINE config 5; no reference counterpart) ------------------------------
# The grid's points, posterior (μ, σ²) and K_ZX stay on the device(s); an `append` is followed by an O(N·M) down-date instead
# of a re-evaluation.  One device: abo_cand_*; a sharded model: abo_mgpu_cand_* (the grid is split over the model's devices).
mutable struct HipCandidates
    ptr::Ptr{Cvoid}
    multi::Bool
    model::HipStandardGP          # the model the stored posterior is in sync with
    M::Int
    d::Int
    function HipCandidates(m::HipStandardGP, zs::AbstractVector)
        m.gpx === nothing && error("HipCandidates: the model has not been fitted (call update first)")
        Z = _pack(zs); d, M = size(Z); h = Ref{Ptr{Cvoid}}()
        if m.gpx.multi
            GC.@preserve Z _check(@abocall LIBABO.abo_mgpu_cand_create(m.gpx.ptr::Ptr{Cvoid}, Z::Ptr{Float64}, M::Int64, d::Int32,
                                                                        h::Ptr{Ptr{Cvoid}})::Int32)
        else
            GC.@preserve Z _check(@abocall LIBABO.abo_cand_create(m.gpx.ptr::Ptr{Cvoid}, Z::Ptr{Float64}, M::Int64, d::Int32, 0::Int32,
                                                                   h::Ptr{Ptr{Cvoid}})::Int32)
        end
        c = new(h[], m.gpx.multi, m, M, d)
        finalizer(c) do x
            x.multi ? (@ccall LIBABO.abo_mgpu_cand_destroy(x.ptr::Ptr{Cvoid})::Int32) : (@ccall LIBABO.abo_cand_destroy(x.ptr::Ptr{Cvoid})::Int32)
        end
        c
    end
end

# after update(model, xs, ys) or a hyper-parameter change: full re-evaluation of the grid
function refresh!(c::HipCandidates, m::HipStandardGP)
    if c.multi
        _check(@abocall LIBABO.abo_mgpu_cand_refresh(m.gpx.ptr::Ptr{Cvoid}, c.ptr::Ptr{Cvoid})::Int32)
    else
        _check(@abocall LIBABO.abo_cand_refresh(m.gpx.ptr::Ptr{Cvoid}, c.ptr::Ptr{Cvoid})::Int32)
    end
    c.model = m; c
end

# append(model, x, y) with the grid's posterior down-dated in the same step; returns the N+1-point model
function append_observation!(c::HipCandidates, x::AbstractVector{Float64}, y::Float64)   # (not `append!`: that name is Base's)
    m = c.model; info = Ref{Int64}(0)
    if c.multi
        n = copy(m)
        GC.@preserve x _check(@abocall(LIBABO.abo_mgpu_append(n.gpx.ptr::Ptr{Cvoid}, x::Ptr{Float64}, length(x)::Int32, y::Float64,
                                                               info::Ptr{Int64}, c.ptr::Ptr{Cvoid})::Int32), info[])
        c.model = n; return n
    end
    n = append(m, x, y)
    _check(@abocall LIBABO.abo_cand_downdate(n.gpx.ptr::Ptr{Cvoid}, c.ptr::Ptr{Cvoid})::Int32)
    c.model = n; n
end

# acquisition epilogue + top-k on the stored posterior (no kernel evaluations): (values, 1-based grid indices)
function top_k(acqf::Union{ExpectedImprovement,UpperConfidenceBound,ProbabilityImprovement}, c::HipCandidates, k::Int)
    kind, p0, best = _acq_args(acqf); tv = Vector{Float64}(undef, k); ti = Vector{Int64}(undef, k)
    if c.multi
        GC.@preserve tv ti _check(@abocall LIBABO.abo_mgpu_cand_acq(c.model.gpx.ptr::Ptr{Cvoid}, c.ptr::Ptr{Cvoid}, kind::Int32, p0::Float64,
                                                                     best::Float64, k::Int32, tv::Ptr{Float64}, ti::Ptr{Int64})::Int32)
    else
        GC.@preserve tv ti _check(@abocall LIBABO.abo_cand_acq(c.model.gpx.ptr::Ptr{Cvoid}, c.ptr::Ptr{Cvoid}, kind::Int32, p0::Float64,
                                                                best::Float64, 0::Int64, C_NULL::Ptr{Float64}, k::Int32, tv::Ptr{Float64},
                                                                ti::Ptr{Int64}, 0::Int32)::Int32)
    end
    tv, ti .+ 1
end

# Greedy (Kriging-believer) q-EI over the grid: q × (EI + arg-max over the grid, condition the grid's posterior on the fantasy
# observation y = μ(x) at the pick); model and grid are as before on return.  One library call: the block form (include/abo_hip.h —
# the posterior covariances of the `block` best candidates to every candidate from ONE pass over the resident K_ZX, rank-1
# corrections between picks, no fantasy appends; block = 0: the library default, < 0: one bordered append and one pass per pick).
# Appending the picks afterwards with their real observations, in order (`append_observation!`), finds each down-date column in
# the chain the batch left with the grid.  (points d × q, 1-based grid indices, EI values)
struct AboQeiStats          # must match `struct abo_qei_stats`
    picks::Int32; block::Int32; block_builds::Int32; block_hits::Int32
    total_ms::Float64; block_ms::Float64; pass_ms::Float64; pass_bytes::Float64; pass_flop::Float64
end
function greedy_qei(c::HipCandidates, q::Int; ξ::Float64=0.01, best_y::Float64, distinct::Bool=false, block::Int=0)
    X = Matrix{Float64}(undef, c.d, q); idx = Vector{Int64}(undef, q); ei = Vector{Float64}(undef, q)
    st = Ref(AboQeiStats(0, 0, 0, 0, 0.0, 0.0, 0.0, 0.0, 0.0))
    if c.multi
        GC.@preserve X idx ei _check(@abocall LIBABO.abo_mgpu_cand_qei(c.model.gpx.ptr::Ptr{Cvoid}, c.ptr::Ptr{Cvoid}, q::Int32, ξ::Float64,
                                                                        best_y::Float64, Int32(distinct)::Int32, X::Ptr{Float64},
                                                                        idx::Ptr{Int64}, ei::Ptr{Float64})::Int32)
        return X, idx .+ 1, ei
    end
    GC.@preserve X idx ei _check(@abocall LIBABO.abo_cand_qei(c.model.gpx.ptr::Ptr{Cvoid}, c.ptr::Ptr{Cvoid}, q::Int32, ξ::Float64,
                                                               best_y::Float64, Int32(distinct)::Int32, 0::Int64, block::Int32,
                                                               X::Ptr{Float64}, idx::Ptr{Int64}, ei::Ptr{Float64}, st::Ptr{AboQeiStats})::Int32)
    X, idx .+ 1, ei            # (st[]: block size, blocks built, picks found in a block, the pass over K_ZX in ms)
end
