# ODEFilterHIP.jl -- the `ccall` binding a ProbNumDiffEq maintainer would add so that
#
#     solve(EnsembleProblem(prob; ...), EK1(order=3), EnsembleHIP(); trajectories=N, dt=..., adaptive=false)
#
# runs the Kalman predict/update/smooth hot path on an MI355X through libodefilter_hip.so
# (C ABI: include/odefilter.h).  NOT EXECUTED in the build image (no Julia toolchain there);
# it mirrors, call for call, what odefilters.jl_amd/host.py does over ctypes, which IS tested.
module ODEFilterHIP

using ProbNumDiffEq            # EK0, EK1 (src/algorithms.jl:23-51)
import DiffEqBase

const LIB = get(ENV, "ODEFILTER_HIP_LIB", "libodefilter_hip.so")

# ---- mirrors of the C structs -------------------------------------------------------------
struct OdefConfig                      # odef_config, 56 bytes
    struct_size::Int32; alg::Int32; order::Int32; diffusion::Int32; smooth::Int32
    rhs_id::Int32; d::Int32; n_params::Int32; params_shared::Int32; save_mode::Int32
    device::Int32; want_loglik::Int32; n_traj::Int64
end

const RHS_IDS = Dict(:fhn => 0, :lorenz63 => 1, :lotka_volterra => 2, :vanderpol => 3, :linear => 4, :pleiades => 5, :lorenz96 => 6)
const DIFFUSIONS = Dict(:dynamic => 0, :fixed => 1, :fixedMAP => 2)   # src/caches.jl:89-96 (the MV models are not on the device)
const F_MEAN, F_COV_TRIL, F_DIFFUSION, F_T, F_LOGLIK, F_NACCEPT, F_NREJECT, F_NF, F_NJAC, F_NSAVED,
      F_RETCODE, F_SMOOTH_MEAN, F_SMOOTH_COV_TRIL = 0:12
const F_SAMPLES = 16
const RETCODES = (:Success, :MaxIters, :DtLessThanMin, :Unstable, :Unstable)

"""Ensemble algorithm: all trajectories of an `EnsembleProblem` on one GPU (`devices` empty / one entry) or sharded
over several GPUs of the node by THIS process (`devices = 0:7`): contiguous blocks of the trajectory index, nothing
exchanged while stepping, one RCCL all-gather of the final posterior means at the end (`odef_group_*`, `odef_allgather`
of include/odefilter.h; SURVEY.md 8e)."""
struct EnsembleHIP <: DiffEqBase.EnsembleAlgorithm
    device::Int
    rhs::Symbol          # which compiled-in vector field `prob.f` corresponds to
    devices::Vector{Int32}
end
EnsembleHIP(rhs::Symbol; device=-1, devices=Int32[]) = EnsembleHIP(device, rhs, collect(Int32, devices))

lasterr(ctx) = unsafe_string(ccall((:odef_last_error, LIB), Cstring, (Ptr{Cvoid},), ctx))
# which kernel the last filter (0) / smoother (1) pass launched, and its device time in ms
function kernel_name(ctx, which::Integer)
    buf = zeros(UInt8, 256)
    GC.@preserve buf ccall((:odef_kernel_name, LIB), Cint, (Ptr{Cvoid}, Cint, Ptr{UInt8}, Csize_t), ctx, which, buf, length(buf))
    return unsafe_string(pointer(buf))
end
function kernel_time_ms(ctx, which::Integer)
    ms = Ref{Cfloat}(0); n = Ref{Cint}(0)
    ccall((:odef_kernel_time_ms, LIB), Cint, (Ptr{Cvoid}, Cint, Ptr{Cfloat}, Ptr{Cint}), ctx, which, ms, n)
    return ms[], n[]
end
check(rc, ctx) = rc == 0 || error("libodefilter_hip: " * lasterr(ctx))

function fetch(ctx, field, ::Type{T}, dims...) where {T}
    out = Array{T}(undef, dims...)
    GC.@preserve out check(ccall((:odef_get, LIB), Cint, (Ptr{Cvoid}, Cint, Ptr{Cvoid}, Csize_t),
                                 ctx, field, out, sizeof(out)), ctx)
    out
end

"""
    compile_rhs(name, source, d, n_params; include_dir) -> rhs_id

Hand a user vector field to the library as HIP C++ source (`odef_rhs_compile`, include/odefilter.h): the stand-in for
the closure `prob.f` (+ `f.jac`) that `perform_step!` calls at src/perform_step.jl:106,116-121.  The returned id goes
into `OdefConfig.rhs_id` (register it in `RHS_IDS` under a symbol to use it with `EnsembleHIP(:name)`).
State dimensions d(q+1) <= 20 (d <= 10) run on the lane / row-team kernels; above that `odef_create` builds the
matrix-core workgroup kernels around the field for the requested order and algorithm (even d <= 32, d(q+1) <= 176).
"""
function compile_rhs(name::AbstractString, source::AbstractString, d::Integer, n_params::Integer;
                     include_dir::Union{Nothing,AbstractString}=nothing)
    id = Ref{Int32}(-1)
    rc = ccall((:odef_rhs_compile, LIB), Cint, (Cstring, Cstring, Int32, Int32, Cstring, Ptr{Int32}),
               name, source, d, n_params, include_dir === nothing ? C_NULL : include_dir, id)
    rc == 0 || error("libodefilter_hip: " * lasterr(C_NULL))   # the compiler log
    RHS_IDS[Symbol(name)] = Int(id[])
    return Int(id[])
end

"""
    __solve(ensembleprob, alg::Union{EK0,EK1}, ::EnsembleHIP; trajectories, dt, adaptive, abstol, reltol)

`u0s` is a d x N matrix (column = trajectory) -- exactly the memory layout `odef_set_problem` expects.
Returns the per-trajectory solution fields as arrays with the trajectory index FIRST (Julia column-major
view of the device layout [n_save][D][N]): `mean[i, k, s]`.
"""
function DiffEqBase.__solve(eprob::DiffEqBase.EnsembleProblem, alg::Union{EK0,EK1}, ealg::EnsembleHIP;
                            trajectories::Int, u0s::Matrix{Float64}, dt=nothing, adaptive=true,
                            abstol=1e-6, reltol=1e-3, max_steps=4096,
                            nsamples::Int=0, sample_seed::UInt64=UInt64(0x5A3B1E), dense_sample_times=nothing, kwargs...)
    prob = eprob.prob
    d, N = size(u0s); @assert N == trajectories
    q = alg.order; D = d * (q + 1); TRI = D * (D + 1) ÷ 2
    p = collect(Float64, prob.p)
    !adaptive && dt === nothing && error("Fixed timestep methods require a choice of dt or choosing the tstops")
    cfg = Ref(OdefConfig(sizeof(OdefConfig), alg isa EK1 ? 1 : 0, q, DIFFUSIONS[alg.diffusionmodel],
                         alg.smooth ? 1 : 0, RHS_IDS[ealg.rhs], d, length(p), 1, 1, ealg.device, 1, N))
    h = Ref{Ptr{Cvoid}}(C_NULL)
    rc = ccall((:odef_create, LIB), Cint, (Ptr{Ptr{Cvoid}}, Ptr{OdefConfig}), h, cfg)
    rc == 0 || error("libodefilter_hip: " * lasterr(C_NULL))
    ctx = h[]
    try
        t0, t1 = Float64.(prob.tspan)
        GC.@preserve u0s p check(ccall((:odef_set_problem, LIB), Cint, (Ptr{Cvoid}, Ptr{Cdouble}, Ptr{Cdouble}, Cdouble),
                                       ctx, u0s, p, t0), ctx)
        if adaptive
            check(ccall((:odef_solve_adaptive, LIB), Cint,
                        (Ptr{Cvoid}, Cdouble, Cdouble, Cdouble, Cdouble, Ptr{Cvoid}, Int64),
                        ctx, t1, abstol, reltol, dt === nothing ? 1e-3 * (t1 - t0) : dt, C_NULL, max_steps), ctx)
        else
            tgrid = collect(t0:dt:t1); tgrid[end] < t1 && push!(tgrid, t1)   # OrdinaryDiffEq's clipped last step
            GC.@preserve tgrid check(ccall((:odef_solve_fixed, LIB), Cint, (Ptr{Cvoid}, Ptr{Cdouble}, Int64),
                                           ctx, tgrid, length(tgrid)), ctx)
        end
        alg.smooth && check(ccall((:odef_smooth, LIB), Cint, (Ptr{Cvoid},), ctx), ctx)
        ns = Int(ccall((:odef_n_save, LIB), Int64, (Ptr{Cvoid},), ctx))
        mean = fetch(ctx, alg.smooth ? F_SMOOTH_MEAN : F_MEAN, Float64, N, D, ns)
        cov  = fetch(ctx, alg.smooth ? F_SMOOTH_COV_TRIL : F_COV_TRIL, Float64, N, TRI, ns)
        tsave = adaptive ? fetch(ctx, F_T, Float64, N, ns) : fetch(ctx, F_T, Float64, ns)
        nsaved = fetch(ctx, F_NSAVED, Int32, N)
        # Adaptive solves hold one record per ATTEMPTED step: a rejected attempt repeats the previous record at the
        # unchanged time (include/odefilter.h, odef_solve_adaptive).  `keep[i, s]` marks the records the reference
        # would have saved (accepted steps only, src/integrator_utils.jl:33-48): x[i, :, keep[i, :]].
        keep = trues(N, ns)
        if adaptive
            for i in 1:N, s in 1:ns
                keep[i, s] = s <= nsaved[i] && (s == 1 || tsave[i, s] != tsave[i, s - 1])
            end
        end
        # sample_states(sol, n) / dense_sample_states(sol, n) (src/solution_sampling.jl:15-75), while the context lives
        samples = nothing; dense_samples = nothing
        if nsamples > 0 && alg.smooth
            check(ccall((:odef_sample, LIB), Cint, (Ptr{Cvoid}, Int64, UInt64, Cdouble), ctx, nsamples, sample_seed, 1.0), ctx)
            samples = fetch(ctx, F_SAMPLES, Float64, N, nsamples, D, ns)
            tq = dense_sample_times === nothing ? collect(range(t0, t1, length=1000)) : collect(Float64, dense_sample_times)
            GC.@preserve tq check(ccall((:odef_dense_sample, LIB), Cint, (Ptr{Cvoid}, Ptr{Cdouble}, Int64, Int64, UInt64, Cdouble),
                                        ctx, tq, length(tq), nsamples, sample_seed, 1.0), ctx)
            dense_samples = (fetch(ctx, F_SAMPLES, Float64, N, nsamples, D, length(tq)), tq)
        end
        return (t = tsave, keep = keep, samples = samples, dense_samples = dense_samples,
                u = view(mean, :, 1:d, :), x_mean = mean, x_cov_tril = cov,
                x_filt_mean = fetch(ctx, F_MEAN, Float64, N, D, ns),
                diffusions = fetch(ctx, F_DIFFUSION, Float64, N, ns)[:, 2:end],
                log_likelihood = fetch(ctx, F_LOGLIK, Float64, N),
                destats = (nf = fetch(ctx, F_NF, Int32, N), njacs = fetch(ctx, F_NJAC, Int32, N),
                           naccept = fetch(ctx, F_NACCEPT, Int32, N), nreject = fetch(ctx, F_NREJECT, Int32, N)),
                nsaved = nsaved,
                retcode = [RETCODES[r + 1] for r in fetch(ctx, F_RETCODE, Int32, N)])
    finally
        ccall((:odef_destroy, LIB), Cvoid, (Ptr{Cvoid},), ctx)
    end
end


grouperr(g) = unsafe_string(ccall((:odef_group_last_error, LIB), Cstring, (Ptr{Cvoid},), g))
gcheck(rc, g) = rc == 0 || error("libodefilter_hip: " * grouperr(g))

"""
    solve_sharded(eprob, alg, ealg; trajectories, u0s, dt, adaptive, ...) -> (final_mean, shards, ctxs...)

The multi-GPU path of `EnsembleHIP(rhs; devices = 0:7)`: one `odef_group` over `ealg.devices`, the whole ensemble handed
over once (`odef_group_set_problem` cuts it into the shards of `odef_shard_range`), the shards' kernels running
concurrently, and ONE collective at the end: `odef_allgather` leaves the final posterior means of all N trajectories,
`final_mean[i, k]`, on every device (returned here from device 1).  Per-shard time series stay on their device and are
read with `fetch(odef_group_ctx(g, k), ...)` exactly as in the single-GPU method -- a full gather of an every-step
record (48.9 GB at the BASELINE size) is deliberately not part of the path.
"""
function solve_sharded(eprob::DiffEqBase.EnsembleProblem, alg::Union{EK0,EK1}, ealg::EnsembleHIP;
                       trajectories::Int, u0s::Matrix{Float64}, dt=nothing, adaptive=true, abstol=1e-6, reltol=1e-3,
                       max_steps=4096)
    prob = eprob.prob
    d, N = size(u0s); @assert N == trajectories
    q = alg.order; D = d * (q + 1)
    p = collect(Float64, prob.p)
    cfg = Ref(OdefConfig(sizeof(OdefConfig), alg isa EK1 ? 1 : 0, q, DIFFUSIONS[alg.diffusionmodel],
                         alg.smooth ? 1 : 0, RHS_IDS[ealg.rhs], d, length(p), 1, 1, -1, 1, N))
    G = length(ealg.devices)
    h = Ref{Ptr{Cvoid}}(C_NULL)
    devs = ealg.devices
    rc = GC.@preserve devs ccall((:odef_group_create, LIB), Cint, (Ptr{Ptr{Cvoid}}, Ptr{OdefConfig}, Int32, Ptr{Int32}), h, cfg, G, devs)
    rc == 0 || error("libodefilter_hip: " * grouperr(C_NULL))
    g = h[]
    try
        t0, t1 = Float64.(prob.tspan)
        GC.@preserve u0s p gcheck(ccall((:odef_group_set_problem, LIB), Cint, (Ptr{Cvoid}, Ptr{Cdouble}, Ptr{Cdouble}, Cdouble), g, u0s, p, t0), g)
        if adaptive
            gcheck(ccall((:odef_group_solve_adaptive, LIB), Cint, (Ptr{Cvoid}, Cdouble, Cdouble, Cdouble, Cdouble, Ptr{Cvoid}, Int64),
                         g, t1, abstol, reltol, dt === nothing ? 1e-3 * (t1 - t0) : dt, C_NULL, max_steps), g)
        else
            tgrid = collect(t0:dt:t1); tgrid[end] < t1 && push!(tgrid, t1)
            GC.@preserve tgrid gcheck(ccall((:odef_group_solve_fixed, LIB), Cint, (Ptr{Cvoid}, Ptr{Cdouble}, Int64), g, tgrid, length(tgrid)), g)
        end
        alg.smooth && gcheck(ccall((:odef_group_smooth, LIB), Cint, (Ptr{Cvoid},), g), g)
        gcheck(ccall((:odef_allgather, LIB), Cint, (Ptr{Cvoid}, Cint), g, alg.smooth ? 1 : 0), g)   # the one collective
        final_mean = Array{Float64}(undef, N, D)       # C layout [D][N] == Julia (N, D)
        GC.@preserve final_mean gcheck(ccall((:odef_group_get_gathered, LIB), Cint,
                                             (Ptr{Cvoid}, Int32, Ptr{Cdouble}, Ptr{Ptr{Cvoid}}, Ptr{Csize_t}), g, 0, final_mean, C_NULL, C_NULL), g)
        shards = map(0:G-1) do k
            first = Ref{Int64}(0); count = Ref{Int64}(0)
            ccall((:odef_group_shard, LIB), Cint, (Ptr{Cvoid}, Int32, Ptr{Int64}, Ptr{Int64}), g, k, first, count)
            (first[] + 1):(first[] + count[])      # 1-based trajectory range of shard k
        end
        return (final_mean = final_mean, u_final = view(final_mean, :, 1:d), shards = shards)
    finally
        ccall((:odef_group_destroy, LIB), Cvoid, (Ptr{Cvoid},), g)
    end
end

end # module
