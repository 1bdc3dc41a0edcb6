# MI355Schur.jl — the reference-side binding of libmi355schur (include/mi355schur.h).
#
# Drop this module next to Fem/ and RecyclingKrylovSolvers/ of venkovic/julia-phd-krylov-spdes and
# `push!(LOAD_PATH, "./MI355Schur/")`. It is purely mechanical: every function is one `ccall`.
#
# It EXTENDS the reference's own generic functions with methods on `MiOperator` and never defines a second
# function of the same name: `cg`, `pcg`, `defcg`, `defpcg`, `eigcg`, … are imported from RecyclingKrylovSolvers
# (exports at RecyclingKrylovSolvers.jl:10-13) and `apply_local_schurs`, `apply_global_schur`,
# `apply_neumann_neumann_schur`, `get_schur_rhs`, `get_subdomain_solutions`, `NeumannNeumannSchurPreconditioner`
# from Fem (exports at Fem/Fem.jl:54-75). A script that does `using Fem; using RecyclingKrylovSolvers; using MI355Schur`
# (Example03:6-7 plus one line) therefore keeps every unqualified call site — `pcg(S, b, x, Πnn)` (Example03:193),
# `defpcg(S, b, x, ϕ, Πnn)` (:214, :224) — and Julia's dispatch picks the device method when `S` is a `MiOperator`.
# Only names the reference does not have are exported from here.
#
# NOTE: there is no `julia` in the build container, so this file has been reviewed, not executed;
# tests/test_julia_shim_cpu.py checks it mechanically (no shadowed export, every `ccall` signature against
# include/mi355schur.h, `index_base = 1` at every create).
module MI355Schur

using LinearAlgebra
using SparseArrays: SparseMatrixCSC
import Base: *, \, size
import LinearAlgebra: mul!, ldiv!
import RecyclingKrylovSolvers
import RecyclingKrylovSolvers: cg, pcg, defcg, defpcg, eigcg, eigpcg, eigdefcg, eigdefpcg, initcg, initpcg
import Fem
import Fem: apply_local_schur, apply_local_schurs, apply_global_schur, apply_neumann_neumann_schur,
            get_schur_rhs, get_subdomain_solutions, NeumannNeumannSchurPreconditioner,
            assemble_local_schurs, prepare_neumann_neumann_schur_precond

# new names only (none of them is exported by Fem or RecyclingKrylovSolvers)
export MiContext, MiOperator, MiPrecond,
       LocalSchurs, LocalSchur, MatrixFreeLocalSchurs, GlobalSchur,
       AssemblyPlan, assemble!, set_values!, SchurSetup, set_blocks!, interior_precond!, interior_iterations,
       keep_levels!, interior_solve, use_level_solver!, peer_handle!, peer_connect!, set_exchange!

const lib = get(ENV, "MI355SCHUR_LIB", "libmi355schur")
const MI_ERR_SINGULAR = Cint(-3)
const MI_ERR_RES_CAPACITY = Cint(-4)
const MI_ERR_BOUNDS = Cint(-8)

function check(rc::Cint)
  rc == 0 && return
  msg = unsafe_string(ccall((:mi_last_error, lib), Cstring, ()))
  rc == MI_ERR_SINGULAR && throw(LinearAlgebra.SingularException(0))   # `WtAW \ mu`, defcg.jl:53,273
  rc == MI_ERR_RES_CAPACITY && throw(BoundsError())                    # res_norm[it], cg.jl:47
  rc == MI_ERR_BOUNDS && throw(BoundsError())                          # V[:, nev+1] / eigvecs(...)[:, 1:nvec], eigcg.jl:101,275
  error("libmi355schur error $rc: $msg")
end

mutable struct MiContext
  h::Ptr{Cvoid}
  function MiContext(device::Integer=0)
    r = Ref{Ptr{Cvoid}}(C_NULL)
    check(ccall((:mi_ctx_create, lib), Cint, (Cint, Ref{Ptr{Cvoid}}), device, r))
    ctx = new(r[])
    finalizer(c -> ccall((:mi_ctx_destroy, lib), Cint, (Ptr{Cvoid},), c.h), ctx)
  end
end

# Anything usable as `A` (A*x, mul!) or as `M` (M \ r, ldiv!). `keep` roots Julia callbacks.
mutable struct MiOperator
  h::Ptr{Cvoid}
  n::Int
  N::Int            # LinearMaps.FunctionMap field read by Example07:273 (`S.N`)
  ctx::MiContext
  keep::Any
end
const MiPrecond = MiOperator

function wrap(ctx::MiContext, r::Ref{Ptr{Cvoid}}, keep=nothing)
  n = Ref{Int64}(0)
  check(ccall((:mi_op_size, lib), Cint, (Ptr{Cvoid}, Ref{Int64}), r[], n))
  op = MiOperator(r[], n[], n[], ctx, keep)
  finalizer(o -> ccall((:mi_op_destroy, lib), Cint, (Ptr{Cvoid},), o.h), op)
end

size(A::MiOperator) = (A.n, A.n)

function mul!(y::Vector{Float64}, A::MiOperator, x::Vector{Float64})     # cg.jl:36,93
  check(ccall((:mi_op_apply, lib), Cint, (Ptr{Cvoid}, Ptr{Float64}, Ptr{Float64}), A.h, x, y))
  y
end
*(A::MiOperator, x::Vector{Float64}) = mul!(Vector{Float64}(undef, A.n), A, x)   # cg.jl:28,83
\(M::MiOperator, r::Vector{Float64}) = M * r                                     # EPDD.jl:1389-1392
ldiv!(z::Vector{Float64}, M::MiOperator, r::Vector{Float64}) = mul!(z, M, r)     # EPDD.jl:1394-1398
ldiv!(M::MiOperator, r::Vector{Float64}) = (r .= M * copy(r))                    # EPDD.jl:1400-1403

# ---------------------------------------------------------------- flattening of the reference's containers
# ind_Γd_Γ2l[d]::Dict{Int,Int} (lΓ => lΓd, EPDD.jl:186-191)  ->  gather_idx[d][lΓd] = lΓ (1-based kept;
# the library is told index_base = 1).
function flatten_maps(ind_Γd_Γ2l::Vector{Dict{Int,Int}})
  g = [Vector{Int64}(undef, length(m)) for m in ind_Γd_Γ2l]
  for (d, m) in enumerate(ind_Γd_Γ2l), (lΓ, lΓd) in m
    g[d][lΓd] = lΓ
  end
  g
end
ptrs(v::Vector{<:Vector{T}}) where {T} = Ptr{T}[pointer(a) for a in v]

"""`LocalSchurs(ctx, Sd, ind_Γd_Γ2l, node_Γ_cnt)`: the operator `x -> apply_local_schurs(Sd, ind_Γd_Γ2l,
node_Γ_cnt, x)` (EPDD.jl:761-785) that Example03:131-135 wraps in a LinearMap."""
function LocalSchurs(ctx::MiContext, Sd::Vector, ind_Γd_Γ2l::Vector{Dict{Int,Int}}, node_Γ_cnt::Vector{Int};
                     dom_range=(0, length(Sd)))
  ndom = length(Sd); g = flatten_maps(ind_Γd_Γ2l); nd = Int64[length(x) for x in g]
  blocks = [Matrix{Float64}(S) for S in Sd]              # `Array(Sd[idom])`, column-major
  r = Ref{Ptr{Cvoid}}(C_NULL)
  GC.@preserve g blocks begin
    check(ccall((:mi_schur_assembled_create, lib), Cint,
                (Ptr{Cvoid}, Int64, Int64, Ptr{Int64}, Ptr{Ptr{Int64}}, Ptr{Ptr{Float64}}, Cint, Int64, Int64, Ref{Ptr{Cvoid}}),
                ctx.h, ndom, length(node_Γ_cnt), nd, ptrs(g), Ptr{Float64}[pointer(b) for b in blocks], 1,
                dom_range[1], dom_range[2], r))
  end
  wrap(ctx, r)
end

"""`NeumannNeumannSchurPreconditioner(ctx, ΠSd, ind_Γd_Γ2l, node_Γ_cnt)`: a method added to the constructor of Fem's
own struct (EPDD.jl:1111-1137; its fields are `ΠSd`, `ind_Γd_Γ2l`, `node_Γ_cnt`): with a `MiContext` in front the blocks
go to the device and a `MiOperator` comes back, usable wherever the reference uses `Πnn \\ r` / `ldiv!` (:1389-1403)."""
function NeumannNeumannSchurPreconditioner(ctx::MiContext, ΠSd::Vector{Matrix{Float64}},
                                           ind_Γd_Γ2l::Vector{Dict{Int,Int}}, node_Γ_cnt::Vector{Int};
                                           dom_range=(0, length(ΠSd)))
  ndom = length(ΠSd); g = flatten_maps(ind_Γd_Γ2l); nd = Int64[length(x) for x in g]
  cnt = Vector{Int64}(node_Γ_cnt)
  r = Ref{Ptr{Cvoid}}(C_NULL)
  GC.@preserve g ΠSd cnt begin
    check(ccall((:mi_nn_create, lib), Cint,
                (Ptr{Cvoid}, Int64, Int64, Ptr{Int64}, Ptr{Ptr{Int64}}, Ptr{Ptr{Float64}}, Ptr{Int64}, Cint, Int64, Int64, Ref{Ptr{Cvoid}}),
                ctx.h, ndom, length(cnt), nd, ptrs(g), Ptr{Float64}[pointer(b) for b in ΠSd], cnt, 1,
                dom_range[1], dom_range[2], r))
  end
  wrap(ctx, r)
end

# Interior solve callback: `interior(idom, rhs) -> A_II[idom] \ rhs`, e.g. a CHOLMOD factor or
# `rhs -> IterativeSolvers.cg(A_IIdd[idom], rhs, Pl=Π_IId[idom], reltol=1e-9)` (EPDD.jl:648-650).
function interior_trampoline(user::Ptr{Cvoid}, idom::Int64, n::Int64, rhs::Ptr{Float64}, sol::Ptr{Float64})::Cint
  f = unsafe_pointer_to_objref(user)[]
  try
    unsafe_wrap(Array, sol, n) .= f(Int(idom) + 1, copy(unsafe_wrap(Array, rhs, n)))
    return Cint(0)
  catch
    return Cint(1)
  end
end

csc_parts(As::Vector{SparseMatrixCSC{Float64,Int}}) =
  ([Vector{Int64}(A.colptr) for A in As], [Vector{Int64}(A.rowval) for A in As], [A.nzval for A in As])

"""`MatrixFreeLocalSchurs(ctx, A_IIdd, A_IΓdd, A_ΓΓdd, ind_Γd_Γ2l, node_Γ_cnt, interior)`: the operator of
Example03:143-150, `apply_local_schurs(A_IIdd, A_IΓdd, A_ΓΓdd, ...)` (EPDD.jl:711-747); sparse products on the
GPU, `A_IIdd^{-1}` through `interior(idom, rhs)` on the host."""
function MatrixFreeLocalSchurs(ctx::MiContext, A_IIdd, A_IΓdd, A_ΓΓdd, ind_Γd_Γ2l, node_Γ_cnt, interior)
  ndom = length(A_IΓdd); g = flatten_maps(ind_Γd_Γ2l)
  nd = Int64[length(x) for x in g]; ni = Int64[A.n for A in A_IIdd]
  igp, igi, igv = csc_parts(A_IΓdd); ggp, ggi, ggv = csc_parts(A_ΓΓdd)
  fref = Ref{Any}(interior)
  cb = @cfunction(interior_trampoline, Cint, (Ptr{Cvoid}, Int64, Int64, Ptr{Float64}, Ptr{Float64}))
  r = Ref{Ptr{Cvoid}}(C_NULL)
  GC.@preserve g igp igi igv ggp ggi ggv fref begin
    check(ccall((:mi_schur_matfree_create, lib), Cint,
                (Ptr{Cvoid}, Int64, Int64, Ptr{Int64}, Ptr{Int64}, Ptr{Ptr{Int64}},
                 Ptr{Ptr{Int64}}, Ptr{Ptr{Int64}}, Ptr{Ptr{Float64}}, Ptr{Ptr{Int64}}, Ptr{Ptr{Int64}}, Ptr{Ptr{Float64}},
                 Ptr{Cvoid}, Ptr{Cvoid}, Cint, Int64, Int64, Ref{Ptr{Cvoid}}),
                ctx.h, ndom, length(node_Γ_cnt), nd, ni, ptrs(g), ptrs(igp), ptrs(igi), ptrs(igv),
                ptrs(ggp), ptrs(ggi), ptrs(ggv), cb, pointer_from_objref(fref), 1, 0, ndom, r))
  end
  wrap(ctx, r, (fref, cb))
end

"""`MatrixFreeLocalSchurs(ctx, A_IIdd, A_IΓdd, A_ΓΓdd, ind_Γd_Γ2l, node_Γ_cnt; reltol=1e-9)`: the same operator with the
interior solve on the device — the library's restatement of `IterativeSolvers.cg(A_IIdd[idom], rhs, reltol=reltol)`
(EPDD.jl:648-650), all local subdomains iterated together."""
function MatrixFreeLocalSchurs(ctx::MiContext, A_IIdd, A_IΓdd, A_ΓΓdd, ind_Γd_Γ2l, node_Γ_cnt; reltol=1e-9)
  ndom = length(A_IΓdd); g = flatten_maps(ind_Γd_Γ2l)
  nd = Int64[length(x) for x in g]; ni = Int64[A.n for A in A_IIdd]
  iip, iii, iiv = csc_parts(A_IIdd); igp, igi, igv = csc_parts(A_IΓdd); ggp, ggi, ggv = csc_parts(A_ΓΓdd)
  r = Ref{Ptr{Cvoid}}(C_NULL)
  GC.@preserve g iip iii iiv igp igi igv ggp ggi ggv begin
    check(ccall((:mi_schur_matfree_device_create, lib), Cint,
                (Ptr{Cvoid}, Int64, Int64, Ptr{Int64}, Ptr{Int64}, Ptr{Ptr{Int64}},
                 Ptr{Ptr{Int64}}, Ptr{Ptr{Int64}}, Ptr{Ptr{Float64}}, Ptr{Ptr{Int64}}, Ptr{Ptr{Int64}}, Ptr{Ptr{Float64}},
                 Ptr{Ptr{Int64}}, Ptr{Ptr{Int64}}, Ptr{Ptr{Float64}}, Float64, Cint, Int64, Int64, Ref{Ptr{Cvoid}}),
                ctx.h, ndom, length(node_Γ_cnt), nd, ni, ptrs(g), ptrs(iip), ptrs(iii), ptrs(iiv),
                ptrs(igp), ptrs(igi), ptrs(igv), ptrs(ggp), ptrs(ggi), ptrs(ggv), reltol, 1, 0, ndom, r))
  end
  wrap(ctx, r)
end

"""`GlobalSchur(ctx, A_IId, A_IΓd, A_ΓΓ, interior)`: `x -> apply_global_schur(A_IId, A_IΓd, A_ΓΓ, x)`
(EPDD.jl:596-625), the operator of Example03:101."""
function GlobalSchur(ctx::MiContext, A_IId, A_IΓd, A_ΓΓ::SparseMatrixCSC{Float64,Int}, interior)
  ndom = length(A_IΓd); ni = Int64[A.n for A in A_IId]
  igp, igi, igv = csc_parts(A_IΓd)
  ggp, ggi = Vector{Int64}(A_ΓΓ.colptr), Vector{Int64}(A_ΓΓ.rowval)
  fref = Ref{Any}(interior)
  cb = @cfunction(interior_trampoline, Cint, (Ptr{Cvoid}, Int64, Int64, Ptr{Float64}, Ptr{Float64}))
  r = Ref{Ptr{Cvoid}}(C_NULL)
  GC.@preserve igp igi igv ggp ggi fref begin
    check(ccall((:mi_schur_global_create, lib), Cint,
                (Ptr{Cvoid}, Int64, Int64, Ptr{Int64}, Ptr{Ptr{Int64}}, Ptr{Ptr{Int64}}, Ptr{Ptr{Float64}},
                 Ptr{Int64}, Ptr{Int64}, Ptr{Float64}, Ptr{Cvoid}, Ptr{Cvoid}, Cint, Ref{Ptr{Cvoid}}),
                ctx.h, ndom, A_ΓΓ.n, ni, ptrs(igp), ptrs(igi), ptrs(igv), ggp, ggi, A_ΓΓ.nzval,
                cb, pointer_from_objref(fref), 1, r))
  end
  wrap(ctx, r, (fref, cb))
end

"""`GlobalSchur(ctx, A_IId, A_IΓd, A_ΓΓ; reltol=sqrt(eps()))`: the same operator with the interior solves on the device
(`IterativeSolvers.cg(A_IId[idom], A_IΓd[idom]*x)` restated, EPDD.jl:609-619; sqrt(eps) is that package's default)."""
function GlobalSchur(ctx::MiContext, A_IId, A_IΓd, A_ΓΓ::SparseMatrixCSC{Float64,Int}; reltol=sqrt(eps(Float64)))
  ndom = length(A_IΓd); ni = Int64[A.n for A in A_IId]
  iip, iii, iiv = csc_parts(A_IId); igp, igi, igv = csc_parts(A_IΓd)
  ggp, ggi = Vector{Int64}(A_ΓΓ.colptr), Vector{Int64}(A_ΓΓ.rowval)
  r = Ref{Ptr{Cvoid}}(C_NULL)
  GC.@preserve iip iii iiv igp igi igv ggp ggi begin
    check(ccall((:mi_schur_global_device_create, lib), Cint,
                (Ptr{Cvoid}, Int64, Int64, Ptr{Int64}, Ptr{Ptr{Int64}}, Ptr{Ptr{Int64}}, Ptr{Ptr{Float64}},
                 Ptr{Ptr{Int64}}, Ptr{Ptr{Int64}}, Ptr{Ptr{Float64}}, Ptr{Int64}, Ptr{Int64}, Ptr{Float64}, Float64, Cint, Ref{Ptr{Cvoid}}),
                ctx.h, ndom, A_ΓΓ.n, ni, ptrs(iip), ptrs(iii), ptrs(iiv), ptrs(igp), ptrs(igi), ptrs(igv),
                ggp, ggi, A_ΓΓ.nzval, reltol, 1, r))
  end
  wrap(ctx, r)
end

"""A symmetric `SparseMatrixCSC` as a device operator (config 2: `pcg(A, b, x, M)` on the full system)."""
function MiOperator(ctx::MiContext, A::SparseMatrixCSC{Float64,Int})
  r = Ref{Ptr{Cvoid}}(C_NULL)
  cp, rv = Vector{Int64}(A.colptr), Vector{Int64}(A.rowval)
  check(ccall((:mi_csr_create, lib), Cint,
              (Ptr{Cvoid}, Int64, Int64, Ptr{Int64}, Ptr{Int64}, Ptr{Float64}, Cint, Ref{Ptr{Cvoid}}),
              ctx.h, A.m, A.n, cp, rv, A.nzval, 1, r))
  wrap(ctx, r)
end

"""`MiOperator(ctx, Πnn::NeumannNeumannSchurPreconditioner)`: the reference's own preconditioner object — what
`prepare_neumann_neumann_schur_precond(Sd_local_mat, ind_Γd_Γ2l, node_Γ_cnt)` returns (EPDD.jl:1201-1220; Example03:139,
Example07:152-154) — copied to the device as it is."""
MiOperator(ctx::MiContext, Πnn::NeumannNeumannSchurPreconditioner) =
  NeumannNeumannSchurPreconditioner(ctx, Πnn.ΠSd, Πnn.ind_Γd_Γ2l, Πnn.node_Γ_cnt)

"""`LocalSchur(ctx, A_IIdd, A_IΓdd, A_ΓΓdd; reltol=1e-9)`: ONE subdomain, `xd -> apply_local_schur(A_IIdd, A_IΓdd,
A_ΓΓdd, xd; reltol)` (EPDD.jl:639-654) in its own Γ_d numbering, interior solve on the device."""
function LocalSchur(ctx::MiContext, A_IIdd::SparseMatrixCSC{Float64,Int}, A_IΓdd::SparseMatrixCSC{Float64,Int},
                    A_ΓΓdd::SparseMatrixCSC{Float64,Int}; reltol=1e-9)
  n = A_ΓΓdd.n
  MatrixFreeLocalSchurs(ctx, [A_IIdd], [A_IΓdd], [A_ΓΓdd], [Dict{Int,Int}(i => i for i in 1:n)], ones(Int, n); reltol=reltol)
end

# Methods added to the reference's own functions (EPDD.jl:639, 711, 761, 596, 1361): the operator handle takes the
# place of the block arrays, the vector argument stays last.
apply_local_schur(S::MiOperator, xd::Vector{Float64}) = S * xd
apply_local_schurs(S::MiOperator, x::Vector{Float64}) = S * x
apply_global_schur(S::MiOperator, x::Vector{Float64}) = S * x
apply_neumann_neumann_schur(Πnn::MiOperator, r::Vector{Float64}) = Πnn * r

# ---------------------------------------------------------------- set-up of the assembled mode on the device
# Methods added to Fem's `assemble_local_schurs` (EPDD.jl:667-695) and `prepare_neumann_neumann_schur_precond` (:1201-1220):
# with a MiContext in front the dense S_d come from the device's exact level elimination (mi_schur_setup_*) and the
# pseudo-inverses from mi_nn_pinv; return types are what LocalSchurs / the device preconditioner take.
mutable struct SchurSetup
  h::Ptr{Cvoid}; n_Γd::Vector{Int64}; ctx::MiContext
end
function SchurSetup(ctx::MiContext, A_IIdd::Vector{SparseMatrixCSC{Float64,Int}}, A_IΓdd::Vector{SparseMatrixCSC{Float64,Int}},
                    A_ΓΓdd::Vector{SparseMatrixCSC{Float64,Int}})
  ndom = length(A_IΓdd)
  nd = Int64[A.n for A in A_ΓΓdd]; ni = Int64[A.n for A in A_IIdd]
  iip, iii, _ = csc_parts(A_IIdd); igp, igi, _ = csc_parts(A_IΓdd); ggp, ggi, _ = csc_parts(A_ΓΓdd)
  r = Ref{Ptr{Cvoid}}(C_NULL)
  GC.@preserve iip iii igp igi ggp ggi begin
    check(ccall((:mi_schur_setup_create, lib), Cint,
                (Ptr{Cvoid}, Int64, Ptr{Int64}, Ptr{Int64}, Ptr{Ptr{Int64}}, Ptr{Ptr{Int64}}, Ptr{Ptr{Int64}}, Ptr{Ptr{Int64}},
                 Ptr{Ptr{Int64}}, Ptr{Ptr{Int64}}, Cint, Ref{Ptr{Cvoid}}),
                ctx.h, ndom, nd, ni, ptrs(iip), ptrs(iii), ptrs(igp), ptrs(igi), ptrs(ggp), ptrs(ggi), 1, r))
  end
  p = SchurSetup(r[], nd, ctx)
  finalizer(q -> ccall((:mi_schur_setup_destroy, lib), Cint, (Ptr{Cvoid},), q.h), p)
end
"""`assemble_local_schurs(plan, A_IIdd, A_IΓdd, A_ΓΓdd[, b_Id])`: one realization on a plan built once for the sparsity; returns
the dense S_d (and, with b_Id, the condensed right-hand sides A_IΓdd' (A_IIdd \\ b_Id) of `get_schur_rhs`, EPDD.jl:853-861)."""
function assemble_local_schurs(p::SchurSetup, A_IIdd, A_IΓdd, A_ΓΓdd, b_Id=nothing)
  ii = reduce(vcat, [A.nzval for A in A_IIdd]); ig = reduce(vcat, [A.nzval for A in A_IΓdd]); gg = reduce(vcat, [A.nzval for A in A_ΓΓdd])
  Sd = Vector{Float64}(undef, sum(p.n_Γd .^ 2)); w = Vector{Float64}(undef, sum(p.n_Γd))
  bI = b_Id === nothing ? C_NULL : reduce(vcat, b_Id)
  check(ccall((:mi_schur_setup_run, lib), Cint, (Ptr{Cvoid}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}),
              p.h, ii, ig, gg, bI, Sd, b_Id === nothing ? C_NULL : w))
  ends = cumsum(p.n_Γd .^ 2)
  S = [reshape(Sd[(e - n * n + 1):e], n, n) for (e, n) in zip(ends, p.n_Γd)]
  b_Id === nothing && return S
  we = cumsum(p.n_Γd)
  return S, [w[(e - n + 1):e] for (e, n) in zip(we, p.n_Γd)]
end
assemble_local_schurs(ctx::MiContext, A_IIdd, A_IΓdd, A_ΓΓdd) = assemble_local_schurs(SchurSetup(ctx, A_IIdd, A_IΓdd, A_ΓΓdd), A_IIdd, A_IΓdd, A_ΓΓdd)
"""`prepare_neumann_neumann_schur_precond(ctx, Sd, ind_Γd_Γ2l, node_Γ_cnt)`: ΠS_d = pinv(S_d, rtol = sqrt(eps)) on the device
(EPDD.jl:1211) and the device preconditioner built from them."""
function prepare_neumann_neumann_schur_precond(ctx::MiContext, Sd::Vector{Matrix{Float64}}, ind_Γd_Γ2l::Vector{Dict{Int,Int}},
                                               node_Γ_cnt::Vector{Int})
  nd = Int64[size(S, 1) for S in Sd]
  cat = reduce(vcat, [vec(S) for S in Sd]); out = similar(cat)
  check(ccall((:mi_nn_pinv, lib), Cint, (Ptr{Cvoid}, Int64, Ptr{Int64}, Ptr{Float64}, Float64, Ptr{Float64}), ctx.h, length(Sd), nd, cat, 0.0, out))
  ends = cumsum(nd .^ 2)
  ΠSd = [reshape(out[(e - n * n + 1):e], n, n) for (e, n) in zip(ends, nd)]
  NeumannNeumannSchurPreconditioner(ctx, ΠSd, ind_Γd_Γ2l, node_Γ_cnt)
end
"""New S_d / ΠS_d on an existing device operator (same maps): Example07's per-realization update without re-creating it."""
set_blocks!(op::MiOperator, blocks::Vector{Matrix{Float64}}) =
  check(ccall((:mi_dense_set_blocks, lib), Cint, (Ptr{Cvoid}, Ptr{Float64}), op.h, reduce(vcat, [vec(B) for B in blocks])))

"""`keep_levels!(plan)`: the next `assemble_local_schurs(plan, …)` keeps the elimination's factors on the device, after which
`interior_solve(plan, f)` is the exact `A_IIdd \\ f_d` of every subdomain (the reference's Cholesky solve, EPDD.jl:649, as
level sweeps) and `use_level_solver!(S, plan)` makes a matrix-free operator use it instead of its inner CG."""
keep_levels!(p::SchurSetup, on::Bool=true) = check(ccall((:mi_schur_setup_keep_levels, lib), Cint, (Ptr{Cvoid}, Cint), p.h, on ? 1 : 0))
function interior_solve(p::SchurSetup, f::Vector{Float64})
  u = similar(f)
  check(ccall((:mi_schur_setup_interior_solve, lib), Cint, (Ptr{Cvoid}, Ptr{Float64}, Ptr{Float64}), p.h, f, u))
  u
end
use_level_solver!(S::MiOperator, p::SchurSetup) = check(ccall((:mi_schur_matfree_interior_levels, lib), Cint, (Ptr{Cvoid}, Ptr{Cvoid}), S.h, p.h))

# ---------------------------------------------------------------- one Julia worker per GPU
# The reference's sketch of the distributed apply is `@sync @distributed (+) for idom` (EllipticPdePllDomainDecomposition.jl:
# 10-14). Here every worker owns a context on its own GPU and a slice of the subdomains (`dom_range` of LocalSchurs /
# NeumannNeumannSchurPreconditioner); the Γ-sum of every iteration is the library's peer exchange (include/mi355schur.h):
#   handles = [remotecall_fetch(() -> peer_handle!(ctx, r, np), workers()[r + 1]) for r in 0:np-1]     # 64 bytes each
#   @everywhere peer_connect!(ctx, myrank, handles)
# after which `pcg(S, b, x, Πnn)` on every worker runs the sharded loop and returns the same bits everywhere.
const MI_PEER_HANDLE_BYTES = 64
function peer_handle!(ctx::MiContext, rank::Integer, n_ranks::Integer)
  check(ccall((:mi_ctx_peer_init, lib), Cint, (Ptr{Cvoid}, Cint, Cint, Int64), ctx.h, rank, n_ranks, 0))
  h = Vector{UInt8}(undef, MI_PEER_HANDLE_BYTES); base = Ref{Ptr{Cvoid}}(C_NULL)
  check(ccall((:mi_ctx_peer_export, lib), Cint, (Ptr{Cvoid}, Ptr{Cvoid}, Ref{Ptr{Cvoid}}), ctx.h, h, base))
  h
end
function peer_connect!(ctx::MiContext, rank::Integer, handles::Vector{Vector{UInt8}})
  for q in 0:length(handles)-1
    q == rank && continue
    check(ccall((:mi_ctx_peer_import, lib), Cint, (Ptr{Cvoid}, Cint, Ptr{Cvoid}, Ptr{Cvoid}), ctx.h, q, handles[q + 1], C_NULL))
  end
  check(ccall((:mi_ctx_peer_ready, lib), Cint, (Ptr{Cvoid},), ctx.h))
end
"""0: off; 1: peer exchange with wait kernels between the launches; 2: the launches wait themselves (one GPU per worker)."""
set_exchange!(ctx::MiContext, mode::Integer) = check(ccall((:mi_ctx_set_exchange, lib), Cint, (Ptr{Cvoid}, Cint), ctx.h, mode))

# ---------------------------------------------------------------- solver drop-ins (whole loop on the GPU)
# Same positional order, keyword and 3-tuple return as RecyclingKrylovSolvers (cg.jl:14,67; defcg.jl:24,242).
function solve(sym::Symbol, A::MiOperator, M, b::Vector{Float64}, x::Vector{Float64}, W, maxit::Int)
  n = A.n
  res = Vector{Float64}(undef, n)                      # cg.jl:23 `res_norm = Array{T,1}(undef, n)`
  it = Ref{Int64}(0)
  rc = if sym == :cg
    ccall((:mi_cg, lib), Cint, (Ptr{Cvoid}, Ptr{Float64}, Ptr{Float64}, Int64, Float64, Ptr{Float64}, Int64, Ref{Int64}),
          A.h, b, x, maxit, 1e-7, res, n, it)
  elseif sym == :pcg
    ccall((:mi_pcg, lib), Cint, (Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Float64}, Ptr{Float64}, Int64, Float64, Ptr{Float64}, Int64, Ref{Int64}),
          A.h, M.h, b, x, maxit, 1e-7, res, n, it)
  elseif sym == :defcg
    ccall((:mi_defcg, lib), Cint, (Ptr{Cvoid}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Int64, Int64, Float64, Ptr{Float64}, Int64, Ref{Int64}),
          A.h, b, x, W, size(W, 2), maxit, 1e-7, res, n, it)
  else
    ccall((:mi_defpcg, lib), Cint, (Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Int64, Int64, Float64, Ptr{Float64}, Int64, Ref{Int64}),
          A.h, M.h, b, x, W, size(W, 2), maxit, 1e-7, res, n, it)
  end
  check(rc)
  return x, Int(it[]), res[1:it[]]
end

cg(A::MiOperator, b::Vector{Float64}, x::Vector{Float64}; maxit=0) = solve(:cg, A, nothing, b, x, nothing, maxit)
pcg(A::MiOperator, b::Vector{Float64}, x::Vector{Float64}, M::MiOperator; maxit=0) = solve(:pcg, A, M, b, x, nothing, maxit)
defcg(A::MiOperator, b::Vector{Float64}, x::Vector{Float64}, W::Matrix{Float64}; maxit=0) = solve(:defcg, A, nothing, b, x, W, maxit)
defpcg(A::MiOperator, b::Vector{Float64}, x::Vector{Float64}, W::Matrix{Float64}, M::MiOperator; maxit=0) = solve(:defpcg, A, M, b, x, W, maxit)

# On-device numeric assembly (the element loop of prepare_local_schurs, EPDD.jl:389-546, for a new coefficient vector).
# `cptr`/`ccode` (0-based, Int64) are recorded ONCE by running that loop symbolically: wherever the reference pushes
# (I, J, ΔKij) or does `b[k] += Δ` for element iel and local pair (i, j), record the code 12*(iel-1) + 3*(i-1) + (j-1)
# (or 12*(iel-1) + 9 + (i-1) for a load term) under the stored entry it lands in, in loop order.
mutable struct AssemblyPlan
  h::Ptr{Cvoid}; n_node::Int; n_entries::Int; ctx
end
function AssemblyPlan(ctx::MiContext, cells::Matrix{Int}, n_node::Int, G::Matrix{Float64}, area::Vector{Float64},
                      ue::Matrix{Float64}, be::Matrix{Float64}, n_matrix_entries::Int, cptr::Vector{Int64}, ccode::Vector{Int64})
  nel = size(cells, 2)
  ct = permutedims(cells)                  # library layout: vertex i of every element contiguous (3 x nel row-major)
  h = Ref{Ptr{Cvoid}}(C_NULL)
  check(ccall((:mi_assembly_plan_create, lib), Cint,
        (Ptr{Cvoid}, Int64, Int64, Ptr{Int64}, Cint, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Int64, Int64, Ptr{Int64}, Ptr{Int64}, Ref{Ptr{Cvoid}}),
        ctx.h, nel, n_node, ct, 1, permutedims(G), area, permutedims(ue), permutedims(be), length(cptr) - 1, n_matrix_entries, cptr, ccode, h))
  p = AssemblyPlan(h[], n_node, length(cptr) - 1, ctx)
  finalizer(q -> ccall((:mi_assembly_plan_destroy, lib), Cint, (Ptr{Cvoid},), q.h), p)
  return p
end
function assemble!(values::Vector{Float64}, p::AssemblyPlan, a::Vector{Float64})
  check(ccall((:mi_assembly_run, lib), Cint, (Ptr{Cvoid}, Ptr{Float64}, Ptr{Float64}), p.h, a, values)); values
end
set_values!(S::MiOperator, ii, ig, gg) =
  check(ccall((:mi_schur_matfree_set_values, lib), Cint, (Ptr{Cvoid}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}),
              S.h, ii === nothing ? C_NULL : ii, ig === nothing ? C_NULL : ig, gg === nothing ? C_NULL : gg))
"""`interior_precond!(S, :diagonal)`: the `precond(s)` keyword of apply_local_schur(s) / apply_global_schur (EPDD.jl:648-650)
for the device interior CG — `:none` (default, plain CG) or `:diagonal` (`Pl = Diagonal(A_IIdd)`)."""
interior_precond!(S::MiOperator, kind::Symbol) =
  check(ccall((:mi_schur_interior_precond, lib), Cint, (Ptr{Cvoid}, Cint), S.h, kind === :diagonal ? 1 : 0))
function interior_iterations(S::MiOperator)
  it = Ref{Int64}(0)
  check(ccall((:mi_schur_interior_iterations, lib), Cint, (Ptr{Cvoid}, Ref{Int64}), S.h, it)); it[]
end
function get_schur_rhs(S::MiOperator, b_I::Vector{Float64}, b_Γ::Vector{Float64})      # EPDD.jl:835-864
  out = similar(b_Γ)
  check(ccall((:mi_schur_matfree_rhs, lib), Cint, (Ptr{Cvoid}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}), S.h, b_I, b_Γ, out)); out
end
# the reference's own container: b_Id::Vector{Vector{Float64}} (one vector per subdomain), concatenated for the library
get_schur_rhs(S::MiOperator, b_Id::Vector{Vector{Float64}}, b_Γ::Vector{Float64}) = get_schur_rhs(S, reduce(vcat, b_Id), b_Γ)
"""`get_subdomain_solutions(S, u_Γ, b_Id)` (EPDD.jl:1014-1025): `u_Id = A_IId \\ (b_Id - A_IΓd u_Γ)` with the operator's
own interior solve; returns one vector per subdomain like the reference."""
function get_subdomain_solutions(S::MiOperator, u_Γ::Vector{Float64}, b_Id::Vector{Vector{Float64}})
  b_I = reduce(vcat, b_Id); u_I = similar(b_I)
  check(ccall((:mi_schur_matfree_interior_solutions, lib), Cint, (Ptr{Cvoid}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}),
              S.h, u_Γ, b_I, u_I))
  ends = cumsum(length.(b_Id))
  [u_I[(e - length(b) + 1):e] for (e, b) in zip(ends, b_Id)]
end

# eigCG family and Init-CG (eigcg.jl:27-33, 143-150; defcg.jl:111-116, 337-343; initcg.jl:28-33, 106-111).
# Same positional orders and 4-tuple / 3-tuple returns as the reference (Example09_..._Functions.jl:314, 364).
function eigsolve(kind::Symbol, A::MiOperator, M, b::Vector{Float64}, x::Vector{Float64}, W, nvec::Int, spdim::Int, maxit::Int)
  n = A.n
  res = Vector{Float64}(undef, n); it = Ref{Int64}(0)
  V = Matrix{Float64}(undef, n, nvec)
  tail = (Int64(spdim), Int64(maxit), 1e-7, res, Int64(n), it, V)
  rc = if kind == :eigcg
    ccall((:mi_eigcg, lib), Cint, (Ptr{Cvoid}, Ptr{Float64}, Ptr{Float64}, Int64, Int64, Int64, Float64, Ptr{Float64}, Int64, Ref{Int64}, Ptr{Float64}),
          A.h, b, x, Int64(nvec), tail...)
  elseif kind == :eigpcg
    ccall((:mi_eigpcg, lib), Cint, (Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Float64}, Ptr{Float64}, Int64, Int64, Int64, Float64, Ptr{Float64}, Int64, Ref{Int64}, Ptr{Float64}),
          A.h, M.h, b, x, Int64(nvec), tail...)
  elseif kind == :eigdefcg
    ccall((:mi_eigdefcg, lib), Cint, (Ptr{Cvoid}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Int64, Int64, Int64, Float64, Ptr{Float64}, Int64, Ref{Int64}, Ptr{Float64}),
          A.h, b, x, W, Int64(nvec), tail...)
  else
    ccall((:mi_eigdefpcg, lib), Cint, (Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Int64, Int64, Int64, Float64, Ptr{Float64}, Int64, Ref{Int64}, Ptr{Float64}),
          A.h, M.h, b, x, W, Int64(nvec), tail...)
  end
  check(rc)
  return x, Int(it[]), res[1:it[]], V
end

eigcg(A::MiOperator, b::Vector{Float64}, x::Vector{Float64}, nvec::Int, spdim::Int; maxit=0) =
  eigsolve(:eigcg, A, nothing, b, x, nothing, nvec, spdim, maxit)
eigpcg(A::MiOperator, b::Vector{Float64}, x::Vector{Float64}, M::MiOperator, nvec::Int, spdim::Int; maxit=0) =
  eigsolve(:eigpcg, A, M, b, x, nothing, nvec, spdim, maxit)
eigdefcg(A::MiOperator, b::Vector{Float64}, x::Vector{Float64}, W::Matrix{Float64}, spdim::Int; maxit=0) =
  eigsolve(:eigdefcg, A, nothing, b, x, W, size(W, 2), spdim, maxit)
eigdefpcg(A::MiOperator, b::Vector{Float64}, x::Vector{Float64}, M::MiOperator, W::Matrix{Float64}, spdim::Int; maxit=0) =
  eigsolve(:eigdefpcg, A, M, b, x, W, size(W, 2), spdim, maxit)

function initsolve(A::MiOperator, M, b::Vector{Float64}, x::Vector{Float64}, W::Matrix{Float64}, maxit::Int)
  n = A.n
  res = Vector{Float64}(undef, n); it = Ref{Int64}(0)
  rc = if M === nothing
    ccall((:mi_initcg, lib), Cint, (Ptr{Cvoid}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Int64, Int64, Float64, Ptr{Float64}, Int64, Ref{Int64}),
          A.h, b, x, W, size(W, 2), maxit, 1e-7, res, n, it)
  else
    ccall((:mi_initpcg, lib), Cint, (Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Int64, Int64, Float64, Ptr{Float64}, Int64, Ref{Int64}),
          A.h, M.h, b, x, W, size(W, 2), maxit, 1e-7, res, n, it)
  end
  check(rc)
  return x, Int(it[]), res[1:it[]]
end
initcg(A::MiOperator, b::Vector{Float64}, x::Vector{Float64}, W::Matrix{Float64}; maxit=0) = initsolve(A, nothing, b, x, W, maxit)
initpcg(A::MiOperator, b::Vector{Float64}, x::Vector{Float64}, M::MiOperator, W::Matrix{Float64}; maxit=0) = initsolve(A, M, b, x, W, maxit)

end # module
