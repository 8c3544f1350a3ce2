# DynamicExpressionsHIPExt.jl — the reference-side binding of libde_hip.so.
#
# This is the Julia half of the boundary (SURVEY.md §8b, INTEGRATION.md): it stays entirely on
# the host, keeps the reference's Node / OperatorEnum / EvalContext API, flattens a tree into
# the post-order tape of include/de_hip.h and `ccall`s the C ABI.  It mirrors, line for line,
# what the Python twin (dynamicexpressions.jl_amd/node.py + api.py) does; the Python twin is the
# one exercised by the test-suite because the build image has no Julia.
#
# Plug-in point: exactly the precedent of the Bumper whole-tree override
#   ext/DynamicExpressionsBumperExt.jl:11-49   _bumper_eval_tree_array(tree, cX, operators, ctx)
#   src/ExtensionInterface.jl:60-64            bumper_eval_tree_array stub
#   src/Evaluate.jl:300-302                    `if bumper isa Val{true} return bumper_eval_tree_array(...)`
# A maintainer adds the analogous stub `hip_eval_tree_array` (INTEGRATION.md §3) and this file
# as a package extension keyed on a weak dependency that provides the shared library.
module DynamicExpressionsHIPExt

using DynamicExpressions:
    AbstractExpressionNode, Node, GraphNode, OperatorEnum, EvalContext, get_child, get_children,
    count_constant_nodes, preserve_sharing
import DynamicExpressions.ExtensionInterfaceModule: is_extension_loaded

const LIBDE = get(ENV, "DE_HIP_LIB", "libde_hip")

# ---- de_hip.h mirrors ---------------------------------------------------------------------
struct TapeNode            # de_tape_node_t
    degree::UInt8
    op::UInt8
    arg::UInt16
end
const DE_OK = Cint(0)
const DE_ERR_UNSUPPORTED_OP = Cint(3)
const DE_LEAF_CONST, DE_LEAF_FEATURE, DE_LEAF_PARAM, DE_LEAF_SHARED = UInt8(0), UInt8(1), UInt8(2), UInt8(3)
const DE_OP_SHARE = UInt8(0xFE)   # include/de_opcodes.h: "the subtree just emitted is shared subtree `arg`"
const DE_F32, DE_F64 = Cint(0), Cint(1)
const DE_OPT_EARLY_EXIT, DE_OPT_FUSE_DEG1, DE_OPT_FUSE_DEG2, DE_OPT_BUMPER_CHECKS, DE_OPT_TURBO, DE_OPT_FULL_EVAL, DE_OPT_FORWARD_GRAD, DE_OPT_REVERSE_GRAD =
    UInt32(1), UInt32(2), UInt32(4), UInt32(8), UInt32(16), UInt32(32), UInt32(64), UInt32(128)
# The ABI this file was written for (include/de_hip.h lists what changed between versions).  Version 2: the rows / gradients of a tree
# with `complete == false` are NOT evaluated to the end (the reference's early exit, src/Evaluate.jl:26-32): with the host arrays this
# shim passes, the library NaN-fills them; `full_eval=true` (DE_OPT_FULL_EVAL) evaluates every tree on every sample instead.
# Version 3: fused loss gradients run forward duals (the reference's flag semantics) unless `reverse_grad=true` (DE_OPT_REVERSE_GRAD).
const DE_HIP_ABI_VERSION = Cint(3)
function __init__()
    v = try
        ccall((:de_abi_version, LIBDE), Cint, ())
    catch
        return nothing   # library not present: every entry point fails loudly when it is used
    end
    v == DE_HIP_ABI_VERSION || error("libde_hip has ABI version $(v), DynamicExpressionsHIPExt was written for $(DE_HIP_ABI_VERSION)")
    return nothing
end
const OPERATOR_LIMIT_BEFORE_SLOWDOWN = 15   # src/Evaluate.jl:14
dtype_code(::Type{Float32}) = DE_F32
dtype_code(::Type{Float64}) = DE_F64

struct UnsupportedOperator <: Exception
    f::Any
    degree::Int
end

# ---- OperatorEnum -> opcode table: by function IDENTITY ----------------------------------------
# An OperatorEnum holds arbitrary Julia functions (src/OperatorEnum.jl:14-49); the device knows a closed set of
# opcodes (include/de_opcodes.h).  A function is mapped onto an opcode only when it IS one of the functions below
# (`===`): Base's own `cos`, `+`, `max` ... .  A user function that merely shares a name with one of them (`cos(x) =
# my_cos(x)`, an anonymous function, a method-extended copy in another module) is NOT the same object and has no opcode:
# `opcode_table` throws `UnsupportedOperator` and the caller keeps the reference CPU path — never wrong values with
# `complete = true`.
const BASE_OPCODES = let t = IdDict{Any,Tuple{Vararg{Pair{Int,String}}}}()
    reg(f, pairs...) = (t[f] = pairs)
    for (f, n) in ((sin, "sin"), (cos, "cos"), (tan, "tan"), (exp, "exp"), (exp2, "exp2"), (log, "log"), (log2, "log2"),
                   (log10, "log10"), (log1p, "log1p"), (sqrt, "sqrt"), (cbrt, "cbrt"), (abs, "abs"), (sinh, "sinh"),
                   (cosh, "cosh"), (tanh, "tanh"), (asin, "asin"), (acos, "acos"), (atan, "atan"), (asinh, "asinh"),
                   (acosh, "acosh"), (atanh, "atanh"), (round, "round"), (floor, "floor"), (ceil, "ceil"), (sign, "sign"),
                   (inv, "inv"), (abs2, "square"))
        reg(f, 1 => n)
    end
    reg(-, 1 => "neg", 2 => "-")                       # unary minus and subtraction are the same Julia function
    reg(+, 2 => "+", 3 => "+")
    reg(max, 2 => "max", 3 => "max")
    for (f, n) in ((*, "*"), (/, "/"), (^, "^"), (min, "min"), (mod, "mod"), (rem, "rem"))
        reg(f, 2 => n)
    end
    reg(fma, 3 => "fma")
    reg(clamp, 3 => "clamp")
    t
end
# Functions that are NOT in Base — the helpers the reference's own tests define (`safe_log`, `relu`, `square`, `cube`,
# `custom_cos`, `pow_abs2`, `greater`, `sub` ..., test/test_params.jl:7-29) or SpecialFunctions.gamma — have opcodes too,
# but only the user can vouch that THEIR function of that name has those semantics (and that gradient):
#     register_hip_opcode(my_safe_log, :safe_log)
# The registration probes the function on a few points against the opcode's definition and refuses a mismatch.
const USER_OPCODES = IdDict{Any,Tuple{Int,String}}()
const USER_OPCODES_LOCK = ReentrantLock()
const PROBES = (0.37, 1.0, 2.5, -0.6, -2.25)
const OPCODE_SEMANTICS = Dict{Symbol,Function}(
    :safe_log => x -> x > 0 ? log(x) : NaN, :safe_log2 => x -> x > 0 ? log2(x) : NaN, :safe_log10 => x -> x > 0 ? log10(x) : NaN,
    :safe_log1p => x -> x > -1 ? log1p(x) : NaN, :safe_sqrt => x -> x >= 0 ? sqrt(x) : NaN, :safe_acosh => x -> x >= 1 ? acosh(x) : NaN,
    :relu => x -> x < 0 ? zero(x) : x, :square => x -> x * x, :cube => x -> x * x * x, :neg => x -> -x, :custom_cos => x -> cos(x)^2,
    :greater => (x, y) -> x > y ? 1.0 : 0.0, :sub => (x, y) -> x - y, :pow_abs2 => (x, y) -> exp(y * log(abs(x))),
    :add => (x, y) -> x + y, :mult => (x, y) -> x * y, :div => (x, y) -> x / y, :pow => (x, y) -> x^y,
)
"""
    register_hip_opcode(f, name::Symbol)

Declare that the user function `f` computes the device operator `name` (one of `keys(OPCODE_SEMANTICS)`, or `:gamma`).
`f` is probed against the operator's definition on a few points (NaN pattern included); a mismatch throws.  Registration
is by identity: other functions of the same name stay on the CPU path.
"""
function register_hip_opcode(f, name::Symbol)
    degree = first(methods(f)).nargs - 1
    code = ccall((:de_opcode_by_name, LIBDE), Cint, (Cstring, Cint), String(name), degree)
    code < 0 && throw(ArgumentError("no device operator $(name) of degree $(degree)"))
    if haskey(OPCODE_SEMANTICS, name)
        g = OPCODE_SEMANTICS[name]
        for x in PROBES, y in (degree == 2 ? PROBES : (0.0,))
            a = try degree == 1 ? f(x) : f(x, y) catch; NaN end     # DomainError == NaN on the device
            b = degree == 1 ? g(x) : g(x, y)
            (isnan(a) && isnan(b)) || isapprox(a, b; rtol=1e-12) ||
                throw(ArgumentError("$(f) is not the operator $(name): f($(x)$(degree == 2 ? ", $(y)" : "")) = $(a), expected $(b)"))
        end
    end
    lock(USER_OPCODES_LOCK) do
        USER_OPCODES[f] = (degree, String(name))
    end
    return f
end

function opcode_of(f, d::Int)
    name = nothing
    if haskey(BASE_OPCODES, f)
        for (deg, n) in BASE_OPCODES[f]
            deg == d && (name = n)
        end
    else
        entry = lock(() -> get(USER_OPCODES, f, nothing), USER_OPCODES_LOCK)
        entry !== nothing && entry[1] == d && (name = entry[2])
    end
    name === nothing && throw(UnsupportedOperator(f, d))
    code = ccall((:de_opcode_by_name, LIBDE), Cint, (Cstring, Cint), name, d)
    code < 0 && throw(UnsupportedOperator(f, d))
    return UInt8(code)
end

"""Opcode of every operator of `operators`, per degree; throws `UnsupportedOperator` for a
function that is neither one of Base's (by identity) nor registered (the caller then keeps the CPU path)."""
function opcode_table(operators::OperatorEnum)
    return ntuple(d -> map(f -> opcode_of(f, d), operators.ops[d]), length(operators.ops))
end

"""EvalContext knobs that change RESULTS -> de_options bits (src/Evaluate.jl:156-181,496,607)."""
function option_bits(operators::OperatorEnum, ctx::EvalContext; full_eval::Bool=false, forward_grad::Bool=false, reverse_grad::Bool=false)
    nops(d) = d <= length(operators.ops) ? length(operators.ops[d]) : 0
    fused = ctx.use_fused isa Val{true}
    bits = UInt32(0)
    ctx.early_exit isa Val{true} && (bits |= DE_OPT_EARLY_EXIT)
    fused && nops(1) <= OPERATOR_LIMIT_BEFORE_SLOWDOWN && (bits |= DE_OPT_FUSE_DEG1)
    fused && nops(2) <= OPERATOR_LIMIT_BEFORE_SLOWDOWN && (bits |= DE_OPT_FUSE_DEG2)
    ctx.bumper isa Val{true} && (bits |= DE_OPT_BUMPER_CHECKS)
    ctx.turbo isa Val{true} && (bits |= DE_OPT_TURBO)   # the LoopVectorization knob = the relaxed-accuracy device operators
    full_eval && (bits |= DE_OPT_FULL_EVAL)             # no early exit at tree granularity: rows of incomplete trees are fully evaluated
    forward_grad && (bits |= DE_OPT_FORWARD_GRAD)       # ABI 2's spelling of what is the default since ABI 3: fused loss gradients by forward duals (the reference's flag semantics exactly)
    reverse_grad && (bits |= DE_OPT_REVERSE_GRAD)       # permission for reverse accumulation (faster from 8 gradient rows per tree on; `ok` may differ in ~0.03 % of Float32 cases)
    return bits
end

# ---- flattening: post-order tape + constant pool (depth-first, left to right) ---------------
# (`cbase` = length(consts) when the tree starts: constant slots are numbered per tree, the pool of a population is shared)
function flatten!(
    nodes::Vector{TapeNode}, consts::Vector{T}, tree::AbstractExpressionNode{T}, optable, cbase::Int=0
) where {T}
    if tree.degree == 0
        if tree.constant
            push!(consts, tree.val)
            push!(nodes, TapeNode(0x00, DE_LEAF_CONST, UInt16(length(consts) - 1 - cbase)))
        elseif hasproperty(tree, :is_parameter) && tree.is_parameter   # ParametricNode
            push!(nodes, TapeNode(0x00, DE_LEAF_PARAM, UInt16(tree.parameter - 1)))
        else
            push!(nodes, TapeNode(0x00, DE_LEAF_FEATURE, UInt16(tree.feature - 1)))
        end
    else
        d = Int(tree.degree)
        for c in get_children(tree, d)
            flatten!(nodes, consts, c, optable, cbase)
        end
        push!(nodes, TapeNode(UInt8(d), optable[d][tree.op], 0x0000))
    end
    return nothing
end

# ---- GraphNode (src/Node.jl:138-166): the CSE tape of de_program_create_cse -------------------------------------
# `flatten!` above walks a GraphNode like the reference's evaluator does — a shared node once per parent — and gives the
# EXPANDED tape with one constant slot per occurrence.  `flatten_cse!` restates the tree with every shared, non-constant
# operator subtree once: DE_OP_SHARE after its first occurrence, DE_LEAF_SHARED afterwards; constant leaves keep the slot
# of their first occurrence in the expanded numbering.  Returns the number of shared subtrees (0: pass an empty range).
# `occurrence_map` gives, per expanded slot, the index of the unique constant (get_scalar_constants order): the shim keeps
# occurrence slots equal (set_population_constants!) and sums their gradient rows.
function flatten_cse!(nodes::Vector{TapeNode}, tree::AbstractExpressionNode{T}, optable) where {T}
    isconst = IdDict{Any,Bool}()
    nparents = IdDict{Any,Int}()
    # every node OBJECT once: nparents[c] = the number of (parent object, child slot) pairs that hold c — more than one
    # distinct parent, or the same parent twice, is what makes a subtree shared in its own right (a node reached only
    # through an already-shared ancestor has one).  Mirrors `multi_use` of the Python twin (node.py flatten_graph).
    function survey(n)
        haskey(isconst, n) && return nothing
        if n.degree == 0
            isconst[n] = n.constant
        else
            cs = get_children(n, Int(n.degree))
            for c in cs
                nparents[c] = get(nparents, c, 0) + 1
                survey(c)
            end
            isconst[n] = all(c -> isconst[c], cs)
        end
        return nothing
    end
    survey(tree)
    defined = IdDict{Any,Int}()
    slot = Ref(0)
    skip(n) = n.degree == 0 ? (n.constant && (slot[] += 1)) : foreach(skip, get_children(n, Int(n.degree)))
    function emit(n, under_ternary::Bool)
        if haskey(defined, n)
            push!(nodes, TapeNode(0x00, DE_LEAF_SHARED, UInt16(defined[n])))
            skip(n)
            return
        end
        if n.degree == 0
            if n.constant
                push!(nodes, TapeNode(0x00, DE_LEAF_CONST, UInt16(slot[]))); slot[] += 1
            else
                push!(nodes, TapeNode(0x00, DE_LEAF_FEATURE, UInt16(n.feature - 1)))
            end
            return
        end
        d = Int(n.degree)
        foreach(c -> emit(c, d == 3), get_children(n, d))
        push!(nodes, TapeNode(UInt8(d), optable[d][n.op], 0x0000))
        if get(nparents, n, 0) > 1 && !isconst[n] && !under_ternary && n !== tree && length(defined) < 12
            defined[n] = length(defined)
            push!(nodes, TapeNode(0x01, DE_OP_SHARE, UInt16(defined[n])))
        end
    end
    emit(tree, false)
    return length(defined)
end
function occurrence_map(tree::AbstractExpressionNode)
    uniq = IdDict{Any,Int}()
    occ = Int[]
    walk(n) = n.degree == 0 ? (n.constant && push!(occ, get!(uniq, n, length(uniq) + 1))) : foreach(walk, get_children(n, Int(n.degree)))
    walk(tree)
    return occ
end

# ---- context: one de_ctx_t per Julia TASK ---------------------------------------------------
# A de_ctx_t is not thread-safe (INTEGRATION.md §4): every call on it — and the destruction of programs that live in it,
# which finalizers run from arbitrary threads — takes the context's lock.  Programs hold a strong reference to their
# context, so a context is finalized only after all of its programs; destroying a program whose context is already gone
# (process exit, finalizer order unspecified) is a no-op.
mutable struct HIPContext
    handle::Ptr{Cvoid}
    lock::ReentrantLock
end
# Finalizers must not block on a lock a running task may hold: when the lock is taken, the SAME cleanup is registered again
# (the object survives this collection and the next one retries) — re-arming with `identity` would leak the handle.
function finalize_context(c::HIPContext)
    if trylock(c.lock)
        try
            c.handle != C_NULL && ccall((:de_ctx_destroy, LIBDE), Cint, (Ptr{Cvoid},), c.handle)
            c.handle = C_NULL
        finally
            unlock(c.lock)
        end
    else
        finalizer(finalize_context, c)
    end
    return nothing
end
# (Several contexts on DIFFERENT devices may live in one process: the library caches its handler tables per device, ABI version 2.)
function HIPContext(device::Integer=0)
    h = Ref{Ptr{Cvoid}}(C_NULL)
    rc = ccall((:de_ctx_create, LIBDE), Cint, (Cint, Ptr{Cvoid}, Ref{Ptr{Cvoid}}), device, C_NULL, h)
    rc == DE_OK || error("de_ctx_create failed: ", unsafe_string(ccall((:de_status_string, LIBDE), Cstring, (Cint,), rc)))
    ctx = HIPContext(h[], ReentrantLock())
    finalizer(finalize_context, ctx)
    return ctx
end
"""The calling task's context (task-local storage: tasks migrate between threads, `Threads.threadid()` is not a key)."""
task_context() = get!(() -> HIPContext(0), task_local_storage(), :DynamicExpressionsHIPExt_context)::HIPContext
"""Free what the context retains between populations (the parked host vectors and recycled device buffers of destroyed populations —
up to 512 MB of host and 256 MB of device memory per context, i.e. per task — and the staging scratch of host arrays): `de_ctx_trim`.
Call it from a task that goes idle; the next population simply allocates afresh."""
function trim_context!(ctx::HIPContext=task_context())
    with_ctx(ctx) do h
        check(ctx, ccall((:de_ctx_trim, LIBDE), Cint, (Ptr{Cvoid},), h))
    end
    return nothing
end

function check(ctx::HIPContext, rc::Cint)
    rc == DE_OK && return true
    rc == DE_ERR_UNSUPPORTED_OP && throw(UnsupportedOperator(nothing, 0))
    error(unsafe_string(ccall((:de_last_error, LIBDE), Cstring, (Ptr{Cvoid},), ctx.handle)))
end
"""Run `f(handle)` holding the context's lock."""
with_ctx(f, ctx::HIPContext) = lock(() -> (ctx.handle == C_NULL && error("HIP context was destroyed"); f(ctx.handle)), ctx.lock)

# ---- the whole-tree override (signature of _bumper_eval_tree_array) --------------------------
"""
    _hip_eval_tree_array(tree, cX, operators, eval_context) -> (result::Vector{T}, ok::Bool)

Drop-in for `eval_tree_array`'s body: same `(output, complete)` tuple, `cX` is the reference's
`[n_features, n_rows]` column-major matrix and is passed zero-copy.  Falls back to the CPU path
(returns `nothing`) when an operator has no device opcode.
"""
function _hip_eval_tree_array(
    tree::AbstractExpressionNode{T}, cX::AbstractMatrix{T}, operators::OperatorEnum,
    eval_context::EvalContext; full_eval::Bool=false,
) where {T<:Union{Float32,Float64}}
    optable = try
        opcode_table(operators)
    catch e
        e isa UnsupportedOperator || rethrow()
        return nothing                       # caller continues on the reference CPU path
    end
    nodes, consts = TapeNode[], T[]
    flatten!(nodes, consts, tree, optable)
    F, N = size(cX)
    X = cX isa Matrix{T} ? cX : Matrix{T}(cX)  # strided views are copied; Matrix is zero-copy
    out = Vector{T}(undef, N)
    ok = Ref{UInt8}(0)
    ctx = task_context()
    with_ctx(ctx) do h
        check(ctx, GC.@preserve nodes consts X out ccall(
            (:de_eval_tree_array, LIBDE), Cint,
            (Ptr{Cvoid}, Cint, Ptr{TapeNode}, Int64, Ptr{Cvoid}, Int64, Ptr{Cvoid}, Int32, Int64, UInt32,
             Ptr{Cvoid}, Ref{UInt8}),
            h, dtype_code(T), nodes, length(nodes), consts, length(consts), X, F, N,
            option_bits(operators, eval_context; full_eval), out, ok))
    end
    return (out, ok[] != 0x00)   # ok == false: `out` is all NaN unless full_eval (only the flag is contractual, SURVEY.md §8a)
end

# ---- population form: lower many trees once, evaluate in one launch -------------------------
# GraphNode populations: the library keeps one constant slot per OCCURRENCE of a constant leaf (the expanded tape); to the user a
# shared constant is ONE constant (count_constant_nodes with f_on_shared, src/NodeUtils.jl:43-51) with ONE gradient row (the shared
# NodeIndex entry, :184-201).  `occ[t]` (occurrence_map) maps tree t's slots to its unique constants — `nothing` when no constant
# of the tree is shared —, `n_slots[t]` / `n_consts[t]` count both: set_population_constants! fans values out, the gradient
# entry points sum the occurrence rows (combine_rows) — the mirror of api.py's `_occ` / `_combine_rows`.
mutable struct HIPPopulation{T}
    ctx::HIPContext
    handle::Ptr{Cvoid}
    n_trees::Int
    n_features::Int
    occ::Vector{Union{Nothing,Vector{Int}}}
    n_slots::Vector{Int}
    n_consts::Vector{Int}
end
function finalize_population(p::HIPPopulation)
    c = p.ctx
    if trylock(c.lock)
        try
            # de_program_destroy synchronises the context's stream: skip it when the context is already gone
            c.handle != C_NULL && p.handle != C_NULL && ccall((:de_program_destroy, LIBDE), Cint, (Ptr{Cvoid},), p.handle)
            p.handle = C_NULL
        finally
            unlock(c.lock)
        end
    else
        finalizer(finalize_population, p)   # lock taken: the same cleanup again at the next collection (see finalize_context)
    end
    return nothing
end
"""Run `f(context handle, program handle)` holding the lock of the population's context (every entry point does: a `de_ctx_t` is
not thread-safe, and `finalize_population` of ANOTHER population of the same context may run on a GC thread at any time)."""
function with_pop(f, pop::HIPPopulation)
    return with_ctx(pop.ctx) do h
        pop.handle == C_NULL && error("HIP population was destroyed")
        f(h, pop.handle)
    end
end
function HIPPopulation(
    trees::AbstractVector{<:AbstractExpressionNode{T}}, operators::OperatorEnum, n_features::Integer;
    eval_context::EvalContext=EvalContext(), n_params::Integer=0, full_eval::Bool=false, forward_grad::Bool=false,
    reverse_grad::Bool=false,
) where {T<:Union{Float32,Float64}}
    optable = opcode_table(operators)
    nodes, consts, cse = TapeNode[], T[], TapeNode[]
    node_off, const_off, cse_off = Int64[0], Int64[0], Int64[0]
    occ = Union{Nothing,Vector{Int}}[]
    n_slots, n_consts = Int[], Int[]
    for t in trees
        c0 = length(consts)
        flatten!(nodes, consts, t, optable, c0)   # constant slots are numbered PER TREE (de_tape_node_t.arg)
        push!(node_off, length(nodes)); push!(const_off, length(consts))
        o = nothing
        if preserve_sharing(typeof(t))            # GraphNode: a second, CSE tape for the eval program
            mark = length(cse)
            flatten_cse!(cse, t, optable) == 0 && resize!(cse, mark)
            om = occurrence_map(t)
            (!isempty(om) && maximum(om) < length(om)) && (o = om)   # some constant leaf occurs more than once
        end
        push!(cse_off, length(cse))
        push!(occ, o)
        push!(n_slots, length(consts) - c0)
        push!(n_consts, o === nothing ? length(consts) - c0 : maximum(o))
    end
    ctx = task_context()
    h = Ref{Ptr{Cvoid}}(C_NULL)
    with_ctx(ctx) do hc
        check(ctx, GC.@preserve nodes consts node_off const_off cse cse_off ccall(
            (:de_program_create_cse, LIBDE), Cint,
            (Ptr{Cvoid}, Cint, Ptr{TapeNode}, Ptr{Int64}, Ptr{TapeNode}, Ptr{Int64}, Int64, Ptr{Cvoid}, Ptr{Int64}, Int32, Int32,
             UInt32, Ref{Ptr{Cvoid}}),
            hc, dtype_code(T), nodes, node_off, isempty(cse) ? C_NULL : pointer(cse), cse_off, length(trees), consts,
            const_off, n_features, n_params, option_bits(operators, eval_context; full_eval, forward_grad, reverse_grad), h))
    end
    pop = HIPPopulation{T}(ctx, h[], length(trees), n_features, occ, n_slots, n_consts)
    finalizer(finalize_population, pop)
    return pop
end
"""Gradient rows (or entries) of tree `t` in the library's per-occurrence layout -> the reference's layout: the rows of a shared
constant are summed.  `g` is a vector of entries or a matrix with one ROW per gradient component; the `lead` rows in front of
the constants (features in `:both` mode) pass through.  `mode == 0` (`variable=true`) has no constant rows."""
function combine_rows(pop::HIPPopulation, t::Integer, g::AbstractVecOrMat, mode::Integer)
    o = pop.occ[t]
    (o === nothing || mode == 0) && return g
    lead = size(g, 1) - length(o)
    nu = pop.n_consts[t]
    out = g isa AbstractVector ? zeros(eltype(g), lead + nu) : zeros(eltype(g), lead + nu, size(g, 2))
    if g isa AbstractVector
        out[1:lead] .= g[1:lead]
        for (k, u) in enumerate(o)
            out[lead + u] += g[lead + k]
        end
    else
        out[1:lead, :] .= g[1:lead, :]
        for (k, u) in enumerate(o)
            out[lead + u, :] .+= g[lead + k, :]
        end
    end
    return out
end

"""`(out::Matrix{T}(N × n_trees), ok::Vector{Bool})`; column t is `eval_tree_array(trees[t], X)`.  Columns with `ok[t] == false`
are NaN (the tree left the kernel at its first flagged workgroup, like the reference's early return) unless the population
was made with `full_eval=true`."""
function eval_population(pop::HIPPopulation{T}, X::Matrix{T}) where {T}
    F, N = size(X)
    @assert F >= pop.n_features
    out = Matrix{T}(undef, N, pop.n_trees)          # row t of the C layout = column t here
    ok = Vector{UInt8}(undef, pop.n_trees)
    with_pop(pop) do hc, hp
        check(pop.ctx, GC.@preserve X out ok ccall(
            (:de_eval, LIBDE), Cint,
            (Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Int64, Int64, Ptr{Cvoid}, Ptr{Cvoid}, Int64, Ptr{UInt8}),
            hc, hp, X, N, F, C_NULL, out, N, ok))
    end
    return out, ok .!= 0x00
end

"""
    population_sum_certificate(pop, X) -> (ok, certified, max_abs)

`de_eval_sum_certificate`: the reference tests `isfinite(sum(x))` (src/ValueInterface.jl:9), the kernels every element; `certified[t]`
says the two provably agree for tree t (some element non-finite, or `N * max|tested value|` below `floatmax(T)`).  Only the trees
with `certified[t] == false` can carry a flag that differs from `eval_tree_array`'s — re-derive those on the CPU when the bit matters.
"""
function population_sum_certificate(pop::HIPPopulation{T}, X::Matrix{T}) where {T}
    F, N = size(X)
    @assert F >= pop.n_features
    ok = Vector{UInt8}(undef, pop.n_trees)
    cert = Vector{UInt8}(undef, pop.n_trees)
    mx = Vector{Float64}(undef, pop.n_trees)
    with_pop(pop) do hc, hp
        check(pop.ctx, GC.@preserve X ok cert mx ccall(
            (:de_eval_sum_certificate, LIBDE), Cint,
            (Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Int64, Int64, Ptr{Cvoid}, Ptr{UInt8}, Ptr{UInt8}, Ptr{Float64}),
            hc, hp, X, N, F, C_NULL, ok, cert, mx))
    end
    return ok .!= 0x00, cert .!= 0x00, mx
end

"""
    eval_population_strict(pop, X) -> (out, ok, uncertified::Vector{Int})

`eval_population` followed by the certificate pass: `uncertified` lists (1-based) the trees whose element-wise flag is NOT provably the
reference's `isfinite(sum(x))` flag — for every other tree `ok[t]` IS `eval_tree_array`'s `complete`.  A caller who needs the
reference's bit re-derives only those trees on the CPU (`strict_flags` of the Python twin).
"""
function eval_population_strict(pop::HIPPopulation{T}, X::Matrix{T}) where {T}
    out, ok = eval_population(pop, X)
    ok2, cert, _ = population_sum_certificate(pop, X)
    ok2 == ok || error("strict flags: the certificate pass and the evaluation disagree on a flag")
    return out, ok, findall(!, cert)
end

struct ParamArgs            # de_param_args_t
    params::Ptr{Cvoid}
    ld_params::Int64
    n_classes::Int64
    classes::Ptr{Cvoid}
    classes_is_i64::Int32
    class_base::Int32
end

"""
    eval_population(pop, X, parameters, classes) -> (out, ok)

ParametricExpression form (src/ParametricExpression.jl:371-390): `parameters` is the `P × C` matrix,
`classes::Vector{Int}` the 1-based class of every column of `X`; the `P × N` gather the reference
materialises is done on the fly in the kernel.  The range assertions of the reference (:378-379) are
repeated here because the library trusts the ids.
"""
function eval_population(
    pop::HIPPopulation{T}, X::Matrix{T}, parameters::Matrix{T}, classes::Vector{Int}
) where {T}
    F, N = size(X)
    @assert length(classes) == N
    @assert isempty(classes) || (minimum(classes) >= 1 && maximum(classes) <= size(parameters, 2))
    out = Matrix{T}(undef, N, pop.n_trees)
    ok = Vector{UInt8}(undef, pop.n_trees)
    with_pop(pop) do hc, hp
        check(pop.ctx, GC.@preserve X parameters classes out ok begin
            pa = Ref(ParamArgs(pointer(parameters), size(parameters, 1), size(parameters, 2), pointer(classes), 1, 1))
            ccall((:de_eval, LIBDE), Cint,
                (Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Int64, Int64, Ref{ParamArgs}, Ptr{Cvoid}, Int64, Ptr{UInt8}),
                hc, hp, X, N, F, pa, out, N, ok)
        end)
    end
    return out, ok .!= 0x00
end

"""
    eval_population_loss(pop, X, y; weights=nothing, loss=:L2) -> (loss::Vector{T}, ok)

`sum(abs2, trees[t](X) .- y)` (test/test_optim.jl:95,99) for every tree, reduced on the device:
the `N × n_trees` output matrix is never written.  `loss[t]` is NaN where `ok[t]` is false.
"""
function eval_population_loss(
    pop::HIPPopulation{T}, X::Matrix{T}, y::Vector{T}; weights::Union{Nothing,Vector{T}}=nothing,
    loss::Symbol=:L2,
) where {T}
    F, N = size(X)
    @assert F >= pop.n_features && length(y) == N
    out = Vector{T}(undef, pop.n_trees)
    ok = Vector{UInt8}(undef, pop.n_trees)
    with_pop(pop) do hc, hp
        check(pop.ctx, GC.@preserve X y weights out ok ccall(
            (:de_eval_loss, LIBDE), Cint,
            (Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Int64, Int64, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Int32,
             Ptr{Cvoid}, Ptr{UInt8}),
            hc, hp, X, N, F, C_NULL, y, weights === nothing ? C_NULL : pointer(weights), loss === :L1 ? 1 : 0, out, ok))
    end
    return out, ok .!= 0x00
end

"""
    set_population_constants!(pop, constants::Vector{T})

New constants for the trees of `pop`, same shapes: `constants` = the trees' `get_scalar_constants` vectors
(src/NodeUtils.jl:99-143: depth-first leaf order; a GraphNode's shared constant ONCE) back to back.  Nothing is
re-flattened or re-lowered — the immediates are patched inside the device programs (`de_program_set_consts`), which is
what an optimiser loop (`ext/DynamicExpressionsOptimExt.jl:182-224`) calls between `eval_population_loss_grad`
evaluations.  For a GraphNode population the values are fanned out to the occurrence slots of the library.
"""
function set_population_constants!(pop::HIPPopulation{T}, constants::Vector{T}) where {T}
    length(constants) == sum(pop.n_consts) || throw(ArgumentError("wrong number of constants"))
    vals = constants
    if any(o -> o !== nothing, pop.occ)          # one value per unique constant -> one per occurrence slot
        vals = Vector{T}(undef, sum(pop.n_slots))
        at, to = 0, 0
        for t in 1:pop.n_trees
            o = pop.occ[t]
            for k in 1:pop.n_slots[t]
                vals[to + k] = constants[at + (o === nothing ? k : o[k])]
            end
            at += pop.n_consts[t]; to += pop.n_slots[t]
        end
    end
    with_pop(pop) do hc, hp
        check(pop.ctx, GC.@preserve vals ccall((:de_program_set_consts, LIBDE), Cint, (Ptr{Cvoid}, Ptr{Cvoid}), hp, vals))
    end
    return pop
end

grad_mode(variable) = variable isa Val{true} || variable === true ? Cint(0) : variable isa Val{:both} ? Cint(2) : Cint(1)
"""Gradient widths of the trees in the LIBRARY's layout (one row per occurrence slot) and their packed offsets."""
function grad_widths(hp::Ptr{Cvoid}, n_trees::Int, mode::Cint)
    ng = [ccall((:de_program_n_grad, LIBDE), Int64, (Ptr{Cvoid}, Int64, Cint), hp, t - 1, mode) for t in 1:n_trees]
    return ng, Int64[0; cumsum(ng)]
end

"""
    eval_population_loss_grad(pop, X, y; weights=nothing, loss=:L2, variable=Val(false))
        -> (loss::Vector{T}, dloss::Vector{Vector{T}}, ok)

The optimiser callback of test/test_optim.jl:42-51 for a whole population in one launch:
`dloss[t][i] = sum_j 2 (ŷ_j - y_j) * dŷ_dconstants[i, j]` without the `n_grad × N` Jacobian.
`loss=:pullback` treats `y` as the cotangent `dY` of the ChainRules pullback
(src/ChainRules.jl:56-77) and returns its `dtree` gradient.
"""
function eval_population_loss_grad(
    pop::HIPPopulation{T}, X::Matrix{T}, y::Vector{T}; weights::Union{Nothing,Vector{T}}=nothing,
    loss::Symbol=:L2, variable=Val(false),
) where {T}
    mode = grad_mode(variable)
    F, N = size(X)
    @assert F >= pop.n_features && length(y) == N
    lossv = Vector{T}(undef, pop.n_trees)
    ok = Vector{UInt8}(undef, pop.n_trees)
    kind = loss === :L1 ? 1 : loss === :pullback ? 2 : 0
    dl, offs = with_pop(pop) do hc, hp
        ng, offs = grad_widths(hp, pop.n_trees, mode)
        dl = Vector{T}(undef, max(offs[end], 1))
        check(pop.ctx, GC.@preserve X y weights lossv dl offs ok ccall(
            (:de_eval_loss_grad, LIBDE), Cint,
            (Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Int64, Int64, Ptr{Cvoid}, Cint, Ptr{Cvoid}, Ptr{Cvoid}, Int32,
             Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Int64}, Ptr{UInt8}),
            hc, hp, X, N, F, C_NULL, mode, y, weights === nothing ? C_NULL : pointer(weights), kind, lossv, dl, offs, ok))
        (dl, offs)
    end
    return lossv, [combine_rows(pop, t, dl[(offs[t] + 1):offs[t + 1]], mode) for t in 1:pop.n_trees], ok .!= 0x00
end

"""
    eval_population_loss_grad_by_class(pop, X, y, parameters, classes; weights=nothing, loss=:L2, variable=Val(:both))
        -> (loss, dloss::Vector{Vector{T}}, dparameters::Vector{Matrix{T}}, ok)

Parametric population: the fused loss gradient with the parameter rows reduced BY CLASS —
`dparameters[t]` is the gradient w.r.t. the `n_params × n_classes` parameter matrix, i.e. Zygote's
`grad.metadata._data.parameters` (test/test_parametric_expression.jl:326-372; the optimiser vector is
`vcat(constants, parameters[:])`, src/ParametricExpression.jl:260-265).  The library reduces class by
class over sample ranges, so the samples are ordered by class here (`sortperm`, stable); a search loop
that keeps its dataset ordered passes `grouped=true`.
"""
function eval_population_loss_grad_by_class(
    pop::HIPPopulation{T}, X::Matrix{T}, y::Vector{T}, parameters::Matrix{T}, classes::Vector{Int64};
    weights::Union{Nothing,Vector{T}}=nothing, loss::Symbol=:L2, variable=Val(:both), grouped::Bool=false,
) where {T}
    mode = variable isa Val{true} || variable === true ? Cint(0) : Cint(2)
    F, N = size(X)
    P, C = size(parameters)
    @assert F >= pop.n_features && length(y) == N && length(classes) == N
    @assert isempty(classes) || (minimum(classes) >= 1 && maximum(classes) <= C)   # src/ParametricExpression.jl:378-379 (the library trusts the ids)
    if !grouped
        perm = sortperm(classes; alg=MergeSort)
        X, y, classes = X[:, perm], y[perm], classes[perm]
        weights = weights === nothing ? nothing : weights[perm]
    end
    starts = zeros(Int64, C + 1)
    for c in classes
        starts[c + 1] += 1
    end
    cumsum!(starts, starts)
    lossv = Vector{T}(undef, pop.n_trees)
    dp = Array{T,3}(undef, P, C, pop.n_trees)
    ok = Vector{UInt8}(undef, pop.n_trees)
    kind = loss === :L1 ? 1 : loss === :pullback ? 2 : 0
    dl, offs = with_pop(pop) do hc, hp
        ng, offs = grad_widths(hp, pop.n_trees, mode)
        dl = Vector{T}(undef, max(offs[end], 1))
        check(pop.ctx, GC.@preserve X y weights parameters classes starts lossv dl dp offs ok begin
            pa = Ref(ParamArgs(pointer(parameters), P, C, pointer(classes), 1, 1))
            ccall((:de_eval_loss_grad_by_class, LIBDE), Cint,
                (Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Int64, Int64, Ref{ParamArgs}, Cint, Ptr{Cvoid}, Ptr{Cvoid}, Int32,
                 Ptr{Int64}, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Int64}, Ptr{Cvoid}, Ptr{UInt8}),
                hc, hp, X, N, F, pa, mode, y, weights === nothing ? C_NULL : pointer(weights), kind, starts, lossv, dl, offs,
                dp, ok)
        end)
        (dl, offs)
    end
    return lossv, [combine_rows(pop, t, dl[(offs[t] + 1):offs[t + 1]], mode) for t in 1:pop.n_trees],
           [dp[:, :, t] for t in 1:pop.n_trees], ok .!= 0x00
end

"""Forward-mode gradient of one tree: `(evaluation, gradient(n_grad × N), complete)` like
`eval_grad_tree_array(tree, cX, operators; variable)` (src/EvaluateDerivative.jl:193-228); a GraphNode's shared constant has
ONE row (the sum over its occurrences)."""
function _hip_eval_grad_tree_array(
    tree::AbstractExpressionNode{T}, cX::Matrix{T}, operators::OperatorEnum; variable=Val(false)
) where {T<:Union{Float32,Float64}}
    mode = grad_mode(variable)
    F, N = size(cX)
    pop = HIPPopulation([tree], operators, F)
    out = Vector{T}(undef, N)
    ok = Ref{UInt8}(0)
    grad = with_pop(pop) do hc, hp
        ng = ccall((:de_program_n_grad, LIBDE), Int64, (Ptr{Cvoid}, Int64, Cint), hp, 0, mode)
        grad = Matrix{T}(undef, ng, N)
        check(pop.ctx, GC.@preserve cX out grad ccall(
            (:de_eval_grad, LIBDE), Cint,
            (Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Int64, Int64, Ptr{Cvoid}, Cint, Ptr{Cvoid}, Int64, Ptr{Cvoid},
             Ptr{Int64}, Ref{UInt8}),
            hc, hp, cX, N, F, C_NULL, mode, out, N, grad, C_NULL, ok))
        grad
    end
    return (out, combine_rows(pop, 1, grad, mode), ok[] != 0x00)
end

"""Single-direction derivative of one tree: `(evaluation, derivative, complete)` like
`eval_diff_tree_array(tree, cX, operators, direction)` (src/EvaluateDerivative.jl:40-53; `direction` is the
1-based feature, `complete` is always true on this path as in the reference, :68-119)."""
function _hip_eval_diff_tree_array(
    tree::AbstractExpressionNode{T}, cX::Matrix{T}, operators::OperatorEnum, direction::Integer
) where {T<:Union{Float32,Float64}}
    F, N = size(cX)
    1 <= direction <= F || throw(ArgumentError("direction must be a feature of cX"))
    pop = HIPPopulation([tree], operators, F)
    out = Vector{T}(undef, N)
    dout = Vector{T}(undef, N)
    ok = Ref{UInt8}(0)
    with_pop(pop) do hc, hp
        check(pop.ctx, GC.@preserve cX out dout ccall(
            (:de_eval_diff, LIBDE), Cint,
            (Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Int64, Int64, Int32, Ptr{Cvoid}, Ptr{Cvoid}, Int64, Ref{UInt8}),
            hc, hp, cX, N, F, Int32(direction - 1), out, dout, N, ok))
    end
    return (out, dout, ok[] != 0x00)
end

"""
    eval_population_pullback_dX(pop, X, dY) -> (dX::Array{T,3}(n_features × N × n_trees), ok)

The `dX` of the ChainRules pullback of `eval_tree_array` (`EvalPullback`, src/ChainRules.jl:56-77): `dX[:, :, t] =
dX_dY .* reshape(dY, 1, :)`, all-NaN where `ok[t]` is false (:62-64).  The `dtree` half is
`eval_population_loss_grad(pop, X, dY; loss=:pullback, variable=Val(false))`.
"""
function eval_population_pullback_dX(pop::HIPPopulation{T}, X::Matrix{T}, dY::Vector{T}) where {T}
    F, N = size(X)
    @assert F >= pop.n_features && length(dY) == N
    dX = Array{T,3}(undef, pop.n_features, N, pop.n_trees)
    ok = Vector{UInt8}(undef, pop.n_trees)
    with_pop(pop) do hc, hp
        check(pop.ctx, GC.@preserve X dY dX ok ccall(
            (:de_eval_pullback_dX, LIBDE), Cint,
            (Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Int64, Int64, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Int64}, Ptr{UInt8}),
            hc, hp, X, N, F, C_NULL, dY, dX, C_NULL, ok))
    end
    return dX, ok .!= 0x00
end

# ---- multi-GPU: one Julia process per GPU (Distributed / MPI), RCCL through the library ----------------------------
"""
    HIPComm(ctx, rank, world, id)  /  hip_unique_id()

`de_dist_*` (include/de_hip.h "multi-GPU"): rank 0 calls `hip_unique_id()` and ships the 128 bytes to the other ranks
(`Distributed.remotecall`, MPI.Bcast ...); every rank builds `HIPComm(task_context(), rank, world, id)`, evaluates its
round-robin shard `trees[rank+1:world:end]`, and `gather_flags(comm, ok_local, n_trees)` returns the completion flags of
ALL trees in global order (one ncclAllGather over xGMI).  Outputs stay sharded.
"""
mutable struct HIPComm
    ctx::HIPContext
    handle::Ptr{Cvoid}
    rank::Int
    world::Int
end
function hip_unique_id()
    id = Vector{UInt8}(undef, 128)
    rc = ccall((:de_dist_unique_id, LIBDE), Cint, (Ptr{UInt8},), id)
    rc == DE_OK || error(unsafe_string(ccall((:de_dist_last_error, LIBDE), Cstring, (Ptr{Cvoid},), C_NULL)))
    return id
end
function HIPComm(ctx::HIPContext, rank::Integer, world::Integer, id::Vector{UInt8}=UInt8[])
    h = Ref{Ptr{Cvoid}}(C_NULL)
    rc = with_ctx(ctx) do hc
        GC.@preserve id ccall((:de_dist_init, LIBDE), Cint, (Ptr{Cvoid}, Cint, Cint, Ptr{UInt8}, Ref{Ptr{Cvoid}}),
                              hc, rank, world, world > 1 ? pointer(id) : C_NULL, h)
    end
    rc == DE_OK || error(unsafe_string(ccall((:de_dist_last_error, LIBDE), Cstring, (Ptr{Cvoid},), C_NULL)))
    return HIPComm(ctx, h[], rank, world)
end
function gather_flags(comm::HIPComm, ok_local::Vector{Bool}, n_trees::Integer)
    loc = UInt8.(ok_local)
    out = Vector{UInt8}(undef, n_trees)
    with_ctx(comm.ctx) do hc
        # the call only QUEUES its copies (host `loc` -> staging, gathered flags -> host `out`): both arrays must stay alive until
        # the stream has drained, so the synchronisation sits inside the same GC.@preserve (ADVICE r5)
        GC.@preserve loc out begin
            rc = ccall((:de_dist_gather_flags, LIBDE), Cint,
                (Ptr{Cvoid}, Ptr{UInt8}, Int64, Ptr{UInt8}), comm.handle, loc, n_trees, out)
            rc == DE_OK || error(unsafe_string(ccall((:de_dist_last_error, LIBDE), Cstring, (Ptr{Cvoid},), comm.handle)))
            check(comm.ctx, ccall((:de_ctx_synchronize, LIBDE), Cint, (Ptr{Cvoid},), hc))
        end
    end
    return out .!= 0x00
end
"""Bound every collective of `comm` (`de_dist_set_timeout`): the calls then wait for what they queued and fail with a message naming
the rank and the collective after `timeout_ms` instead of hanging on a peer that is down; 0 = asynchronous calls (the default)."""
function set_comm_timeout!(comm::HIPComm, timeout_ms::Integer)
    with_ctx(comm.ctx) do _
        rc = ccall((:de_dist_set_timeout, LIBDE), Cint, (Ptr{Cvoid}, Int64), comm.handle, timeout_ms)
        rc == DE_OK || error(unsafe_string(ccall((:de_dist_last_error, LIBDE), Cstring, (Ptr{Cvoid},), comm.handle)))
    end
    return nothing
end
close_comm(comm::HIPComm) = with_ctx(comm.ctx) do _
    comm.handle != C_NULL && ccall((:de_dist_destroy, LIBDE), Cint, (Ptr{Cvoid},), comm.handle)
    comm.handle = C_NULL
    nothing
end

is_extension_loaded(::Val{:HIP}) = true

end # module
