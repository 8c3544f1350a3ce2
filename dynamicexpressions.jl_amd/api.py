"""Host-side mirror of the reference's evaluation interface on top of the C ABI.

Same names, argument meaning and error behaviour as the reference for this path:

* ``eval_tree_array(tree, cX, operators; eval_context)``  -> ``(out, complete)``
  (src/Evaluate.jl:279-309)
* ``eval_grad_tree_array(tree, cX, operators; variable)`` -> ``(out, grad, complete)``
  (src/EvaluateDerivative.jl:193-228)
* ``eval_diff_tree_array(tree, cX, operators, direction)``-> ``(out, dout, complete)``
  (src/EvaluateDerivative.jl:40-53)
* ``ParametricExpression`` / ``eval_tree_array(ex, X, classes)``
  (src/ParametricExpression.jl:371-390)
* ``tree(X, operators)`` sugar with NaN-fill (src/EvaluationHelpers.jl:29-33) = ``Expression``.

plus the population form the MI355X kernels are built for: ``Population(trees, operators)``
lowers many trees once and evaluates them all in one launch.

All arithmetic happens in ``csrc/libde_hip.so`` (hand-written gfx950 kernels).  There is NO
CPU fallback: if the library or a GPU is missing, calls raise ``DeviceError``.
PyTorch is used only as the owner of device memory / streams.
"""
from __future__ import annotations

import ctypes as C
import os
from dataclasses import dataclass
from typing import Optional, Sequence, Tuple, Union

import numpy as np

from .node import Node, TAPE_DTYPE, count_constant_nodes, flatten_graph, flatten_population, max_feature, preserve_sharing
from .operators import OperatorEnum

_HERE = os.path.dirname(os.path.abspath(__file__))
# DE_HIP_LIB (the variable the Julia shim reads too) selects another build of the same library
LIB_PATH = os.environ.get("DE_HIP_LIB") or os.path.join(_HERE, "csrc", "libde_hip.so")

DE_F32, DE_F64 = 0, 1
GRAD_VARIABLE, GRAD_CONSTANT, GRAD_BOTH = 0, 1, 2
ABI_VERSION = 3  # DE_HIP_ABI_VERSION of include/de_hip.h this module was written for
OPT_EARLY_EXIT, OPT_FUSE_DEG1, OPT_FUSE_DEG2, OPT_BUMPER_CHECKS, OPT_TURBO, OPT_FULL_EVAL, OPT_FORWARD_GRAD, OPT_REVERSE_GRAD = 1, 2, 4, 8, 16, 32, 64, 128

EXPORTS = [
    "de_abi_version", "de_opcode_table_version", "de_opcode_by_name", "de_opcode_name",
    "de_opcode_degree", "de_status_string", "de_ctx_create", "de_ctx_destroy", "de_ctx_set_stream",
    "de_ctx_synchronize", "de_ctx_declare_dataset", "de_ctx_stream", "de_last_error", "de_program_create", "de_program_create_cse",
    "de_program_set_consts", "de_program_destroy", "de_program_n_trees", "de_program_n_nodes",
    "de_program_n_grad", "de_program_dump", "de_program_verify", "de_program_stream_hash", "de_host_pool_selftest", "de_lower_tape", "de_lower_tape_stage", "de_eval", "de_eval_grad", "de_eval_diff", "de_eval_loss", "de_eval_loss_grad", "de_eval_loss_grad_by_class",
    "de_eval_pullback_dX", "de_eval_tree_array", "de_eval_plan", "de_prio_tiles_wanted", "de_program_last_live_trees", "de_dist_unique_id", "de_dist_init", "de_dist_destroy", "de_dist_shard_size", "de_dist_world_size",
    "de_dist_broadcast", "de_dist_gather_flags", "de_dist_last_error", "de_ctx_last_kernel_ms", "de_ctx_last_kernel_name",
    "de_ctx_device", "de_ctx_timing_ring", "de_ctx_timing_read", "de_dist_reorder_selftest", "de_eval_sum_certificate",
    "de_dist_set_timeout", "de_ctx_trim",
]


class DeviceError(RuntimeError):
    """The HIP library/GPU is unavailable or a de_* call failed.  Never swallowed."""


class UncertifiedFlag(DeviceError):
    """EvalContext(strict_flags=True): the device's element-wise validity flag of this tree is not PROVABLY the reference's
    `isfinite(sum(x))` flag (a finite array whose sum may overflow): the caller keeps the CPU path for it."""


class ParamArgs(C.Structure):
    _fields_ = [("params", C.c_void_p), ("ld_params", C.c_int64), ("n_classes", C.c_int64),
                ("classes", C.c_void_p), ("classes_is_i64", C.c_int32), ("class_base", C.c_int32)]


_lib: Optional[C.CDLL] = None


def library() -> C.CDLL:
    """Load libde_hip.so (built by ``__graft_entry__.build()`` / ``csrc/build.sh``)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise DeviceError(f"{LIB_PATH} is missing: run `python -c 'import __graft_entry__ as g; g.build()'` "
                          "(there is no CPU fallback)")
    try:
        # torch ships its own libamdhip64; import it FIRST so that the process has exactly one HIP
        # runtime (two runtimes in one process do not see the device).  Without torch (e.g. the Julia
        # shim) libde_hip.so simply uses /opt/rocm's runtime.
        try:
            import torch  # noqa: F401
        except ImportError:  # pragma: no cover
            pass
        lib = C.CDLL(LIB_PATH)
    except OSError as e:  # pragma: no cover
        raise DeviceError(f"cannot load {LIB_PATH}: {e}") from e
    vp, i32, i64, u32 = C.c_void_p, C.c_int32, C.c_int64, C.c_uint32
    lib.de_abi_version.restype = C.c_int
    if lib.de_abi_version() != ABI_VERSION:  # include/de_hip.h lists what changed between versions
        raise DeviceError(f"{LIB_PATH} has ABI version {lib.de_abi_version()}, this module was written for {ABI_VERSION}: rebuild (csrc/build.sh)")
    lib.de_opcode_table_version.restype = C.c_int
    lib.de_opcode_by_name.argtypes = [C.c_char_p, C.c_int]
    lib.de_opcode_name.restype = C.c_char_p
    lib.de_opcode_name.argtypes = [C.c_int]
    lib.de_opcode_degree.argtypes = [C.c_int]
    lib.de_status_string.restype = C.c_char_p
    lib.de_status_string.argtypes = [C.c_int]
    lib.de_ctx_create.argtypes = [C.c_int, vp, C.POINTER(vp)]
    lib.de_ctx_destroy.argtypes = [vp]
    lib.de_ctx_set_stream.argtypes = [vp, vp]
    lib.de_ctx_synchronize.argtypes = [vp]
    lib.de_ctx_declare_dataset.argtypes = [vp, C.c_int, vp, i64, i64, i32]
    lib.de_ctx_stream.restype = vp
    lib.de_ctx_stream.argtypes = [vp]
    lib.de_last_error.restype = C.c_char_p
    lib.de_last_error.argtypes = [vp]
    lib.de_program_create.argtypes = [vp, C.c_int, vp, vp, i64, vp, vp, i32, i32, u32, C.POINTER(vp)]
    lib.de_program_create_cse.argtypes = [vp, C.c_int, vp, vp, vp, vp, i64, vp, vp, i32, i32, u32, C.POINTER(vp)]
    lib.de_program_set_consts.argtypes = [vp, vp]
    lib.de_program_destroy.argtypes = [vp]
    lib.de_program_n_trees.restype = i64
    lib.de_program_n_trees.argtypes = [vp]
    lib.de_program_n_nodes.restype = i64
    lib.de_program_n_nodes.argtypes = [vp]
    lib.de_program_n_grad.restype = i64
    lib.de_program_n_grad.argtypes = [vp, i64, C.c_int]
    lib.de_program_verify.argtypes = [vp]
    lib.de_host_pool_selftest.restype = C.c_int64
    lib.de_host_pool_selftest.argtypes = [C.c_int64, C.POINTER(C.c_int32)]
    lib.de_program_stream_hash.restype = C.c_uint64
    lib.de_program_stream_hash.argtypes = [vp]
    lib.de_program_dump.restype = i64
    lib.de_program_dump.argtypes = [vp, i64, vp, i64, C.c_int]
    lib.de_lower_tape.restype = i64
    lib.de_lower_tape.argtypes = [C.c_int, vp, i64, vp, i64, i32, i32, u32, vp, i64, vp]
    lib.de_lower_tape_stage.restype = i64
    lib.de_lower_tape_stage.argtypes = [C.c_int, vp, i64, vp, i64, i32, i32, u32, C.c_int, vp, i64]
    lib.de_eval.argtypes = [vp, vp, vp, i64, i64, C.POINTER(ParamArgs), vp, i64, vp]
    lib.de_eval_loss.argtypes = [vp, vp, vp, i64, i64, C.POINTER(ParamArgs), vp, vp, C.c_int32, vp, vp]
    lib.de_eval_loss_grad.argtypes = [vp, vp, vp, i64, i64, C.POINTER(ParamArgs), C.c_int, vp, vp, C.c_int32, vp, vp, vp, vp]
    lib.de_eval_loss_grad_by_class.argtypes = [vp, vp, vp, i64, i64, C.POINTER(ParamArgs), C.c_int, vp, vp, C.c_int32, vp, vp, vp, vp, vp, vp]
    lib.de_eval_grad.argtypes = [vp, vp, vp, i64, i64, C.POINTER(ParamArgs), C.c_int, vp, i64, vp, vp, vp]
    lib.de_eval_diff.argtypes = [vp, vp, vp, i64, i64, i32, vp, vp, i64, vp]
    lib.de_eval_pullback_dX.argtypes = [vp, vp, vp, i64, i64, C.POINTER(ParamArgs), vp, vp, vp, vp]
    lib.de_eval_tree_array.argtypes = [vp, C.c_int, vp, i64, vp, i64, vp, i32, i64, u32, vp, vp]
    lib.de_eval_plan.argtypes = [vp, i64, vp]
    lib.de_prio_tiles_wanted.argtypes = [i64, i32, i64]
    lib.de_program_last_live_trees.argtypes = [vp, C.POINTER(i64)]
    lib.de_dist_unique_id.argtypes = [vp]
    lib.de_dist_init.argtypes = [vp, C.c_int, C.c_int, vp, C.POINTER(vp)]
    lib.de_dist_destroy.argtypes = [vp]
    lib.de_dist_set_timeout.argtypes = [vp, i64]
    lib.de_dist_world_size.argtypes = [vp]
    lib.de_dist_shard_size.restype = i64
    lib.de_dist_shard_size.argtypes = [i64, C.c_int, C.c_int]
    lib.de_dist_broadcast.argtypes = [vp, vp, C.c_size_t, C.c_int]
    lib.de_dist_gather_flags.argtypes = [vp, vp, i64, vp]
    lib.de_dist_last_error.restype = C.c_char_p
    lib.de_dist_last_error.argtypes = [vp]
    lib.de_ctx_last_kernel_ms.argtypes = [vp, C.POINTER(C.c_float)]
    lib.de_ctx_last_kernel_name.restype = C.c_char_p
    lib.de_ctx_last_kernel_name.argtypes = [vp]
    lib.de_ctx_device.argtypes = [vp]
    lib.de_ctx_timing_ring.argtypes = [vp, i32]
    lib.de_ctx_timing_read.argtypes = [vp, vp, i32, C.POINTER(i32)]
    lib.de_dist_reorder_selftest.argtypes = [vp, vp, i64, C.c_int, vp, C.POINTER(C.c_float)]
    lib.de_eval_sum_certificate.argtypes = [vp, vp, vp, i64, i64, C.POINTER(ParamArgs), vp, vp, vp]
    _lib = lib
    return lib


def _dtype_code(dtype) -> int:
    dtype = np.dtype(dtype)
    if dtype == np.float32:
        return DE_F32
    if dtype == np.float64:
        return DE_F64
    # the reference asserts T in (Float32, Float64) for its accelerated back-ends
    # (src/Evaluate.jl:287-289)
    raise TypeError(f"MI355X back-end supports Float32/Float64, got {dtype}")


@dataclass
class EvalContext:
    """``EvalContext(; turbo, bumper, early_exit, buffer, use_fused)`` (src/Evaluate.jl:156-181).

    ``turbo=True`` (the LoopVectorization path of the reference) selects the relaxed-accuracy Float32 operators
    (``DE_OPT_TURBO``: <= 1e-6 relative, documented domain edges; Float64 and the gradient entry points run the exact
    operators); ``buffer`` (ArrayBuffer arena) is a CPU allocation detail, accepted for signature compatibility;
    ``bumper=True`` selects the Bumper path's flag semantics."""
    turbo: bool = False
    bumper: bool = False
    early_exit: bool = True
    buffer: object = None
    use_fused: bool = True
    full_eval: bool = False  # DE_OPT_FULL_EVAL: evaluate incomplete trees to the end as well (no early exit at tree granularity)
    forward_grad: bool = False  # DE_OPT_FORWARD_GRAD: round 5's spelling of what is the default since ABI 3 (forward duals); wins over reverse_grad
    # DE_OPT_REVERSE_GRAD: PERMISSION to run fused loss gradients by reverse accumulation (faster from 8 gradient rows per tree on; its
    # products are associated leaf-wards: `ok` may differ from the reference's forward-mode flag in ~0.03 % of Float32 fuzz cases)
    reverse_grad: bool = False
    # strict_flags: after every Population.eval the certificate pass (de_eval_sum_certificate) runs as well and `Population.uncertified`
    # lists the trees whose element-wise flag is NOT provably the reference's isfinite(sum(x)) flag (src/ValueInterface.jl:9) — the only
    # trees a caller who needs the reference's bit has to re-derive on the CPU; the one-tree sugar raises UncertifiedFlag for such a tree
    strict_flags: bool = False

    def option_bits(self, operators: OperatorEnum) -> int:
        f1, f2 = operators.fuse_flags(self.use_fused)
        return ((OPT_EARLY_EXIT if self.early_exit else 0) | (OPT_FUSE_DEG1 if f1 else 0) |
                (OPT_FUSE_DEG2 if f2 else 0) | (OPT_BUMPER_CHECKS if self.bumper else 0) | (OPT_TURBO if self.turbo else 0) |
                (OPT_FULL_EVAL if self.full_eval else 0) | (OPT_FORWARD_GRAD if self.forward_grad else 0) |
                (OPT_REVERSE_GRAD if self.reverse_grad else 0))


class Context:
    """One ``de_ctx_t``: a device + the stream the kernels are launched on.  By default the
    stream is torch's current stream for that device, so torch ops and de_* calls order
    naturally."""

    def __init__(self, device: int = 0, stream: Optional[int] = None):
        lib = library()
        follow = False
        if stream is None:
            follow = True
            try:
                import torch
                if torch.cuda.is_available():
                    # 0 = HIP's null stream (torch's default stream) -> DE_STREAM_NULL
                    stream = torch.cuda.current_stream(device).cuda_stream or -1
            except ImportError:  # pragma: no cover
                stream = None
        self._h = C.c_void_p()
        self._follow_torch = follow  # launch on torch's CURRENT stream of each call (see use_torch_stream)
        rc = lib.de_ctx_create(device, C.c_void_p(stream) if stream else None, C.byref(self._h))
        if rc != 0:
            raise DeviceError(f"de_ctx_create(device={device}) failed: {lib.de_status_string(rc).decode()} "
                              "(no MI355X visible? there is no CPU fallback)")
        self.device = device
        self._stream = stream

    def use_torch_stream(self) -> None:
        """Called before every de_* call that takes torch tensors: inside ``with torch.cuda.stream(s):`` torch
        allocates the outputs on ``s`` and expects the kernels there too, so the context follows torch's current
        stream (one pointer compare when it has not changed; de_ctx_set_stream orders the new stream behind the
        work already queued).  A context created with an explicit ``stream=`` keeps it."""
        if not self._follow_torch:
            return
        import torch
        cur = torch.cuda.current_stream(self.device).cuda_stream or -1
        if cur != self._stream:
            self.check(library().de_ctx_set_stream(self._h, C.c_void_p(cur)))
            self._stream = cur

    def check(self, rc: int) -> None:
        if rc != 0:
            lib = library()
            msg = lib.de_last_error(self._h).decode()
            name = lib.de_status_string(rc).decode()
            if rc in (1, 2, 6):
                raise ValueError(f"{name}: {msg}")
            if rc == 3:
                from .operators import UnsupportedOperatorError
                raise UnsupportedOperatorError(f"{name}: {msg}")
            raise DeviceError(f"{name}: {msg}")

    def synchronize(self) -> None:
        self.check(library().de_ctx_synchronize(self._h))

    def trim(self) -> None:
        """Free what the context retains between programs (parked host vectors, recycled device buffers, staging scratch)."""
        self.check(library().de_ctx_trim(self._h))

    def declare_dataset(self, X, dtype=None) -> None:
        """``X``: a device tensor ``[F, N]`` (feature index fastest, as ``eval`` takes it) that stays unchanged between calls — the
        library computes its per-dataset statistics (the priority-tile keys) once instead of in every call; ``None`` withdraws."""
        lib = library()
        if X is None:
            self.check(lib.de_ctx_declare_dataset(self._h, 0, None, 0, 0, 0))
            return
        # the declaration is matched by (pointer, N, ldX): it must be the very tensor `eval` will use IN PLACE — a device tensor of the
        # feature-fastest layout.  Anything `eval` would copy first (host array, wrong layout) can never match: refuse it loudly.
        if not _is_torch(X) or not X.is_cuda or X.dim() != 2:
            raise ValueError("declare_dataset needs the 2-d DEVICE tensor [F, N] that eval() is called with")
        name = str(X.dtype)
        if name not in ("torch.float32", "torch.float64"):
            raise ValueError(f"declare_dataset: dtype {name} is neither float32 nor float64")
        F, N = int(X.shape[0]), int(X.shape[1])
        if N > 1 and (int(X.stride(0)) != 1 or int(X.stride(1)) < F):
            raise ValueError("declare_dataset: X must be feature-fastest ([N, F] storage viewed as [F, N], e.g. Xs.t()): "
                             f"strides {tuple(X.stride())} would be copied by eval() and the declaration would never match")
        ldX = int(X.stride(1)) if N > 1 else F
        dt = DE_F32 if name == "torch.float32" else DE_F64
        self.check(lib.de_ctx_declare_dataset(self._h, dt, X.data_ptr(), N, ldX, F))

    def last_kernel_ms(self) -> float:
        ms = C.c_float(0)
        self.check(library().de_ctx_last_kernel_ms(self._h, C.byref(ms)))
        return float(ms.value)

    def timing_ring(self, n: int) -> None:
        """Keep the event pairs of the last ``n`` timed calls (0: off): ``timing_read`` then returns the device time of every call of a
        free-running loop without a synchronisation per call (``last_kernel_ms`` blocks)."""
        self.check(library().de_ctx_timing_ring(self._h, int(n)))

    def timing_read(self, cap: int = 4096) -> list:
        """Device ms of the timed calls since ``timing_ring`` / the last read, oldest first (waits for the last one)."""
        buf = (C.c_float * int(cap))()
        n = C.c_int32(0)
        self.check(library().de_ctx_timing_read(self._h, C.cast(buf, C.c_void_p), int(cap), C.byref(n)))
        return [float(buf[i]) for i in range(n.value)]

    def last_kernel_name(self) -> str:
        return library().de_ctx_last_kernel_name(self._h).decode()

    def close(self) -> None:
        if self._h:
            library().de_ctx_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):  # pragma: no cover
        try:
            self.close()
        except Exception:
            pass


_default_ctx = {}


def default_context(device: int = 0) -> Context:
    if device not in _default_ctx:
        _default_ctx[device] = Context(device)
    return _default_ctx[device]


def _is_torch(x) -> bool:
    return type(x).__module__.startswith("torch")


def _prep_X(X, dtype):
    """Return (pointer, F, N, ldX, keepalive, is_torch).  X is [n_features, N]; memory must
    be feature-fastest (src/Evaluate.jl:251), i.e. Fortran order for numpy / stride (1, F)
    for torch."""
    if _is_torch(X):
        import torch
        if X.dim() == 1:  # eval_tree_array(tree, cX::AbstractVector) (src/Evaluate.jl:311-315)
            X = X.reshape(-1, 1)
        tdt = torch.float32 if np.dtype(dtype) == np.float32 else torch.float64
        if X.dtype != tdt:
            X = X.to(tdt)
        F, N = X.shape
        if not (X.stride(0) == 1 and (N <= 1 or X.stride(1) >= F)) and X.numel() > 0:
            X = X.t().contiguous().t()
        ld = X.stride(1) if N > 1 else max(F, 1)
        return X.data_ptr(), F, N, ld, X, True
    X = np.asarray(X)
    if X.ndim == 1:
        X = X.reshape(-1, 1)
    Xf = np.asfortranarray(X, dtype=dtype)
    F, N = Xf.shape
    return Xf.ctypes.data, F, N, max(F, 1), Xf, False


def lower_tape(tape, consts, n_features: int, n_params: int = 0, options: int = 7, dtype=np.float32):
    """Host-only: lower one tape, return (instr[n,4] uint32 words, meta dict).  No GPU needed."""
    lib = library()
    dtype = np.dtype(dtype)
    tape = np.ascontiguousarray(tape)
    consts = np.ascontiguousarray(consts, dtype=dtype)
    meta = np.zeros(4, dtype=np.int32)
    cp = consts.ctypes.data if consts.size else None
    n = lib.de_lower_tape(_dtype_code(dtype), tape.ctypes.data, len(tape), cp, consts.size, n_features,
                          n_params, options, None, 0, meta.ctypes.data)
    if n < 0:
        code = int(-n)
        name = lib.de_status_string(code).decode()
        if code == 3:
            from .operators import UnsupportedOperatorError
            raise UnsupportedOperatorError(name)
        raise ValueError(name)
    w = np.zeros(max(int(n), 1), dtype=np.uint32)
    lib.de_lower_tape(_dtype_code(dtype), tape.ctypes.data, len(tape), cp, consts.size, n_features,
                      n_params, options, w.ctypes.data, w.size, meta.ctypes.data)
    return w[:int(n)].reshape(-1, 4), dict(n_slots=int(meta[0]), host_ok_eval=bool(meta[1]),
                                           host_ok_grad=bool(meta[2]), uses_params=bool(meta[3]))


def lower_tape_stage(tape, consts, n_features: int, stage: int, n_params: int = 0, options: int = 7,
                     dtype=np.float32) -> np.ndarray:
    """Host-only: the bound (stage 2) or fused (stage 3) instruction words of one tape, [n, 4] uint32."""
    lib = library()
    dtype = np.dtype(dtype)
    tape = np.ascontiguousarray(tape)
    consts = np.ascontiguousarray(consts, dtype=dtype)
    cp = consts.ctypes.data if consts.size else None
    args = (_dtype_code(dtype), tape.ctypes.data, len(tape), cp, consts.size, n_features, n_params, options, stage)
    n = lib.de_lower_tape_stage(*args, None, 0)
    if n < 0:
        raise ValueError(lib.de_status_string(int(-n)).decode())
    w = np.zeros(max(int(n), 1), dtype=np.uint32)
    lib.de_lower_tape_stage(*args, w.ctypes.data, w.size)
    return w[:int(n)].reshape(-1, 4)


class Population:
    """A population of trees lowered once to a device program (``de_program_t``).

    ``eval(X)`` evaluates every tree on every sample in one launch and returns
    ``(out[n_trees, N], ok[n_trees])``.  numpy in -> numpy out (staged through the
    library's device scratch, PCIe-inclusive); torch CUDA tensor in -> torch CUDA tensors out
    (zero-copy, asynchronous on the current stream)."""

    def __init__(self, trees: Sequence[Node], operators: OperatorEnum, dtype=np.float32,
                 n_features: Optional[int] = None, n_params: int = 0,
                 eval_context: Optional[EvalContext] = None, ctx: Optional[Context] = None):
        self.ctx = ctx or default_context()
        self.dtype = np.dtype(dtype)
        self.operators = operators
        self.eval_context = eval_context or EvalContext()
        self.n_trees = len(trees)
        nodes, noff, consts, coff = flatten_population(trees, operators, self.dtype)
        if n_features is None:
            n_features = max((max_feature(t) for t in trees), default=0)
        self.n_features, self.n_params = int(n_features), int(n_params)
        self.n_consts = np.diff(coff).astype(np.int64)  # per tree, as the user counts them (GraphNode: unique constants)
        self._classes_checked = set()
        self._h = C.c_void_p()
        lib = library()
        # GraphNode trees (src/Node.jl:138-166): the library gets the expanded tape (one constant slot per OCCURRENCE) plus
        # a CSE tape for the eval program; a shared constant is ONE constant to the user: `_occ[t]` maps occurrence slots
        # to unique constants, set_constants fans values out, the gradient entry points sum the occurrence rows.
        self._occ = None
        cse_ptrs = (None, None)
        if any(preserve_sharing(t) for t in trees):
            occ, cse_tapes = [], []
            for t in trees:
                _, _, cse, o = flatten_graph(t, operators, self.dtype) if preserve_sharing(t) else (None, None, None, None)
                occ.append(o if o is not None and len(o) and len(np.unique(o)) < len(o) else None)
                cse_tapes.append(cse if cse is not None else np.zeros(0, dtype=TAPE_DTYPE))
            if any(o is not None for o in occ):
                self._occ = occ
                self.n_consts = np.array([len(np.unique(o)) if o is not None else int(n) for o, n in zip(occ, np.diff(coff))], dtype=np.int64)
            if any(len(c) for c in cse_tapes):
                self._cse_nodes = np.concatenate(cse_tapes) if cse_tapes else np.zeros(0, dtype=TAPE_DTYPE)
                self._cse_off = np.zeros(len(trees) + 1, dtype=np.int64)
                np.cumsum([len(c) for c in cse_tapes], out=self._cse_off[1:])
                cse_ptrs = (self._cse_nodes.ctypes.data, self._cse_off.ctypes.data)
        self._slots_per_tree = np.diff(coff).astype(np.int64)
        if cse_ptrs[0] is not None:
            self.ctx.check(lib.de_program_create_cse(
                self.ctx._h, _dtype_code(self.dtype), nodes.ctypes.data, noff.ctypes.data, cse_ptrs[0], cse_ptrs[1], self.n_trees,
                consts.ctypes.data if len(consts) else None, coff.ctypes.data, self.n_features, self.n_params,
                self.eval_context.option_bits(operators), C.byref(self._h)))
        else:
            self.ctx.check(lib.de_program_create(
                self.ctx._h, _dtype_code(self.dtype), nodes.ctypes.data, noff.ctypes.data, self.n_trees,
                consts.ctypes.data if len(consts) else None, coff.ctypes.data, self.n_features, self.n_params,
                self.eval_context.option_bits(operators), C.byref(self._h)))
        self.n_nodes = int(lib.de_program_n_nodes(self._h))
        self.uncertified = np.zeros(0, dtype=np.int64)  # EvalContext(strict_flags=True): set by every eval()

    # -- constants (optimiser inner loop, src/NodeUtils.jl:99-143) ------------------
    def set_constants(self, consts: np.ndarray) -> None:
        consts = np.ascontiguousarray(consts, dtype=self.dtype)
        if consts.size != int(self.n_consts.sum()):
            raise ValueError("wrong number of constants")
        if self._occ is not None:  # one value per unique constant -> one per occurrence slot
            parts, at = [], 0
            for o, nu, ns in zip(self._occ, self.n_consts, self._slots_per_tree):
                vals = consts[at:at + int(nu)]
                parts.append(vals[o] if o is not None else vals)
                at += int(nu)
            consts = np.ascontiguousarray(np.concatenate(parts) if parts else consts, dtype=self.dtype)
        self.ctx.check(library().de_program_set_consts(self._h, consts.ctypes.data if consts.size else None))

    def verify(self) -> None:
        """Program sanitizer (``de_program_verify``): raises ValueError naming the offending instruction."""
        self.ctx.check(library().de_program_verify(self._h))

    def stream_hash(self) -> int:
        """Test hook (``de_program_stream_hash``): one number over every host-side stream ``de_program_create`` built."""
        return int(library().de_program_stream_hash(self._h))

    def _combine_rows(self, t: int, g, mode: int):
        """Gradient rows (or entries) of tree t in the library's per-occurrence layout -> the reference's layout for a
        GraphNode: the rows of a shared constant are summed (its NodeIndex entry is shared, src/NodeUtils.jl:184-201)."""
        if self._occ is None or self._occ[t] is None or mode == GRAD_VARIABLE:
            return g
        o = self._occ[t]
        lead = g.shape[0] - len(o)  # (params,) features rows of the :both mode come first
        nu = int(o.max()) + 1
        if _is_torch(g):
            import torch
            out = torch.zeros((lead + nu,) + tuple(g.shape[1:]), dtype=g.dtype, device=g.device)
            out[:lead] = g[:lead]
            out.index_add_(0, torch.as_tensor(o + lead, device=g.device), g[lead:])
            return out
        out = np.zeros((lead + nu,) + g.shape[1:], dtype=g.dtype)
        out[:lead] = g[:lead]
        np.add.at(out, o + lead, g[lead:])
        return out

    def plan(self, N: int) -> dict:
        """Launch plan of ``eval`` for N samples (tile size, tree chunks, trees per chunk)."""
        pl = np.zeros(3, dtype=np.int32)
        self.ctx.check(library().de_eval_plan(self._h, N, pl.ctypes.data))
        return dict(tile=int(pl[0]), n_chunks=int(pl[1]), trees_per_chunk=int(pl[2]))

    def last_live_trees(self) -> int:
        """Trees still live behind the probe launch of the priority tiles in the most recent ``eval`` / ``eval_loss`` (the launch proper
        ran dense chunks over them); -1 when that call did not compact (small launch, full_eval, early_exit off)."""
        n = C.c_int64(-1)
        self.ctx.check(library().de_program_last_live_trees(self._h, C.byref(n)))
        return int(n.value)

    def n_grad(self, tree: int, mode: int) -> int:
        return int(library().de_program_n_grad(self._h, tree, mode))

    def _n_grad_all(self, mode: int) -> np.ndarray:
        """n_grad of every tree in ``mode`` (cached: it depends on the tree shapes only)."""
        cache = self.__dict__.setdefault("_ng_cache", {})
        if mode not in cache:
            lib = library()
            cache[mode] = np.array([lib.de_program_n_grad(self._h, t, mode) for t in range(self.n_trees)], dtype=np.int64)
        return cache[mode]

    def dump(self, tree: int) -> np.ndarray:
        """Lowered instruction words of one tree ([n_instr, 4] uint32) — test hook."""
        lib = library()
        n = lib.de_program_dump(self._h, tree, None, 0, 0)
        w = np.zeros(max(int(n), 1), dtype=np.uint32)
        got = lib.de_program_dump(self._h, tree, w.ctypes.data, w.size, 0)
        return w[:int(got)].reshape(-1, 4)

    def meta(self, tree: int) -> dict:
        w = np.zeros(5, dtype=np.uint32)
        library().de_program_dump(self._h, tree, w.ctypes.data, 5, 1)
        return dict(n_slots=int(w[0]), host_ok_eval=bool(w[1]), host_ok_grad=bool(w[2]), uses_params=bool(w[3]), waves=int(w[4]))

    # -- evaluation -------------------------------------------------------------------
    def _param_args(self, params, classes, class_base, N, keep):
        if self.n_params == 0 and params is None:
            return None
        if params is None or classes is None:
            # src/ParametricExpression.jl:357-359
            raise ValueError("Incorrect call. You must pass the `classes::Vector` argument when calling `eval_tree_array`.")
        pa = ParamArgs()
        if _is_torch(params):
            import torch
            tdt = torch.float32 if self.dtype == np.float32 else torch.float64
            params = params.to(tdt)
            if params.stride(0) != 1 and params.numel() > 0:
                params = params.t().contiguous().t()
            P, ncls = params.shape
            pa.params, pa.ld_params = params.data_ptr(), (params.stride(1) if ncls > 1 else max(P, 1))
        else:
            params = np.asfortranarray(params, dtype=self.dtype)
            P, ncls = params.shape
            pa.params, pa.ld_params = params.ctypes.data, max(P, 1)
        if P != self.n_params:
            raise ValueError("parameter matrix has the wrong number of rows")
        if _is_torch(classes):
            import torch
            if classes.dtype not in (torch.int32, torch.int64):
                classes = classes.to(torch.int64)
            classes = classes.contiguous()
            n_c = classes.numel()
            # @assert maximum(classes) <= n_classes (:378-379).  Reading min/max of a device tensor synchronises the
            # stream, so the verdict is cached per (storage, version, shape): the hot path of a search loop — the same
            # `classes` tensor call after call — stays asynchronous.
            key = (classes.data_ptr(), n_c, classes._version, ncls, class_base)
            if n_c and key not in self._classes_checked:
                mn, mx = int(classes.min().item()), int(classes.max().item())
                if mn < class_base or mx - class_base >= ncls:
                    raise ValueError(f"class ids must lie in [{class_base}, {class_base + ncls}): got [{mn}, {mx}] "
                                     "(maximum(classes) <= size(parameters, 2))")
                if len(self._classes_checked) > 64:
                    self._classes_checked.clear()
                self._classes_checked.add(key)
            pa.classes, pa.classes_is_i64 = classes.data_ptr(), int(classes.dtype == torch.int64)
        else:
            classes = np.ascontiguousarray(classes)
            if classes.dtype not in (np.int32, np.int64):
                classes = classes.astype(np.int64)
            n_c = classes.size
            if n_c:
                mn, mx = int(classes.min()), int(classes.max())
                if mn < class_base or mx - class_base >= ncls:
                    raise ValueError(f"class ids must lie in [{class_base}, {class_base + ncls}): got [{mn}, {mx}] "
                                     "(maximum(classes) <= size(parameters, 2))")
            pa.classes, pa.classes_is_i64 = classes.ctypes.data, int(classes.dtype == np.int64)
        # @assert length(classes) == size(X, 2)  (:378)
        if n_c != N:
            raise ValueError(f"length(classes) == size(X, 2) violated: {n_c} class ids for {N} samples")
        pa.n_classes, pa.class_base = ncls, class_base
        keep.extend([params, classes])
        return pa

    def eval(self, X, params=None, classes=None, class_base: int = 1):
        ptr, F, N, ldX, keep_x, is_t = _prep_X(X, self.dtype)
        if is_t:
            self.ctx.use_torch_stream()
        if F < self.n_features:
            raise ValueError(f"X has {F} features but the trees use feature {self.n_features}")
        keep = [keep_x]
        pa = self._param_args(params, classes, class_base, N, keep)
        lib = library()
        if is_t:
            import torch
            out = torch.empty((self.n_trees, N), dtype=keep_x.dtype, device=keep_x.device)
            ok = torch.empty(self.n_trees, dtype=torch.uint8, device=keep_x.device)
            self.ctx.check(lib.de_eval(self.ctx._h, self._h, ptr, N, ldX, C.byref(pa) if pa else None,
                                       out.data_ptr(), N, ok.data_ptr()))
            if self.eval_context.strict_flags:
                self._certify(ptr, N, ldX, pa, ok.cpu().numpy())
            return out, ok.bool()
        out = np.empty((self.n_trees, N), dtype=self.dtype)
        ok = np.zeros(self.n_trees, dtype=np.uint8)
        self.ctx.check(lib.de_eval(self.ctx._h, self._h, ptr, N, ldX, C.byref(pa) if pa else None,
                                   out.ctypes.data, N, ok.ctypes.data))
        if self.eval_context.strict_flags:
            self._certify(ptr, N, ldX, pa, ok)
        return out, ok.astype(bool)

    def _certify(self, ptr, N, ldX, pa, ok_eval) -> None:
        """strict_flags: the certificate pass behind an evaluation; `self.uncertified` = indices of the trees whose flag is not
        provably the reference's (and the pass must reproduce the evaluation's flags: it runs the same tests)."""
        ok = np.zeros(self.n_trees, dtype=np.uint8)
        cert = np.zeros(self.n_trees, dtype=np.uint8)
        self.ctx.check(library().de_eval_sum_certificate(self.ctx._h, self._h, ptr, N, ldX, C.byref(pa) if pa else None,
                                                         ok.ctypes.data, cert.ctypes.data, None))
        if not np.array_equal(ok.astype(bool), np.asarray(ok_eval).astype(bool)):
            raise DeviceError("strict_flags: the certificate pass and the evaluation disagree on a flag")
        self.uncertified = np.nonzero(cert == 0)[0]

    def sum_certificate(self, X, params=None, classes=None, class_base: int = 1):
        """``(ok, certified, max_abs)`` (numpy, per tree): the certificate of ``de_eval_sum_certificate`` — ``certified[t]`` says that the
        reference's ``complete`` (``isfinite(sum(x))`` per tested array, src/ValueInterface.jl:9) provably equals the element-wise flag
        ``ok[t]`` the kernels compute; ``max_abs[t]`` = the largest |tested value or constant operand| of the tree."""
        ptr, F, N, ldX, keep_x, is_t = _prep_X(X, self.dtype)
        if is_t:
            self.ctx.use_torch_stream()
        if F < self.n_features:
            raise ValueError(f"X has {F} features but the trees use feature {self.n_features}")
        keep = [keep_x]
        pa = self._param_args(params, classes, class_base, N, keep)
        ok = np.zeros(self.n_trees, dtype=np.uint8)
        cert = np.zeros(self.n_trees, dtype=np.uint8)
        mx = np.zeros(self.n_trees, dtype=np.float64)
        self.ctx.check(library().de_eval_sum_certificate(self.ctx._h, self._h, ptr, N, ldX, C.byref(pa) if pa else None,
                                                         ok.ctypes.data, cert.ctypes.data, mx.ctypes.data))
        return ok.astype(bool), cert.astype(bool), mx

    def eval_loss(self, X, y, weights=None, loss: str = "L2", params=None, classes=None, class_base: int = 1):
        """Fused ``sum_j w_j * l(tree_t(X[:, j]) - y[j])`` for every tree (l = abs2 for "L2", abs for
        "L1") without materialising the [n_trees, N] output: what the reference's optimisation
        consumers compute right after eval_tree_array (``sum(abs2, tree(X, operators) .- y)``,
        test/test_optim.jl:95,99).  Returns (loss[n_trees], ok[n_trees]); loss is NaN where not ok."""
        kind = {"L2": 0, "L1": 1}[loss]
        ptr, F, N, ldX, keep_x, is_t = _prep_X(X, self.dtype)
        if is_t:
            self.ctx.use_torch_stream()
        if F < self.n_features:
            raise ValueError(f"X has {F} features but the trees use feature {self.n_features}")
        keep = [keep_x]
        pa = self._param_args(params, classes, class_base, N, keep)
        lib = library()

        def vec(v, name):
            if v is None:
                return None
            if is_t:
                import torch
                v = torch.as_tensor(v, dtype=keep_x.dtype, device=keep_x.device).contiguous()
                if v.numel() != N:
                    raise ValueError(f"{name} must have {N} entries")
                keep.append(v)
                return v.data_ptr()
            v = np.ascontiguousarray(v, dtype=self.dtype)
            if v.size != N:
                raise ValueError(f"{name} must have {N} entries")
            keep.append(v)
            return v.ctypes.data

        yp, wp = vec(y, "y"), vec(weights, "weights")
        if is_t:
            import torch
            out = torch.empty(self.n_trees, dtype=keep_x.dtype, device=keep_x.device)
            ok = torch.empty(self.n_trees, dtype=torch.uint8, device=keep_x.device)
            self.ctx.check(lib.de_eval_loss(self.ctx._h, self._h, ptr, N, ldX, C.byref(pa) if pa else None,
                                            yp, wp, kind, out.data_ptr(), ok.data_ptr()))
            return out, ok.bool()
        out = np.empty(self.n_trees, dtype=self.dtype)
        ok = np.zeros(self.n_trees, dtype=np.uint8)
        self.ctx.check(lib.de_eval_loss(self.ctx._h, self._h, ptr, N, ldX, C.byref(pa) if pa else None,
                                        yp, wp, kind, out.ctypes.data, ok.ctypes.data))
        return out, ok.astype(bool)

    def eval_loss_grad(self, X, y, weights=None, loss: str = "L2", variable: Union[bool, str] = False,
                       params=None, classes=None, class_base: int = 1):
        """Fused loss and its gradient w.r.t. the rows ``variable`` selects (default: the constants) —
        the body of the reference's optimiser callback (test/test_optim.jl:42-51: ``G[i] = sum_j
        2(yhat_j - y_j) * dyhat_dconstants[i, j]``) without the [n_grad, N] Jacobian.  ``loss="pullback"``
        treats ``y`` as the cotangent dY of the ChainRules pullback (src/ChainRules.jl:56-77).
        Returns (loss[n_trees], [dloss_t[n_grad_t] per tree], ok)."""
        kind = {"L2": 0, "L1": 1, "pullback": 2}[loss]
        mode = _grad_mode(variable)
        ptr, F, N, ldX, keep_x, is_t = _prep_X(X, self.dtype)
        if is_t:
            self.ctx.use_torch_stream()
        if F < self.n_features:
            raise ValueError(f"X has {F} features but the trees use feature {self.n_features}")
        keep = [keep_x]
        pa = self._param_args(params, classes, class_base, N, keep)
        lib = library()
        ng = self._n_grad_all(mode)
        offs = np.zeros(self.n_trees + 1, dtype=np.int64)
        np.cumsum(ng, out=offs[1:])
        total = max(int(offs[-1]), 1)

        def vec(v, name):
            if v is None:
                return None
            if is_t:
                import torch
                v = torch.as_tensor(v, dtype=keep_x.dtype, device=keep_x.device).contiguous()
                n = v.numel()
                p_ = v.data_ptr()
            else:
                v = np.ascontiguousarray(v, dtype=self.dtype)
                n = v.size
                p_ = v.ctypes.data
            if n != N:
                raise ValueError(f"{name} must have {N} entries")
            keep.append(v)
            return p_

        yp, wp = vec(y, "y"), vec(weights, "weights")
        if is_t:
            import torch
            lo = torch.empty(self.n_trees, dtype=keep_x.dtype, device=keep_x.device)
            dl = torch.empty(total, dtype=keep_x.dtype, device=keep_x.device)
            ok = torch.empty(self.n_trees, dtype=torch.uint8, device=keep_x.device)
            self.ctx.check(lib.de_eval_loss_grad(self.ctx._h, self._h, ptr, N, ldX, C.byref(pa) if pa else None, mode,
                                                 yp, wp, kind, lo.data_ptr(), dl.data_ptr(), offs.ctypes.data, ok.data_ptr()))
            return lo, [self._combine_rows(t, d, mode) for t, d in enumerate(torch.split(dl[:int(offs[-1])], ng.tolist()))], ok.bool()
        lo = np.empty(self.n_trees, dtype=self.dtype)
        dl = np.empty(total, dtype=self.dtype)
        ok = np.zeros(self.n_trees, dtype=np.uint8)
        self.ctx.check(lib.de_eval_loss_grad(self.ctx._h, self._h, ptr, N, ldX, C.byref(pa) if pa else None, mode,
                                             yp, wp, kind, lo.ctypes.data, dl.ctypes.data, offs.ctypes.data, ok.ctypes.data))
        return lo, [self._combine_rows(t, d, mode) for t, d in enumerate(np.split(dl[:int(offs[-1])], offs[1:-1]))], ok.astype(bool)

    def eval_loss_grad_by_class(self, X, y, params, classes, weights=None, loss: str = "L2",
                                variable: Union[bool, str] = "both", class_base: int = 1, grouped: bool = False):
        """Fused loss + gradient of a parametric population with the parameter rows reduced by class:
        returns (loss[n_trees], [dloss_t], dparams[n_trees, n_params, n_classes], ok) where
        ``dparams[t]`` is the gradient w.r.t. the parameter MATRIX — Zygote's
        ``grad.metadata._data.parameters`` (test/test_parametric_expression.jl:326-372).
        The library reduces class by class over sample ranges, so the samples are ordered by class
        first (a stable sort, done here unless ``grouped=True`` says the caller already did: classes are
        part of the dataset, so a search loop orders it once)."""
        kind = {"L2": 0, "L1": 1, "pullback": 2}[loss]
        mode = _grad_mode(variable)
        is_t = _is_torch(X)
        if is_t:
            import torch
            classes = torch.as_tensor(classes, device=X.device)
            if not grouped:
                order = torch.argsort(classes, stable=True)
                X, classes = X[:, order], classes[order]  # _prep_X makes the column gather feature-fastest again
                y = torch.as_tensor(y, device=X.device)[order]
                weights = None if weights is None else torch.as_tensor(weights, device=X.device)[order]
            counts = torch.bincount((classes - class_base).to(torch.int64), minlength=params.shape[1]).cpu().numpy()
        else:
            classes = np.asarray(classes)
            if not grouped:
                order = np.argsort(classes, kind="stable")
                X, classes = np.asfortranarray(np.asarray(X)[:, order]), classes[order]
                y = np.asarray(y)[order]
                weights = None if weights is None else np.asarray(weights)[order]
            counts = np.bincount((classes - class_base).astype(np.int64), minlength=params.shape[1])
        ptr, F, N, ldX, keep_x, is_t = _prep_X(X, self.dtype)
        if is_t:
            self.ctx.use_torch_stream()
        if F < self.n_features:
            raise ValueError(f"X has {F} features but the trees use feature {self.n_features}")
        keep = [keep_x]
        pa = self._param_args(params, classes, class_base, N, keep)
        if pa is None:
            raise ValueError("not a parametric population")
        n_cls = int(pa.n_classes)
        starts = np.zeros(n_cls + 1, dtype=np.int64)
        np.cumsum(counts[:n_cls], out=starts[1:])
        lib = library()
        ng = self._n_grad_all(mode)
        offs = np.zeros(self.n_trees + 1, dtype=np.int64)
        np.cumsum(ng, out=offs[1:])
        total = max(int(offs[-1]), 1)
        P = self.n_params

        def vec(v, name):
            if v is None:
                return None
            if is_t:
                import torch
                v = torch.as_tensor(v, dtype=keep_x.dtype, device=keep_x.device).contiguous()
                n, p_ = v.numel(), v.data_ptr()
            else:
                v = np.ascontiguousarray(v, dtype=self.dtype)
                n, p_ = v.size, v.ctypes.data
            if n != N:
                raise ValueError(f"{name} must have {N} entries")
            keep.append(v)
            return p_

        yp, wp = vec(y, "y"), vec(weights, "weights")
        if is_t:
            import torch
            kw = dict(dtype=keep_x.dtype, device=keep_x.device)
            lo, dl = torch.empty(self.n_trees, **kw), torch.empty(total, **kw)
            dp = torch.empty((self.n_trees, n_cls, P), **kw)
            ok = torch.empty(self.n_trees, dtype=torch.uint8, device=keep_x.device)
            self.ctx.check(lib.de_eval_loss_grad_by_class(self.ctx._h, self._h, ptr, N, ldX, C.byref(pa), mode, yp, wp, kind,
                                                          starts.ctypes.data, lo.data_ptr(), dl.data_ptr(), offs.ctypes.data,
                                                          dp.data_ptr(), ok.data_ptr()))
            return lo, list(torch.split(dl[:int(offs[-1])], ng.tolist())), dp.transpose(1, 2), ok.bool()
        lo, dl = np.empty(self.n_trees, dtype=self.dtype), np.empty(total, dtype=self.dtype)
        dp = np.empty((self.n_trees, n_cls, P), dtype=self.dtype)
        ok = np.zeros(self.n_trees, dtype=np.uint8)
        self.ctx.check(lib.de_eval_loss_grad_by_class(self.ctx._h, self._h, ptr, N, ldX, C.byref(pa), mode, yp, wp, kind,
                                                      starts.ctypes.data, lo.ctypes.data, dl.ctypes.data, offs.ctypes.data,
                                                      dp.ctypes.data, ok.ctypes.data))
        return lo, np.split(dl[:int(offs[-1])], offs[1:-1]), dp.transpose(0, 2, 1), ok.astype(bool)

    def eval_grad(self, X, variable: Union[bool, str] = False, params=None, classes=None,
                  class_base: int = 1):
        """All trees' forward-mode gradients.  Returns (out[n_trees,N], grads, ok) where
        grads is a list of per-tree [n_grad_t, N] Fortran-ordered arrays (views of one
        packed buffer)."""
        mode = _grad_mode(variable)
        ptr, F, N, ldX, keep_x, is_t = _prep_X(X, self.dtype)
        if is_t:
            self.ctx.use_torch_stream()
        if F < self.n_features:
            raise ValueError(f"X has {F} features but the trees use feature {self.n_features}")
        keep = [keep_x]
        pa = self._param_args(params, classes, class_base, N, keep)
        lib = library()
        ng = self._n_grad_all(mode)
        offs = np.zeros(self.n_trees + 1, dtype=np.int64)
        np.cumsum(ng * N, out=offs[1:])
        total = int(offs[-1])
        if is_t:
            import torch
            out = torch.empty((self.n_trees, N), dtype=keep_x.dtype, device=keep_x.device)
            grad = torch.empty(max(total, 1), dtype=keep_x.dtype, device=keep_x.device)
            ok = torch.empty(self.n_trees, dtype=torch.uint8, device=keep_x.device)
            self.ctx.check(lib.de_eval_grad(self.ctx._h, self._h, ptr, N, ldX, C.byref(pa) if pa else None, mode,
                                            out.data_ptr(), N, grad.data_ptr(), offs.ctypes.data, ok.data_ptr()))
            grads = [self._combine_rows(t, grad[offs[t]:offs[t + 1]].view(N, int(ng[t])).t(), mode) for t in range(self.n_trees)]
            return out, grads, ok.bool()
        out = np.empty((self.n_trees, N), dtype=self.dtype)
        grad = np.empty(max(total, 1), dtype=self.dtype)
        ok = np.zeros(self.n_trees, dtype=np.uint8)
        self.ctx.check(lib.de_eval_grad(self.ctx._h, self._h, ptr, N, ldX, C.byref(pa) if pa else None, mode,
                                        out.ctypes.data, N, grad.ctypes.data, offs.ctypes.data, ok.ctypes.data))
        grads = [self._combine_rows(t, grad[offs[t]:offs[t + 1]].reshape((int(ng[t]), N), order="F"), mode) for t in range(self.n_trees)]
        return out, grads, ok.astype(bool)

    def eval_pullback_dX(self, X, dY, params=None, classes=None, class_base: int = 1):
        """The ``dX`` of the ChainRules pullback (``EvalPullback``, src/ChainRules.jl:56-77) for every tree:
        ``dX[t][f, j] = d tree_t / d x_f (x_j) * dY[j]`` — ``dX_dY .* reshape(dY, 1, :)`` — NaN-filled where the
        evaluation is incomplete (:62-64).  Returns (dX[n_trees, n_rows, N], ok); rows = (params,) features.  The
        other half of the pullback, ``dtree``, is ``eval_loss_grad(..., loss="pullback")``."""
        ptr, F, N, ldX, keep_x, is_t = _prep_X(X, self.dtype)
        if is_t:
            self.ctx.use_torch_stream()
        if F < self.n_features:
            raise ValueError(f"X has {F} features but the trees use feature {self.n_features}")
        keep = [keep_x]
        pa = self._param_args(params, classes, class_base, N, keep)
        lib = library()
        G = self.n_params + self.n_features
        if is_t:
            import torch
            dy = torch.as_tensor(dY, dtype=keep_x.dtype, device=keep_x.device).contiguous()
            if dy.numel() != N:
                raise ValueError(f"dY must have {N} entries")
            dX = torch.empty((self.n_trees, N, G), dtype=keep_x.dtype, device=keep_x.device)
            ok = torch.empty(self.n_trees, dtype=torch.uint8, device=keep_x.device)
            self.ctx.check(lib.de_eval_pullback_dX(self.ctx._h, self._h, ptr, N, ldX, C.byref(pa) if pa else None,
                                                   dy.data_ptr(), dX.data_ptr(), None, ok.data_ptr()))
            return dX.transpose(1, 2), ok.bool()
        dy = np.ascontiguousarray(dY, dtype=self.dtype)
        if dy.size != N:
            raise ValueError(f"dY must have {N} entries")
        dX = np.empty((self.n_trees, N, G), dtype=self.dtype)
        ok = np.zeros(self.n_trees, dtype=np.uint8)
        self.ctx.check(lib.de_eval_pullback_dX(self.ctx._h, self._h, ptr, N, ldX, C.byref(pa) if pa else None,
                                               dy.ctypes.data, dX.ctypes.data, None, ok.ctypes.data))
        return dX.transpose(0, 2, 1), ok.astype(bool)

    def eval_diff(self, X, direction: int):
        """``direction`` is the 1-based feature index, as in the reference."""
        ptr, F, N, ldX, keep_x, is_t = _prep_X(X, self.dtype)
        if is_t:
            self.ctx.use_torch_stream()
        lib = library()
        if is_t:
            import torch
            out = torch.empty((self.n_trees, N), dtype=keep_x.dtype, device=keep_x.device)
            dout = torch.empty_like(out)
            ok = torch.empty(self.n_trees, dtype=torch.uint8, device=keep_x.device)
            self.ctx.check(lib.de_eval_diff(self.ctx._h, self._h, ptr, N, ldX, direction - 1, out.data_ptr(),
                                            dout.data_ptr(), N, ok.data_ptr()))
            return out, dout, ok.bool()
        out = np.empty((self.n_trees, N), dtype=self.dtype)
        dout = np.empty_like(out)
        ok = np.zeros(self.n_trees, dtype=np.uint8)
        self.ctx.check(lib.de_eval_diff(self.ctx._h, self._h, ptr, N, ldX, direction - 1, out.ctypes.data,
                                        dout.ctypes.data, N, ok.ctypes.data))
        return out, dout, ok.astype(bool)

    def close(self) -> None:
        if self._h:
            library().de_program_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):  # pragma: no cover
        try:
            self.close()
        except Exception:
            pass


def _grad_mode(variable) -> int:
    # variable::Union{Bool,Val}: true | false | Val(:both)   (src/EvaluateDerivative.jl:200-202)
    if variable is True:
        return GRAD_VARIABLE
    if variable is False:
        return GRAD_CONSTANT
    if variable in ("both", ":both"):
        return GRAD_BOTH
    raise ValueError("variable must be True, False or 'both'")


def _x_dtype(X, tree_dtype=None):
    if _is_torch(X):
        import torch
        return np.dtype(np.float64 if X.dtype == torch.float64 else np.float32)
    dt = np.asarray(X).dtype
    if dt == np.float64:
        return np.dtype(np.float64)
    if dt == np.float32:
        return np.dtype(np.float32)
    if dt.kind in "iu":
        return np.dtype(np.float64)
    raise TypeError(f"MI355X back-end supports Float32/Float64, got {dt}")


def eval_tree_array(tree: Node, cX, operators: OperatorEnum, eval_context: Optional[EvalContext] = None,
                    ctx: Optional[Context] = None):
    """``eval_tree_array(tree, cX, operators; eval_context) -> (output, complete)``."""
    if isinstance(tree, ParametricExpression):
        raise ValueError("Incorrect call. You must pass the `classes::Vector` argument when calling `eval_tree_array`.")
    F = np.asarray(cX.shape)[0] if not _is_torch(cX) else cX.shape[0]
    pop = Population([tree], operators, _x_dtype(cX), n_features=int(F), eval_context=eval_context, ctx=ctx)
    try:
        out, ok = pop.eval(cX)
        if pop.eval_context.strict_flags and len(pop.uncertified):
            raise UncertifiedFlag("strict_flags: the flag of this tree is not provably the reference's isfinite(sum(x)) flag on this X "
                                  "(a finite tested array whose sum may overflow): keep the CPU path for it")
        return out[0], bool(ok[0])
    finally:
        pop.close()


def eval_grad_tree_array(tree: Node, cX, operators: OperatorEnum, variable: Union[bool, str] = False,
                         turbo: bool = False, ctx: Optional[Context] = None):
    """``eval_grad_tree_array(tree, cX, operators; variable) -> (evaluation, gradient, complete)``."""
    F = cX.shape[0]
    pop = Population([tree], operators, _x_dtype(cX), n_features=int(F), ctx=ctx)
    try:
        out, grads, ok = pop.eval_grad(cX, variable)
        return out[0], grads[0], bool(ok[0])
    finally:
        pop.close()


def eval_diff_tree_array(tree: Node, cX, operators: OperatorEnum, direction: int, turbo: bool = False,
                         ctx: Optional[Context] = None):
    """``eval_diff_tree_array(tree, cX, operators, direction) -> (evaluation, derivative, complete)``."""
    F = cX.shape[0]
    pop = Population([tree], operators, _x_dtype(cX), n_features=int(F), ctx=ctx)
    try:
        out, dout, ok = pop.eval_diff(cX, direction)
        return out[0], dout[0], bool(ok[0])
    finally:
        pop.close()


class Expression:
    """``Expression(tree; operators)`` callable sugar (src/Expression.jl:435-520,
    src/EvaluationHelpers.jl:29-33): ``ex(X)`` returns the output with NaN-fill when
    incomplete; validates ``max_feature(ex) <= size(X, 1)`` (:401-409)."""

    def __init__(self, tree: Node, operators: OperatorEnum):
        self.tree, self.operators = tree, operators

    def __call__(self, X, eval_context: Optional[EvalContext] = None):
        if max_feature(self.tree) > X.shape[0]:
            raise ValueError("expression references a feature beyond size(X, 1)")
        out, ok = eval_tree_array(self.tree, X, self.operators, eval_context)
        if not ok:
            out[...] = float("nan")  # set_nan!, src/Utils.jl:73-76
        return out

    def grad(self, X, variable: Union[bool, str] = True):
        """``ex'(X)`` (src/EvaluationHelpers.jl:56-62): gradient with NaN-fill."""
        _, g, ok = eval_grad_tree_array(self.tree, X, self.operators, variable)
        if not ok:
            g[...] = float("nan")
        return g


class ParametricExpression:
    """``ParametricExpression(tree; operators, parameters)`` (src/ParametricExpression.jl:83-116)."""

    def __init__(self, tree: Node, operators: OperatorEnum, parameters):
        self.tree, self.operators = tree, operators
        self.parameters = np.asfortranarray(parameters)

    def eval_tree_array(self, X, classes, eval_context: Optional[EvalContext] = None,
                        ctx: Optional[Context] = None):
        """``eval_tree_array(ex, X, classes) -> (output, complete)``; classes are 1-based."""
        dt = _x_dtype(X)
        pop = Population([self.tree], self.operators, dt, n_features=int(X.shape[0]),
                         n_params=self.parameters.shape[0], eval_context=eval_context, ctx=ctx)
        try:
            out, ok = pop.eval(X, self.parameters.astype(dt), classes, class_base=1)
            return out[0], bool(ok[0])
        finally:
            pop.close()

    def __call__(self, X, classes=None, **kw):
        if classes is None:
            raise ValueError("Incorrect call. You must pass the `classes::Vector` argument when calling `eval_tree_array`.")
        out, ok = self.eval_tree_array(X, classes, **kw)
        if not ok:
            out[...] = float("nan")
        return out

    def get_scalar_constants(self):
        """tree constants then parameters[:] (src/ParametricExpression.jl:258-267)."""
        from .node import get_scalar_constants
        cs, refs = get_scalar_constants(self.tree)
        return np.concatenate([cs, self.parameters.reshape(-1, order="F")]), refs
