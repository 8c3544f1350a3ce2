"""ctypes wrapper of the CPU oracle (oracle/libde_oracle.so).

TEST INFRASTRUCTURE ONLY: imported by tests/, by bench.py's cpu_baseline leg and by
__graft_entry__.smoke() as the checker — never by the product package.  See the header of
oracle/de_oracle.c for what the oracle restates and how far it is pinned.

All functions take the same (tape, consts) the C ABI takes, so a test compares
``libde_hip`` and the oracle on literally the same bytes.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess
from typing import Optional, Tuple

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "libde_oracle.so")

OPT_EARLY_EXIT, OPT_FUSE_DEG1, OPT_FUSE_DEG2, OPT_BUMPER = 1, 2, 4, 8
OPT_DEFAULT = 7
GRAD_VARIABLE, GRAD_CONSTANT, GRAD_BOTH = 0, 1, 2


def build(force: bool = False) -> str:
    """Compile the oracle with gcc (oracle/Makefile)."""
    if force or not os.path.exists(_LIB_PATH) or any(
            os.path.getmtime(os.path.join(_HERE, f)) > os.path.getmtime(_LIB_PATH)
            for f in ("de_oracle.c", "de_oracle_impl.h", "de_oracle_ops.h")):
        subprocess.run(["make", "-C", _HERE, "-s"] + (["-B"] if force else []), check=True)
    return _LIB_PATH


_lib: Optional[C.CDLL] = None


def lib() -> C.CDLL:
    global _lib
    if _lib is None:
        if not os.path.exists(_LIB_PATH):
            build()
        _lib = C.CDLL(_LIB_PATH)
        for sfx, ct in (("f32", C.c_float), ("f64", C.c_double)):
            getattr(_lib, f"de_oracle_unary_{sfx}").restype = ct
            getattr(_lib, f"de_oracle_unary_{sfx}").argtypes = [C.c_int, ct]
            getattr(_lib, f"de_oracle_binary_{sfx}").restype = ct
            getattr(_lib, f"de_oracle_binary_{sfx}").argtypes = [C.c_int, ct, ct]
            getattr(_lib, f"de_oracle_ternary_{sfx}").restype = ct
            getattr(_lib, f"de_oracle_ternary_{sfx}").argtypes = [C.c_int, ct, ct, ct]
            getattr(_lib, f"de_oracle_unary_grad_{sfx}").argtypes = [C.c_int, ct, C.c_void_p]
            getattr(_lib, f"de_oracle_binary_grad_{sfx}").argtypes = [C.c_int, ct, ct, C.c_void_p]
    return _lib


def _sfx(dtype) -> str:
    dtype = np.dtype(dtype)
    if dtype == np.float32:
        return "f32"
    if dtype == np.float64:
        return "f64"
    raise TypeError("oracle supports float32/float64")


def _p(a: Optional[np.ndarray]):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def _prep(tape, consts, X, dtype):
    dtype = np.dtype(dtype)
    tape = np.ascontiguousarray(tape)
    consts = np.ascontiguousarray(consts, dtype=dtype)
    X = np.asarray(X, dtype=dtype)
    if X.ndim != 2:
        raise ValueError("X must be [n_features, N]")
    Xf = np.asfortranarray(X)  # element (f, j) at f + F*j
    return tape, consts, Xf


def _check(rc: int):
    if rc != 0:
        raise ValueError({-2: "bad tape", -3: "unsupported opcode", -6: "index out of range",
                          -100: "out of memory"}.get(rc, f"oracle error {rc}"))


def eval_tree_array(tape, consts, X, options: int = OPT_DEFAULT, elementwise: bool = False
                    ) -> Tuple[np.ndarray, bool]:
    """Reference eval_tree_array (src/Evaluate.jl:279-309) on one tape.  ``elementwise``
    replaces is_valid_array's isfinite(sum(x)) by all(isfinite, x)."""
    dtype = np.asarray(X).dtype
    tape, consts, Xf = _prep(tape, consts, X, dtype)
    F, N = Xf.shape
    out = np.empty(N, dtype=dtype)
    ok = C.c_uint8(0)
    fn = getattr(lib(), f"de_oracle_eval_{_sfx(dtype)}")
    rc = fn(_p(tape), C.c_int64(len(tape)), _p(consts), C.c_int64(len(consts)), _p(Xf),
            C.c_int32(F), C.c_int64(N), C.c_int64(F), C.c_uint32(options),
            C.c_int32(int(elementwise)), _p(out), C.byref(ok))
    _check(rc)
    return out, bool(ok.value)


def eval_tree_array_parametric(tape, consts, X, params, classes, class_base: int = 1,
                               options: int = OPT_DEFAULT, elementwise: bool = False):
    """eval_tree_array(ex::ParametricExpression, X, classes) (src/ParametricExpression.jl:371-390)."""
    dtype = np.asarray(X).dtype
    tape, consts, Xf = _prep(tape, consts, X, dtype)
    F, N = Xf.shape
    params = np.asfortranarray(np.asarray(params, dtype=dtype))
    P, ncls = params.shape
    classes = np.ascontiguousarray(classes, dtype=np.int32)
    out = np.empty(N, dtype=dtype)
    ok = C.c_uint8(0)
    fn = getattr(lib(), f"de_oracle_eval_param_{_sfx(dtype)}")
    rc = fn(_p(tape), C.c_int64(len(tape)), _p(consts), C.c_int64(len(consts)), _p(Xf),
            C.c_int32(F), C.c_int64(N), C.c_int64(F), _p(params), C.c_int32(P), C.c_int64(ncls),
            C.c_int64(P), _p(classes), C.c_int32(class_base), C.c_uint32(options),
            C.c_int32(int(elementwise)), _p(out), C.byref(ok))
    _check(rc)
    return out, bool(ok.value)


def parametric_to_plain(tape, X, params, classes, class_base: int = 1):
    """The reference's own reduction of the parametric case to the plain one
    (src/ParametricExpression.jl:381-389): gather parameters by class above X and re-index
    leaves (parameter p -> feature p, feature f -> f+P).  Used to run the gradient oracle on
    parametric trees."""
    X = np.asarray(X)
    params = np.asarray(params, dtype=X.dtype)
    P = params.shape[0]
    cls = np.asarray(classes, dtype=np.int64) - class_base
    PX = np.asfortranarray(np.vstack([params[:, cls], X]))
    t2 = np.array(tape, copy=True)
    leaf = t2["degree"] == 0
    isp = leaf & (t2["op"] == 2)
    isf = leaf & (t2["op"] == 1)
    t2["arg"][isf] += P
    t2["op"][isp] = 1
    return t2, PX


def eval_grad_tree_array(tape, consts, X, mode: int, elementwise: bool = False):
    """Reference eval_grad_tree_array (src/EvaluateDerivative.jl:193-228):
    returns (out[N], grad[n_grad, N], ok)."""
    dtype = np.asarray(X).dtype
    tape, consts, Xf = _prep(tape, consts, X, dtype)
    F, N = Xf.shape
    nc = int(np.sum((tape["degree"] == 0) & (tape["op"] == 0)))
    G = {GRAD_VARIABLE: F, GRAD_CONSTANT: nc, GRAD_BOTH: F + nc}[mode]
    out = np.empty(N, dtype=dtype)
    grad = np.zeros((G, N), dtype=dtype, order="F")
    ok = C.c_uint8(0)
    ng = C.c_int64(0)
    fn = getattr(lib(), f"de_oracle_grad_{_sfx(dtype)}")
    rc = fn(_p(tape), C.c_int64(len(tape)), _p(consts), C.c_int64(len(consts)), _p(Xf),
            C.c_int32(F), C.c_int64(N), C.c_int64(F), C.c_int32(mode), C.c_int32(int(elementwise)),
            _p(out), _p(grad), C.byref(ok), C.byref(ng))
    _check(rc)
    assert ng.value == G
    return out, grad, bool(ok.value)


def eval_diff_tree_array(tape, consts, X, direction0: int):
    """Reference eval_diff_tree_array (src/EvaluateDerivative.jl:40-53); 0-based direction."""
    dtype = np.asarray(X).dtype
    tape, consts, Xf = _prep(tape, consts, X, dtype)
    F, N = Xf.shape
    out = np.empty(N, dtype=dtype)
    dout = np.empty(N, dtype=dtype)
    ok = C.c_uint8(0)
    fn = getattr(lib(), f"de_oracle_diff_{_sfx(dtype)}")
    rc = fn(_p(tape), C.c_int64(len(tape)), _p(consts), C.c_int64(len(consts)), _p(Xf),
            C.c_int32(F), C.c_int64(N), C.c_int64(F), C.c_int32(direction0), _p(out), _p(dout),
            C.byref(ok))
    _check(rc)
    return out, dout, bool(ok.value)


def unary(op: int, x, dtype=np.float64):
    f = getattr(lib(), f"de_oracle_unary_{_sfx(dtype)}")
    return f(op, float(x))


def binary(op: int, x, y, dtype=np.float64):
    f = getattr(lib(), f"de_oracle_binary_{_sfx(dtype)}")
    return f(op, float(x), float(y))


def ternary(op: int, x, y, z, dtype=np.float64):
    f = getattr(lib(), f"de_oracle_ternary_{_sfx(dtype)}")
    return f(op, float(x), float(y), float(z))


def unary_grad(op: int, x, dtype=np.float64) -> float:
    g = np.zeros(1, dtype=dtype)
    getattr(lib(), f"de_oracle_unary_grad_{_sfx(dtype)}")(op, float(x), _p(g))
    return float(g[0])


def binary_grad(op: int, x, y, dtype=np.float64):
    g = np.zeros(2, dtype=dtype)
    getattr(lib(), f"de_oracle_binary_grad_{_sfx(dtype)}")(op, float(x), float(y), _p(g))
    return float(g[0]), float(g[1])
