"""OperatorEnum mirror + the function-name -> opcode registry.

Reference: ``OperatorEnum{OPS}`` is a tuple-of-tuples of Julia functions indexed
``[degree][op_idx]`` (src/OperatorEnum.jl:14-49), ``Node.op`` is the 1-based index into
the tuple of the node's degree (src/Node.jl:80).  A GPU kernel cannot call a Julia
closure, so at the boundary every function is mapped, by its Julia name, onto the closed
opcode set of ``include/de_opcodes.h``; an unknown function raises
``UnsupportedOperatorError`` (the Julia shim then keeps the reference CPU path).

The authoritative table lives in the C library (``de_opcode_by_name``);
``tests/test_abi.py`` checks this Python copy against it.
"""
from __future__ import annotations

from typing import Dict, Iterable, Sequence, Tuple

OPERATOR_LIMIT_BEFORE_SLOWDOWN = 15  # src/Evaluate.jl:14

# (julia name, degree) -> opcode id  (ids: include/de_opcodes.h, table version 1)
_U = [
    "neg", "abs", "square", "cube", "relu", "sign", "round", "floor", "ceil", "inv", "sqrt",
    "cbrt", "exp", "exp2", "log", "log2", "log10", "log1p", "sin", "cos", "tan", "sinh", "cosh",
    "tanh", "asin", "acos", "atan", "asinh", "acosh", "atanh", "safe_log", "safe_log2",
    "safe_log10", "safe_log1p", "safe_sqrt", "safe_acosh", "custom_cos", "gamma",
]
_B = ["+", "-", "*", "/", "^", "max", "min", "mod", "rem", "greater", "pow_abs2"]
_T = ["fma", "clamp", "+", "max"]

OPCODES: Dict[Tuple[str, int], int] = {}
for _i, _n in enumerate(_U):
    OPCODES[(_n, 1)] = 1 + _i
for _i, _n in enumerate(_B):
    OPCODES[(_n, 2)] = 64 + _i
for _i, _n in enumerate(_T):
    OPCODES[(_n, 3)] = 128 + _i
# aliases: unary minus is the Julia function `-` of degree 1; `sub` is test_params.jl:14
OPCODES[("-", 1)] = OPCODES[("neg", 1)]
OPCODES[("sub", 2)] = OPCODES[("-", 2)]
OPCODES[("add", 2)] = OPCODES[("+", 2)]
OPCODES[("mult", 2)] = OPCODES[("*", 2)]
OPCODES[("div", 2)] = OPCODES[("/", 2)]
OPCODES[("pow", 2)] = OPCODES[("^", 2)]

OPCODE_NAMES: Dict[int, str] = {}
for (_n, _d), _c in OPCODES.items():
    OPCODE_NAMES.setdefault(_c, _n)


class UnsupportedOperatorError(ValueError):
    """A function of the OperatorEnum has no device opcode (DE_ERR_UNSUPPORTED_OP)."""


class OperatorEnum:
    """``OperatorEnum(; binary_operators, unary_operators)`` / ``OperatorEnum(1 => (...), 2 => (...))``.

    Operators are given by their Julia names (``"cos"``, ``"+"``, ``"safe_log"`` ...).
    ``ops[degree-1][op_idx-1]`` is the name of ``operators[degree][op_idx]``.
    """

    def __init__(self, binary_operators: Sequence[str] = (), unary_operators: Sequence[str] = (),
                 ternary_operators: Sequence[str] = ()):
        self.ops: Tuple[Tuple[str, ...], ...] = (
            tuple(unary_operators), tuple(binary_operators), tuple(ternary_operators))
        for d, names in enumerate(self.ops, start=1):
            if len(names) > 255:  # op::UInt8, src/Node.jl:80
                raise ValueError("at most 255 operators per degree")

    @property
    def unaops(self) -> Tuple[str, ...]:  # src/OperatorEnum.jl:40-49
        return self.ops[0]

    @property
    def binops(self) -> Tuple[str, ...]:
        return self.ops[1]

    def nops(self, degree: int) -> int:  # get_nops, src/Evaluate.jl:331-335
        return len(self.ops[degree - 1]) if 1 <= degree <= len(self.ops) else 0

    def opcode(self, degree: int, op_idx: int) -> int:
        """1-based (degree, op_idx) -> device opcode; raises like get_op (src/Evaluate.jl:408-419)."""
        names = self.ops[degree - 1] if 1 <= degree <= len(self.ops) else ()
        if not names:
            raise ValueError(
                f"Invalid access: a node has degree {degree}, but no operators were passed for this degree.")
        if not (1 <= op_idx <= len(names)):
            raise IndexError(f"operator index {op_idx} out of range for degree {degree}")
        name = names[op_idx - 1]
        code = OPCODES.get((name, degree))
        if code is None:
            raise UnsupportedOperatorError(
                f"operator {name!r} of degree {degree} has no MI355X opcode; keep the CPU path")
        return code

    def index(self, name: str, degree: int) -> int:
        """1-based op index of a function name (what Node(; op=...) stores)."""
        return self.ops[degree - 1].index(name) + 1

    def fuse_flags(self, use_fused: bool = True) -> Tuple[bool, bool]:
        """(fuse_deg1, fuse_deg2): fusion is skipped when a degree has more than 15
        operators (src/Evaluate.jl:496,607) or use_fused=false (:511,619)."""
        return (use_fused and self.nops(1) <= OPERATOR_LIMIT_BEFORE_SLOWDOWN,
                use_fused and self.nops(2) <= OPERATOR_LIMIT_BEFORE_SLOWDOWN)


def opcode_degree(code: int) -> int:
    return 1 if code < 64 else (2 if code < 128 else 3)


def all_opcodes() -> Iterable[Tuple[str, int, int]]:
    for (n, d), c in sorted(OPCODES.items(), key=lambda kv: kv[1]):
        yield n, d, c
