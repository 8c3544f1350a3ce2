"""A numpy model of the device accumulator machine (csrc/de_program.h), used ONLY by the CPU
unit tests of the host lowering (tests/test_lowering.py): it executes the instruction words
``de_lower_tape`` returns, so operand order / SWAP / spill slots / check flags can be
verified against the oracle without a GPU.  It is test infrastructure — the product never
imports it and it is not a fallback path.
"""
import numpy as np

DOP_LOAD, DOP_RSUB, DOP_RDIV = 0xF0, 0xF1, 0xF2
SRC_ACC, SRC_ROW, SRC_CONST, SRC_PARAM = 0, 1, 2, 4


def _jlmax(x, y):
    r = np.where((y > x) | (np.signbit(x) & ~np.signbit(y)), y, x)
    r = np.where(np.isnan(y), y, r)
    return np.where(np.isnan(x), x, r)


def _jlmin(x, y):
    r = np.where((y < x) | (np.signbit(y) & ~np.signbit(x)), y, x)
    r = np.where(np.isnan(y), y, r)
    return np.where(np.isnan(x), x, r)


def _jlmod(x, y):
    r = np.fmod(x, y)
    out = np.where((r > 0) != (y > 0), r + y, r)
    return np.where(r == 0, np.copysign(r, y), out)


def _sign(x):
    return np.where(x > 0, 1, np.where(x < 0, -1, x)).astype(x.dtype)


def _nan_where(cond, val):
    return np.where(cond, np.nan, val).astype(val.dtype)


UNARY = {
    1: lambda x: -x, 2: np.abs, 3: lambda x: x * x, 4: lambda x: (x * x) * x,
    5: lambda x: np.where(x < 0, 0, x).astype(x.dtype), 6: _sign, 7: np.rint, 8: np.floor, 9: np.ceil,
    10: lambda x: 1 / x, 11: np.sqrt, 12: np.cbrt, 13: np.exp, 14: np.exp2, 15: np.log, 16: np.log2,
    17: np.log10, 18: np.log1p, 19: np.sin, 20: np.cos, 21: np.tan, 22: np.sinh, 23: np.cosh,
    24: np.tanh, 25: np.arcsin, 26: np.arccos, 27: np.arctan, 28: np.arcsinh, 29: np.arccosh,
    30: np.arctanh,
    31: lambda x: _nan_where(x <= 0, np.log(np.where(x <= 0, 1, x))),
    32: lambda x: _nan_where(x <= 0, np.log2(np.where(x <= 0, 1, x))),
    33: lambda x: _nan_where(x <= 0, np.log10(np.where(x <= 0, 1, x))),
    34: lambda x: _nan_where(x <= -1, np.log1p(np.where(x <= -1, 0, x))),
    35: lambda x: _nan_where(x < 0, np.sqrt(np.where(x < 0, 0, x))),
    36: lambda x: _nan_where(x < 1, np.arccosh(np.where(x < 1, 1, x))),
    37: lambda x: np.cos(x) * np.cos(x),
    38: lambda x: np.vectorize(__import__('math').gamma, otypes=[np.float64])(x.astype(np.float64)),
}
BINARY = {
    64: lambda x, y: x + y, 65: lambda x, y: x - y, 66: lambda x, y: x * y, 67: lambda x, y: x / y,
    68: lambda x, y: np.power(x, y), 69: _jlmax, 70: _jlmin, 71: _jlmod, 72: np.fmod,
    73: lambda x, y: (x > y).astype(x.dtype), 74: lambda x, y: np.exp(y * np.log(np.abs(x))),
    DOP_RSUB: lambda x, y: y - x, DOP_RDIV: lambda x, y: y / x,
}
for _r, _f in ((0xF3, 68), (0xF4, 71), (0xF5, 72), (0xF6, 73), (0xF7, 74)):
    BINARY[_r] = (lambda f: lambda x, y: f(y, x))(BINARY[_f])
TERNARY = {
    128: lambda x, y, z: x * y + z,  # (fma: single rounding on device; tolerance in the test)
    129: lambda x, y, z: np.where(x > z, z, np.where(x < y, y, x)),
    130: lambda x, y, z: (x + y) + z,
    131: lambda x, y, z: _jlmax(_jlmax(x, y), z),
}


def run(words, X, early_exit=True, params=None, classes0=None, host_ok=True, noise_eps=0.0, rng=None, select_log=None, value_log=None, overflow_at=None):
    """Execute instruction words ([n,4] uint32) on X [F, N].  Returns (out, ok).

    ``noise_eps`` > 0 multiplies every operator result by (1 +- u*noise_eps), u uniform in [1/4, 1], with a random sign
    per sample (discrete stochastic arithmetic): the spread of the outputs over a few such runs
    measures how strongly a sample amplifies a one-rounding-error difference between two
    implementations of the same operator (used for the parity tolerance, helpers.py).

    ``select_log`` (a list) receives, in program order, the operand pair (x, y) of every SELECTING operator (abs / relu / sign / floor / ceil / round: the operand and the edge it is compared with; max, min,
    greater, clamp, max3): helpers.unstable_selections compares the pairs of a clean and of the perturbed runs.

    ``overflow_at`` (the element type's largest finite value when X is wider than the element type being modelled): an operator result
    beyond it becomes +-Inf, as it does in the element type — exp(x1 log|x3|) = 4.6e38 is Inf in Float32, and everything downstream of it
    (min(., safe_log(Inf)) ...) follows the Float32 evaluation, not the float64 one (fuzz seed 61, profiles/r5_value_findings_61.jsonl).

    ``value_log`` (a list) receives every operator RESULT in program order (float64): helpers.parity_tolerance compares the intermediates
    of the perturbed runs with the clean ones to find the samples that sit behind a chaotic intermediate."""
    dt = X.dtype
    N = X.shape[1]
    acc = np.zeros(N, dtype=dt)
    stack = {}
    bad = False
    with np.errstate(all="ignore"):
        for w in words:
            hdr, feat = int(w[0]), int(w[1])
            op, src = hdr & 0xFF, (hdr >> 8) & 7
            F = X.shape[0]
            if hdr & (1 << 11):
                stack[F + ((hdr >> 20) & 15)] = acc.copy()
            if src == SRC_ROW:
                row = feat & 0xFFFF
                b = X[row].copy() if row < F else stack[row].copy()
            elif src == SRC_CONST:
                imm = np.array([w[2], w[3]], dtype=np.uint32)
                c = imm[:1].view(np.float32)[0] if dt == np.float32 else imm.view(np.float64)[0]
                b = np.full(N, c, dtype=dt)
            elif src == SRC_PARAM:
                b = params[feat & 0xFFFF, classes0].astype(dt)
            else:
                b = acc.copy()
            if early_exit and (hdr & (1 << 12)):
                bad |= bool(np.any(~np.isfinite(b)))
            if 128 <= op < DOP_LOAD:
                c2 = stack[F + ((hdr >> 24) & 15)]
                if select_log is not None and op in (129, 131):
                    select_log.append((b.astype(np.float64), c2.astype(np.float64)))
                    select_log.append((_jlmax(b, c2).astype(np.float64) if op == 131 else b.astype(np.float64), acc.astype(np.float64)))
                acc = TERNARY[op](b, c2, acc).astype(dt)
            else:
                if op == DOP_LOAD:
                    acc = b
                elif op < 64:
                    if select_log is not None and op in (2, 5, 6, 7, 8, 9):
                        # unary operators that SELECT: abs / relu / sign against 0 (their derivative — for sign the value — jumps there),
                        # floor / ceil against the nearest integer, round against the nearest half-integer (the value jumps)
                        b64 = b.astype(np.float64)
                        edge = np.zeros_like(b64) if op in (2, 5, 6) else (np.rint(b64) if op in (8, 9) else np.floor(b64) + 0.5)
                        select_log.append((b64, edge))
                    acc = UNARY[op](b).astype(dt)
                else:
                    if select_log is not None and op in (69, 70, 73, 0xF6):
                        select_log.append((acc.astype(np.float64), b.astype(np.float64)))
                    acc = BINARY[op](acc, b).astype(dt)
                if (not early_exit) and (hdr & (1 << 14)):
                    acc = np.where(np.isfinite(b), acc, np.inf).astype(dt)
            if overflow_at is not None and op != DOP_LOAD:
                acc = np.where(np.abs(acc) > overflow_at, np.copysign(np.inf, acc), acc).astype(dt)
            if noise_eps and op != DOP_LOAD:
                # random sign AND magnitude in [1/4, 1] of noise_eps: a fixed step can land on a period of the
                # function downstream (1.05e8 * 2^-23 = 12.55 ~ 4*pi hid a chaotic cos(exp(x)) sample)
                step = rng.choice(np.array([-1.0, 1.0]), size=N) * rng.uniform(0.25, 1.0, size=N)
                acc = (acc * (1.0 + noise_eps * step)).astype(dt)
                if op in (74, 0xF7):  # pow_abs2 = exp(y * log|x|) is three roundings; exp amplifies the inner two by |m|
                    with np.errstate(all="ignore"):
                        m = np.abs(np.log(np.abs(acc.astype(np.float64))))
                    m = np.where(np.isfinite(m), m, 0.0)
                    step2 = rng.choice(np.array([-1.0, 1.0]), size=N) * rng.uniform(0.25, 1.0, size=N)
                    acc = (acc * (1.0 + noise_eps * 2.0 * m * step2)).astype(dt)
            if value_log is not None and op != DOP_LOAD:
                value_log.append(acc.astype(np.float64))
            check = bool(hdr & (1 << 15)) if early_exit else bool(hdr & (1 << 13))
            if check:
                bad |= bool(np.any(~np.isfinite(acc)))
    return acc, (host_ok and not bad)
