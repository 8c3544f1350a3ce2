"""A linter for the Julia half of the boundary, which this image cannot execute (no Julia): every
``ccall((:name, LIBDE), ret, (types...), args...)`` of ``dynamicexpressions.jl_amd/julia/*.jl`` is parsed and checked against the
prototype of the same name in ``include/de_hip.h`` — the symbol exists, the arity agrees (types AND arguments), the return type and
every argument has the width / pointer-ness the C side declares — and the Julia mirrors of the two C structs that cross the ABI
(``TapeNode`` = ``de_tape_node_t``, ``ParamArgs`` = ``de_param_args_t``) have the same fields in the same order with the same sizes.
(the row SURVEY §8f-2 stays "partial" until somebody runs julia/runtests_hip.jl.)"""
import glob
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "de_hip.h")
JULIA = sorted(glob.glob(os.path.join(ROOT, "dynamicexpressions.jl_amd", "julia", "*.jl")))

# class of a C type: ("ptr", None) or ("int" | "float", bytes) or ("void", 0)
C_SCALARS = {"int": ("int", 4), "int32_t": ("int", 4), "uint32_t": ("int", 4), "int64_t": ("int", 8), "uint64_t": ("int", 8),
             "size_t": ("int", 8), "uint8_t": ("int", 1), "uint16_t": ("int", 2), "float": ("float", 4), "double": ("float", 8),
             "void": ("void", 0)}
JL_TYPES = {"Cint": ("int", 4), "Int32": ("int", 4), "UInt32": ("int", 4), "Int64": ("int", 8), "UInt64": ("int", 8), "Csize_t": ("int", 8),
            "UInt8": ("int", 1), "UInt16": ("int", 2), "Cfloat": ("float", 4), "Float32": ("float", 4), "Cdouble": ("float", 8),
            "Float64": ("float", 8), "Cvoid": ("void", 0), "Cstring": ("ptr", None)}


def c_class(t):
    t = re.sub(r"\bconst\b", "", t).strip()
    if "*" in t:
        return ("ptr", None)
    return C_SCALARS[t.split()[-1] if t.split()[-1] in C_SCALARS else t]


def jl_class(t):
    t = t.strip()
    if t.startswith(("Ptr{", "Ref{")):
        return ("ptr", None)
    return JL_TYPES[t]


def strip_c_comments(s):
    return re.sub(r"/\*.*?\*/", " ", s, flags=re.S)


def prototypes():
    src = strip_c_comments(open(HEADER).read())
    out = {}
    for m in re.finditer(r"^\s*((?:const\s+)?[A-Za-z_][\w ]*?[\s\*]+)(de_\w+)\s*\(([^;{]*?)\)\s*;", src, re.M):
        ret, name, params = m.group(1).strip(), m.group(2), " ".join(m.group(3).split())
        if params in ("void", ""):
            plist = []
        else:
            plist = []
            for prm in params.split(","):
                prm = prm.strip()
                mt = re.match(r"(.*?)(\w+)$", prm)  # type + parameter name
                plist.append(mt.group(1).strip())
        out[name] = (c_class(ret), [c_class(t) for t in plist])
    return out


def split_top(s):
    """Split at the top-level commas of `s` (parentheses, brackets and braces nest)."""
    parts, depth, cur = [], 0, []
    for ch in s:
        if ch in "([{":
            depth += 1
        elif ch in ")]}":
            depth -= 1
        if ch == "," and depth == 0:
            parts.append("".join(cur).strip())
            cur = []
        else:
            cur.append(ch)
    if "".join(cur).strip():
        parts.append("".join(cur).strip())
    return parts


def ccalls(path):
    src = re.sub(r"#[^\n]*", "", open(path).read())  # (no string in these files holds a '#')
    for m in re.finditer(r"ccall\(", src):
        i, depth = m.end(), 1
        while depth:
            depth += {"(": 1, ")": -1}.get(src[i], 0)
            i += 1
        body = src[m.end():i - 1]
        parts = split_top(body)
        sym = re.match(r"\(\s*:(\w+)\s*,\s*LIBDE\s*\)", parts[0])
        if not sym:
            continue
        types = parts[2].strip()
        assert types.startswith("(") and types.endswith(")"), (path, parts[0], types)
        tl = split_top(types[1:-1])
        yield sym.group(1), parts[1], tl, parts[3:], src.count("\n", 0, m.start()) + 1


def test_header_parses():
    protos = prototypes()
    assert len(protos) >= 40 and "de_eval" in protos and "de_eval_loss_grad_by_class" in protos
    assert protos["de_eval"][1][0] == ("ptr", None) and protos["de_eval"][1][3] == ("int", 8)


@pytest.mark.parametrize("path", JULIA, ids=[os.path.basename(p) for p in JULIA])
def test_every_ccall_matches_its_prototype(path):
    protos = prototypes()
    n = 0
    for name, ret, types, args, line in ccalls(path):
        where = f"{os.path.basename(path)}:{line} ccall(:{name})"
        assert name in protos, f"{where}: no such function in include/de_hip.h"
        cret, cparams = protos[name]
        assert len(types) == len(cparams), f"{where}: {len(types)} argument types, the prototype has {len(cparams)}"
        assert len(args) == len(types), f"{where}: {len(args)} arguments for {len(types)} argument types"
        assert jl_class(ret) == cret, f"{where}: return type {ret} against {cret}"
        for k, (jt, ct) in enumerate(zip(types, cparams)):
            assert jl_class(jt) == ct, f"{where}: argument {k + 1} is {jt}, the prototype wants {ct}"
        n += 1
    if os.path.basename(path) == "DynamicExpressionsHIPExt.jl":
        assert n >= 25, n  # the shim's entry points are all there


def c_struct(name):
    src = strip_c_comments(open(HEADER).read())
    m = re.search(r"typedef\s+struct\s+\w+\s*\{([^}]*)\}\s*" + name + r"\s*;", src)
    fields = []
    for decl in m.group(1).split(";"):
        decl = " ".join(decl.split())
        if decl:
            mt = re.match(r"(.*?)(\w+)$", decl)
            fields.append((mt.group(2), c_class(mt.group(1))))
    return fields


def jl_struct(name):
    src = open(os.path.join(ROOT, "dynamicexpressions.jl_amd", "julia", "DynamicExpressionsHIPExt.jl")).read()
    m = re.search(r"^struct\s+" + name + r"\b[^\n]*\n(.*?)^end", src, re.M | re.S)
    fields = []
    for line in m.group(1).splitlines():
        line = line.split("#")[0].strip()
        if line:
            fname, ftype = line.split("::")
            fields.append((fname.strip(), jl_class(ftype)))
    return fields


@pytest.mark.parametrize("jl,c", [("TapeNode", "de_tape_node_t"), ("ParamArgs", "de_param_args_t")])
def test_struct_mirrors_have_the_c_layout(jl, c):
    jf, cf = jl_struct(jl), c_struct(c)
    assert [n for n, _ in jf] == [n for n, _ in cf], (jf, cf)
    assert [t for _, t in jf] == [t for _, t in cf], (jf, cf)


def test_every_population_entry_point_takes_the_context_lock():
    """ a de_ctx_t is not thread-safe and finalizers run on GC threads: outside `with_ctx` / `with_pop` (and the
    two finalizers, which trylock) no ccall may touch a context or a program."""
    src = open(os.path.join(ROOT, "dynamicexpressions.jl_amd", "julia", "DynamicExpressionsHIPExt.jl")).read()
    free = {"de_abi_version", "de_opcode_by_name", "de_status_string", "de_dist_unique_id", "de_dist_last_error", "de_ctx_create"}
    for fn in re.finditer(r"^function\s+([\w!]+)[^\n]*\n(.*?)^end", src, re.M | re.S):
        body = fn.group(2)
        for m in re.finditer(r"ccall\(\(:(\w+), LIBDE\)", body):
            if m.group(1) in free:
                continue
            before = body[:m.start()]
            locked = ("with_ctx(" in before or "with_pop(" in before or "trylock(" in before or fn.group(1) in ("check", "grad_widths"))
            assert locked, f"{fn.group(1)}: ccall(:{m.group(1)}) outside the context's lock"


# ---- a block-balance check: what a parser would refuse first ------------------------------------------------------------------------
OPENERS = {"function", "if", "for", "while", "try", "let", "begin", "struct", "module", "do", "quote", "macro"}


def _julia_tokens(src):
    """(token, bracket depth) of the identifiers / keywords outside strings, characters and comments; brackets tracked on the way."""
    i, n, depth = 0, len(src), 0
    while i < n:
        ch = src[i]
        if ch == "#":
            if src.startswith("#=", i):
                i = src.index("=#", i) + 2
            else:
                while i < n and src[i] != "\n":
                    i += 1
        elif src.startswith('"""', i):
            i = src.index('"""', i + 3) + 3
        elif ch == '"':
            i += 1
            while src[i] != '"':
                i += 2 if src[i] == "\\" else 1
            i += 1
        elif ch == "'" and i + 2 < n and (src[i + 2] == "'" or (src[i + 1] == "\\" and src[i + 3] == "'")):
            i += 3 if src[i + 2] == "'" else 4  # a character literal (not the adjoint operator)
        elif ch in "([{":
            depth += 1
            i += 1
        elif ch in ")]}":
            depth -= 1
            i += 1
        elif ch.isalpha() or ch == "_" or ch == "@":
            j = i + 1
            while j < n and (src[j].isalnum() or src[j] in "_!"):
                j += 1
            yield src[i:j], depth, (src[i - 1] if i else "\n")
            i = j
        else:
            i += 1
    assert depth == 0, "unbalanced brackets"


@pytest.mark.parametrize("path", [p for p in JULIA if not p.endswith("golden_cases.jl")], ids=lambda p: os.path.basename(p))
def test_julia_blocks_balance(path):
    """Every block opener outside brackets has its `end` (comprehensions / generators live inside brackets and have none; `a[end]` is an
    index): the first thing a parser would refuse.  The build image has no Julia; this keeps an edit from leaving a block open."""
    opened = closed = 0
    stack = []
    for tok, depth, prev in _julia_tokens(open(path).read()):
        if depth:
            continue
        if tok in OPENERS and prev != ":" and prev != ".":  # (:if / x.begin are not keywords)
            opened += 1
            stack.append(tok)
        elif tok == "end":
            closed += 1
            assert stack, f"{os.path.basename(path)}: `end` without an open block"
            stack.pop()
    assert not stack, f"{os.path.basename(path)}: unclosed block(s): {stack[-3:]}"
    assert opened == closed and opened > 20
