"""Synthetic workloads of SURVEY.md §8d: seeded PRNG + the reference's random-tree recipe.

``gen_random_tree_fixed_size`` restates test/tree_gen_utils.jl:27-91 (the generator the
reference's own tests AND benchmark/benchmarks.jl:76-108 use).  Julia's MersenneTwister
streams cannot be reproduced without Julia, so trees are drawn from an in-repo
xoshiro256** (SplitMix64-seeded) generator: same distribution, different stream.
"""
from __future__ import annotations

import math
from typing import List, Optional, Sequence, Type

import numpy as np

from .node import Node, ParametricNode, count_depth, count_nodes
from .operators import OperatorEnum

_M64 = (1 << 64) - 1


class Xoshiro256ss:
    """xoshiro256** seeded through SplitMix64 (Blackman & Vigna); pure Python, exact on
    every platform."""

    def __init__(self, seed: int):
        z = seed & _M64
        self.s: List[int] = []
        for _ in range(4):
            z = (z + 0x9E3779B97F4A7C15) & _M64
            x = z
            x = ((x ^ (x >> 30)) * 0xBF58476D1CE4E5B9) & _M64
            x = ((x ^ (x >> 27)) * 0x94D049BB133111EB) & _M64
            self.s.append(x ^ (x >> 31))
        self._spare: Optional[float] = None

    def next_u64(self) -> int:
        s = self.s
        r = (((s[1] * 5) & _M64) << 7 | ((s[1] * 5) & _M64) >> 57) & _M64
        r = (r * 9) & _M64
        t = (s[1] << 17) & _M64
        s[2] ^= s[0]
        s[3] ^= s[1]
        s[1] ^= s[2]
        s[0] ^= s[3]
        s[2] ^= t
        s[3] = ((s[3] << 45) | (s[3] >> 19)) & _M64
        return r

    def rand(self) -> float:  # uniform [0, 1)
        return (self.next_u64() >> 11) * (1.0 / (1 << 53))

    def randbool(self) -> bool:
        return bool(self.next_u64() >> 63)

    def randint(self, n: int) -> int:
        """Uniform integer in 1..n (Julia ``rand(rng, 1:n)``), rejection sampled."""
        lim = _M64 - (_M64 + 1) % n
        while True:
            x = self.next_u64()
            if x <= lim:
                return 1 + x % n

    def randn(self) -> float:  # Box-Muller
        if self._spare is not None:
            v, self._spare = self._spare, None
            return v
        u1 = 1.0 - self.rand()
        u2 = self.rand()
        r = math.sqrt(-2.0 * math.log(u1))
        self._spare = r * math.sin(2.0 * math.pi * u2)
        return r * math.cos(2.0 * math.pi * u2)


def make_random_leaf(nfeatures: int, rng: Xoshiro256ss, dtype=np.float32,
                     node_type: Type[Node] = Node, nparams: int = 0) -> Node:
    """test/tree_gen_utils.jl:27-35: constant ``randn(T)`` w.p. 1/2 else feature U{1..F}.
    With ``nparams > 0`` (ParametricNode populations): constant / feature /
    parameter each w.p. 1/3."""
    if nparams > 0:
        k = rng.randint(3)
        if k == 1:
            return node_type(val=float(dtype(rng.randn())))
        if k == 2:
            return node_type(feature=rng.randint(nfeatures))
        return node_type(parameter=rng.randint(nparams))
    if rng.randbool():
        return node_type(val=float(dtype(rng.randn())))
    return node_type(feature=rng.randint(nfeatures))


def _set_node(dst: Node, src: Node) -> None:
    """set_node!(tree, new_tree): overwrite ``dst`` in place (src/Node.jl set_node!)."""
    for slot in ("degree", "constant", "val", "feature", "op", "children"):
        setattr(dst, slot, getattr(src, slot))
    if isinstance(dst, ParametricNode):
        dst.is_parameter = getattr(src, "is_parameter", False)
        dst.parameter = getattr(src, "parameter", 0)


def gen_random_tree_fixed_size(node_count: int, operators: OperatorEnum, nfeatures: int,
                               rng: Xoshiro256ss, dtype=np.float32,
                               node_type: Type[Node] = Node, nparams: int = 0) -> Node:
    """test/tree_gen_utils.jl:71-91 (+ append_random_op :37-69): start from a random leaf;
    repeatedly replace a uniformly chosen leaf by a random unary/binary operator with fresh
    random leaves, P(binary) = nbin/(nuna+nbin); when one node short only a unary fits."""
    nuna, nbin = len(operators.unaops), len(operators.binops)
    tree = make_random_leaf(nfeatures, rng, dtype, node_type, nparams)
    leaves = [tree]
    cur = 1
    while cur < node_count:
        if cur == node_count - 1:
            if nuna == 0:
                break
            make_bin = False
        else:
            make_bin = rng.rand() < nbin / (nuna + nbin)
        li = rng.randint(len(leaves)) - 1  # rand(NodeSampler(; tree, filter=degree==0))
        node = leaves[li]
        if make_bin:
            new = node_type(rng.randint(nbin),
                            make_random_leaf(nfeatures, rng, dtype, node_type, nparams),
                            make_random_leaf(nfeatures, rng, dtype, node_type, nparams))
        else:
            new = node_type(rng.randint(nuna),
                            make_random_leaf(nfeatures, rng, dtype, node_type, nparams))
        _set_node(node, new)
        leaves[li:li + 1] = list(node.children)
        cur += node.degree
    return tree


BENCH_OPERATORS = OperatorEnum(binary_operators=("+", "-", "/", "*"),
                               unary_operators=("cos", "exp"))  # benchmark/benchmarks.jl:32-35


def random_population(n_trees: int, seed: int, node_count: int = 20, nfeatures: int = 5,
                      max_depth: int = 15, operators: OperatorEnum = BENCH_OPERATORS,
                      dtype=np.float32, node_type: Type[Node] = Node, nparams: int = 0) -> List[Node]:
    """``n_trees`` random ``node_count``-node trees, redrawn while count_depth > max_depth
    (BASELINE.json: "random depth<=15 trees")."""
    rng = Xoshiro256ss(seed)
    out: List[Node] = []
    while len(out) < n_trees:
        t = gen_random_tree_fixed_size(node_count, operators, nfeatures, rng, dtype, node_type, nparams)
        if count_depth(t) <= max_depth:
            out.append(t)
    return out


def random_X(nfeatures: int, n: int, seed: int, dtype=np.float32) -> np.ndarray:
    """``randn(T, F, N)``: returns an (F, N) Fortran-ordered array (feature index fastest,
    src/Evaluate.jl:251).  numpy's PCG64 stream (stable across numpy versions)."""
    g = np.random.Generator(np.random.PCG64(seed))
    return np.asfortranarray(g.standard_normal((n, nfeatures), dtype=np.float64).astype(dtype).T)


# The BASELINE.json configurations.  seed = 0xDE00 + config id.
CONFIGS = {
    "C2": dict(n_trees=1000, N=10**6, nfeatures=5, seed=0xDE02, dtype="float32"),
    "C3": dict(n_trees=1000, N=10**6, nfeatures=5, seed=0xDE02, dtype="float32", grad="variable"),
    "headline": dict(n_trees=1000, N=10**7, nfeatures=5, seed=0xDE02, dtype="float32"),
    "C4": dict(n_trees=10000, N=10**7, nfeatures=5, seed=0xDE04, dtype="float32"),
    "C5": dict(n_trees=1000, N=10**6, nfeatures=5, nparams=8, n_classes=16, seed=0xDE05, dtype="float32"),
}
