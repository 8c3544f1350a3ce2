"""dynamicexpressions.jl_amd — MI355X-native batched tree evaluation for DynamicExpressions.jl.

Host-side mirror of the reference interface for the hot path (eval_tree_array /
eval_grad_tree_array / eval_diff_tree_array over Node + OperatorEnum, and the
ParametricExpression wrapper) on top of the C ABI of ``csrc/libde_hip.so``
(include/de_hip.h).  The directory name carries a dot, so it is imported through the
root-level shim ``dynamicexpressions_jl_amd.py``.
"""
from .operators import OperatorEnum, UnsupportedOperatorError, OPCODES  # noqa: F401
from .node import (  # noqa: F401
    Node, ParametricNode, GraphNode, flatten_graph, preserve_sharing, break_sharing, count_nodes, count_depth, count_constant_nodes, flatten,
    flatten_population, get_scalar_constants, set_scalar_constants, string_tree, postorder,
)
from .simplify import simplify_tree, combine_operators  # noqa: F401
from . import synth  # noqa: F401

__all__ = [
    "OperatorEnum", "UnsupportedOperatorError", "Node", "ParametricNode", "GraphNode", "flatten_graph", "preserve_sharing", "break_sharing", "count_nodes",
    "count_depth", "count_constant_nodes", "flatten", "flatten_population",
    "get_scalar_constants", "set_scalar_constants", "string_tree", "simplify_tree", "combine_operators", "synth",
]


def __getattr__(name):
    # The device API needs libde_hip.so (and torch); import it lazily so that the host
    # logic (flattening, synthesis) stays usable in tools that never touch the GPU.
    if name in ("api", "dist"):
        import importlib
        return importlib.import_module(f"{__name__}.{name}")
    _api_names = {
        "EvalContext", "eval_tree_array", "eval_grad_tree_array", "eval_diff_tree_array",
        "Population", "ParametricExpression", "Expression", "library", "DeviceError", "UncertifiedFlag",
    }
    if name in _api_names:
        from . import api
        return getattr(api, name)
    raise AttributeError(name)
