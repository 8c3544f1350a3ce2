"""Import shim: the package directory is named ``dynamicexpressions.jl_amd`` (with a dot, as
the project layout prescribes), which Python cannot import by name.  ``import
dynamicexpressions_jl_amd`` loads that directory as a regular package under this name."""
import importlib.util
import os
import sys

_dir = os.path.join(os.path.dirname(os.path.abspath(__file__)), "dynamicexpressions.jl_amd")
_spec = importlib.util.spec_from_file_location(
    __name__, os.path.join(_dir, "__init__.py"), submodule_search_locations=[_dir])
_mod = importlib.util.module_from_spec(_spec)
sys.modules[__name__] = _mod
_spec.loader.exec_module(_mod)
