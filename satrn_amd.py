"""Import shim: the product package lives in `p4-fr-sorry-math-but-love-you_amd/` (not a valid Python identifier),
so `import satrn_amd` loads that directory as the package `satrn_amd`."""
import importlib.util
import os
import sys

_d = os.path.join(os.path.dirname(os.path.abspath(__file__)), "p4-fr-sorry-math-but-love-you_amd")
_spec = importlib.util.spec_from_file_location(__name__, os.path.join(_d, "__init__.py"), submodule_search_locations=[_d])
_mod = importlib.util.module_from_spec(_spec)
sys.modules[__name__] = _mod
_spec.loader.exec_module(_mod)
