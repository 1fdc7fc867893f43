"""Import shim: the product package lives in the directory ``abstractbayesopt.jl_amd/`` (the name
the build contract fixes); a dot is not a legal Python package character, so this tiny namespace
package registers that directory as the submodule ``abstractbayesopt.jl_amd``:

    import abstractbayesopt.jl_amd as abo
"""
import importlib.util as _ilu
import os as _os
import sys as _sys

_pkg_dir = _os.path.join(_os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))), "abstractbayesopt.jl_amd")
_name = __name__ + ".jl_amd"
if _name not in _sys.modules:
    _spec = _ilu.spec_from_file_location(_name, _os.path.join(_pkg_dir, "__init__.py"),
                                         submodule_search_locations=[_pkg_dir])
    _mod = _ilu.module_from_spec(_spec)
    _sys.modules[_name] = _mod
    _spec.loader.exec_module(_mod)
jl_amd = _sys.modules[_name]
