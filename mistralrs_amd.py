"""Import shim: the package directory is named `mistral.rs_amd/` (not a valid Python identifier),
so `import mistralrs_amd` loads that directory as the package `mistralrs_amd`."""
import importlib.util as _u
import os as _os
import sys as _sys

_d = _os.path.join(_os.path.dirname(_os.path.abspath(__file__)), "mistral.rs_amd")
_spec = _u.spec_from_file_location("mistralrs_amd", _os.path.join(_d, "__init__.py"),
                                   submodule_search_locations=[_d])
_mod = _u.module_from_spec(_spec)
_sys.modules["mistralrs_amd"] = _mod
_spec.loader.exec_module(_mod)
