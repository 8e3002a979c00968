"""Import alias: the package directory is named `go-kzg_amd` (not a valid Python identifier), so
`import gokzg_amd` loads it from there.  Nothing else lives in this file."""
import importlib.util
import os
import sys

_dir = os.path.join(os.path.dirname(os.path.abspath(__file__)), "go-kzg_amd")
_spec = importlib.util.spec_from_file_location("gokzg_amd", os.path.join(_dir, "__init__.py"), submodule_search_locations=[_dir])
_mod = importlib.util.module_from_spec(_spec)
sys.modules["gokzg_amd"] = _mod
_spec.loader.exec_module(_mod)
