"""Import shim: `import ai_economist_amd` loads the package that lives in the
(hyphenated, hence not directly importable) directory `ai-economist_amd/`."""
import importlib.util
import os
import sys

_dir = os.path.join(os.path.dirname(os.path.abspath(__file__)), "ai-economist_amd")
_spec = importlib.util.spec_from_file_location(
    "ai_economist_amd", os.path.join(_dir, "__init__.py"), submodule_search_locations=[_dir]
)
_mod = importlib.util.module_from_spec(_spec)
sys.modules["ai_economist_amd"] = _mod
_spec.loader.exec_module(_mod)
