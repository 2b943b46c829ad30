"""Import alias: `import sse_amd` loads the package kept in the directory
`sequence-semantic-embedding_amd/` (a name Python cannot import directly)."""
import importlib.util
import os
import sys

_dir = os.path.join(os.path.dirname(os.path.abspath(__file__)), "sequence-semantic-embedding_amd")
_spec = importlib.util.spec_from_file_location(
    "sse_amd", os.path.join(_dir, "__init__.py"), submodule_search_locations=[_dir])
_mod = importlib.util.module_from_spec(_spec)
sys.modules["sse_amd"] = _mod
_spec.loader.exec_module(_mod)
