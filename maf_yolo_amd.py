"""Import alias: `import maf_yolo_amd` loads the package that lives in ./maf-yolo_amd/
(the directory name carries the reference's hyphen, which Python cannot import directly)."""
import importlib.util as _u
import os as _os
import sys as _sys

_dir = _os.path.join(_os.path.dirname(_os.path.abspath(__file__)), "maf-yolo_amd")
_spec = _u.spec_from_file_location("maf_yolo_amd", _os.path.join(_dir, "__init__.py"),
                                   submodule_search_locations=[_dir])
_mod = _u.module_from_spec(_spec)
_sys.modules["maf_yolo_amd"] = _mod
_spec.loader.exec_module(_mod)
