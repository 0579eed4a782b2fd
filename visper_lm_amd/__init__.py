"""Import shim: the package lives in `visper-lm_amd/` (not a valid Python identifier); this makes it
importable as `visper_lm_amd` without symlinks."""
import os as _os

__path__ = [_os.path.join(_os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))), "visper-lm_amd")]
with open(_os.path.join(__path__[0], "__init__.py")) as _f:
    exec(compile(_f.read(), _os.path.join(__path__[0], "__init__.py"), "exec"))
