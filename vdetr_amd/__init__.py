"""Import alias: the package directory is ``v-detr_amd/`` (not a valid Python identifier), so
``import vdetr_amd`` resolves to it through this shim.  No code lives here."""
import os as _os

_real = _os.path.join(_os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))), "v-detr_amd")
__path__ = [_real]
with open(_os.path.join(_real, "__init__.py")) as _fh:
    exec(compile(_fh.read(), _os.path.join(_real, "__init__.py"), "exec"))
del _fh
