"""Importable alias of the package that lives in ``nerf-pytorch_amd/`` (a hyphen is not a valid Python identifier).

``import nerf_pytorch_amd`` executes ``nerf-pytorch_amd/__init__.py`` in this module's namespace and points
``__path__`` at that directory, so ``nerf_pytorch_amd.models`` etc. resolve to the files there.
"""
import os as _os

_real = _os.path.join(_os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))), "nerf-pytorch_amd")
__path__ = [_real]
with open(_os.path.join(_real, "__init__.py")) as _f:
    exec(compile(_f.read(), _os.path.join(_real, "__init__.py"), "exec"))
del _f
