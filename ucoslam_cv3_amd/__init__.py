"""Importable alias of the `ucoslam-cv3_amd/` package directory (a hyphen cannot appear in `import`).

`import ucoslam_cv3_amd` resolves every submodule from ../ucoslam-cv3_amd/.
"""
import os as _os

_real = _os.path.join(_os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))), "ucoslam-cv3_amd")
__path__.insert(0, _real)
with open(_os.path.join(_real, "__init__.py")) as _f:
    exec(compile(_f.read(), _os.path.join(_real, "__init__.py"), "exec"))
del _f
