"""Import shim: the product package lives in the directory `neo-360_amd/`
(the name the build contract fixes), which is not a valid Python identifier.
This package re-roots itself there so `import neo360_amd.<module>` resolves to
`neo-360_amd/<module>.py`."""
import os as _os

_real = _os.path.join(_os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))), "neo-360_amd")
__path__.insert(0, _real)
with open(_os.path.join(_real, "__init__.py")) as _f:
    exec(compile(_f.read(), _os.path.join(_real, "__init__.py"), "exec"))
del _f
