"""Import shim: the package directory is named `cmpt-eigenex_amd/` (not a valid
Python identifier); `import cmpt_eigenex_amd` resolves to it through this file."""
import os as _os

__path__ = [_os.path.join(_os.path.dirname(_os.path.abspath(__file__)), "cmpt-eigenex_amd")]
__file__ = _os.path.join(__path__[0], "__init__.py")
with open(__file__) as _f:
    exec(compile(_f.read(), __file__, "exec"))
