"""Import alias: the package lives in the directory `r-super_amd/` (not a valid Python identifier); `import rsuper_amd` and every
`rsuper_amd.<sub>` import resolve to it through this package's search path -- no code is executed from the other directory's `__init__`
(its docstring describes the layout; the version is repeated here)."""
import os as _os

__path__ = [_os.path.join(_os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))), 'r-super_amd')]
__version__ = '0.1.0'
