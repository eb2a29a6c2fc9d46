"""Drop-in `spconv` for a DODA checkout: put doda_amd/shims on PYTHONPATH."""
from doda_amd.spconv import *  # noqa: F401,F403
from doda_amd.spconv import __all__, __version__, functional, modules, ops  # noqa: F401
import sys as _sys
_sys.modules[__name__ + ".modules"] = modules
_sys.modules[__name__ + ".functional"] = functional
_sys.modules[__name__ + ".ops"] = ops
