"""Import alias: the product package lives in ``multi-uav-ta-gym-env_amd/`` (not a valid Python
identifier), so ``import muavta_amd`` resolves its submodules there."""
import os as _os

_real = _os.path.join(_os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))), "multi-uav-ta-gym-env_amd")
__path__.insert(0, _real)
with open(_os.path.join(_real, "__init__.py")) as _f:
    exec(compile(_f.read(), _os.path.join(_real, "__init__.py"), "exec"))
