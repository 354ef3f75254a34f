"""Import alias: ``import fluid_amd`` == the package in ``vulkan-3d-fluid-simulation_amd/``
(the directory name has hyphens and cannot be written in an import statement).  Submodules are
aliased too, so ``fluid_amd.params`` and ``vulkan-3d-fluid-simulation_amd.params`` are one module."""
import importlib
import os
import sys

_root = os.path.dirname(os.path.abspath(__file__))
if _root not in sys.path:
    sys.path.insert(0, _root)
_REAL = "vulkan-3d-fluid-simulation_amd"
_pkg = importlib.import_module(_REAL)
for _name, _mod in list(sys.modules.items()):
    if _name.startswith(_REAL + "."):
        sys.modules[__name__ + _name[len(_REAL):]] = _mod
sys.modules[__name__] = _pkg
