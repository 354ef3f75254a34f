"""Import alias: ``import fluid_amd`` == the package in ``vulkan-3d-fluid-simulation_amd/``
(the directory name has hyphens and cannot be written in an import statement)."""
import importlib
import os
import sys

_root = os.path.dirname(os.path.abspath(__file__))
if _root not in sys.path:
    sys.path.insert(0, _root)
_pkg = importlib.import_module("vulkan-3d-fluid-simulation_amd")
sys.modules[__name__] = _pkg
