"""MI355X-native 3D Eulerian fluid-step engine — drop-in for sections 00…14 of
Matezzzz/vulkan-3d-fluid-simulation (see include/fluid_engine.h).

The directory name carries hyphens (it mirrors the reference repository's name), so import it with
``importlib.import_module("vulkan-3d-fluid-simulation_amd")`` or via the ``fluid_amd`` alias that
``fluid_amd.py`` at the repository root installs.
"""
from . import params  # noqa: F401
from .params import FluidParams, dam_break_params, default_params  # noqa: F401
from . import engine  # noqa: F401
from . import scenes  # noqa: F401
from .engine import FluidEngine, FluidEngineError, load_library  # noqa: F401
from .build import build_engine  # noqa: F401
