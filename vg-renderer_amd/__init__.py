"""vg-renderer_amd: MI355X-native batch implementation of vg-renderer's Path flattener + Stroker.

The directory name contains a hyphen (task contract), so import it with
    vgr = importlib.import_module("vg-renderer_amd")
"""
from . import capi  # noqa: F401
from . import pathset  # noqa: F401
from .pathset import PathSetBuilder, PathSetArrays, make_draws  # noqa: F401
