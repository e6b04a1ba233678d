"""lc2is_amd — MI355X-native (gfx950) kernels and drop-in modules for the LC2IS hot path.

See DESIGN.md for the scope and INTEGRATION.md for the C-ABI boundary.
"""
from . import _lib  # noqa: F401

__all__ = ["_lib"]
