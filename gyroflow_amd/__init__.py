"""gfwarp — MI355X-native rolling-shutter lens warp (gyroflow-core ``process_pixels`` hot path).

The product is the HIP library ``libgfwarp.so`` (C ABI: include/gfwarp.h); this
package is its host-side mirror of the reference operator surface.
"""
from . import abi  # noqa: F401

__all__ = ["abi"]
