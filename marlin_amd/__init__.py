"""marlin_amd: MI355X-native FFT spectral-solver inner loop for Marlin (HIP kernels behind a C ABI)."""
from . import _lib  # noqa: F401

__all__ = ["_lib"]
