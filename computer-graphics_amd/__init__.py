"""MI355X path-tracing hot path: Python host side over the C ABI of libmipt.so.

Importing the package never loads the HIP library; the first call that needs it
does, and fails loudly if it has not been built (there is no CPU fallback).
"""
from . import abi  # noqa: F401
from . import api  # noqa: F401

__all__ = ["abi", "api"]
