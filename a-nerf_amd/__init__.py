"""a-nerf_amd: MI355X-native ray-march hot path for A-NeRF.

The directory name carries a hyphen, so import it as ``importlib.import_module("a-nerf_amd")`` or
through the ``anerf_amd`` alias module at the repo root.
"""
from . import synth  # noqa: F401

__all__ = ["synth"]
