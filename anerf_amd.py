"""Alias so that ``import anerf_amd`` resolves to the hyphen-named package directory ``a-nerf_amd/``."""
import importlib
import sys

sys.modules[__name__] = importlib.import_module("a-nerf_amd")
