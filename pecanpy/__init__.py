"""Drop-in alias: ``from pecanpy import pecanpy`` resolves to the MI355X-native implementation."""
import sys

from pecanpy_amd import graph, pecanpy, wrappers  # noqa: F401
from pecanpy_amd import version  # noqa: F401

sys.modules[__name__ + ".graph"] = graph
sys.modules[__name__ + ".pecanpy"] = pecanpy
sys.modules[__name__ + ".wrappers"] = wrappers
__all__ = ["graph", "pecanpy"]
