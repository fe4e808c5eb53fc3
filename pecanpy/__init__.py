"""Drop-in alias: ``from pecanpy import pecanpy`` resolves to the MI355X-native implementation."""
import sys

from pecanpy_amd import cli, graph, pecanpy, wrappers  # noqa: F401
from pecanpy_amd import version  # noqa: F401

sys.modules[__name__ + ".graph"] = graph
sys.modules[__name__ + ".pecanpy"] = pecanpy
sys.modules[__name__ + ".wrappers"] = wrappers
sys.modules[__name__ + ".cli"] = cli
__all__ = ["graph", "pecanpy"]
