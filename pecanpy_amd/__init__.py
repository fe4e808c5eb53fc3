"""pecanpy_amd: MI355X-native node2vec walk engine behind PecanPy's API.

``from pecanpy_amd import pecanpy`` mirrors ``from pecanpy import pecanpy`` of the reference
(krishnanlab/PecanPy); the top-level ``pecanpy`` shim package in this repo makes the original
import path work unchanged.
"""
from . import graph
from . import pecanpy

version = "0.1.0"
__all__ = ["graph", "pecanpy"]
