"""Packaging: ``pip install -e .`` gives the ``pecanpy`` console script of the reference
(reference setup.cfg:64-66) on top of the MI355X engine.  The native library is built in-tree by
``python -c 'import __graft_entry__ as g; g.build()'`` (hipcc, gfx950)."""
from setuptools import setup

setup(
    name="pecanpy-amd",
    version="0.1.0",
    packages=["pecanpy_amd", "pecanpy"],
    package_data={"pecanpy_amd": ["libpecanpy_amd.so", "csrc/*"]},
    entry_points={"console_scripts": ["pecanpy=pecanpy_amd.cli:main"]},
)
