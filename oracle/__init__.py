"""CPU oracle of the walk-generation path -- TEST INFRASTRUCTURE ONLY.

Nothing under ``pecanpy_amd/`` may import this package.  See ``oracle/pecan_oracle.c``.
"""
