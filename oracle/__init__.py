"""CPU oracle for the GaussianFormer hot path -- TEST INFRASTRUCTURE ONLY.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may
import this package, and only as the checker / reported CPU baseline.  The product
(``gaussianformer_amd``) never imports it.  Parity is unpinned by the reference (it ships
no tests or golden vectors for this path); see ``gf_oracle.c`` header and DESIGN.md.
"""
from .oracle import *  # noqa: F401,F403
