"""CPU oracle (TEST INFRASTRUCTURE ONLY).

Python bindings of ``oracle/libmisoracle.so`` -- the plain-C restatement of the reference's hot
path (see ``oracle/mo_common.h`` for the scope statement and ``PARITY UNPINNED`` notice).

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import
this package; the product (``image_stitching_amd``) never does.
"""
from .bindings import *  # noqa: F401,F403
