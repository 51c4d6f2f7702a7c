"""ORACLE — TEST INFRASTRUCTURE ONLY.

CPU restatements of the reference's hot-path algorithms.  Only `tests/`, `__graft_entry__.smoke()`
and `bench.py`'s `cpu_baseline` leg may import anything under this directory — and there only as the
checker, never as the thing measured or shipped.  `wast3d_amd` must never import it.
"""
