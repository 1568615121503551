"""ORACLE — test infrastructure only.

CPU restatements of the reference's hot path.  Nothing under
``viewformer_amd/`` imports this package; only ``tests/``,
``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg do.
"""
