"""CPU oracle for the V-DETR hot path.

TEST INFRASTRUCTURE ONLY: nothing under ``v-detr_amd/`` may import this package.  It is the checker for
``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of ``bench.py``.
"""
