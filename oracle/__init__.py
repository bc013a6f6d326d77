"""TEST INFRASTRUCTURE ONLY — CPU restatement of the ProNeRF rendering hot path.

Nothing under ``oracle/`` is part of the shipped product.  Only ``tests/``,
``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import it,
and there only as the checker / the CPU baseline, never as the measured path.
"""
