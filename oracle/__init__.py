"""TEST INFRASTRUCTURE ONLY -- CPU oracle for the DeepPhysiNet physics-informed training step.

Nothing under ``oracle/`` is part of the product path.  Only ``tests/``,
``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import it,
and only as the checker / the timed CPU baseline.  ``deepphysinet_amd`` never
imports this package and raises if its HIP library is missing.

Parity status: PINNED.  ``oracle/dpn_oracle.py`` is checked against golden
vectors captured from the reference itself (``tests/golden/make_golden.py`` imports
``/root/reference`` in the build container; the vectors are committed under
``tests/golden/``; see ``tests/test_oracle_golden.py``).
"""
