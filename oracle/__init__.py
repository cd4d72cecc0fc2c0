"""CPU oracle for the DiGA training hot path -- TEST INFRASTRUCTURE ONLY.

This package restates, on PyTorch-CPU / numpy, the algorithms of the reference
hot path.  It is the *checker*: only
``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of
``bench.py`` may import it.  Nothing under ``diga_amd/`` imports it; the
product path is the HIP library behind ``include/diga_hip.h`` and fails loudly
when that library is missing.

Parity pinning: the reference has no tests of its own, so
the oracle is pinned by golden vectors captured from the reference itself,
imported in the build container (``tools/gen_golden.py``; fixtures in
``tests/golden/*.npz``).  ``tests/test_oracle_golden.py`` checks every oracle
function against those captures.

Path abbreviation used in citations: ``G5/`` = ``domain_adaptation/GTA5/`` of
the reference tree.
"""
