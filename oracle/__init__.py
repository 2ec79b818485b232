"""CPU oracle for the SQFA pairwise SPD-distance hot path.

TEST INFRASTRUCTURE ONLY.  Nothing under ``sqfa_amd/`` imports this package.
Only ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of
``bench.py`` may import it, and only as the checker / the timed CPU baseline.

Parity status: PINNED.  Both restatements are checked in ``tests/test_oracle.py``
against golden vectors produced by importing the reference itself
(``tests/golden/make_golden.py``).
"""
from . import closed_form, reference_path  # noqa: F401
