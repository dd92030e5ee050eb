"""CPU oracle for the EKF-SLAM hot path — TEST INFRASTRUCTURE ONLY.

Nothing under ``oracle/`` is part of the product.  Only ``tests/``,
``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import,
call, link or execute anything in this package, and only as the checker.

PARITY UNPINNED: the reference (SamShue/EKF_SLAM, pure MATLAB) ships no golden
vectors, assertions or fixtures, and neither MATLAB nor Octave exists in the
build image, so these restatements are pinned only by hand-derived known-answer
tests (tests/test_oracle_kat.py) and by agreeing with each other.
"""
