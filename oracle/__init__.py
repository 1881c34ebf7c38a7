"""CPU oracle: TEST INFRASTRUCTURE ONLY.

A plain numpy/SciPy restatement of the reference's hot path
(parapint.linalg.{Schur,MPISchur}ComplementLinearSolver and its SciPy / MA27-style
sub-block solvers).  Only ``tests/``, ``__graft_entry__.smoke()`` and the
``cpu_baseline`` leg of ``bench.py`` may import it; the product package
``parapint_amd`` never does, and has no CPU fallback.

Parity pinning: the restatement is checked in
``tests/test_oracle.py`` against every fixture the reference's own tests hold for
this path -- the 3x3 sub-solver contract (linalg/tests/test_linear_solvers.py:13-23,
63-99), the two 8x8 bordered systems (linalg/schur_complement/tests/
test_explicit_schur_complement.py:13-55, test_mpi_explicit_schur_complement.py:22-115)
and the synthetic-KKT known answer 0.3163456780448639 (examples/tests/
test_examples.py:76-99) -- and against golden vectors produced by the reference's own
solver files run in the build container (tests/golden/make_golden.py).
"""
