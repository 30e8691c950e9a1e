"""ORACLE — CPU restatement of the reference's algorithms for the hot path.

TEST INFRASTRUCTURE ONLY.  Only tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg may import this package; the product (yoloseries_amd/) never does.
Parity status: PINNED — every function here is checked against golden vectors that
tools/gen_golden.py produced by running the reference itself (tests/golden/*.npz),
see tests/test_oracle_golden.py.
"""
