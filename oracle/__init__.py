"""ORACLE -- CPU restatement of the reference algorithm (AaltoML/nonstationary-audio-gp, matlab/).

TEST INFRASTRUCTURE ONLY.  Only tests/, __graft_entry__.smoke() and bench.py's `cpu_baseline`
leg may import this package; the product path (nonstationary-audio-gp_amd/) never does.

PARITY UNPINNED: the reference ships no tests, golden vectors or fixtures for the hot path
and is 100 % MATLAB (no MATLAB/Octave in the build container, nothing was denied -- the
interpreters simply do not exist), so this restatement is pinned only by the mathematical
self-checks in tests/test_oracle_*.py.
"""
