"""CPU restatements of the reference hot path.  TEST INFRASTRUCTURE ONLY: imported by tests/,
__graft_entry__.smoke() and bench.py's cpu_baseline leg, never by the product package."""
