"""CPU parity oracle for the FastDepth MobileNetSkipAdd hot path.  TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's `cpu_baseline` leg may import this package; the
product package (fast-depth_amd/) never does and has no CPU execution path of its own.
"""
