"""CPU oracle for the `thrifty detect` hot path -- TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import
this package.  The product (thrifty_amd) never does.
"""
