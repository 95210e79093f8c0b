"""CPU oracle for the pbrs path-integrator hot path — TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this package.
Nothing under pbrs_amd/ does; the product path fails loudly without its HIP library instead.
"""
