"""CPU parity oracle — TEST INFRASTRUCTURE ONLY.

Importable only from tests/, __graft_entry__.smoke() and bench.py's cpu_baseline
leg.  Nothing under omnifusion_amd/ imports this package (tests/test_boundary.py
greps for it).
"""
