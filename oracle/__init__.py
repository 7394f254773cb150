"""TEST INFRASTRUCTURE ONLY -- the CPU oracle.  See oracle/oracle.py."""
