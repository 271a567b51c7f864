"""CPU oracle — TEST INFRASTRUCTURE ONLY (see oracle/i2p_oracle.c).  Importable only from
tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg."""
