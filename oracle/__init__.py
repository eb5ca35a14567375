"""CPU oracle package — TEST INFRASTRUCTURE ONLY (see oracle/vs_oracle.h, oracle/README.md).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this package.
The product package (pgvectorscale_amd) never does.
"""
