"""ORACLE package -- test infrastructure only (see oracle/ref_chain.c header).

Importers allowed: tests/, __graft_entry__.smoke(), bench.py's cpu_baseline leg.
The product package (tetraear_amd) never imports this.
"""
