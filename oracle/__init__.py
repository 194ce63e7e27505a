"""ORACLE -- CPU restatements of the reference hot path (test infrastructure only).

Importers allowed: tests/, __graft_entry__.smoke(), bench.py's cpu_baseline leg.
The product package (simple-hrnet_amd) must never import from here.
"""
