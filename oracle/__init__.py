"""CPU oracle -- TEST INFRASTRUCTURE ONLY (see oracle/mgar_oracle.c).  Never imported by
multimodal_gar_amd/; used by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline."""
