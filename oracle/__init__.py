"""CPU restatements of the reference algorithms for this path -- TEST INFRASTRUCTURE (see oracle/neube_oracle.py and
oracle/painting_oracle.py): imported only by tests/, __graft_entry__.smoke() and the cpu_baseline leg of bench.py."""
