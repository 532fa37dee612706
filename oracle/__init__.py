"""CPU oracle of the EDRL hot path — TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's `cpu_baseline` leg may import this package;
the product (the *_amd package) never does.  See oracle/edrl_oracle.py and oracle/resnet_oracle.py.
"""
