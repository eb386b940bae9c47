"""CPU oracle package (TEST INFRASTRUCTURE).  Import only from tests/, __graft_entry__.smoke() and
bench.py's cpu_baseline leg - never from ppo_cpp_amd (the product)."""
