"""benchlib -- the pieces of bench.py, one module per workload (bench.py was one 63 KB file with seven
workloads in one main()).  bench.py at the repo root keeps the contract (arguments, the timed region of the headline
workload, the ONE JSON line); everything here is imported by it and by nothing in the product."""
