"""Randomised parity stress beyond the fixed-seed tests: RoIBBox / Decoder / RPN targets / RoIPooling / top_k / RoIDelta / combined NMS against the CPU oracle with fresh
seeds, shapes and score distributions for a given number of seconds (the generators live in tests/stress_cases.py; a short
fixed-seed slice of them runs in tests/test_gpu_stress.py).  Test infrastructure (uses oracle/); prints the first mismatch and
exits 1, or a summary.

    python tools/stress_parity.py [seconds] [seed]
"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

from tests import stress_cases

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
seed = int(sys.argv[2]) if len(sys.argv) > 2 else int(time.time())
print("seed", seed, flush=True)
t0 = time.time()
try:
    n = stress_cases.run(np.random.default_rng(seed), budget, progress=lambda s: print(s, flush=True))
except stress_cases.Mismatch as exc:
    print("MISMATCH", exc, flush=True)
    sys.exit(1)
print("ok", n, f"in {time.time() - t0:.0f} s", flush=True)
