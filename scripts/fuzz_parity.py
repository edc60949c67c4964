"""Randomised parity sweep: HIP found set / values / levels vs the CPU oracle on many block shapes, densities and seeds
(case generator: tests/fuzz_cases.parity_case; a seeded slice runs under `pytest -m gpu`, tests/test_gpu_fuzz.py).
    python scripts/fuzz_parity.py [n_cases]        (GPU box; ~1 s per case; FUZZ_SEED=... draws another sweep)"""
import os, sys
import numpy as np
sys.path.insert(0, "."); sys.path.insert(0, "tests")
import fuzz_cases
from mustache_amd.engine import ScaleSpaceEngine
ncases = int(sys.argv[1]) if len(sys.argv) > 1 else 40
rng = np.random.default_rng(int(os.environ.get('FUZZ_SEED', 2024)))
eng = ScaleSpaceEngine(fuzz_cases.OCT)
bad = total = 0
for case in range(ncases):
    ok, n, d = fuzz_cases.parity_case(rng, eng)
    total += n
    if not ok:
        bad += 1
        print("MISMATCH case", case, d, flush=True)
    elif case % 10 == 0:
        print("case", case, d, flush=True)
print("done:", ncases, "cases,", bad, "mismatches; loops compared:", total)
