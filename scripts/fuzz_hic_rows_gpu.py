#!/usr/bin/env python3
"""Random `.hic` files through both streamed reads on the GPU: rows decoded on the device (mst_band_scatter_hic_rows) against the
host decoder -- the cases of tests/test_gpu_hic_rows.py's seeded slice, as many as asked for.  GPU box.
    python scripts/fuzz_hic_rows_gpu.py [cases] [seed]"""
import os
import sys
import tempfile
import time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np      # noqa: E402
import torch            # noqa: E402
import fuzz_cases       # noqa: E402

cases = int(sys.argv[1]) if len(sys.argv) > 1 else 100
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
rng = np.random.default_rng(seed)
dev = torch.device("cuda", 0)
d = tempfile.mkdtemp(prefix="mst_fuzz_rows_")
t0, total, bad = time.time(), 0, 0
for case in range(cases):
    try:
        total += fuzz_cases.hic_rows_case(rng, os.path.join(d, "f.hic"), dev)
    except AssertionError as e:
        bad += 1
        print("case %d MISMATCH %s" % (case, e), flush=True)
    if case % 25 == 24:
        print("case %d: %d records so far, %d mismatches, %.0f s" % (case, total, bad, time.time() - t0), flush=True)
print("done: %d cases (seed %d), %d records, %d mismatches" % (cases, seed, total, bad))
