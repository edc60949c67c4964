#!/usr/bin/env python3
"""Wall time of the whole CLI on BASELINE config 4's shape: `python -m mustache_amd -f chr1_1kb.hic -ch chr1 -r 1kb ...` as a
subprocess (interpreter start, imports, GPU context, read, normalisation, kernels, tail, TSV), on the synthetic 650 MB file of
bench.py's file leg.  GPU box.   python scripts/cli_hic_time.py [runs]"""
import os
import shutil
import subprocess
import sys
import tempfile
import time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch                 # noqa: E402
import hic_writer            # noqa: E402

runs = int(sys.argv[1]) if len(sys.argv) > 1 else 3
d = tempfile.mkdtemp(prefix="mst_cli_")
try:
    f, o = os.path.join(d, "chr1_1kb.hic"), os.path.join(d, "out.tsv")
    t0 = time.time()
    nrec = hic_writer.write_synthetic_hic(f, 248957, 2000, 1000, 400.0, 8000, 1, 200.0, torch.device("cuda:0"))
    print("wrote %d records, %d bytes in %.1f s" % (nrec, os.path.getsize(f), time.time() - t0), flush=True)
    env = dict(os.environ, PYTHONPATH=ROOT)
    for i in range(runs):
        t0 = time.time()
        r = subprocess.run([sys.executable, "-X", "importtime", "-m", "mustache_amd", "-f", f, "-ch", "chr1", "-r", "1kb", "-norm", "KR",
                            "-pt", "0.1", "-st", "0.88", "-o", o, "-v", "True"] if os.environ.get("CLI_IMPORTTIME") else
                           [sys.executable, "-m", "mustache_amd", "-f", f, "-ch", "chr1", "-r", "1kb", "-norm", "KR",
                            "-pt", "0.1", "-st", "0.88", "-o", o], env=env, capture_output=True, text=True)
        dt = time.time() - t0
        lines = sum(1 for _ in open(o)) if os.path.exists(o) else -1
        print("run %d: CLI wall %.2f s, rc %d, %d output lines; last stdout line: %s" % (
            i, dt, r.returncode, lines, (r.stdout.strip().split("\n") or [""])[-1]), flush=True)
        if r.returncode:
            print(r.stderr[-800:])
    # the same with the stages timed inside the process
    code = r'''
import sys, time
t0 = time.time()
import torch
t1 = time.time()
torch.cuda.init(); torch.zeros(1, device="cuda"); torch.cuda.synchronize()
t2 = time.time()
from mustache_amd.mustache import main
t3 = time.time()
main(sys.argv[1:])
t4 = time.time()
print("STAGES import torch %.2f s | GPU context %.2f s | import mustache_amd %.2f s | main() %.2f s" % (t1 - t0, t2 - t1, t3 - t2, t4 - t3))
'''
    r = subprocess.run([sys.executable, "-c", code, "-f", f, "-ch", "chr1", "-r", "1kb", "-norm", "KR", "-pt", "0.1", "-st", "0.88",
                        "-o", o], env=env, capture_output=True, text=True)
    print((r.stdout.strip().split("\n") or [""])[-1], r.stderr[-500:] if r.returncode else "")
finally:
    shutil.rmtree(d, ignore_errors=True)
