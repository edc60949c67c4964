#!/usr/bin/env python3
"""cProfile of the CLI's main() on the synthetic chr1 @ 1 kb `.hic` (torch and the GPU context warm): where the program's own
0.2 s go.  GPU box.   python scripts/cli_hic_profile.py"""
import cProfile
import io
import os
import pstats
import shutil
import sys
import tempfile
import time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch                 # noqa: E402
import hic_writer            # noqa: E402
from mustache_amd.mustache import main      # noqa: E402

d = tempfile.mkdtemp(prefix="mst_clip_")
try:
    f, o = os.path.join(d, "chr1_1kb.hic"), os.path.join(d, "out.tsv")
    hic_writer.write_synthetic_hic(f, 248957, 2000, 1000, 400.0, 8000, 1, 200.0, torch.device("cuda:0"))
    argv = ["-f", f, "-ch", "chr1", "-r", "1kb", "-norm", "KR", "-pt", "0.1", "-st", "0.88", "-o", o]
    for run in range(3):
        pr = cProfile.Profile()
        t0 = time.time()
        pr.enable()
        main(argv)
        pr.disable()
        print("run %d: main() %.3f s" % (run, time.time() - t0), flush=True)
        if run in (0, 2):
            s = io.StringIO()
            pstats.Stats(pr, stream=s).sort_stats("cumulative").print_stats(22)
            print("\n".join(ln[:150] for ln in s.getvalue().splitlines()[:45]))
finally:
    shutil.rmtree(d, ignore_errors=True)
