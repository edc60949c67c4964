"""Wall time of the whole CLI on BASELINE config 1's shape (chr21 @ 5 kb, text input + bias vector): writes a synthetic
RAWobserved / KRnorm pair (3 M lines), then times `python -m mustache_amd ...` as a subprocess (imports included)."""
import os, subprocess, sys, tempfile, time
import numpy as np
sys.path.insert(0, ".")
from mustache_amd.synth import synth_coo
n, dpx, res = 9630, 400, 5000
x, y, v = synth_coo(n, dpx, depth=300.0, seed=0, nloops=300)
d = tempfile.mkdtemp()
f, b, o = os.path.join(d, "chr21.RAWobserved"), os.path.join(d, "chr21.KRnorm"), os.path.join(d, "out.tsv")
t0 = time.time()
np.savetxt(f, np.column_stack([x * res, y * res, np.round(v) + 1]), fmt="%d\t%d\t%.1f")
open(b, "w").write("\n".join(repr(float(t)) for t in np.random.default_rng(1).uniform(0.6, 1.6, n + 1)) + "\n")
print("wrote %d lines in %.1f s" % (len(v), time.time() - t0))
for backend in ("native", "pandas"):
    env = dict(os.environ, MUSTACHE_TEXT_BACKEND=backend, PYTHONPATH=os.getcwd())
    t0 = time.time()
    r = subprocess.run([sys.executable, "-m", "mustache_amd", "-f", f, "-b", b, "-ch", "21", "-r", "5kb", "-pt", "0.1", "-st",
                        "0.8", "-o", o], env=env, capture_output=True, text=True)
    print(backend, "CLI wall %.2f s" % (time.time() - t0), r.stdout.strip().split("\n")[-1], r.stderr[-200:] if r.returncode else "")
