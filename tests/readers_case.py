"""TEST INFRASTRUCTURE shared by tests/golden/make_golden.py (which drives the REFERENCE's read_cooler / read_mcooler /
read_hic_file with it) and tests/test_readers_ref.py (which drives mustache_amd.readers with the same objects): one
deterministic three-chromosome contact container plus the two stand-ins the readers talk to -- the file-backed `cooler`
module of tests/standins and a list-backed `hicstraw` that answers straw() window queries.

What the container holds on purpose (res 5 kb, distance limit 2 Mb -> windows of 2000 bins advancing by 1600):
  chrA  27.3 Mb  several windows; the last one ends at size - 1 (the walk's stop rule); records exactly ON window seams
                 (first / last bin of a window, inside the 400-bin overlap of two windows), records farther apart than the
                 limit, NaN values (a NaN balancing weight), zero and negative values
  chrB   6.0 Mb  shorter than one window
  chrC   0.8 Mb  below the 1 Mb threshold of the whole-genome chromosome list (mustache.py:1019-1028)
A second resolution (10 kb) of the same contacts serves the `.mcool` case.
"""
import os
import sys
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
RES = 5000
DIST = 2_000_000
CHROMS = [("chrA", 27_300_000), ("chrB", 6_000_000), ("chrC", 800_000)]


def contacts(size_bp, res, seed):
    """Upper-triangular (x <= y) bin contacts with the features listed in the module docstring."""
    n = -(-size_bp // res)
    rng = np.random.default_rng(seed)
    m = 2 * n
    x = rng.integers(0, n, m)
    span = int(DIST // res)
    y = np.minimum(x + rng.integers(0, span + 60, m), n - 1)        # some beyond the distance limit
    W, step = max(2 * span, 2000), max(2 * span, 2000) - span
    seams = []
    for s in range(0, n, step):                                     # window starts, window ends, overlap zone edges
        for b in (s, s + 1, s + W - 1, s + W, s + span, s + step - 1):
            if 0 <= b < n:
                for d in (0, 1, 7, span, span + 1):
                    if b + d < n:
                        seams.append((b, b + d))
                    if b - d >= 0:
                        seams.append((b - d, b))
    if seams:
        sx, sy = np.array(seams).T
        x, y = np.concatenate([x, sx]), np.concatenate([y, sy])
    key = np.unique(x.astype(np.int64) * n + y)
    x, y = key // n, key % n
    v = np.round(rng.uniform(0.05, 30.0, len(x)) * 64.0) / 64.0       # dyadic: exact through any float path, compresses well
    v[rng.integers(0, len(v), len(v) // 23)] = np.nan               # pixels of a bin whose weight is NaN
    v[rng.integers(0, len(v), len(v) // 41)] = 0.0
    v[rng.integers(0, len(v), len(v) // 97)] = -1.5
    return x, y, v


def container(res=RES):
    """[(name, size_bp, x, y, v)] at resolution `res`."""
    return [(name, size) + contacts(size, res, 100 + i + res % 97) for i, (name, size) in enumerate(CHROMS)]


def standin_cooler():
    """tests/standins/cooler.py loaded by path (whatever module, if any, is registered under the name `cooler`)."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("_standin_cooler", os.path.join(HERE, "standins", "cooler.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def write_cool_files(tmpdir):
    """-> (path of the 5 kb `.cool`, path of the `.mcool` holding the 10 kb resolution)."""
    cooler = standin_cooler()
    cool = os.path.join(str(tmpdir), "case.cool")
    mcool = os.path.join(str(tmpdir), "case.mcool")
    cooler.write_cool(cool, RES, container(RES))
    cooler.write_cool(mcool, 2 * RES, container(2 * RES))
    return cool, mcool


def fake_hicstraw(res=RES):
    """A module object with the two names the readers use: HiCFile(f).getChromosomes() and straw(...) -- the latter returns
    the records whose two positions (bin * res, as straw reports them) lie inside the queried base-pair window."""
    data = {name: (x, y, v) for name, _, x, y, v in container(res)}
    chrom = [types.SimpleNamespace(name="ALL", length=0)] + [types.SimpleNamespace(name=n, length=s) for n, s in CHROMS]
    calls = []

    def straw(kind, norm, f, loc1, loc2, unit, r):
        assert kind == "observed" and unit == "BP" and int(r) == res
        c1, s, e = loc1.split(":")
        c2, s2, e2 = loc2.split(":")
        assert (c1, s, e) == (c2, s2, e2)
        s, e = int(s), int(e)
        calls.append((norm, c1, s, e))
        x, y, v = data[c1]
        sel = (x * res >= s) & (x * res <= e) & (y * res >= s) & (y * res <= e)
        return [types.SimpleNamespace(binX=int(a * res), binY=int(b * res), counts=float(c))
                for a, b, c in zip(x[sel], y[sel], v[sel])]

    mod = types.ModuleType("hicstraw")
    mod.straw = straw
    mod.HiCFile = lambda f: types.SimpleNamespace(getChromosomes=lambda: chrom)
    mod.calls = calls
    return mod


def as_sorted(x, y, v):
    x, y, v = np.asarray(x, dtype=np.int64), np.asarray(y, dtype=np.int64), np.asarray(v, dtype=np.float64)
    o = np.lexsort((v, y, x))
    return x[o], y[o], v[o]


# ---- the whole-genome GPU test's container (tests/test_gpu_pipeline.py): realistic synthetic maps, so that loops come out
GENOME_RES, GENOME_DPX = 5000, 400
GENOME = (("chr1", 5200, 61), ("chr2", 4300, 62), ("chrX", 2900, 63))


def genome_container():
    """[(name, size_bp, x, y, v)]: three chromosomes at 5 kb with NaN-weight pixels + one below 1 Mb (never enumerated)."""
    root = os.path.dirname(HERE)
    if root not in sys.path:
        sys.path.insert(0, root)
    from mustache_amd.synth import synth_coo
    chroms = []
    for name, n, seed in GENOME:
        x, y, v = synth_coo(n, GENOME_DPX, depth=150.0, seed=seed)
        v = v.copy()
        v[::53] = np.nan                      # unbalanceable bins: NaN -> 0 -> dropped (mustache.py:463, :487)
        chroms.append((name, n * GENOME_RES, x, y, v))
    chroms.append(("chrM", 16571, np.array([0, 1]), np.array([1, 2]), np.array([3.0, 4.0])))
    return chroms


def digest(x, y, v):
    """sha256 over the sorted record set (int64 x, int64 y, float64 v bytes)."""
    import hashlib
    x, y, v = as_sorted(x, y, v)
    h = hashlib.sha256()
    for a in (x, y, v):
        h.update(np.ascontiguousarray(a).tobytes())
    return h.hexdigest()
