"""Binary-format readers against in-memory stand-ins for hic-straw / cooler (the real modules are optional and absent
offline): window walk, de-duplication across overlapping windows, distance filter, error behaviour."""
import sys
import types

import numpy as np
import pytest
from scipy import sparse


def _truth(n, dpx, seed):
    rng = np.random.default_rng(seed)
    x = rng.integers(0, n, 6000)
    y = np.minimum(x + rng.integers(0, dpx + 40, 6000), n - 1)
    key = np.unique(x * n + y)
    x, y = key // n, key % n
    v = rng.uniform(0.1, 9, len(x))
    v[::17] = np.nan
    return x, y, v


def test_window_ranges_match_reference_walk():
    from mustache_amd.readers import window_ranges
    # res 5 kb, dist 2 Mb -> W = 2000 bins = 10 Mb, step 8 Mb
    r = window_ranges(48_129_895, 2_000_000, 5000)
    assert r[0] == (0, 10_000_000) and r[1] == (8_000_000, 18_000_000)
    assert r[-1][1] == 48_129_894 and all(b[0] - a[0] == 8_000_000 for a, b in zip(r, r[1:]))
    one = window_ranges(3_000_000, 2_000_000, 5000)      # chromosome shorter than one window
    assert one[0] == (0, 3_000_000)


def test_read_hic_with_fake_straw(monkeypatch):
    n, res, dist = 9000, 5000, 2_000_000
    tx, ty, tv = _truth(n, dist // res, 1)
    calls = []

    def straw(kind, norm, f, loc1, loc2, unit, r):
        c, s, e = loc1.split(":")
        s, e = int(s), int(e)
        calls.append((norm, s, e))
        sel = (tx * res >= s) & (tx * res <= e) & (ty * res >= s) & (ty * res <= e)
        return [types.SimpleNamespace(binX=int(a * res), binY=int(b * res), counts=float(c_))
                for a, b, c_ in zip(tx[sel], ty[sel], tv[sel])]

    chrom = [types.SimpleNamespace(name="ALL", length=0), types.SimpleNamespace(name="7", length=n * res)]
    fake = types.ModuleType("hicstraw")
    fake.straw = straw
    fake.HiCFile = lambda f: types.SimpleNamespace(getChromosomes=lambda: chrom)
    monkeypatch.setitem(sys.modules, "hicstraw", fake)
    monkeypatch.setenv("MUSTACHE_HIC_BACKEND", "hicstraw")
    from mustache_amd.readers import read_hic_file, list_chromosomes
    x, y, v = read_hic_file("x.hic", False, False, dist, "7", "7", res)
    assert calls[0][0] == "KR" and len(calls) > 3
    exp = (~np.isnan(tv)) & (np.abs(tx - ty) <= dist / res) & (tv > 0)
    got = set(zip(x.tolist(), y.tolist(), v.tolist()))
    assert got == set(zip(tx[exp].tolist(), ty[exp].tolist(), tv[exp].tolist())), "every record once, filtered"
    assert list_chromosomes("x.hic", res) == ["7"]
    with pytest.raises(NameError):
        read_hic_file("x.hic", False, False, dist, "chr9", "chr9", res)


def test_read_cooler_with_fake_cooler(monkeypatch):
    n, res, dist = 7000, 5000, 2_000_000
    tx, ty, tv = _truth(n, dist // res, 2)
    full = sparse.coo_matrix((tv, (tx, ty)), shape=(n, n))
    full = (full + sparse.triu(full, 1).T).tocsr()

    class Fetcher:
        def fetch(self, region, region2=None):
            c, s, e = region
            a, b = s // res, -(-e // res)
            return full[a:b, a:b].tocoo()

    clr = types.SimpleNamespace(binsize=res, chromnames=["chr3", "chrM"], chromsizes={"chr3": n * res, 0: n * res, 1: 16000},
                                matrix=lambda balance, sparse: Fetcher())
    fake = types.ModuleType("cooler")
    fake.Cooler = lambda uri: clr
    monkeypatch.setitem(sys.modules, "cooler", fake)
    from mustache_amd.readers import read_cooler, read_mcooler, list_chromosomes
    x, y, v, r = read_cooler("a.cool", dist, "chr3", "chr3", False)
    assert r == res
    exp = (~np.isnan(tv)) & (np.abs(tx - ty) <= dist / res) & (tv > 0)
    assert set(zip(x.tolist(), y.tolist())) == set(zip(tx[exp].tolist(), ty[exp].tolist()))
    assert len(x) == exp.sum()
    x2, y2, v2 = read_mcooler("a.mcool", dist, "chr3", "chr3", res, False)
    assert len(x2) == len(x)
    with pytest.raises(NameError):
        read_cooler("a.cool", dist, "chrZ", "chrZ", False)
    assert list_chromosomes("a.cool", res) == ["chr3"]


def test_missing_optional_module_is_named(monkeypatch):
    monkeypatch.setitem(sys.modules, "hicstraw", None)
    monkeypatch.setenv("MUSTACHE_HIC_BACKEND", "hicstraw")
    from mustache_amd.readers import read_hic_file
    with pytest.raises(ImportError, match="hicstraw"):
        read_hic_file("x.hic", False, 1000, 2_000_000, "1", "1", 5000)
