"""TEST INFRASTRUCTURE: a file-backed stand-in for the third-party `cooler` package (absent offline, needs HDF5), importable
by sub-processes through PYTHONPATH=tests/standins.  It implements exactly the surface the reference calls
(mustache.py:399-493, :1019-1036):

    clr = cooler.Cooler(uri)            # uri = path  or  path::/resolutions/<res>
    clr.binsize, clr.chromnames, clr.chromsizes[name | index]
    clr.matrix(balance=..., sparse=True).fetch((chrom, start_bp, end_bp))  -> scipy COO of the bin window, SYMMETRIC
    clr.matrix(balance=True, sparse=True).fetch(chrom1, chrom2)           (not provided: inter-chromosomal is dead code)

The "file" is an .npz written by `write_cool`: per chromosome the upper-triangular balanced contacts (bin i <= j, value;
NaN where a real cooler's weight is NaN)."""
import numpy as np
from scipy import sparse


class _Sizes:
    def __init__(self, names, sizes):
        self._n, self._s = list(names), [int(s) for s in sizes]

    def __getitem__(self, key):
        return self._s[key] if isinstance(key, (int, np.integer)) else self._s[self._n.index(key)]


class _Matrix:
    def __init__(self, clr):
        self._c = clr

    def fetch(self, region, region2=None):
        if region2 is not None or not isinstance(region, tuple):
            raise NotImplementedError("stand-in: only (chrom, start, end) windows")
        name, s, e = region
        res = self._c.binsize
        n = -(-self._c.chromsizes[name] // res)
        a, b = int(s) // res, min(n, -(-int(e) // res))
        x, y, v = self._c._coo[name]
        sel = (x >= a) & (x < b) & (y >= a) & (y < b)
        xs, ys, vs = x[sel] - a, y[sel] - a, v[sel]
        off = xs != ys
        rows = np.concatenate([xs, ys[off]])
        cols = np.concatenate([ys, xs[off]])
        return sparse.coo_matrix((np.concatenate([vs, vs[off]]), (rows, cols)), shape=(b - a, b - a))


class Cooler:
    def __init__(self, uri):
        path, _, group = str(uri).partition("::")
        z = np.load(path, allow_pickle=False)
        self.binsize = int(z["binsize"])
        if group and int(group.rsplit("/", 1)[-1]) != self.binsize:
            raise KeyError("resolution %s not in this file" % group)
        self.chromnames = [str(s) for s in z["chromnames"]]
        self.chromsizes = _Sizes(self.chromnames, z["chromsizes"])
        self._coo = {nm: (z["x_%d" % i].astype(np.int64), z["y_%d" % i].astype(np.int64), z["v_%d" % i])
                     for i, nm in enumerate(self.chromnames)}

    def matrix(self, balance=True, sparse=True):
        return _Matrix(self)


def write_cool(path, binsize, chroms):
    """chroms: [(name, size_bp, x, y, v)] with bin coordinates x <= y."""
    d = dict(binsize=binsize, chromnames=np.array([c[0] for c in chroms]), chromsizes=np.array([c[1] for c in chroms]))
    for i, (_, _, x, y, v) in enumerate(chroms):
        d["x_%d" % i], d["y_%d" % i], d["v_%d" % i] = np.asarray(x), np.asarray(y), np.asarray(v, dtype=np.float64)
    with open(path, "wb") as fh:          # np.savez would append ".npz" to a name ending in ".cool"
        np.savez(fh, **d)
