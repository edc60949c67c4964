"""ctypes binding of libmustache_io.so (include/mustache_io.h): the native `.hic` reader.

Host-only (g++ + zlib): it loads without a GPU.  `HicFile` mirrors the two things the reference asks of hic-straw in
read_hic_file() (reference mustache/mustache.py:308-312, :328-333): the chromosome table and the observed, normalised
intra-chromosomal records near the diagonal.
"""
import ctypes
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_lib = None
MST_IO_ABI_VERSION = 3          # include/mustache_io.h


class HicError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__("libmustache_io error %d: %s" % (code, msg))
        self.code = code


_P = ctypes.c_void_p
_SIGNATURES = {
    "mst_io_abi_version": (ctypes.c_int, []),
    "mst_io_last_error": (ctypes.c_char_p, []),
    "mst_io_free": (None, [_P]),
    "mst_io_inflate": (ctypes.c_int64, [_P, ctypes.c_int64, _P, ctypes.c_int64]),
    "mst_hic_open": (ctypes.c_int, [ctypes.c_char_p, ctypes.POINTER(_P)]),
    "mst_hic_close": (None, [_P]),
    "mst_hic_version": (ctypes.c_int32, [_P]),
    "mst_hic_master_offset": (ctypes.c_int64, [_P]),
    "mst_hic_genome": (ctypes.c_char_p, [_P]),
    "mst_hic_n_chromosomes": (ctypes.c_int32, [_P]),
    "mst_hic_chromosome": (ctypes.c_int, [_P, ctypes.c_int32, ctypes.POINTER(ctypes.c_char_p),
                                          ctypes.POINTER(ctypes.c_int64)]),
    "mst_hic_n_resolutions": (ctypes.c_int32, [_P]),
    "mst_hic_resolution": (ctypes.c_int32, [_P, ctypes.c_int32]),
    "mst_hic_read_intra": (ctypes.c_int64, [_P, ctypes.c_char_p, ctypes.c_int32, ctypes.c_char_p, ctypes.c_int64,
                                            ctypes.c_int32, ctypes.POINTER(_P), ctypes.POINTER(_P), ctypes.POINTER(_P)]),
    "mst_hic_read_intra_packed": (ctypes.c_int64, [_P, ctypes.c_char_p, ctypes.c_int32, ctypes.c_char_p, ctypes.c_int64,
                                                   ctypes.c_int64, ctypes.c_int32, ctypes.POINTER(_P), ctypes.POINTER(_P),
                                                   ctypes.POINTER(_P), ctypes.POINTER(ctypes.c_int64)]),
    "mst_hic_decode_intra_packed": (ctypes.c_int64, [_P, ctypes.c_char_p, ctypes.c_int32, ctypes.c_char_p, ctypes.c_int64,
                                                     ctypes.c_int64, ctypes.c_int32, ctypes.POINTER(ctypes.c_int64)]),
    "mst_hic_fetch_packed": (ctypes.c_int, [_P, _P, _P, _P, ctypes.c_int64, ctypes.c_int32]),
    "mst_hic_decode_intra_packed_part": (ctypes.c_int64, [_P, ctypes.c_char_p, ctypes.c_int32, ctypes.c_char_p,
                                                          ctypes.c_int64, ctypes.c_int64, ctypes.c_int32, ctypes.c_int32,
                                                          ctypes.c_int32, ctypes.POINTER(ctypes.c_int64),
                                                          ctypes.POINTER(ctypes.c_int32), ctypes.POINTER(ctypes.c_int32)]),
    "mst_hic_stream_open": (ctypes.c_int, [_P, ctypes.c_char_p, ctypes.c_int32, ctypes.c_char_p, ctypes.c_int64, ctypes.c_int64,
                                           ctypes.c_int32, ctypes.c_int32, ctypes.c_int32, _P, ctypes.c_int32, ctypes.c_int64,
                                           ctypes.c_int32, ctypes.POINTER(_P)]),
    "mst_hic_stream_next": (ctypes.c_int, [_P, ctypes.c_int32, ctypes.POINTER(ctypes.c_int32), ctypes.POINTER(ctypes.c_int64)]),
    "mst_hic_stream_release": (ctypes.c_int, [_P, ctypes.c_int32]),
    "mst_hic_stream_close": (ctypes.c_int, [_P, ctypes.POINTER(ctypes.c_int64), ctypes.POINTER(ctypes.c_int64),
                                            ctypes.POINTER(ctypes.c_int32), ctypes.POINTER(ctypes.c_int32)]),
    "mst_hic_rawstream_open": (ctypes.c_int, [_P, ctypes.c_char_p, ctypes.c_int32, ctypes.c_char_p, ctypes.c_int64, ctypes.c_int32,
                                              ctypes.c_int32, ctypes.c_int32, _P, ctypes.c_int32, ctypes.c_int64, ctypes.POINTER(_P)]),
    "mst_hic_rawstream_info": (ctypes.c_int, [_P, ctypes.POINTER(_P), ctypes.POINTER(ctypes.c_int64), ctypes.POINTER(ctypes.c_int64)]),
    "mst_hic_rawstream_next": (ctypes.c_int, [_P, ctypes.c_int32, ctypes.POINTER(ctypes.c_int32), ctypes.POINTER(ctypes.c_int64),
                                              ctypes.POINTER(ctypes.c_int32)]),
    "mst_hic_rawstream_release": (ctypes.c_int, [_P, ctypes.c_int32]),
    "mst_hic_rawstream_close": (ctypes.c_int, [_P, ctypes.POINTER(ctypes.c_int64), ctypes.POINTER(ctypes.c_int64),
                                               ctypes.POINTER(ctypes.c_int32), ctypes.POINTER(ctypes.c_int32)]),
    "mst_text_read_contacts": (ctypes.c_int64, [ctypes.c_char_p, ctypes.c_char, ctypes.c_char_p, ctypes.c_int32,
                                                ctypes.POINTER(ctypes.c_int32), ctypes.POINTER(_P), ctypes.POINTER(_P),
                                                ctypes.POINTER(_P)]),
    "mst_host_fill_block": (ctypes.c_int, [_P, ctypes.c_int64, ctypes.c_int64, ctypes.c_int32, ctypes.c_int32, ctypes.c_int32]),
}


def load():
    """Load libmustache_io.so once; raise if it has not been built (`make -C mustache_amd/csrc`)."""
    global _lib
    if _lib is None:
        path = os.environ.get("MUSTACHE_IO_LIB") or os.path.join(_HERE, "libmustache_io.so")
        if not os.path.exists(path):
            raise RuntimeError("libmustache_io.so is missing (%s): build it with `make -C mustache_amd/csrc`" % path)
        lib = ctypes.CDLL(path)
        for name, (res, args) in _SIGNATURES.items():
            fn = getattr(lib, name)
            fn.restype, fn.argtypes = res, args
        if lib.mst_io_abi_version() != MST_IO_ABI_VERSION:
            raise RuntimeError("libmustache_io.so ABI %d != %d expected (%s): rebuild with `make -C mustache_amd/csrc`"
                               % (lib.mst_io_abi_version(), MST_IO_ABI_VERSION, path))
        _lib = lib
    return _lib


def _check(lib, rc):
    if rc < 0:
        raise HicError(int(rc), lib.mst_io_last_error().decode("utf-8", "replace"))
    return rc


class HicFile:
    def __init__(self, path):
        self._lib = load()
        self._h = _P()
        _check(self._lib, self._lib.mst_hic_open(os.fsencode(path), ctypes.byref(self._h)))
        self.path = path

    def close(self):
        if self._h:
            self._lib.mst_hic_close(self._h)
            self._h = _P()

    __del__ = close

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    @property
    def version(self):
        return int(self._lib.mst_hic_version(self._h))

    @property
    def master_offset(self):
        return int(self._lib.mst_hic_master_offset(self._h))

    @property
    def genome(self):
        return self._lib.mst_hic_genome(self._h).decode()

    def chromosomes(self):
        """[(name, length)] in file order; entry 0 is usually the pseudo-chromosome "All"."""
        out = []
        for i in range(self._lib.mst_hic_n_chromosomes(self._h)):
            name, length = ctypes.c_char_p(), ctypes.c_int64()
            _check(self._lib, self._lib.mst_hic_chromosome(self._h, i, ctypes.byref(name), ctypes.byref(length)))
            out.append((name.value.decode(), int(length.value)))
        return out

    def resolutions(self):
        return [int(self._lib.mst_hic_resolution(self._h, i)) for i in range(self._lib.mst_hic_n_resolutions(self._h))]

    def read_intra(self, chrom, resolution, norm="KR", max_dist_bins=-1, threads=0):
        """(x, y, v): bin indices (x <= y, int64) and normalised observed values (float64 holding straw's float32) of the
        records with y - x <= max_dist_bins, value > 0, not NaN."""
        px, py, pv = _P(), _P(), _P()
        n = _check(self._lib, self._lib.mst_hic_read_intra(self._h, str(chrom).encode(), int(resolution),
                                                           str(norm).encode(), int(max_dist_bins), int(threads),
                                                           ctypes.byref(px), ctypes.byref(py), ctypes.byref(pv)))
        try:
            if n == 0:
                return np.zeros(0, np.int64), np.zeros(0, np.int64), np.zeros(0, np.float64)
            x = np.ctypeslib.as_array(ctypes.cast(px, ctypes.POINTER(ctypes.c_int64)), shape=(n,)).copy()
            y = np.ctypeslib.as_array(ctypes.cast(py, ctypes.POINTER(ctypes.c_int64)), shape=(n,)).copy()
            v = np.ctypeslib.as_array(ctypes.cast(pv, ctypes.POINTER(ctypes.c_double)), shape=(n,)).copy()
        finally:
            for p in (px, py, pv):
                self._lib.mst_io_free(p)
        return x, y, v


class HicStream:
    """mst_hic_stream_*: the packed records of one chromosome (or of share `part` of its blocks) delivered slab by slab into
    caller-owned memory while later blocks are still being inflated.  `memory_ptr` points to n_slabs * slab_records *
    (8 + dist_bytes) bytes (page-locked for GPU uploads); see include/mustache_io.h for the slab layout."""

    def __init__(self, hic, chrom, resolution, norm, max_dist_bins, chrom_size_bp, memory_ptr, n_slabs, slab_records,
                 dist_bytes=2, threads=0, part=(0, 1)):
        self._lib, self._hic = hic._lib, hic
        self._s = _P()
        self.slab_records, self.dist_bytes, self.n_slabs = int(slab_records), int(dist_bytes), int(n_slabs)
        _check(self._lib, self._lib.mst_hic_stream_open(hic._h, str(chrom).encode(), int(resolution), str(norm).encode(),
                                                        int(max_dist_bins), int(chrom_size_bp), int(threads), int(part[0]),
                                                        int(part[1]), _P(memory_ptr), int(n_slabs), int(slab_records),
                                                        int(dist_bytes), ctypes.byref(self._s)))
        self.n = self.total = self.blocks_total = self.blocks_mine = None

    def next(self, timeout_ms=-1):
        """(slab index, record count) of a filled slab; None when nothing was ready within timeout_ms; False at the end."""
        slab, count = ctypes.c_int32(), ctypes.c_int64()
        rc = _check(self._lib, self._lib.mst_hic_stream_next(self._s, int(timeout_ms), ctypes.byref(slab), ctypes.byref(count)))
        if rc == 1:
            return int(slab.value), int(count.value)
        return None if rc == 2 else False

    def release(self, slab):
        _check(self._lib, self._lib.mst_hic_stream_release(self._s, int(slab)))

    def close(self):
        if self._s:
            n, tot, bt, bm = ctypes.c_int64(), ctypes.c_int64(), ctypes.c_int32(), ctypes.c_int32()
            s, self._s = self._s, _P()
            _check(self._lib, self._lib.mst_hic_stream_close(s, ctypes.byref(n), ctypes.byref(tot), ctypes.byref(bt),
                                                             ctypes.byref(bm)))
            self.n, self.total, self.blocks_total, self.blocks_mine = int(n.value), int(tot.value), int(bt.value), int(bm.value)

    __del__ = close


class HicRawStream:
    """mst_hic_rawstream_*: the RAW rows of one chromosome's near-diagonal blocks (or of share `part` of them) delivered slab by
    slab into caller-owned memory while later blocks are still being inflated -- record bytes as the file stores them plus one
    16-byte directory entry per row (include/mustache_hicrow.h); the rows are decoded on the GPU (mst_band_scatter_hic_rows).
    `memory_ptr` points to n_slabs * slab_bytes bytes (page-locked).  Versions 7-9 only."""

    def __init__(self, hic, chrom, resolution, norm, max_dist_bins, memory_ptr, n_slabs, slab_bytes, threads=0, part=(0, 1)):
        self._lib, self._hic = hic._lib, hic
        self._s = _P()
        self.slab_bytes, self.n_slabs = int(slab_bytes), int(n_slabs)
        _check(self._lib, self._lib.mst_hic_rawstream_open(hic._h, str(chrom).encode(), int(resolution), str(norm).encode(),
                                                           int(max_dist_bins), int(threads), int(part[0]), int(part[1]),
                                                           _P(memory_ptr), int(n_slabs), int(slab_bytes), ctypes.byref(self._s)))
        self.rows_total = self.bytes_total = self.blocks_total = self.blocks_mine = None

    def info(self):
        """(the chromosome's normalisation vector as a float64 array (a copy) or None for norm NONE, its length in bp)"""
        v, n, length = _P(), ctypes.c_int64(), ctypes.c_int64()
        _check(self._lib, self._lib.mst_hic_rawstream_info(self._s, ctypes.byref(v), ctypes.byref(n), ctypes.byref(length)))
        if n.value < 0:
            return None, int(length.value)
        if n.value == 0:
            return np.zeros(0, np.float64), int(length.value)
        return np.ctypeslib.as_array(ctypes.cast(v, ctypes.POINTER(ctypes.c_double)), shape=(n.value,)).copy(), int(length.value)

    def next(self, timeout_ms=-1):
        """(slab index, payload bytes, rows) of a filled slab; None when nothing was ready within timeout_ms; False at the end."""
        slab, nbytes, rows = ctypes.c_int32(), ctypes.c_int64(), ctypes.c_int32()
        rc = _check(self._lib, self._lib.mst_hic_rawstream_next(self._s, int(timeout_ms), ctypes.byref(slab), ctypes.byref(nbytes),
                                                                ctypes.byref(rows)))
        if rc == 1:
            return int(slab.value), int(nbytes.value), int(rows.value)
        return None if rc == 2 else False

    def release(self, slab):
        _check(self._lib, self._lib.mst_hic_rawstream_release(self._s, int(slab)))

    def close(self):
        if self._s:
            rt, bt_, bt, bm = ctypes.c_int64(), ctypes.c_int64(), ctypes.c_int32(), ctypes.c_int32()
            s, self._s = self._s, _P()
            _check(self._lib, self._lib.mst_hic_rawstream_close(s, ctypes.byref(rt), ctypes.byref(bt_), ctypes.byref(bt),
                                                                ctypes.byref(bm)))
            self.rows_total, self.bytes_total = int(rt.value), int(bt_.value)
            self.blocks_total, self.blocks_mine = int(bt.value), int(bm.value)

    __del__ = close


class PackedContacts:
    """Records of one chromosome as the native reader hands them to the GPU loader (mst_band_from_packed): x = binX (int32),
    dist = binY - binX (int32), v = straw's float32 value; `n` = max(binY) + 1 (mustache.py:894), `res` the resolution.
    `pinned`: the three torch tensors (page-locked host memory) the arrays are views of, when the caller's allocator
    provided such -- the upload then runs at the full PCIe rate; None for plain NumPy arrays."""

    def __init__(self, x, dist, v, n, res, pinned=None, part=0, n_parts=1, blocks_total=None, blocks_mine=None, count=None):
        self.x, self.dist, self.v = x, dist, v
        self.count, self.n, self.res, self.pinned = int(len(v) if count is None else count), int(n), int(res), pinned
        # n_parts > 1: this object holds only the records of part `part` of the chromosome's `.hic` blocks (one process per
        # GPU, each rank decodes its share: read_intra_packed(part=...)); `n` is then this part's max(binY) + 1 and the
        # device loader (normalize.band_from_packed) exchanges the parts between the ranks before it scatters
        self.part, self.n_parts = int(part), int(n_parts)
        self.blocks_total, self.blocks_mine = blocks_total, blocks_mine
        self.read_s = None
        # streamed reads (normalize.read_hic_stream_to_device): the records already sit in device memory as a list of
        # (x int32, dist uint16 | int32, v float32, count) tensors, one per slab; x / dist / v above are then None
        self.device_parts = None
        # raw streamed reads (normalize.read_hic_stream_to_device, `.hic` v7-9): the rows were decoded ON THE DEVICE straight
        # into `device_band` ([dpx + 2][n_alloc] float64, this rank's share scattered, `band_stats` = the kernel's uint64 [4]
        # counters); `raw_parts` = the device copies of the slabs (payload bytes, row directory, rows), kept only when other
        # ranks need them (n_parts > 1) or a read-back check was asked for; `raw_ctx` = what the kernel call needs again
        self.device_band = self.band_stats = self.raw_parts = self.raw_ctx = None

    def __len__(self):
        return self.count

    def coo(self):
        """(x, y, v) as the reference's int64 / float64 COO (copies) -- for callers that want the classic triple."""
        if self.device_band is not None:
            import torch
            if self.n_parts > 1:
                raise RuntimeError("coo() of one rank's share of a raw streamed read: call normalize.band_from_packed first")
            d, x = torch.nonzero(self.device_band[:, :self.n], as_tuple=True)
            v = self.device_band[d, x]
            order = torch.argsort(x * (self.device_band.shape[0]) + d)
            x, d, v = x[order].cpu().numpy(), d[order].cpu().numpy(), v[order].cpu().numpy()
            return x, x + d, v
        if self.device_parts is not None:
            x = np.concatenate([p[0][:p[3]].cpu().numpy() for p in self.device_parts] or [np.zeros(0, np.int32)]).astype(np.int64)
            d = np.concatenate([p[1][:p[3]].cpu().numpy() for p in self.device_parts] or [np.zeros(0, np.int32)]).astype(np.int64)
            v = np.concatenate([p[2][:p[3]].cpu().numpy() for p in self.device_parts] or [np.zeros(0, np.float32)])
            return x, x + d, v.astype(np.float64)
        x = self.x.astype(np.int64)
        return x, x + self.dist.astype(np.int64), self.v.astype(np.float64)


def read_intra_packed(hic, chrom, resolution, norm="KR", max_dist_bins=-1, chrom_size_bp=0, threads=0, alloc=None,
                      part=(0, 1)):
    """HicFile -> PackedContacts: mst_hic_decode_intra_packed (records stay in the handle's per-thread arenas) +
    mst_hic_fetch_packed into arrays from `alloc(count)` -> (x int32, dist int32, v float32, keepalive) -- NumPy arrays by
    default; mustache_amd.normalize.pinned_packed_alloc hands out views of page-locked torch tensors.
    part = (p, n): decode only share p of n of the chromosome's blocks (one process per GPU: the ranks read the file once
    between them and exchange the records afterwards)."""
    import time
    t0 = time.time()
    nb = ctypes.c_int64()
    bt, bm = ctypes.c_int32(), ctypes.c_int32()
    n = _check(hic._lib, hic._lib.mst_hic_decode_intra_packed_part(hic._h, str(chrom).encode(), int(resolution),
                                                                   str(norm).encode(), int(max_dist_bins),
                                                                   int(chrom_size_bp), int(threads), int(part[0]),
                                                                   int(part[1]), ctypes.byref(nb), ctypes.byref(bt),
                                                                   ctypes.byref(bm)))
    if alloc is None:
        x, d, v, keep = np.empty(n, np.int32), np.empty(n, np.int32), np.empty(n, np.float32), None
    else:
        x, d, v, keep = alloc(n)
    if n:
        _check(hic._lib, hic._lib.mst_hic_fetch_packed(hic._h, x.ctypes.data_as(_P), d.ctypes.data_as(_P),
                                                       v.ctypes.data_as(_P), int(n), int(threads)))
    pc = PackedContacts(x, d, v, nb.value, resolution, pinned=keep, part=part[0], n_parts=part[1], blocks_total=bt.value,
                        blocks_mine=bm.value)
    pc.read_s = time.time() - t0
    return pc


def read_text_contacts(path, sep, chromosome=None, threads=0):
    """(n_cols, pos1, pos2, count): the numeric columns of a 3- or 5-column contact text file as float64 arrays, the rows
    pandas.read_csv(path, sep=sep, header=None).dropna() keeps (5 columns: those whose two chromosome fields match
    `chromosome`), parsed bit for bit like pandas' default converter.  Raises HicError(code -3) for files the native parser
    does not cover -- the caller then uses pandas itself."""
    lib = load()
    ncols = ctypes.c_int32()
    pa, pb, pc = _P(), _P(), _P()
    chrom = None if chromosome is None else str(chromosome).encode()
    n = _check(lib, lib.mst_text_read_contacts(os.fsencode(path), str(sep).encode()[:1], chrom, int(threads),
                                               ctypes.byref(ncols), ctypes.byref(pa), ctypes.byref(pb), ctypes.byref(pc)))
    try:
        if n == 0:
            z = np.zeros(0, np.float64)
            return int(ncols.value), z, z.copy(), z.copy()
        out = [np.ctypeslib.as_array(ctypes.cast(p, ctypes.POINTER(ctypes.c_double)), shape=(n,)).copy() for p in (pa, pb, pc)]
    finally:
        for p in (pa, pb, pc):
            lib.mst_io_free(p)
    return (int(ncols.value),) + tuple(out)
