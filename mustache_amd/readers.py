"""Binary contact-map readers: `.hic` (native reader, or hic-straw), `.cool` / `.mcool` (cooler) -> upper-triangular COO
in bin units.

Host I/O, not part of the GPU hot path.  `.hic` files are read by libmustache_io.so
(include/mustache_io.h) by default; the third-party modules (hic-straw as an optional cross-check backend, cooler) are
imported lazily -- they are absent from the offline build image; a missing module raises an ImportError that names it.
Semantics follow reference mustache/mustache.py:300-396 (read_hic_file), :399-493 (read_cooler), :496-592
(read_mcooler): the chromosome is fetched in overlapping windows of max(2*dist/res, 2000) bins advancing by window - dist,
records seen in several windows are kept once, and intra-chromosomal output keeps |x - y| <= dist/res with value > 0.
Where the reference de-duplicates with Python set differences between consecutive windows, this module de-duplicates on
the (x, y) key with NumPy (same set of records; the reference's output order is set-iteration order and is not
reproduced -- nothing downstream depends on it here).
"""
import atexit
import importlib
import os
import threading

import numpy as np


def _need(module):
    try:
        return importlib.import_module(module)
    except ImportError as e:
        raise ImportError("reading this file type needs the optional third-party module %r (pip install %s)"
                          % (module, {"hicstraw": "hic-straw"}.get(module, module))) from e


def window_ranges(size_bp, distance_in_bp, res):
    """(start, end) base-pair windows exactly as the reference walks them (mustache.py:319-363): the first window ends at
    min(size, W), later ones are capped at size - 1, and the walk stops after the window that ends at size - 1."""
    W = max(2 * distance_in_bp / res, 2000) * res
    start, end = 0, min(size_bp, W)
    out = []
    while start < size_bp:
        out.append((int(start), int(end)))
        start = min(start + W - distance_in_bp, size_bp)
        if end == size_bp - 1:
            break
        end = min(end + W - distance_in_bp, size_bp - 1)
    return out


def _finish(x, y, v, distance_in_bp, res, intra, label):
    """Common tail (mustache.py:365-396, :477-493): drop NaN rows, NaN values -> 0, distance filter, value > 0."""
    x, y = np.asarray(x), np.asarray(y)
    v = np.asarray(v, dtype=np.float64)
    if len(v) == 0:
        print(f'There is no contact in chrmosome {label} to work on.')
        return [], [], []
    ok = ~(np.isnan(x.astype(np.float64)) | np.isnan(y.astype(np.float64)) | np.isnan(v))
    x, y, v = x[ok].astype(np.int64), y[ok].astype(np.int64), v[ok]
    if intra:
        keep = (np.abs(x - y) <= distance_in_bp / res) & (v > 0)
        x, y, v = x[keep], y[keep], v[keep]
    if len(v) == 0:
        print(f'There is no contact in chrmosome {label} to work on.')
        return [], [], []
    return x, y, v


def _dedup(xs, ys, vs):
    x, y, v = np.concatenate(xs), np.concatenate(ys), np.concatenate(vs)
    key = x.astype(np.int64) * (1 << 32) + y.astype(np.int64)
    _, first = np.unique(key, return_index=True)
    first.sort()
    return x[first], y[first], v[first]


def hic_backend():
    """Which reader serves `.hic` files: "hicstraw" (the reference's own dependency) or "native" (libmustache_io.so).
    Default ("auto"): hic-straw when that module is importable, the native reader otherwise -- the native reader's block
    decoding has no real `.hic` file or hic-straw dump to be pinned on in this build environment (only its header parse
    is pinned on reference-held code, include/mustache_io.h), so wherever the reference's dependency exists it stays the
    source of truth.  MUSTACHE_HIC_BACKEND=native|hicstraw forces one."""
    b = os.environ.get("MUSTACHE_HIC_BACKEND", "auto").lower()
    if b not in ("auto", "native", "hicstraw"):
        raise ValueError("MUSTACHE_HIC_BACKEND must be 'auto', 'native' or 'hicstraw'")
    if b == "auto":
        import importlib.util
        try:
            b = "hicstraw" if importlib.util.find_spec("hicstraw") is not None else "native"
        except (ImportError, ValueError):
            b = "native"
    return b


def _hic_chromosomes(f):
    """[(name, length)] without the leading pseudo-chromosome (mustache.py:308-312 skips index 0)."""
    if hic_backend() == "hicstraw":
        return [(str(c.name), c.length) for c in _need("hicstraw").HiCFile(f).getChromosomes()[1:]]
    from .hicfile import HicFile
    with HicFile(f) as h:
        return h.chromosomes()[1:]


def read_hic_file(f, norm_method, CHRM_SIZE, distance_in_bp, chr1, chr2, res):
    if not CHRM_SIZE:
        sizes = {"chr" + name.replace("chr", ''): length for name, length in _hic_chromosomes(f)}
        key = "chr" + str(chr1).replace("chr", '')
        if key not in sizes:
            raise NameError('wrong chromosome name!')
        CHRM_SIZE = sizes[key]
    norm = "KR" if not norm_method else str(norm_method)          # (:328-333)
    backend = hic_backend()
    print("reading %s through the %s .hic reader (MUSTACHE_HIC_BACKEND=auto|native|hicstraw)"
          % (os.path.basename(str(f)), backend if chr1 == chr2 or backend == "hicstraw" else "hicstraw"))
    if backend == "native" and chr1 == chr2:
        # one pass over the zlib blocks near the diagonal: the union of the reference's overlapping straw windows is every
        # record with |x - y| <= dist/res (consecutive windows overlap by exactly `dist`), already de-duplicated
        from .hicfile import HicFile
        with HicFile(f) as h:
            x, y, v = h.read_intra(chr1, res, norm, int(distance_in_bp // res))
        # NaN rows, value > 0 and the distance filter (mustache.py:370-388) are already applied by the reader; what is left
        # is straw's window end: no position past the chromosome size the caller gave
        if len(v) and int(y.max()) * res >= CHRM_SIZE:
            keep = y * res < CHRM_SIZE                                # x <= y
            x, y, v = x[keep], y[keep], v[keep]
        if len(v) == 0:
            print(f'There is no contact in chrmosome {chr1} to work on.')
            return [], [], []
        return x, y, v
    hicstraw = _need("hicstraw")
    xs, ys, vs = [], [], []
    for start, end in window_ranges(CHRM_SIZE, distance_in_bp, res):
        recs = hicstraw.straw("observed", norm, f, "%s:%d:%d" % (chr1, start, end), "%s:%d:%d" % (chr2, start, end),
                              "BP", res)
        if len(recs) == 0:
            continue
        xs.append(np.array([r.binX for r in recs], dtype=np.float64) // res)
        ys.append(np.array([r.binY for r in recs], dtype=np.float64) // res)
        vs.append(np.array([r.counts for r in recs], dtype=np.float64))
    if not xs:
        print(f'There is no contact in chrmosome {chr1} to work on.')
        return [], [], []
    x, y, v = _dedup(xs, ys, vs)
    return _finish(x, y, v, distance_in_bp, res, chr1 == chr2, chr1)


_HIC_LOCK = threading.RLock()
_HIC_HANDLE = [None, None]           # (path, HicFile): the packed reader's arenas live in the handle -- keep the last one open


def _hic_handle(f):
    from .hicfile import HicFile
    path = os.path.abspath(str(f))
    if _HIC_HANDLE[0] != path:
        if _HIC_HANDLE[1] is not None:
            _HIC_HANDLE[1].close()
        _HIC_HANDLE[0], _HIC_HANDLE[1] = path, HicFile(path)
    return _HIC_HANDLE[1]


def close_hic_handle():
    """Release the cached `.hic` handle (its mmap and the per-thread arenas sized to the largest chromosome read)."""
    with _HIC_LOCK:
        if _HIC_HANDLE[1] is not None:
            _HIC_HANDLE[1].close()
            import sys
            norm = sys.modules.get("mustache_amd.normalize")      # only if the GPU loader was ever imported
            if norm is not None:
                norm.release_slab_pool()                          # the streamed reads' page-locked slabs (up to 512 MB)
        _HIC_HANDLE[0], _HIC_HANDLE[1] = None, None


atexit.register(close_hic_handle)


def read_hic_packed(f, norm_method, CHRM_SIZE, distance_in_bp, chr1, res, part=(0, 1), device=None):
    """read_hic_file's records for an intra-chromosomal run as hicfile.PackedContacts (int32 bin, int32 distance, float32
    value: what the GPU loader mst_band_scatter_packed takes) -- same record set as read_hic_file through the native reader,
    without the int64 / float64 COO triple; None when the chromosome has no contact.  Native backend only.
    part = (rank, ranks): one process per GPU on ONE chromosome -- this rank decodes only its share of the file's blocks
    (the ranks read the file once between them; normalize.band_from_packed exchanges the shares).  A share may be empty;
    it is still returned, so that every rank takes part in the exchange.
    With a GPU the read is STREAMED (normalize.read_hic_stream_to_device): slabs of records are copied to `device` (default:
    the calling thread's current device -- pass it explicitly from a reader thread) while later blocks are still being
    inflated, and the result holds device tensors; MUSTACHE_HIC_STREAM=0 keeps the one-shot read into page-locked arrays."""
    from .hicfile import read_intra_packed
    norm = "KR" if not norm_method else str(norm_method)
    alloc = None
    try:
        import torch
        if torch.cuda.is_available():
            from .normalize import pinned_packed_alloc as alloc
    except ImportError:
        pass
    with _HIC_LOCK:                       # decode + fetch use the handle's arenas: one reader at a time per process
        h = _hic_handle(f)
        if not CHRM_SIZE:
            sizes = {"chr" + name.replace("chr", ''): length for name, length in h.chromosomes()[1:]}
            key = "chr" + str(chr1).replace("chr", '')
            if key not in sizes:
                raise NameError('wrong chromosome name!')
            CHRM_SIZE = sizes[key]
        print("reading %s through the native .hic reader, packed records (MUSTACHE_HIC_BACKEND=auto|native|hicstraw)"
              % os.path.basename(str(f)))
        if alloc is not None and os.environ.get("MUSTACHE_HIC_STREAM", "1") != "0":
            import torch
            from .normalize import read_hic_stream_to_device
            dev = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
            pc = read_hic_stream_to_device(h, chr1, res, norm, int(distance_in_bp // res), int(CHRM_SIZE), dev, part=part)
        else:
            pc = read_intra_packed(h, chr1, res, norm, int(distance_in_bp // res), int(CHRM_SIZE), alloc=alloc, part=part)
    if part[1] > 1:
        print("rank %d of %d decoded %d of the chromosome's %d .hic blocks (%d records) in %.3f s"
              % (part[0], part[1], pc.blocks_mine, pc.blocks_total, len(pc), pc.read_s), flush=True)
        return pc
    if len(pc) == 0:
        print(f'There is no contact in chrmosome {chr1} to work on.')
        return None
    return pc


def _read_cooler_obj(clr, distance_in_bp, chr1, chr2, res, cooler_balance):
    sparse = _need("scipy.sparse")
    if chr1 not in clr.chromnames or chr2 not in clr.chromnames:
        raise NameError('wrong chromosome name!')
    balance = True if not cooler_balance else cooler_balance     # (:422-425)
    if chr1 != chr2:
        m = sparse.triu(clr.matrix(balance=True, sparse=True).fetch(chr1, chr2)).tocoo()
        return _finish(m.row, m.col, np.nan_to_num(m.data, nan=0, posinf=0, neginf=0), distance_in_bp, res, False, chr1)
    size = clr.chromsizes[chr1]
    xs, ys, vs = [], [], []
    for start, end in window_ranges(size, distance_in_bp, res):
        m = sparse.triu(clr.matrix(balance=balance, sparse=True).fetch((chr1, start, end))).tocoo()
        if len(m.row) == 0:
            continue
        off = int(start / res)
        xs.append(off + m.row.astype(np.int64))
        ys.append(off + m.col.astype(np.int64))
        vs.append(np.nan_to_num(m.data.astype(np.float64), nan=0, posinf=0, neginf=0))
    if not xs:
        print(f'There is no contact in chrmosome {chr1} to work on.')
        return [], [], []
    x, y, v = _dedup(xs, ys, vs)
    return _finish(x, y, v, distance_in_bp, res, True, chr1)


def read_cooler(f, distance_in_bp, chr1, chr2, cooler_balance):
    cooler = _need("cooler")
    clr = cooler.Cooler(f)
    res = clr.binsize
    print(f'Your cooler data resolution is {res}')
    x, y, v = _read_cooler_obj(clr, distance_in_bp, chr1, chr2, res, cooler_balance)
    return x, y, v, res


def read_mcooler(f, distance_in_bp, chr1, chr2, res, cooler_balance):
    cooler = _need("cooler")
    clr = cooler.Cooler('%s::/resolutions/%s' % (f, res))
    try:
        return _read_cooler_obj(clr, distance_in_bp, chr1, chr2, res, cooler_balance)
    except NameError:
        raise
    except Exception as e:                      # (:558-559)
        raise NameError('Reading from the file failed!') from e


def chromosome_sizes(f, res):
    """{name: length in bp} for a .hic / .cool / .mcool file ({} when it cannot be told) -- only used to balance a
    whole-genome run over several GPUs."""
    try:
        if f.endswith(".hic"):
            return dict(_hic_chromosomes(f))
        if f.endswith(".cool") or f.endswith(".mcool"):
            cooler = _need("cooler")
            clr = cooler.Cooler(f if f.endswith(".cool") else '%s::/resolutions/%s' % (f, res))
            return {name: int(clr.chromsizes[name]) for name in clr.chromnames}
    except Exception:
        pass
    return {}


def list_chromosomes(f, res):
    """Chromosomes main() iterates when -ch is omitted (mustache.py:1019-1036)."""
    if f.endswith(".hic"):
        return [name for name, _ in _hic_chromosomes(f)]
    cooler = _need("cooler")
    clr = cooler.Cooler(f if f.endswith(".cool") else '%s::/resolutions/%s' % (f, res))
    return [name for i, name in enumerate(clr.chromnames) if clr.chromsizes[i] > 1000000]
