"""Diagonal-distance normalisation on the GPU (reference mustache/mustache.py:622-686).

Device layout: the diagonal-major band (see csrc/mst_band.hip).  `normalize_band` works on device tensors and is
what the pipeline uses; `normalize_sparse_device` is the drop-in with the reference's host-array signature."""
import numpy as np
import torch

from . import _lib
from .engine import _ptr, _stream, require_gpu


def band_from_coo(x, y, v, n, dpx):
    """x, y int64 / v float64 device tensors -> band [dpx+2, n] float64 (device)."""
    lib = require_gpu()
    band = torch.empty((dpx + 2, n), dtype=torch.float64, device=v.device)
    with torch.cuda.device(v.device):
        _lib.check(lib.mst_band_from_coo(_ptr(x), _ptr(y), _ptr(v), int(v.numel()), int(n), int(dpx), _ptr(band),
                                         _stream()))
    return band


def band_from_host_coo(x, y, v, n, dpx, device):
    """Host COO (the readers' output) -> band on `device`, with the reference's rule for repeated pixels.  The reference
    writes the entries of a diagonal in input order (`vals[x[indices]] = v[indices]`, mustache.py:633-635; `cc[xc, yc] = vc`,
    :921-924): the LAST entry of a repeated (x, y) wins.  The device scatter is a plain racing store, so repeated pixels are
    looked for first (every entry reads its pixel back from the band: a mismatch means two entries with different values
    share a pixel) and, only if there are any, removed on the host keeping the last one.
    (The reference's per-diagonal mean / std run over ALL entries, repeats included, :636-639; after the de-duplication a
    repeated pixel counts once here -- stated in DESIGN.md; a contact list with repeats is malformed input for both.)"""
    import warnings
    xd, yd, vd = (torch.from_numpy(np.ascontiguousarray(a)).to(device) for a in (x, y, v))
    band = band_from_coo(xd, yd, vd, n, dpx)
    # every entry reads its pixel back: a pixel written by two entries with DIFFERENT values (zero included) shows up as a
    # mismatch for one of them whichever store won the race; repeats of equal value change nothing.  nnz-sized, exact.
    back = vd.clone()
    band_to_coo(band, xd, yd, back, n, dpx)
    if bool((back != vd).any().item()):
        xs, ys, vs = np.asarray(x, dtype=np.int64), np.asarray(y, dtype=np.int64), np.asarray(v, dtype=np.float64)
        key = np.minimum(xs, ys) * np.int64(n) + np.maximum(xs, ys)
        order = np.argsort(key, kind="stable")
        last = np.ones(len(key), bool)
        last[:-1] = key[order][1:] != key[order][:-1]          # last entry of every run of equal keys, in input order
        keep = np.sort(order[last])
        warnings.warn("contact list holds %d repeated pixel(s): the last entry of each is used (as the reference's scatter does)"
                      % (len(key) - len(keep)))
        del band, xd, yd, vd
        xd, yd, vd = (torch.from_numpy(np.ascontiguousarray(a[keep])).to(device) for a in (xs, ys, vs))
        band = band_from_coo(xd, yd, vd, n, dpx)
    return band


def pinned_packed_alloc(count):
    """Allocator for hicfile.read_intra_packed: three page-locked torch tensors (PyTorch's caching host allocator recycles
    them from chromosome to chromosome) and NumPy views of them for the reader to fill."""
    ts = (torch.empty(count, dtype=torch.int32, pin_memory=True), torch.empty(count, dtype=torch.int32, pin_memory=True),
          torch.empty(count, dtype=torch.float32, pin_memory=True))
    return ts[0].numpy(), ts[1].numpy(), ts[2].numpy(), ts


_SLAB_POOL = {}
_SLAB_POOL_CAP = 512 << 20          # page-locked bytes the streamed reads may keep (MUSTACHE_HIC_POOL_MB overrides)
# ONE page-locked buffer serves every streamed read of the process, so the reads take turns: the pipeline's reader thread may be
# streaming the next chromosome while the main thread re-reads the current one (band_from_packed's check / duplicate-pixel
# fallback calls `reread` outside readers._HIC_LOCK).  Re-entrant: the raw read's own fallback to the packed read nests.
import threading
_SLAB_LOCK = threading.RLock()


def _slab_pool(nbytes):
    """Page-locked slab memory for the streaming `.hic` reads, kept until release_slab_pool() (a whole-genome run reuses it
    chromosome after chromosome; pinning fresh pages costs ~0.3 s per GB).  One buffer, grown when a read asks for more."""
    buf = _SLAB_POOL.get("buf")
    if buf is None or buf.numel() < nbytes:
        _SLAB_POOL.clear()
        buf = _SLAB_POOL["buf"] = torch.empty(int(nbytes), dtype=torch.uint8, pin_memory=True)
    return buf


def release_slab_pool():
    """give the page-locked slab memory back (readers.close_hic_handle calls this when the last `.hic` file is closed)"""
    _SLAB_POOL.clear()


def _slab_count(threads, slab_bytes):
    """every worker thread fills one slab at a time; as many again (+ 8) keep uploads in flight while they do -- within the
    pool's cap, but never fewer than the workers + 2"""
    import os
    env = int(os.environ.get("MUSTACHE_HIC_SLABS", "0"))
    if env:
        return env
    cap = int(os.environ.get("MUSTACHE_HIC_POOL_MB", "0")) << 20 or _SLAB_POOL_CAP
    return max(8, min(256, 2 * threads + 8, max(threads + 2, cap // slab_bytes)))


def read_hic_stream_to_device(hic, chrom, res, norm, dpx, chrom_size_bp, device, part=(0, 1), threads=0,
                              slab_records=1 << 19, n_slabs=None, raw=None, keep_raw=False):
    """hicfile.HicFile -> hicfile.PackedContacts whose contacts already sit in DEVICE memory when this returns; the PCIe
    transfer runs under the inflate of the later blocks instead of after it.  part = (rank, ranks): this rank's share of the
    blocks only.  band_from_packed() takes the result (and exchanges the shares between the ranks first when there are several).

    raw (default: `.hic` versions 7-9 unless MUSTACHE_HIC_RAW=0): the host ONLY INFLATES -- worker threads copy each row's record
    bytes as the file stores them (6 bytes per record) into page-locked slabs with a 16-byte directory entry per row
    (mst_hic_rawstream_*), every slab goes to the device on the copy stream and mst_band_scatter_hic_rows decodes the rows,
    divides by the normalisation vector, filters and scatters straight into the band (pc.device_band) right behind the copy.
    Otherwise (v6 files, raw=False): the workers decode into slabs of packed records (binX int32, value float32, distance
    uint16: 10 bytes per record, pc.device_parts) and band_from_packed scatters them."""
    import os
    import time
    from .hicfile import HicStream, PackedContacts
    require_gpu()
    if raw is None:
        raw = hic.version >= 7 and os.environ.get("MUSTACHE_HIC_RAW", "1") != "0"
    if not threads:
        threads = max(4, _reader_threads() // max(1, int(part[1])))       # ranks of one node share its cores
    if raw:
        return _read_hic_raw_to_band(hic, chrom, res, norm, dpx, chrom_size_bp, device, part, threads, slab_records, n_slabs,
                                     keep_raw)
    t0 = time.time()
    dist_bytes = 2 if dpx + 1 <= 65535 else 4
    slab_records = int(os.environ.get("MUSTACHE_HIC_SLAB_RECORDS", "0")) or slab_records
    slab_records += slab_records & 1              # even: every array of a slab stays 4-byte aligned
    slab_bytes = slab_records * (8 + dist_bytes)
    if n_slabs is None:
        n_slabs = _slab_count(threads, slab_bytes)
    from .engine import device_streams
    side = device_streams(torch.device(device))[2]          # the process's one copy stream of this device
    ddt = torch.uint16 if dist_bytes == 2 else torch.int32
    _SLAB_LOCK.acquire()
    st = None
    parts, pending = [], []
    try:
        pool = _slab_pool(n_slabs * slab_bytes)
        st = HicStream(hic, chrom, res, norm, int(dpx), int(chrom_size_bp), pool.data_ptr(), n_slabs, slab_records, dist_bytes,
                       threads=threads, part=part)
        while True:
            got = st.next(2 if pending else -1)
            # slabs whose copies have completed go back to the workers
            while pending and pending[0][0].query():
                st.release(pending.pop(0)[1])
            if got is None:
                continue
            if got is False:
                break
            slab, cnt = got
            base = slab * slab_bytes
            full = cnt == slab_records
            with torch.cuda.stream(side):
                if full:
                    # a full slab (all but each worker's last) goes over in ONE copy: its three arrays are contiguous
                    dv = pool[base:base + slab_bytes].to(device, non_blocking=True)
                    xd = dv[:4 * cnt].view(torch.int32)
                    vd = dv[4 * slab_records:4 * slab_records + 4 * cnt].view(torch.float32)
                    dd = dv[8 * slab_records:8 * slab_records + dist_bytes * cnt].view(ddt)
                else:
                    hx = pool[base:base + 4 * cnt].view(torch.int32)
                    hv = pool[base + 4 * slab_records:base + 4 * slab_records + 4 * cnt].view(torch.float32)
                    hd = pool[base + 8 * slab_records:base + 8 * slab_records + dist_bytes * cnt].view(ddt)
                    xd, dd, vd = (t.to(device, non_blocking=True) for t in (hx, hd, hv))
                ev = side.record_event()
            parts.append((xd, dd, vd, cnt))
            pending.append((ev, slab))
    finally:
        side.synchronize()          # also on the error path: no copy out of the pool may be in flight when the next read fills it
        if st is not None:
            st.close()
        _SLAB_LOCK.release()
    torch.cuda.current_stream(device).wait_stream(side)
    pc = PackedContacts(None, None, None, st.n, res, part=part[0], n_parts=part[1], blocks_total=st.blocks_total,
                        blocks_mine=st.blocks_mine, count=st.total)
    pc.device_parts = parts
    pc.read_s = time.time() - t0
    return pc


def _scatter_rows(lib, pay, rows, n_rows, ctx, band, stats, stream, verify=0):
    _lib.check(lib.mst_band_scatter_hic_rows(_ptr(pay), _ptr(rows), int(n_rows), _ptr(ctx["norm"]) if ctx["norm"] is not None else None,
                                             ctx["n_norm"], ctx["max_dist"], ctx["y_limit"], ctx["n_alloc"], ctx["dpx"], _ptr(band),
                                             _ptr(stats), int(verify), stream))


def _read_hic_raw_to_band(hic, chrom, res, norm, dpx, chrom_size_bp, device, part, threads, slab_records, n_slabs, keep_raw):
    """the raw form of read_hic_stream_to_device (see there).  The band is allocated for every bin the chromosome can hold --
    ceil(length / res) from the file's header, or the caller's smaller size -- and trimmed by band_from_packed to the
    reference's n = max(binY) + 1 over the records that survive the filters (mustache.py:894), which only the kernel knows."""
    import os
    import time
    from .engine import device_streams
    from .hicfile import HicRawStream, PackedContacts
    lib = require_gpu()
    t0 = time.time()
    device = torch.device(device)
    slab_bytes = int(os.environ.get("MUSTACHE_HIC_SLAB_BYTES", "0")) or 10 * int(os.environ.get("MUSTACHE_HIC_SLAB_RECORDS", "0")) \
        or 10 * int(slab_records)
    slab_bytes = max(4096, (slab_bytes + 15) // 16 * 16)
    if n_slabs is None:
        n_slabs = _slab_count(threads, slab_bytes)
    side = device_streams(device)[2]                        # the process's one copy stream of this device
    keep = keep_raw or part[1] > 1 or bool(os.environ.get("MUSTACHE_CHECK_PACKED"))
    parts, pending = [], []
    _SLAB_LOCK.acquire()
    st = None
    try:
        pool = _slab_pool(n_slabs * slab_bytes)
        st = HicRawStream(hic, chrom, res, norm, int(dpx), pool.data_ptr(), n_slabs, slab_bytes, threads=threads, part=part)
        normv, length = st.info()
        y_limit = -(-int(chrom_size_bp) // int(res)) if chrom_size_bp and chrom_size_bp > 0 else 0
        n_alloc = max(1, -(-int(length) // int(res)))
        if y_limit:
            n_alloc = max(1, min(n_alloc, y_limit))
        with torch.cuda.stream(side):
            band = torch.zeros((dpx + 2, n_alloc), dtype=torch.float64, device=device)
            stats = torch.zeros(4, dtype=torch.int64, device=device)
            ctx = {"norm": None if normv is None else torch.from_numpy(normv).to(device), "n_norm": -1 if normv is None else len(normv),
                   "max_dist": int(dpx), "y_limit": y_limit, "n_alloc": n_alloc, "dpx": int(dpx)}
        while True:
            got = st.next(2 if pending else -1)
            while pending and pending[0][0].query():          # slabs whose copies have completed go back to the workers
                st.release(pending.pop(0)[1])
            if got is None:
                continue
            if got is False:
                break
            slab, nbytes, rows = got
            base = slab * slab_bytes
            with torch.cuda.stream(side), torch.cuda.device(device):
                if slab_bytes - nbytes - 16 * rows <= slab_bytes // 16:
                    dv = pool[base:base + slab_bytes].to(device, non_blocking=True)       # a full slab goes over in ONE copy
                    pay, dr = dv[:nbytes], dv[slab_bytes - 16 * rows:]
                else:
                    pay = pool[base:base + nbytes].to(device, non_blocking=True)
                    dr = pool[base + slab_bytes - 16 * rows:base + slab_bytes].to(device, non_blocking=True)
                ev = side.record_event()
                _scatter_rows(lib, pay, dr, rows, ctx, band, stats, side.cuda_stream)
            if keep:
                parts.append((pay, dr, rows))
            pending.append((ev, slab))
    finally:
        side.synchronize()          # also on the error path: no copy out of the pool may be in flight when the next read fills it
        if st is not None:
            st.close()
        _SLAB_LOCK.release()
    ymax1, kept, beyond, _ = (int(a) for a in stats.cpu().numpy())
    if beyond and part[1] == 1:
        import warnings
        warnings.warn("%s: %d record(s) lie beyond the chromosome length the file's header gives (%d bp): reading it again through "
                      "the host decoder" % (chrom, beyond, length))
        del band, parts
        return read_hic_stream_to_device(hic, chrom, res, norm, dpx, chrom_size_bp, device, part=part, threads=threads,
                                         slab_records=slab_records, raw=False)
    # (several ranks: a rank that alone fell back here would enter all_gather_packed while the others enter all_gather_raw -- two
    #  collectives with different payloads.  The decision is taken in _band_from_raw AFTER the exchange, where every rank has
    #  decoded every rank's rows and therefore holds the same `beyond` count: all of them re-read, or none.)
    pc = PackedContacts(None, None, None, ymax1, res, part=part[0], n_parts=part[1], blocks_total=st.blocks_total,
                        blocks_mine=st.blocks_mine, count=kept)
    pc.device_band, pc.band_stats, pc.raw_parts, pc.raw_ctx = band, stats, parts if keep else None, ctx

    def reread():             # the whole chromosome through the host decoder, from a handle of its own (the caller's may be closed)
        from .hicfile import HicFile
        with HicFile(hic.path) as h2:
            return read_hic_stream_to_device(h2, chrom, res, norm, dpx, chrom_size_bp, device, part=(0, 1), threads=threads,
                                             slab_records=slab_records, raw=False)
    pc.raw_ctx["reread"] = reread
    pc.read_s = time.time() - t0
    return pc


def _reader_threads():
    """The native reader's default worker count: the hardware threads, at most 128, and at most FOUR times the container's CPU
    quota when there is one.  (Measured on the 16-CPU-quota GPU box, 1.8 core-seconds of inflate per read: 32 threads 0.086 s
    mean over the steady passes, 48: 0.066, 64: 0.055 with 0.040 in the passes the quota does not interrupt, 96: 0.095 -- a
    read that burns a 100 ms period's quota in its first 20 ms waits out the rest of the period.)"""
    import os
    env = os.environ.get("MUSTACHE_HIC_THREADS")
    if env:
        return max(1, int(env))
    hw = min(os.cpu_count() or 1, 128)
    try:
        q, p = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if q != "max":
            hw = min(hw, max(1, int(4.0 * float(q) / float(p) + 0.5)))
    except (OSError, ValueError):
        pass
    return hw


def band_from_packed(pc, dpx, device, check=None):
    """hicfile.PackedContacts -> raw band [dpx+2, n] on `device`: three uploads of 4 bytes per record each (from page-locked
    memory when the records were read into it) and one scatter (mst_band_scatter_packed).  With one process per GPU
    (pc.n_parts > 1: this rank decoded only its share of the `.hic` blocks) the ranks exchange their records first
    (sharding.all_gather_packed) and every rank scatters all shares -- the band is the same on every rank and the same as in
    a 1-rank run, bit for bit (a matrix holds every pixel once, so the order of the scatters is immaterial).
    `check` (default: the environment variable MUSTACHE_CHECK_PACKED): read every record back from the band afterwards; a
    mismatch means two records with different values share a pixel -- a malformed file; the reference's scatter keeps the
    last one (mustache.py:921-924) -- and the band is rebuilt through band_from_host_coo, which applies that rule
    deterministically, with a warning."""
    import os
    lib = require_gpu()
    if check is None:
        check = bool(os.environ.get("MUSTACHE_CHECK_PACKED"))
    if getattr(pc, "device_band", None) is not None:
        return _band_from_raw(lib, pc, dpx, device, check)
    dev_parts = getattr(pc, "device_parts", None)
    if getattr(pc, "n_parts", 1) > 1:
        from .sharding import all_gather_packed
        if dev_parts is not None:             # streamed read: this rank's records are on the device already
            cat = lambda k, dt: (torch.cat([p[k][:p[3]] for p in dev_parts]) if dev_parts
                                 else torch.zeros(0, dtype=dt, device=device))
            own = (cat(0, torch.int32), cat(1, torch.uint16 if dpx + 1 <= 65535 else torch.int32), cat(2, torch.float32))
        else:
            own = (pc.x, pc.dist, pc.v)
        parts, n = all_gather_packed(own[0], own[1], own[2], pc.n, device)
        pc.n_all = n
    elif dev_parts is not None:
        n = int(pc.n)
        parts = dev_parts
    else:
        n = int(pc.n)
        if pc.pinned is not None:
            xd, dd, vd = (t.to(device, non_blocking=True) for t in pc.pinned)
        else:
            xd, dd, vd = (torch.from_numpy(a).to(device) for a in (pc.x, pc.dist, pc.v))
        parts = [(xd, dd, vd, int(pc.count))]
    band = torch.zeros((dpx + 2, n), dtype=torch.float64, device=device)
    with torch.cuda.device(device):
        for xd, dd, vd, cnt in parts:
            if cnt:
                _lib.check(lib.mst_band_scatter_packed(_ptr(xd), _ptr(dd), dd.element_size(), _ptr(vd), cnt, n, int(dpx),
                                                       _ptr(band), _stream()))
        if check:
            bad = torch.zeros(1, dtype=torch.int64, device=device)
            for xd, dd, vd, cnt in parts:
                if cnt:
                    _lib.check(lib.mst_band_verify_packed(_ptr(xd), _ptr(dd), dd.element_size(), _ptr(vd), cnt, n, int(dpx),
                                                          _ptr(band), _ptr(bad), _stream()))
            if int(bad.item()):
                xs = np.concatenate([p[0][:p[3]].cpu().numpy().astype(np.int64) for p in parts])
                ds = np.concatenate([p[1][:p[3]].cpu().numpy().astype(np.int64) for p in parts])
                vs = np.concatenate([p[2][:p[3]].cpu().numpy().astype(np.float64) for p in parts])
                del band
                return band_from_host_coo(xs, xs + ds, vs, n, dpx, device)
    if dev_parts is not None:
        # the slabs were allocated on the copy stream and have just been read by kernels queued on THIS stream: tell the
        # allocator, or a caller that drops `pc` while those kernels wait behind earlier work hands the blocks back to the copy
        # stream's pool and the next chromosome's slabs (the reader thread runs ahead) overwrite records not yet scattered
        cur = torch.cuda.current_stream(device)
        for part in dev_parts:
            for t in part[:3]:
                t.record_stream(cur)
    return band


def _band_from_raw(lib, pc, dpx, device, check):
    """band_from_packed for a raw streamed read (read_hic_stream_to_device, `.hic` v7-9): this rank's rows are in the band
    already (scattered by the kernel behind each slab's copy).  With several ranks the RAW slabs are exchanged
    (sharding.all_gather_raw: 6 bytes per record over xGMI) and every rank decodes the other ranks' rows into its own band --
    the same band on every rank, the 1-rank run's bit for bit.  Then the band is trimmed to n = max(binY) + 1 over the kept
    records (mustache.py:894).  `check`: every record is read back (two records with different values sharing a pixel = a
    malformed file); on a mismatch the chromosome is read again through the host decoder, whose loader applies the
    reference's last-entry-wins rule."""
    band, stats, ctx = pc.device_band, pc.band_stats, pc.raw_ctx
    if check and pc.raw_parts is None:                   # the slabs were not kept: the check runs on a second read
        return band_from_packed(ctx["reread"](), dpx, device, check=True)
    cur = torch.cuda.current_stream(device)
    # allocated and filled on the copy stream (synchronised before the read returned), used on this one from here on
    for t in [band, stats] + ([ctx["norm"]] if ctx["norm"] is not None else []) + [t for p in (pc.raw_parts or []) for t in p[:2]]:
        t.record_stream(cur)
    parts = pc.raw_parts or []
    with torch.cuda.device(device):
        if pc.n_parts > 1 and not getattr(pc, "_exchanged", False):
            from .sharding import all_gather_raw
            rank, others = all_gather_raw(parts, device)
            for r, plist in enumerate(others):
                if r != rank:
                    for pay, dr, rows in plist:
                        _scatter_rows(lib, pay, dr, rows, ctx, band, stats, _stream())
            pc._exchanged = True
            pc.raw_parts = parts = [p for plist in others for p in plist]
        if check:
            for pay, dr, rows in parts:
                _scatter_rows(lib, pay, dr, rows, ctx, band, stats, _stream(), verify=1)
        ymax1, kept, beyond, bad = (int(a) for a in stats.cpu().numpy())
    if beyond:
        # the same count on every rank (each has decoded all shares by now): every rank reads the whole chromosome again through
        # the host decoder -- no collective involved -- as the one-rank read does
        import warnings
        warnings.warn("%d record(s) lie beyond the chromosome length the file's header gives: reading the chromosome again through "
                      "the host decoder" % beyond)
        del band
        return band_from_packed(ctx["reread"](), dpx, device, check=check)
    if check and bad:
        import warnings
        warnings.warn("%d record(s) of the .hic file share a pixel with another value: reading the chromosome again through the host "
                      "decoder" % bad)
        del band
        again = ctx["reread"]()
        return band_from_packed(again, dpx, device, check=True)
    pc.n_all = pc.n = n = ymax1 if kept else 0
    pc.count_all = kept
    return band if n == band.shape[1] else _trim_band_in_place(band, n)


def _trim_band_in_place(band, n, temp_bytes=64 << 20):
    """band [rows, n_alloc] -> [rows, n] (n < n_alloc) in the SAME allocation: the rows are moved up group by group through a
    temporary of at most `temp_bytes` (row d goes from offset d * n_alloc to d * n, so a group's destination never reaches a row
    that has not been moved yet), and the result is a view of the first rows * n elements.  `band[:, :n].contiguous()` would
    hold a second band next to the first -- 4 GB + 4 GB for chr1 at 1 kb, for the handful of empty trailing bins almost every
    chromosome has."""
    rows = band.shape[0]
    if n == 0:
        return band.new_zeros((rows, 0))
    flat = band.view(-1)
    group = max(1, min(rows, temp_bytes // (8 * n)))
    for d0 in range(0, rows, group):
        g = min(group, rows - d0)
        tmp = band[d0:d0 + g, :n].clone(memory_format=torch.contiguous_format)
        flat[d0 * n:(d0 + g) * n].copy_(tmp.view(-1))
        del tmp
    return flat[:rows * n].view(rows, n)


def band_to_coo(band, x, y, v_out, n, dpx):
    lib = require_gpu()
    with torch.cuda.device(band.device):
        _lib.check(lib.mst_band_to_coo(_ptr(band), _ptr(x), _ptr(y), int(v_out.numel()), int(n), int(dpx),
                                       _ptr(v_out), _stream()))
    return v_out


_KERNELS = {"auto": 1, "blocked": 2, "segment": 3}


def normalize_band(band, n, dpx, resolution, blocked=False, kernel=None):
    """Returns (normalised band, diag_stats [dpx+2, 4] = mean, std, weight, count).  Branch selection and window
    size follow mustache.py:628, :631.  `kernel` ("blocked" / "segment"; `blocked=True` is short for the former) asks for one
    of the two cross-check formulations of branch A's window sums -- only a PROFILE build of the library carries them
    (make PROFILE=1, MUSTACHE_HIP_LIB=.../libmustache_hip_profile.so); the product library picks its kernel itself."""
    lib = require_gpu()
    local = (n - dpx) * resolution > 2000000
    window = int(2000000 / resolution)
    out = torch.empty_like(band)
    stats = torch.empty((dpx + 2, 4), dtype=torch.float64, device=band.device)
    with torch.cuda.device(band.device):
        _lib.check(lib.mst_normalize_band(_ptr(band), _ptr(out), int(n), int(dpx), window, _KERNELS[kernel or ("blocked" if blocked else "auto")] if local else 0,
                                          _ptr(stats), _stream()))
    return out, stats, local


def normalize_sparse_device(x, y, v, resolution, distance_in_px, blocked=False, kernel=None):
    """Host COO in, `v` overwritten in place, weights returned -- the reference's call shape."""
    require_gpu()
    xh = np.ascontiguousarray(np.asarray(x), dtype=np.int64)
    yh = np.ascontiguousarray(np.asarray(y), dtype=np.int64)
    n = int(max(xh.max(), yh.max())) + 1
    dev = torch.device("cuda:%d" % torch.cuda.current_device())
    xd, yd = torch.from_numpy(xh).to(dev), torch.from_numpy(yh).to(dev)
    vd = torch.from_numpy(np.ascontiguousarray(v, dtype=np.float64)).to(dev)
    band = band_from_coo(xd, yd, vd, n, distance_in_px)
    out, stats, local = normalize_band(band, n, distance_in_px, resolution, blocked=blocked, kernel=kernel)
    band_to_coo(out, xd, yd, vd, n, distance_in_px)
    v[...] = vd.cpu().numpy()
    st = stats.cpu().numpy()
    return [float(w) for w in st[:, 2]] if local else []
