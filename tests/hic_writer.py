"""Test infrastructure: a minimal writer of Juicer `.hic` files (versions 8 and 9, intra-chromosomal matrices, BP
resolutions, optional normalisation vectors) -- the published layout that straw reads, written independently of
mustache_amd/csrc/hic_reader.cpp so the reader is exercised against something it did not produce itself.

Only what read_hic_file needs is written: header, one matrix per chromosome with one or more resolutions, blocks of
type 1 (row lists; short or float counts, short or int coordinates in v9) or type 2 (dense), expected-value sections
(present but empty, or with one dummy vector so the reader has to skip real bytes) and the normalisation-vector index.
"""
import math
import struct
import zlib

import numpy as np


def _s(text):
    return text.encode() + b"\0"


def _block_v8(recs, x_off, y_off, float_counts, dense):
    """recs: list of (x, y, count) with block-relative... absolute bins; returns the uncompressed block body."""
    out = [struct.pack("<iii", len(recs), x_off, y_off), struct.pack("<B", 1 if float_counts else 0)]
    if dense:
        w = max(r[0] - x_off for r in recs) + 1
        hgt = max(r[1] - y_off for r in recs) + 1
        grid = {(r[0] - x_off, r[1] - y_off): r[2] for r in recs}
        out.append(struct.pack("<B", 2))
        out.append(struct.pack("<ih", w * hgt, w))
        for i in range(w * hgt):
            row, col = divmod(i, w)
            c = grid.get((col, row))
            if float_counts:
                out.append(struct.pack("<f", float("nan") if c is None else c))
            else:
                out.append(struct.pack("<h", -32768 if c is None else int(c)))
        return b"".join(out)
    out.append(struct.pack("<B", 1))
    rows = {}
    for x, y, c in recs:
        rows.setdefault(y - y_off, []).append((x - x_off, c))
    out.append(struct.pack("<h", len(rows)))
    for y in sorted(rows):
        out.append(struct.pack("<hh", y, len(rows[y])))
        for x, c in sorted(rows[y]):
            out.append(struct.pack("<h", x) + (struct.pack("<f", c) if float_counts else struct.pack("<h", int(c))))
    return b"".join(out)


def _block_v9(recs, x_off, y_off, float_counts, short_x, short_y):
    out = [struct.pack("<iii", len(recs), x_off, y_off),
           struct.pack("<BBB", 1 if float_counts else 0, 0 if short_x else 1, 0 if short_y else 1), struct.pack("<B", 1)]
    fx, fy = ("<h" if short_x else "<i"), ("<h" if short_y else "<i")
    rows = {}
    for x, y, c in recs:
        rows.setdefault(y - y_off, []).append((x - x_off, c))
    out.append(struct.pack(fy, len(rows)))
    for y in sorted(rows):
        out.append(struct.pack(fy, y) + struct.pack(fx, len(rows[y])))
        for x, c in sorted(rows[y]):
            out.append(struct.pack(fx, x) + (struct.pack("<f", c) if float_counts else struct.pack("<h", int(c))))
    return b"".join(out)


def write_hic(path, chroms, matrices, norms=None, version=8, block_bin_count=64, float_counts=True, dense_blocks=False,
              short_coords=True, dummy_expected=True):
    """chroms: [(name, length)] (index 0 should be ("All", ...)).  matrices: {chrom_index: {resolution: (x, y, counts)}}
    with bin coordinates x <= y.  norms: {(type, chrom_index, resolution): vector}."""
    assert version in (8, 9)
    norms = norms or {}
    v9 = version == 9
    body = bytearray()
    body += _s("HIC") + struct.pack("<i", version) + struct.pack("<q", 0) + _s("synthetic")
    nvi_at = None
    if v9:
        nvi_at = len(body)
        body += struct.pack("<qq", 0, 0)
    body += struct.pack("<i", 1) + _s("software") + _s("tests/hic_writer.py")
    body += struct.pack("<i", len(chroms))
    for name, length in chroms:
        body += _s(name) + (struct.pack("<q", length) if v9 else struct.pack("<i", length))
    all_res = sorted({r for m in matrices.values() for r in m}, reverse=True)
    body += struct.pack("<i", len(all_res)) + b"".join(struct.pack("<i", r) for r in all_res)
    body += struct.pack("<i", 0)                                       # no fragment resolutions

    master = {}
    for ci, per_res in matrices.items():
        zooms = []
        for zi, res in enumerate(sorted(per_res, reverse=True)):
            x, y, cnt = (np.asarray(a) for a in per_res[res])
            nbins = chroms[ci][1] // res + 1
            bcc = nbins // block_bin_count + 1
            groups = {}
            for xi, yi, c in zip(x.tolist(), y.tolist(), cnt.tolist()):
                if v9:
                    depth = int(math.log2(1 + abs(xi - yi) / math.sqrt(2) / block_bin_count))
                    pad = (xi + yi) // 2 // block_bin_count
                    bn = depth * bcc + pad
                else:
                    bn = (yi // block_bin_count) * bcc + (xi // block_bin_count)
                groups.setdefault(bn, []).append((xi, yi, c))
            blocks = []
            for bn in sorted(groups):
                recs = groups[bn]
                x_off, y_off = min(r[0] for r in recs), min(r[1] for r in recs)
                raw = (_block_v9(recs, x_off, y_off, float_counts, short_coords, short_coords) if v9
                       else _block_v8(recs, x_off, y_off, float_counts, dense_blocks))
                comp = zlib.compress(raw)
                blocks.append((bn, len(body), len(comp)))
                body += comp
            zooms.append((res, zi, bcc, blocks))
        pos = len(body)
        body += struct.pack("<iii", ci, ci, len(zooms))
        for res, zi, bcc, blocks in zooms:
            body += _s("BP") + struct.pack("<i", zi) + struct.pack("<ffff", 0, 0, 0, 0)
            body += struct.pack("<iiii", res, block_bin_count, bcc, len(blocks))
            for bn, bpos, bsize in blocks:
                body += struct.pack("<iqi", bn, bpos, bsize)
        master["%d_%d" % (ci, ci)] = (pos, len(body) - pos)

    norm_pos = {}
    for key, vec in norms.items():
        vec = np.asarray(vec, dtype=np.float64)
        p = len(body)
        if v9:
            body += struct.pack("<q", len(vec)) + vec.astype("<f4").tobytes()
        else:
            body += struct.pack("<i", len(vec)) + vec.astype("<f8").tobytes()
        norm_pos[key] = (p, len(body) - p)

    master_at = len(body)
    foot = bytearray()
    foot += struct.pack("<i", len(master))
    for k, (p, sz) in master.items():
        foot += _s(k) + struct.pack("<qi", p, sz)
    # expected values (one dummy vector so a reader has to skip real bytes), then normalised expected values
    if dummy_expected:
        vals = np.arange(5, dtype=np.float64)
        foot += struct.pack("<i", 1) + _s("BP") + struct.pack("<i", all_res[0])
        foot += (struct.pack("<q", 5) + vals.astype("<f4").tobytes()) if v9 else (struct.pack("<i", 5) + vals.tobytes())
        foot += struct.pack("<i", 1) + struct.pack("<i", 1) + (struct.pack("<f", 1.0) if v9 else struct.pack("<d", 1.0))
        foot += struct.pack("<i", 1) + _s("KR") + _s("BP") + struct.pack("<i", all_res[0])
        foot += (struct.pack("<q", 5) + vals.astype("<f4").tobytes()) if v9 else (struct.pack("<i", 5) + vals.tobytes())
        foot += struct.pack("<i", 0)
    else:
        foot += struct.pack("<i", 0) + struct.pack("<i", 0)
    nbytes_field = 8 if v9 else 4
    nvi_pos = master_at + nbytes_field + len(foot)
    nvi = bytearray(struct.pack("<i", len(norm_pos)))
    for (typ, ci, res), (p, sz) in norm_pos.items():
        nvi += _s(typ) + struct.pack("<i", ci) + _s("BP") + struct.pack("<i", res) + struct.pack("<q", p)
        nvi += struct.pack("<q", sz) if v9 else struct.pack("<i", sz)
    foot += nvi
    body += (struct.pack("<q", len(foot)) if v9 else struct.pack("<i", len(foot))) + foot
    struct.pack_into("<q", body, 8, master_at)
    if v9:
        struct.pack_into("<qq", body, nvi_at, nvi_pos, len(nvi))
    with open(path, "wb") as fh:
        fh.write(bytes(body))


# ---- bulk writer (bench.py's end-to-end-from-a-file leg): version 8, one chromosome, one resolution, float counts, row-list
# blocks assembled with NumPy and deflated on a thread pool -- tens of millions of records in seconds instead of the
# per-record Python loops above.  Same layout, so the same reader paths are exercised.
def block_v8_rows_float(x, y, c, x_off, y_off):
    """Uncompressed body of a type-1 block with float counts; x, y absolute bins sorted by (y, x), all inside the block."""
    n = len(x)
    xr = (np.asarray(x) - x_off).astype("<i2")
    yr = (np.asarray(y) - y_off).astype("<i2")
    newrow = np.ones(n, bool)
    newrow[1:] = yr[1:] != yr[:-1]
    first = np.flatnonzero(newrow)
    R = len(first)
    row_id = np.cumsum(newrow) - 1
    counts = np.diff(np.append(first, n))
    words = np.empty(2 * R + 3 * n, "<i2")
    hdr = 2 * np.arange(R) + 3 * first
    words[hdr] = yr[first]
    words[hdr + 1] = counts.astype("<i2")
    pos = 2 * (row_id + 1) + 3 * np.arange(n)
    fb = np.asarray(c).astype("<f4").view("<i2").reshape(n, 2)
    words[pos] = xr
    words[pos + 1] = fb[:, 0]
    words[pos + 2] = fb[:, 1]
    return struct.pack("<iii", n, int(x_off), int(y_off)) + struct.pack("<BB", 1, 1) + struct.pack("<h", R) + words.tobytes()


def write_hic_bulk(path, chrom, length, res, blocks, block_bin_count, norm_ones=True, level=1, threads=16):
    """blocks: iterable of (block_x, block_y, x, y, c) with absolute bins sorted by (y, x) inside each block (x <= y).
    Writes chromosome index 1 (index 0 = "All") with a KR vector of ones when norm_ones.  Returns the record count."""
    from concurrent.futures import ThreadPoolExecutor
    nbins = length // res + 1
    bcc = nbins // block_bin_count + 1
    total = 0
    with open(path, "wb") as fh:
        head = bytearray()
        head += _s("HIC") + struct.pack("<i", 8) + struct.pack("<q", 0) + _s("synthetic")
        head += struct.pack("<i", 1) + _s("software") + _s("tests/hic_writer.py bulk")
        head += struct.pack("<i", 2) + _s("All") + struct.pack("<i", 1) + _s(chrom) + struct.pack("<i", int(length))
        head += struct.pack("<i", 1) + struct.pack("<i", int(res)) + struct.pack("<i", 0)
        fh.write(head)
        pos = len(head)
        index = []

        def pack(b):
            bx, by, x, y, c = b
            raw = block_v8_rows_float(x, y, c, int(np.min(x)), int(np.min(y)))
            return by * bcc + bx, len(x), zlib.compress(raw, level)

        with ThreadPoolExecutor(max_workers=threads) as pool:
            for bn, cnt, comp in pool.map(pack, blocks):
                fh.write(comp)
                index.append((bn, pos, len(comp)))
                pos += len(comp)
                total += cnt
        mat_at = pos
        mat = bytearray(struct.pack("<iii", 1, 1, 1))
        mat += _s("BP") + struct.pack("<i", 0) + struct.pack("<ffff", 0, 0, 0, 0)
        mat += struct.pack("<iiii", int(res), int(block_bin_count), int(bcc), len(index))
        for bn, bpos, bsize in sorted(index):
            mat += struct.pack("<iqi", bn, bpos, bsize)
        fh.write(mat)
        pos += len(mat)
        norm_at = pos
        if norm_ones:
            vec = np.ones(nbins + 1, "<f8")
            nv = struct.pack("<i", len(vec)) + vec.tobytes()
            fh.write(nv)
            pos += len(nv)
        master_at = pos
        foot = bytearray(struct.pack("<i", 1) + _s("1_1") + struct.pack("<qi", mat_at, len(mat)))
        foot += struct.pack("<i", 0) + struct.pack("<i", 0)
        if norm_ones:
            foot += struct.pack("<i", 1) + _s("KR") + struct.pack("<i", 1) + _s("BP") + struct.pack("<i", int(res))
            foot += struct.pack("<q", norm_at) + struct.pack("<i", pos - norm_at)
        else:
            foot += struct.pack("<i", 0)
        fh.write(struct.pack("<i", len(foot)) + foot)
        fh.seek(8)
        fh.write(struct.pack("<q", master_at))
    return total


def write_synthetic_hic(path, n, dpx, res, depth, nloops, seed, keep, device, block_bins=1000):
    """A config-4-shaped `.hic` file (one chromosome at `res`, version 8, float counts, KR vector of ones) holding the
    synthetic chromosome with pixel (i, i + d) kept with probability min(1, keep / (d + 1)) -- generated slab by slab on the
    GPU, written with write_hic_bulk (bench.py's file leg; test infrastructure, untimed).  Returns the record count."""
    import os
    import torch
    from mustache_amd.synth import band_counts, _uniform

    def blocks():
        d = torch.arange(dpx + 2, dtype=torch.int64, device=device)[:, None]
        for bx in range(-(-n // block_bins)):
            i0, i1 = bx * block_bins, min(n, (bx + 1) * block_bins)
            i = torch.arange(i0, i1, dtype=torch.int64, device=device)[None, :]
            val = band_counts(n, dpx, depth, nloops, seed, i0=i0, i1=i1, device=device)
            # thinning grows with the distance (dense near the diagonal, sparse far out, like a real map at 1 kb)
            val = torch.where(_uniform(seed + 5, d, i, 9) * (d + 1).to(torch.float64) < keep, val, torch.zeros_like(val))
            dd, cc = torch.nonzero(val > 0, as_tuple=True)
            x = cc + i0
            y = x + dd
            v = val[dd, cc].to(torch.float32)
            by = y // block_bins
            order = torch.argsort((by << 42) | (y << 21) | x)
            x, y, v, by = x[order].to(torch.int32).cpu().numpy(), y[order].to(torch.int32).cpu().numpy(), \
                v[order].cpu().numpy(), by[order].cpu().numpy()
            cuts = np.flatnonzero(np.r_[True, by[1:] != by[:-1]]) if len(by) else np.zeros(0, np.int64)
            cuts = np.append(cuts, len(by))
            for a, b in zip(cuts[:-1], cuts[1:]):
                yield bx, int(by[a]), x[a:b], y[a:b], v[a:b]

    return write_hic_bulk(path, "chr1", n * res, res, blocks(), block_bins, threads=min(32, os.cpu_count() or 4))
