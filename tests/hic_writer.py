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
