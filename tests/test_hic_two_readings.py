"""The native `.hic` reader (libmustache_io.so) against a SECOND independent reading of the format (tests/hic_pyreader.py: pure
Python, written from the format description, sharing nothing with tests/hic_writer.py or the C++ reader).  hic-straw and real
files are absent offline, so block decoding is pinned on two independent readings agreeing:
  * everything tests/hic_writer.py can write (v8 / v9, short / float counts, row lists / dense grids, short / long coordinates)
    decodes identically through both readers;
  * hand-assembled files (a third, minimal container writer below, again independent) with blocks of kinds the writer never
    emits -- version 6 plain records, version 7, empty rows, rows out of order, non-zero offsets with relative number 0,
    relative numbers past 32767 stored in 16 bits (they wrap negative in BOTH readers, as they do in straw's `short`), dense
    grids with NaN / -32768 holes and a ragged last row, an empty block, v9 mixed coordinate widths -- decode identically too.
CPU only."""
import math
import os
import struct
import sys
import zlib

import numpy as np
import pytest

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from hic_pyreader import PyHic          # noqa: E402
from hic_writer import write_hic        # noqa: E402


def _filtered(records, max_dist):
    """What mst_hic_read_intra documents: (min, max) bin order, |dist| <= max_dist, value not NaN and > 0."""
    out = []
    for bx, by, c in records:
        if bx > by:
            bx, by = by, bx
        if max_dist >= 0 and by - bx > max_dist:
            continue
        if math.isnan(c) or not c > 0:
            continue
        out.append((bx, by, np.float32(c)))
    return sorted(out)


def _native(path, chrom, res, norm, max_dist):
    from mustache_amd.hicfile import HicFile
    with HicFile(path) as h:
        x, y, v = h.read_intra(chrom, res, norm, max_dist)
    return sorted(zip(x.tolist(), y.tolist(), v.astype(np.float32)))


def _native_raw(path, chrom, res, norm, max_dist, slab_bytes=4096, n_slabs=4):
    """the same records through the RAW stream (mst_hic_rawstream_*: the host only inflates and copies the rows) decoded by
    the NumPy restatement of the device kernel (tests/hic_rows_numpy.py) -- versions 7-9"""
    from hic_rows_numpy import decode_slab
    from mustache_amd.hicfile import HicFile, HicRawStream
    mem = np.zeros(n_slabs * slab_bytes + 16, np.uint8)
    base0 = (-mem.ctypes.data) % 16
    out = []
    with HicFile(path) as h:
        st = HicRawStream(h, chrom, res, norm, max_dist, mem.ctypes.data + base0, n_slabs, slab_bytes, threads=2)
        nv, _ = st.info()
        while True:
            r = st.next(50)
            if r is None:
                continue
            if r is False:
                break
            slab, nbytes, rows = r
            base = base0 + slab * slab_bytes
            x, y, v = decode_slab(mem[base:base + nbytes].copy(), mem[base + slab_bytes - 16 * rows:base + slab_bytes].copy(), nv,
                                  max_dist)
            out += list(zip(x.tolist(), y.tolist(), v))
            st.release(slab)
        st.close()
    return sorted(out)


def _same(a, b):
    assert len(a) == len(b), (len(a), len(b))
    for (ax, ay, av), (bx, by, bv) in zip(a, b):
        assert ax == bx and ay == by and np.float32(av) == np.float32(bv), ((ax, ay, av), (bx, by, bv))


@pytest.mark.parametrize("version,float_counts,dense,short_coords,bbc", [
    (8, False, False, True, 64), (8, True, False, True, 37), (8, False, True, True, 16), (8, True, True, True, 16),
    (9, False, False, True, 64), (9, True, False, False, 50), (9, True, False, True, 23)])
def test_everything_the_writer_emits_decodes_identically_in_both_readers(tmp_path, version, float_counts, dense, short_coords, bbc):
    rng = np.random.default_rng(version * 100 + bbc)
    n, res = 900, 5000
    x = rng.integers(0, n, 12000)
    y = np.minimum(x + rng.integers(0, 300, 12000), n - 1)
    key = np.unique(x * 100003 + y)
    x, y = key // 100003, key % 100003
    c = rng.integers(1, 900, len(x)).astype(np.float64) if not float_counts else \
        rng.uniform(0.25, 40, len(x)).astype(np.float32).astype(np.float64)
    norm = rng.uniform(0.4, 2.5, n + 1)
    norm[[5, 77]] = np.nan
    p = str(tmp_path / "w.hic")
    write_hic(p, [("All", 7500), ("chr1", n * res), ("chrX", 12345)], {1: {res: (x, y, c)}},
              {("KR", 1, res): norm, ("VC", 1, res): np.ones(n + 1)}, version=version, block_bin_count=bbc,
              float_counts=float_counts, dense_blocks=dense, short_coords=short_coords)
    py = PyHic(p)
    assert py.version == version and [nm for nm, _ in py.chromosomes] == ["All", "chr1", "chrX"] and py.resolutions == [res]
    for norm_name, max_dist in (("NONE", -1), ("KR", -1), ("KR", 120), ("VC", 40)):
        _same(_filtered(py.records("chr1", res, norm_name), max_dist), _native(p, "chr1", res, norm_name, max_dist))
        _same(_native_raw(p, "chr1", res, norm_name, max_dist), _native(p, "chr1", res, norm_name, max_dist))
    assert len(_native(p, "chr1", res, "KR", -1)) > 5000


# ---- a third, minimal container writer: just enough structure around hand-made block bodies -----------------------------------
def _z(text):
    return text.encode() + b"\x00"


def _container(path, version, length, res, per_block, columns, blocks, normvec):
    """blocks: {blockNumber: uncompressed body}.  One chromosome "c" (index 1; index 0 = "All"), one resolution, no expected
    values, one normalisation vector "KR"."""
    wide = version >= 9
    out = bytearray(_z("HIC") + struct.pack("<i", version) + struct.pack("<q", 0) + _z("handmade"))
    nvi_field = len(out)
    if wide:
        out += struct.pack("<qq", 0, 0)
    out += struct.pack("<i", 0)                                                      # no attributes
    out += struct.pack("<i", 2) + _z("All") + (struct.pack("<q", 1) if wide else struct.pack("<i", 1))
    out += _z("c") + (struct.pack("<q", length) if wide else struct.pack("<i", length))
    out += struct.pack("<ii", 1, res) + struct.pack("<i", 0)
    where = {}
    for number, body in blocks.items():
        comp = zlib.compress(body, 6)
        where[number] = (len(out), len(comp))
        out += comp
    matrix_at = len(out)
    out += struct.pack("<iii", 1, 1, 1) + _z("BP") + struct.pack("<i", 0) + struct.pack("<ffff", 1, 2, 3, 4)
    out += struct.pack("<iiii", res, per_block, columns, len(where))
    for number in sorted(where):
        out += struct.pack("<iqi", number, where[number][0], where[number][1])
    matrix_size = len(out) - matrix_at
    vec_at = len(out)
    out += (struct.pack("<q", len(normvec)) + np.asarray(normvec, "<f4").tobytes()) if wide else \
        (struct.pack("<i", len(normvec)) + np.asarray(normvec, "<f8").tobytes())
    vec_size = len(out) - vec_at
    footer_at = len(out)
    foot = bytearray(struct.pack("<i", 1) + _z("1_1") + struct.pack("<qi", matrix_at, matrix_size))
    foot += struct.pack("<i", 0) + struct.pack("<i", 0)                              # expected, normalised expected: none
    index = bytearray(struct.pack("<i", 1) + _z("KR") + struct.pack("<i", 1) + _z("BP") + struct.pack("<i", res))
    index += struct.pack("<q", vec_at) + (struct.pack("<q", vec_size) if wide else struct.pack("<i", vec_size))
    index_at = footer_at + (8 if wide else 4) + len(foot)
    foot += index
    out += (struct.pack("<q", len(foot)) if wide else struct.pack("<i", len(foot))) + foot
    struct.pack_into("<q", out, 8, footer_at)
    if wide:
        struct.pack_into("<qq", out, nvi_field, index_at, len(index))
    with open(path, "wb") as fh:
        fh.write(bytes(out))


def _rows(version, x0, y0, rows, short_counts, wide_x=False, wide_y=False, n_records=None):
    """rows: [(rowNumber, [(column, value), ...])] in the order given (not sorted, may be empty)."""
    total = sum(len(r[1]) for r in rows) if n_records is None else n_records
    out = bytearray(struct.pack("<iii", total, x0, y0) + struct.pack("<B", 0 if short_counts else 1))
    if version >= 9:
        out += struct.pack("<BB", 1 if wide_x else 0, 1 if wide_y else 0)
    out += struct.pack("<B", 1)
    fx, fy = ("<i" if wide_x else "<h"), ("<i" if wide_y else "<h")
    wrap = lambda v, f: struct.pack(f, v if f == "<i" else ((v + 32768) % 65536) - 32768)     # 16 bits keep the low 16 bits
    out += wrap(len(rows), fy)
    for row, cells in rows:
        out += wrap(row, fy) + wrap(len(cells), fx)
        for col, val in cells:
            out += wrap(col, fx) + (struct.pack("<h", int(val)) if short_counts else struct.pack("<f", val))
    return bytes(out)


def _grid(version, x0, y0, width, values, short_counts):
    out = bytearray(struct.pack("<iii", sum(1 for v in values if v is not None), x0, y0))
    out += struct.pack("<B", 0 if short_counts else 1)
    if version >= 9:
        out += struct.pack("<BB", 0, 0)
    out += struct.pack("<B", 2) + struct.pack("<ih", len(values), width)
    for v in values:
        if short_counts:
            out += struct.pack("<h", -32768 if v is None else int(v))
        else:
            out += struct.pack("<f", float("nan") if v is None else v)
    return bytes(out)


def _plain_v6(records):
    return struct.pack("<i", len(records)) + b"".join(struct.pack("<iif", x, y, v) for x, y, v in records)


@pytest.mark.parametrize("version", [6, 7, 8, 9])
def test_hand_assembled_corner_cases_decode_identically_in_both_readers(tmp_path, version):
    res, per_block, columns = 1000, 50000, 3
    nbins = 120000
    norm = np.linspace(0.5, 2.0, nbins)
    blocks = {}
    if version == 6:
        blocks[0] = _plain_v6([(3, 9, 1.5), (9, 3, 2.5), (7, 7, 0.0), (11, 40000, 3.25), (100, 90, -1.0), (5, 6, float("nan"))])
        blocks[4] = _plain_v6([])
    else:
        # row list: an empty row, rows out of order, relative number 0 at a non-zero offset, a lower-triangle record
        blocks[0] = _rows(version, 37, 41, [(5, [(0, 7), (3, 2)]), (2, []), (0, [(0, 1), (1, 0), (9, 5)]), (1, [(30, 4)])], True)
        # float counts with a NaN and a negative value
        blocks[1] = _rows(version, 1000, 1000, [(0, [(0, 0.5), (1, float("nan")), (2, -3.0)]), (7, [(7, 123.25)])], False)
        # relative numbers past 32767 in 16-bit fields: 40000 -> -25536 in both readers (as in straw's `short`)
        blocks[2] = _rows(version, 50000, 50000, [(40000, [(40000, 6)]), (10, [(32767, 8), (32768, 9)])], True)
        # dense grids: holes (-32768 / NaN), a ragged last row, width 1
        blocks[3] = _grid(version, 200, 300, 4, [1, None, 3, 4, None, None, 7, 8, 9, 10], True)
        blocks[4] = _grid(version, 60000, 60010, 3, [0.25, None, 2.5, None, 4.75], False)
        blocks[5] = _grid(version, 5, 5, 1, [2, None, 4], True)
        blocks[6] = _rows(version, 0, 0, [], True)                                     # a block without rows
        if version == 9:
            blocks[7] = _rows(version, 10, 20, [(70000, [(66000, 2.5), (5, 1.5)]), (3, [(90000, 4.0)])], False, wide_x=True,
                              wide_y=True)
            blocks[8] = _rows(version, 10, 20, [(9, [(70001, 3)])], True, wide_x=True, wide_y=False)
            blocks[9] = _rows(version, 10, 20, [(80000, [(2, 3)])], True, wide_x=False, wide_y=True)
    p = str(tmp_path / "h.hic")
    _container(p, version, nbins * res, res, per_block, columns, blocks, norm)
    py = PyHic(p)
    assert py.version == version and py.chromosomes[1] == ("c", nbins * res)
    raw = py.records("c", res, "NONE")
    if version == 6:
        assert len(raw) == 6
    else:
        assert (37, 46, 7.0) in raw and (37, 41, 1.0) in raw and (67, 42, 4.0) in raw            # offsets, relative 0, x > y
        assert (50000 - 25536, 50000 - 25536, 6.0) in raw and (50000 + 32767, 50010, 8.0) in raw and (50000 - 32768, 50010, 9.0) in raw
        assert (202, 301, 7.0) in raw and (201, 302, 10.0) in raw and (60002, 60010, 2.5) in raw and (60001, 60011, 4.75) in raw
    # (no distance limit on the native side: with one it also SKIPS blocks by their number's position in the block grid, and
    # these hand-numbered blocks do not sit where their numbers say -- block selection is covered by the writer test above)
    for norm_name in ("NONE", "KR"):
        _same(_filtered(py.records("c", res, norm_name), -1), _native(p, "c", res, norm_name, -1))
        if version >= 7:          # ... and through the raw rows (every corner case above reaches the device decoder's restatement)
            _same(_native_raw(p, "c", res, norm_name, -1), _native(p, "c", res, norm_name, -1))
    if version == 6:              # plain records have no rows: the raw stream refuses them, the caller uses the host decoder
        from mustache_amd.hicfile import HicError
        with pytest.raises(HicError, match="plain records"):
            _native_raw(p, "c", res, "NONE", -1)
    assert len(_native(p, "c", res, "NONE", -1)) >= (3 if version == 6 else 18)
    # the packed form agrees with the classic one on the same file
    from mustache_amd.hicfile import HicFile, read_intra_packed
    with HicFile(p) as h:
        pc = read_intra_packed(h, "c", res, "KR", -1)
    want = _native(p, "c", res, "KR", -1)
    got = sorted(zip(pc.x.tolist(), (pc.x.astype(np.int64) + pc.dist).tolist(), pc.v))
    _same(got, want)
