"""Native `.hic` reader (libmustache_io.so, include/mustache_io.h) against files written by tests/hic_writer.py -- an
independent writer of the published layout (no `.hic` file and no hic-straw exist offline, so parity with hic-straw on a
real file is unpinned; the hic-straw backend of mustache_amd.readers stays available for that cross-check)."""
import ctypes
import os
import re
import sys
import types

import numpy as np
import pytest

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from hic_writer import write_hic     # noqa: E402

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _contacts(n, spread, m, seed, integer=True):
    rng = np.random.default_rng(seed)
    x = rng.integers(0, n, m)
    y = np.minimum(x + rng.integers(0, spread, m), n - 1)
    key = np.unique(x * 1000003 + y)
    x, y = key // 1000003, key % 1000003
    c = rng.integers(1, 900, len(x)).astype(np.float64) if integer else rng.uniform(0.25, 40, len(x)).astype(np.float32).astype(np.float64)
    return x, y, c


def _sorted(x, y, v):
    o = np.lexsort((y, x))
    return x[o], y[o], v[o]


def test_io_library_exports_every_declared_symbol():
    lib = ctypes.CDLL(os.path.join(ROOT, "mustache_amd", "libmustache_io.so"))
    header = open(os.path.join(ROOT, "include", "mustache_io.h")).read()
    names = set(re.findall(r"\b(mst_(?:io|hic|text|host)_\w+)\s*\(", header))
    assert len(names) == 29
    for n in names:
        assert hasattr(lib, n), n
    assert lib.mst_io_abi_version() == 3


@pytest.mark.parametrize("version,float_counts,dense,short_coords,bbc", [
    (8, False, False, True, 64), (8, True, False, True, 37), (8, False, True, True, 16), (8, True, True, True, 16),
    (9, False, False, True, 64), (9, True, False, False, 50), (9, True, False, True, 23)])
def test_round_trip_all_block_encodings(tmp_path, version, float_counts, dense, short_coords, bbc):
    from mustache_amd.hicfile import HicFile
    n, res = 1500, 5000
    x, y, c = _contacts(n, 400, 30000, version * 10 + bbc, integer=not float_counts)
    rng = np.random.default_rng(3)
    norm = rng.uniform(0.4, 2.5, n + 1)
    norm[[5, 77]] = np.nan                               # bins without a normalisation factor -> records dropped
    p = str(tmp_path / "a.hic")
    write_hic(p, [("All", 7500), ("chr1", n * res), ("chrX", 12345)],
              {1: {res: (x, y, c), 25000: (x[:50] // 5, np.maximum(y[:50] // 5, x[:50] // 5), c[:50])}},
              {("KR", 1, res): norm, ("VC", 1, res): np.ones(n + 1)}, version=version, block_bin_count=bbc,
              float_counts=float_counts, dense_blocks=dense, short_coords=short_coords)
    with HicFile(p) as h:
        assert h.version == version
        assert h.chromosomes() == [("All", 7500), ("chr1", n * res), ("chrX", 12345)]
        assert h.resolutions() == [25000, 5000]
        nv = norm.astype(np.float32).astype(np.float64) if version == 9 else norm      # v9 stores float32 vectors
        for name in ("chr1", "1"):                       # with or without the prefix, as users type it
            for norm_name, want in (("NONE", c.astype(np.float32)), ("VC", c.astype(np.float32)),
                                    ("KR", (c.astype(np.float32).astype(np.float64) / (nv[x] * nv[y])).astype(np.float32))):
                for md in (-1, 0, 3, 150, 399, 10000):
                    ok = ~np.isnan(want) & (want > 0) & ((y - x <= md) if md >= 0 else True)
                    gx, gy, gv = _sorted(*h.read_intra(name, res, norm_name, md, threads=3))
                    ex, ey, ev = _sorted(x[ok], y[ok], want[ok].astype(np.float64))
                    assert np.array_equal(gx, ex) and np.array_equal(gy, ey) and np.array_equal(gv, ev), (norm_name, md)


def test_error_codes(tmp_path):
    from mustache_amd.hicfile import HicFile, HicError
    p = str(tmp_path / "a.hic")
    x, y, c = _contacts(300, 50, 2000, 1)
    write_hic(p, [("All", 1), ("chr2", 300 * 1000)], {1: {1000: (x, y, c)}}, {("KR", 1, 1000): np.ones(301)})
    with HicFile(p) as h:
        for args, code in ((("chr9", 1000, "KR"), -4), (("chr2", 5000, "KR"), -4), (("chr2", 1000, "SCALE"), -4)):
            with pytest.raises(HicError) as e:
                h.read_intra(*args)
            assert e.value.code == code
    bad = str(tmp_path / "bad.hic")
    open(bad, "wb").write(b"HDF\0" + bytes(100))
    with pytest.raises(HicError) as e:
        HicFile(bad)
    assert e.value.code == -3
    raw = open(p, "rb").read()
    open(bad, "wb").write(raw[:len(raw) // 2])           # truncated: the master index lies outside the file
    with pytest.raises(HicError) as e:
        HicFile(bad)
    assert e.value.code == -3
    with pytest.raises(HicError) as e:
        HicFile(str(tmp_path / "missing.hic"))
    assert e.value.code == -2
    # a corrupted zlib stream is reported, not crashed on
    blk = bytearray(raw)
    start = raw.index(b"\x78\x9c")                       # first zlib header
    blk[start + 8:start + 24] = bytes(16)
    open(bad, "wb").write(bytes(blk))
    with HicFile(bad) as h:
        with pytest.raises(HicError) as e:
            h.read_intra("chr2", 1000, "NONE")
        assert e.value.code in (-5, -3)


def test_read_hic_file_native_equals_straw_backend(tmp_path, monkeypatch):
    """mustache_amd.readers.read_hic_file through the native reader == through the hic-straw code path (served by an
    in-memory stand-in that answers straw() window queries from the same contacts): same records, each once."""
    from mustache_amd.readers import read_hic_file, list_chromosomes
    n, res, dist = 5200, 5000, 2_000_000
    x, y, c = _contacts(n, 460, 60000, 9)
    norm = np.random.default_rng(4).uniform(0.5, 2.0, n + 1)
    norm[100:110] = np.nan
    p = str(tmp_path / "a.hic")
    write_hic(p, [("All", 1), ("chr7", n * res), ("chr8", 999)], {1: {res: (x, y, c)}}, {("KR", 1, res): norm},
              version=8, block_bin_count=128)
    val = (c.astype(np.float32).astype(np.float64) / (norm[x] * norm[y])).astype(np.float32).astype(np.float64)

    def straw(kind, nm, f, loc1, loc2, unit, r):
        _, s, e = loc1.split(":")
        s, e = int(s), int(e)
        sel = (x * res >= s) & (x * res <= e) & (y * res >= s) & (y * res <= e)
        return [types.SimpleNamespace(binX=int(a * res), binY=int(b * res), counts=float(v))
                for a, b, v in zip(x[sel], y[sel], val[sel])]

    fake = types.ModuleType("hicstraw")
    fake.straw = straw
    fake.HiCFile = lambda f: types.SimpleNamespace(getChromosomes=lambda: [
        types.SimpleNamespace(name="All", length=1), types.SimpleNamespace(name="chr7", length=n * res),
        types.SimpleNamespace(name="chr8", length=999)])
    monkeypatch.setitem(sys.modules, "hicstraw", fake)
    out = {}
    for backend in ("native", "hicstraw"):
        monkeypatch.setenv("MUSTACHE_HIC_BACKEND", backend)
        assert list_chromosomes(p, res) == ["chr7", "chr8"]
        out[backend] = _sorted(*read_hic_file(p, False, False, dist, "chr7", "chr7", res))
    for a, b in zip(out["native"], out["hicstraw"]):
        assert np.array_equal(a, b)
    assert len(out["native"][0]) > 20000
    monkeypatch.setenv("MUSTACHE_HIC_BACKEND", "native")
    with pytest.raises(NameError):
        read_hic_file(p, False, False, dist, "chr9", "chr9", res)


def test_integration_md_hic_snippet_runs(tmp_path):
    """INTEGRATION.md section 3's ctypes example, executed as written against a file made by tests/hic_writer.py."""
    text = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    sec = text[text.index("## 3."):]
    code = re.search(r"```python\n(.*?)```", sec, re.S).group(1)
    n, res = 3000, 1000
    x, y, c = _contacts(n, 2500, 40000, 5)
    p = str(tmp_path / "sample.hic")
    write_hic(p, [("All", 1), ("chr1", n * res)], {1: {res: (x, y, c)}}, {("KR", 1, res): np.full(n + 1, 2.0)})
    code = code.replace('ctypes.CDLL("mustache_amd/libmustache_io.so")',
                        'ctypes.CDLL(os.path.join(ROOT, "mustache_amd", "libmustache_io.so"))').replace('b"sample.hic"', "PATH")
    ns = dict(os=os, ROOT=ROOT, PATH=os.fsencode(p))
    exec(compile(code, "INTEGRATION.md", "exec"), ns)
    keep = (y - x) <= 2000
    assert ns["n"] == int(keep.sum()) and sorted(ns["x"].tolist()) == sorted(x[keep].tolist())


def test_corrupted_files_never_crash(tmp_path):
    """Random byte / word flips and truncations of a valid file: the reader either returns records or a HicError -- it must
    not crash or hang (every structure is read through a bounds-checked cursor)."""
    import random
    from mustache_amd.hicfile import HicFile, HicError
    n, res = 700, 5000
    x, y, c = _contacts(n, 300, 6000, 12)
    good = str(tmp_path / "g.hic")
    write_hic(good, [("All", 1), ("chr1", n * res)], {1: {res: (x, y, c)}}, {("KR", 1, res): np.full(n + 1, 1.5)}, version=9,
              block_bin_count=64)
    raw = open(good, "rb").read()
    rnd = random.Random(4)
    outcomes = {"ok": 0, "err": 0}
    for i in range(120):
        b = bytearray(raw)
        for _ in range(rnd.randint(1, 5)):
            pos = rnd.randrange(len(b) - 4)
            if rnd.random() < 0.5:
                b[pos] = rnd.randrange(256)
            else:
                b[pos:pos + 4] = rnd.randrange(2 ** 32).to_bytes(4, "little")
        if rnd.random() < 0.15:
            b = b[:rnd.randrange(8, len(b))]
        p = str(tmp_path / "m.hic")
        open(p, "wb").write(bytes(b))
        try:
            with HicFile(p) as h:
                h.chromosomes()
                h.read_intra("chr1", res, "KR", 100, threads=2)
            outcomes["ok"] += 1
        except HicError:
            outcomes["err"] += 1
    assert outcomes["ok"] + outcomes["err"] == 120 and outcomes["err"] > 20


def test_header_parse_equals_reference_held_parser(tmp_path, golden_dir):
    """tests/golden/hic_header_v8.npz is what the reference's own header parser (diff_mustache.py:182-249 read_header, imported
    by make_golden.py) returns for the version-8 file that `hic_header_case` writes: version, master-index offset, genome id,
    the chromosome table and the resolutions.  mst_hic_open must agree field for field.  (Block decoding has no such pin:
    the reference reads blocks through hic-straw only, include/mustache_io.h.)"""
    import hashlib
    sys.path.insert(0, golden_dir)
    try:
        from make_golden import hic_header_case
    finally:
        sys.path.remove(golden_dir)
    from mustache_amd.hicfile import HicFile
    g = np.load(os.path.join(golden_dir, "hic_header_v8.npz"))
    path = str(tmp_path / "hdr.hic")
    hic_header_case(path)
    blob = open(path, "rb").read()
    assert len(blob) == int(g["file_bytes"]) and hashlib.sha256(blob).hexdigest() == str(g["sha256"]), "writer drifted"
    with HicFile(path) as h:
        assert h.version == int(g["version"]) == 8
        assert h.master_offset == int(g["masterindex"])
        assert h.genome == str(g["genome"])
        chroms = h.chromosomes()
        # read_header keeps only entries with a name and a non-zero length; every entry of this file has both
        assert [c[0] for c in chroms] == [str(s) for s in g["chr_name"]]
        assert [c[1] for c in chroms] == [int(v) for v in g["chr_length"]]
        assert list(range(len(chroms))) == [int(i) for i in g["chr_index"]]
        assert h.resolutions() == [int(r) for r in g["resolutions"]]


@pytest.mark.parametrize("version,n_parts", [(8, 2), (8, 3), (9, 2), (8, 8)])
def test_part_decodes_partition_the_chromosome(tmp_path, version, n_parts):
    """One process per GPU on one chromosome: rank p of n decodes share p of the near-diagonal blocks
    (mst_hic_decode_intra_packed_part).  The shares are disjoint, every block is decoded by exactly one of them, and their
    union is the record set of the whole read -- the file is read once between the ranks."""
    from mustache_amd.hicfile import HicFile, read_intra_packed
    n, res = 3000, 5000
    x, y, c = _contacts(n, 300, 60000, 11 * version + n_parts, integer=False)
    p = str(tmp_path / "p.hic")
    write_hic(p, [("All", 7500), ("chr1", n * res)], {1: {res: (x, y, c)}}, {("KR", 1, res): np.ones(n + 1)},
              version=version, block_bin_count=100)
    with HicFile(p) as h:
        whole = read_intra_packed(h, "chr1", res, "KR", 250)
        key_w = np.sort(whole.x.astype(np.int64) * (1 << 32) + whole.dist)
        assert whole.blocks_total == whole.blocks_mine > n_parts and whole.n_parts == 1
        keys, blocks, top = [], 0, 0
        for part in range(n_parts):
            pc = read_intra_packed(h, "chr1", res, "KR", 250, part=(part, n_parts))
            assert pc.blocks_total == whole.blocks_total and (pc.part, pc.n_parts) == (part, n_parts)
            blocks += pc.blocks_mine
            top = max(top, pc.n)
            k = pc.x.astype(np.int64) * (1 << 32) + pc.dist
            keys.append(k)
            # values travel with their pixel
            order = np.argsort(k)
            pos = np.searchsorted(key_w, k[order])
            vw = whole.v[np.argsort(whole.x.astype(np.int64) * (1 << 32) + whole.dist)]
            assert np.array_equal(vw[pos], pc.v[order])
        assert blocks == whole.blocks_total and top == whole.n
        allk = np.concatenate(keys)
        assert len(allk) == len(key_w) and np.array_equal(np.sort(allk), key_w)      # disjoint and complete
        assert min(len(k) for k in keys) > 0.5 * len(key_w) / n_parts                # shares balanced by compressed bytes


def test_own_inflate_equals_zlib_on_arbitrary_streams():
    """csrc/mst_inflate.h (the block reader's zlib-stream decoder) against zlib itself: stored / fixed / dynamic blocks at every
    compression level on random, repetitive, low-entropy and row-list-like data; a flipped bit or a truncation is either
    rejected or -- when zlib accepts the stream too -- decodes to zlib's bytes (the Adler-32 is verified)."""
    import zlib
    from mustache_amd import hicfile
    lib = hicfile.load()

    def inf(comp, cap):
        out = (ctypes.c_uint8 * max(cap, 1))()
        n = lib.mst_io_inflate(comp, len(comp), out, cap)
        return n, bytes(out[:max(n, 0)])

    rng = np.random.default_rng(0)
    rows = np.cumsum(rng.integers(0, 3, 300000)).astype("<i2").tobytes()
    for case in range(300):
        kind = case % 5
        m = int(rng.integers(0, 150000))
        if kind == 0:
            data = rng.integers(0, 256, m, dtype=np.uint8).tobytes()
        elif kind == 1:
            data = bytes(rng.integers(0, 4, m, dtype=np.uint8))
        elif kind == 2:
            data = (b"abcdefgh" * (m // 8 + 1))[:m]
        elif kind == 3:
            data = rows[:m]
        else:
            data = (rng.integers(0, 256, 40, dtype=np.uint8).tobytes() * (m // 40 + 1))[:m]      # long matches, distance 40
        comp = zlib.compress(data, int(rng.integers(0, 10)))
        n, out = inf(comp, len(data) + 300)
        assert n == len(data) and out == data, (case, kind, m, n)
        if len(comp) > 8:
            bad = bytearray(comp)
            bad[int(rng.integers(0, len(bad)))] ^= 1 << int(rng.integers(0, 8))
            n2, out2 = inf(bytes(bad), len(data) + 300)
            try:
                z = zlib.decompress(bytes(bad))
            except zlib.error:
                z = None
            assert n2 < 0 if z is None else (n2 >= 0 and out2 == z), (case, "flipped bit")
            n3, _ = inf(comp[:int(rng.integers(0, len(comp)))], len(data) + 300)
            assert n3 < 0, (case, "truncated stream accepted")
    assert inf(zlib.compress(b"x" * 1000), 100)[0] == -1            # MST_IO_E_ARG: the output does not fit


@pytest.mark.parametrize("version,float_counts,dense,short_coords,n_parts,dist_bytes", [
    (8, True, False, True, 1, 2), (8, False, False, True, 2, 2), (9, True, False, False, 3, 4), (8, True, True, True, 1, 2),
    (9, False, False, True, 1, 4)])
def test_streamed_read_equals_packed_read(tmp_path, version, float_counts, dense, short_coords, n_parts, dist_bytes):
    """mst_hic_stream_*: the slabs delivered by the streaming read hold exactly the records of the one-shot packed read
    (which inflates through zlib here: MUSTACHE_HIC_ZLIB is honoured per process, so the comparison partner is produced by a
    subprocess), for row-list and dense blocks, short and long coordinates, both distance widths, several parts."""
    import subprocess
    from mustache_amd.hicfile import HicFile, HicStream, read_intra_packed
    n, res, dpx = 2500, 5000, 260
    x, y, c = _contacts(n, 320, 50000, 7 * version + n_parts, integer=not float_counts)
    norm = np.random.default_rng(5).uniform(0.5, 2.0, n + 1)
    norm[[9, 300]] = np.nan
    p = str(tmp_path / "s.hic")
    write_hic(p, [("All", 7500), ("chr1", n * res)], {1: {res: (x, y, c)}}, {("KR", 1, res): norm}, version=version,
              block_bin_count=64, float_counts=float_counts, dense_blocks=dense, short_coords=short_coords)
    with HicFile(p) as h:
        whole = read_intra_packed(h, "chr1", res, "KR", dpx, (n - 3) * res)
        key_w = whole.x.astype(np.int64) * (1 << 20) + whole.dist
        order_w = np.argsort(key_w)
        # slabs larger than any block, and slabs much smaller than a block (a slab is handed over when it is full, in the
        # middle of a block if need be): the same record set either way
        for cap, n_slabs in ((6000, 5), (258, 9)):
            mem = np.zeros(n_slabs * cap * (8 + dist_bytes), np.uint8)
            got_k, got_v, blocks, top = [], [], 0, 0
            for part in range(n_parts):
                st = HicStream(h, "chr1", res, "KR", dpx, (n - 3) * res, mem.ctypes.data, n_slabs, cap, dist_bytes, threads=3,
                               part=(part, n_parts))
                while True:
                    r = st.next(50)
                    if r is None:
                        continue
                    if r is False:
                        break
                    slab, cnt = r
                    assert 0 < cnt <= cap
                    base = slab * cap * (8 + dist_bytes)
                    sx = mem[base:base + 4 * cnt].view(np.int32).astype(np.int64)
                    sv = mem[base + 4 * cap:base + 4 * cap + 4 * cnt].view(np.float32).copy()
                    sd = mem[base + 8 * cap:base + 8 * cap + dist_bytes * cnt].view(np.uint16 if dist_bytes == 2 else np.int32)
                    got_k.append(sx * (1 << 20) + sd.astype(np.int64))
                    got_v.append(sv)
                    st.release(slab)
                st.close()
                blocks += st.blocks_mine
                top = max(top, st.n)
                assert st.blocks_total == whole.blocks_total
            k, v = np.concatenate(got_k), np.concatenate(got_v)
            o = np.argsort(k)
            assert blocks == whole.blocks_total and top == whole.n and len(k) == len(key_w) > 10000
            assert np.array_equal(k[o], key_w[order_w]) and np.array_equal(v[o], whole.v[order_w])
    # the same one-shot read through zlib (cross-check of the own inflate on real block payloads)
    code = ("import sys, numpy as np; sys.path.insert(0, %r)\n"
            "from mustache_amd.hicfile import HicFile, read_intra_packed\n"
            "h = HicFile(%r); pc = read_intra_packed(h, 'chr1', %d, 'KR', %d, %d)\n"
            "print(len(pc), int(pc.x.astype(np.int64).sum()), int(pc.dist.astype(np.int64).sum()), repr(float(pc.v.astype(np.float64).sum())))\n"
            % (ROOT, p, res, dpx, (n - 3) * res))
    r = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, MUSTACHE_HIC_ZLIB="1"), capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-2000:]
    want = "%d %d %d %r" % (len(whole), int(whole.x.astype(np.int64).sum()), int(whole.dist.astype(np.int64).sum()),
                            float(whole.v.astype(np.float64).sum()))
    assert r.stdout.strip() == want


@pytest.mark.parametrize("version,float_counts,dense,short_coords,n_parts", [
    (8, True, False, True, 1), (8, False, False, True, 2), (9, True, False, False, 3), (8, True, True, True, 1),
    (9, False, False, True, 1), (8, False, True, True, 2), (9, False, True, False, 1)])
def test_raw_streamed_rows_decode_to_the_packed_read(tmp_path, version, float_counts, dense, short_coords, n_parts):
    """mst_hic_rawstream_*: the host only inflates and copies each row's record bytes + a 16-byte directory entry; the rows,
    decoded by the NumPy restatement of the device kernel (tests/hic_rows_numpy.py: columns, counts, the division by the norm
    vector, every filter), are exactly the records of the host decoder's packed read -- row-list and dense blocks, short and long
    coordinates, short and float counts, slabs larger than a block and slabs of a few rows, several parts."""
    from hic_rows_numpy import decode_slab
    from mustache_amd.hicfile import HicFile, HicRawStream, read_intra_packed
    n, res, dpx = 2500, 5000, 260
    x, y, c = _contacts(n, 320, 50000, 11 * version + n_parts, integer=not float_counts)
    norm = np.random.default_rng(5).uniform(0.5, 2.0, n + 1)
    norm[[9, 300]] = np.nan
    p = str(tmp_path / "s.hic")
    write_hic(p, [("All", 7500), ("chr1", n * res)], {1: {res: (x, y, c)}}, {("KR", 1, res): norm}, version=version,
              block_bin_count=64, float_counts=float_counts, dense_blocks=dense, short_coords=short_coords)
    size_bp = (n - 3) * res
    with HicFile(p) as h:
        whole = read_intra_packed(h, "chr1", res, "KR", dpx, size_bp)
        key_w = whole.x.astype(np.int64) * (1 << 20) + whole.dist
        order_w = np.argsort(key_w)
        for slab_bytes, n_slabs in ((1 << 16, 5), (4096, 9)):
            mem = np.zeros(n_slabs * slab_bytes + 16, np.uint8)
            base0 = (-mem.ctypes.data) % 16                        # 16-byte aligned slab memory
            got, blocks, rows_seen = [], 0, 0
            for part in range(n_parts):
                st = HicRawStream(h, "chr1", res, "KR", dpx, mem.ctypes.data + base0, n_slabs, slab_bytes, threads=3,
                                  part=(part, n_parts))
                nv, length = st.info()
                assert length == n * res and len(nv) == n + 1
                assert np.array_equal(nv, norm.astype(np.float32).astype(np.float64) if version > 8 else norm, equal_nan=True)
                while True:
                    r = st.next(50)
                    if r is None:
                        continue
                    if r is False:
                        break
                    slab, nbytes, rows = r
                    assert rows > 0 and nbytes + 16 * rows <= slab_bytes and nbytes % 2 == 0
                    base = base0 + slab * slab_bytes
                    got.append(decode_slab(mem[base:base + nbytes].copy(), mem[base + slab_bytes - 16 * rows:base + slab_bytes].copy(),
                                           nv, dpx, -(-size_bp // res)))
                    rows_seen += rows
                    st.release(slab)
                st.close()
                blocks += st.blocks_mine
                assert st.blocks_total == whole.blocks_total
            assert st.rows_total > 0 and blocks == whole.blocks_total
            gx, gy, gv = (np.concatenate([g[i] for g in got]) for i in range(3))
            k = gx * (1 << 20) + (gy - gx)
            o = np.argsort(k)
            assert len(k) == len(key_w) > 10000 and np.array_equal(k[o], key_w[order_w]) and np.array_equal(gv[o], whole.v[order_w])
            assert int(gy.max()) + 1 == whole.n


def test_own_inflate_accepts_the_incomplete_codes_zlib_accepts():
    """zlib's rule for incomplete Huffman codes (inftrees.c): a literal/length or distance code may be incomplete when its
    longest code is ONE bit long -- e.g. a dynamic block whose only symbol is end-of-block -- and a distance alphabet may hold no
    code at all; every other incomplete code, e.g. a single distance code of two bits, is an error.  Hand-assembled streams:
    mst_inflate.h decides each of them as zlib does."""
    import zlib
    from mustache_amd import hicfile
    lib = hicfile.load()

    class Bits:
        def __init__(self):
            self.acc, self.n, self.out = 0, 0, bytearray()

        def put(self, value, nbits):                 # LSB first
            self.acc |= value << self.n
            self.n += nbits
            while self.n >= 8:
                self.out.append(self.acc & 255)
                self.acc >>= 8
                self.n -= 8

        def code(self, code, length):                # Huffman codes go in MSB first
            for i in range(length - 1, -1, -1):
                self.put((code >> i) & 1, 1)

        def done(self):
            if self.n:
                self.out.append(self.acc & 255)
            return bytes(self.out)

    def stream(dist_len):
        """one final dynamic block: literal/length lengths = {256: 1} (incomplete, longest code one bit), distance lengths =
        {0: dist_len}; code-length code: 18 -> 1 bit '0', 0 -> 2 bits '10', 1 -> '110', 2 -> '111'; then the end-of-block symbol"""
        b = Bits()
        b.put(1, 1); b.put(2, 2)                     # BFINAL, BTYPE = dynamic
        b.put(0, 5); b.put(0, 5); b.put(14, 4)       # HLIT = 257, HDIST = 1, HCLEN = 18 code-length codes
        cl = {18: 1, 0: 2, 1: 3, 2: 3}
        for sym in (16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1):
            b.put(cl.get(sym, 0), 3)
        codes = {18: (0b0, 1), 0: (0b10, 2), 1: (0b110, 3), 2: (0b111, 3)}
        b.code(*codes[18]); b.put(138 - 11, 7)       # 138 zeros
        b.code(*codes[18]); b.put(118 - 11, 7)       # 118 zeros: symbols 0..255
        b.code(*codes[1])                            # symbol 256: one bit
        b.code(*codes[dist_len])                     # the one distance symbol
        b.code(0, 1)                                 # end of block
        raw = b.done()
        return b"\x78\x01" + raw + (1).to_bytes(4, "big")        # zlib header, Adler-32 of the empty output

    def ours(comp):
        out = (ctypes.c_uint8 * 1024)()                # (the call wants 266 bytes of working margin)
        return lib.mst_io_inflate(comp, len(comp), out, 1024)

    for dist_len, ok in ((0, True), (1, True), (2, False)):
        comp = stream(dist_len)
        try:
            z = zlib.decompress(comp)
        except zlib.error:
            z = None
        assert (z == b"") == ok, ("zlib itself", dist_len, z)
        assert (ours(comp) == 0) == ok and (ok or ours(comp) < 0), (dist_len, ours(comp))
