#!/usr/bin/env python3
"""Randomised check of the streamed `.hic` reads (mst_hic_stream_* and, since round 5, the RAW form mst_hic_rawstream_* whose
rows are decoded here by the NumPy restatement of the device kernel, tests/hic_rows_numpy.py; CPU only): random files (versions 8-9, row-list / dense
blocks, short / long coordinates, integer / float counts, block sizes), random slab sizes from 2 records to larger than any
block, random slab and thread counts, random part splits, a consumer that releases slabs late and out of order -- the records
delivered must be exactly those of the one-shot packed read.  argv: cases [seed]"""
import os
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np                                                     # noqa: E402
from hic_writer import write_hic                                       # noqa: E402
from hic_rows_numpy import decode_slab                                 # noqa: E402
from mustache_amd.hicfile import HicFile, HicRawStream, HicStream, read_intra_packed  # noqa: E402

cases = int(sys.argv[1]) if len(sys.argv) > 1 else 40
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
bad = 0
tmp = tempfile.mkdtemp(prefix="mst_fuzz_hic_")
for case in range(cases):
    n = int(rng.integers(200, 3000))
    res = int(rng.choice([1000, 5000, 25000]))
    spread = int(rng.integers(20, 400))
    dpx = int(rng.integers(10, spread + 20))
    m = int(rng.integers(100, 60000))
    x = rng.integers(0, n, m)
    y = np.minimum(x + rng.integers(0, spread, m), n - 1)
    key = np.unique(x * 1000003 + y)
    x, y = key // 1000003, key % 1000003
    float_counts = bool(rng.integers(2))
    c = rng.uniform(0.25, 40, len(x)).astype(np.float32).astype(np.float64) if float_counts else rng.integers(1, 900, len(x)).astype(np.float64)
    version = int(rng.choice([8, 9]))
    dense = bool(rng.integers(4) == 0)
    short_coords = bool(rng.integers(2)) if version == 9 else True
    bbc = int(rng.choice([16, 64, 128, 500]))
    norm = rng.uniform(0.5, 2.0, n + 1)
    norm[rng.integers(0, n, 3)] = np.nan
    p = os.path.join(tmp, "f.hic")
    write_hic(p, [("All", 7500), ("chr1", n * res)], {1: {res: (x, y, c)}}, {("KR", 1, res): norm}, version=version,
              block_bin_count=bbc, float_counts=float_counts, dense_blocks=dense, short_coords=short_coords)
    dist_bytes = int(rng.choice([2, 4]))
    n_parts = int(rng.integers(1, 4))
    cap = 2 * int(rng.choice([1, 3, 17, 129, 1000, 40000]))
    n_slabs = int(rng.integers(2, 12))
    threads = int(rng.integers(1, 9))
    hold = int(rng.integers(0, max(1, n_slabs - 1)))          # slabs the consumer keeps back before releasing the oldest
    with HicFile(p) as h:
        whole = read_intra_packed(h, "chr1", res, "KR", dpx, 0)
        key_w = whole.x.astype(np.int64) * (1 << 20) + whole.dist
        mem = np.zeros(n_slabs * cap * (8 + dist_bytes), np.uint8)
        got_k, got_v, blocks = [], [], 0
        for part in range(n_parts):
            st = HicStream(h, "chr1", res, "KR", dpx, 0, mem.ctypes.data, n_slabs, cap, dist_bytes, threads=threads,
                           part=(part, n_parts))
            held = []
            while True:
                r = st.next(20)
                if r is None:
                    if held:                                  # nothing ready: the workers may be waiting for a slab
                        st.release(held.pop(int(rng.integers(len(held)))))
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
                held.append(slab)
                while len(held) > hold:
                    st.release(held.pop(int(rng.integers(len(held)))))
            st.close()
            blocks += st.blocks_mine
        k = np.concatenate(got_k) if got_k else np.zeros(0, np.int64)
        v = np.concatenate(got_v) if got_v else np.zeros(0, np.float32)
        o, ow = np.argsort(k, kind="stable"), np.argsort(key_w, kind="stable")
        ok = blocks == whole.blocks_total and len(k) == len(key_w) and np.array_equal(k[o], key_w[ow]) and \
            np.array_equal(v[o], whole.v[ow])
        # the raw form: slabs of row payloads + directories (from 4 KB, i.e. blocks staged and cut at rows, to slabs that take
        # whole blocks inflated in place), the same late / out-of-order releases
        slab_bytes = 16 * int(rng.choice([256, 300, 1024, 4096, 65536]))
        rmem = np.zeros(n_slabs * slab_bytes + 16, np.uint8)
        base0 = (-rmem.ctypes.data) % 16
        rk, rv, rblocks = [], [], 0
        for part in range(n_parts):
            st = HicRawStream(h, "chr1", res, "KR", dpx, rmem.ctypes.data + base0, n_slabs, slab_bytes, threads=threads,
                              part=(part, n_parts))
            nv, _ = st.info()
            held = []
            while True:
                r = st.next(20)
                if r is None:
                    if held:
                        st.release(held.pop(int(rng.integers(len(held)))))
                    continue
                if r is False:
                    break
                slab, nbytes, rows = r
                base = base0 + slab * slab_bytes
                gx, gy, gv = decode_slab(rmem[base:base + nbytes].copy(), rmem[base + slab_bytes - 16 * rows:base + slab_bytes].copy(),
                                         nv, dpx)
                rk.append(gx * (1 << 20) + (gy - gx))
                rv.append(gv)
                held.append(slab)
                while len(held) > hold:
                    st.release(held.pop(int(rng.integers(len(held)))))
            st.close()
            rblocks += st.blocks_mine
        k = np.concatenate(rk) if rk else np.zeros(0, np.int64)
        v = np.concatenate(rv) if rv else np.zeros(0, np.float32)
        o = np.argsort(k, kind="stable")
        ok = ok and rblocks == whole.blocks_total and len(k) == len(key_w) and np.array_equal(k[o], key_w[ow]) and \
            np.array_equal(v[o], whole.v[ow])
    desc = dict(n=n, res=res, dpx=dpx, records=len(key_w), version=version, dense=dense, short=short_coords, floats=float_counts,
                bbc=bbc, dist_bytes=dist_bytes, parts=n_parts, cap=cap, slabs=n_slabs, threads=threads, hold=hold, raw_slab=slab_bytes)
    print("case %3d %s %s" % (case, "ok " if ok else "MISMATCH", desc), flush=True)
    bad += not ok
print("done: %d cases, %d mismatches" % (cases, bad))
sys.exit(1 if bad else 0)
