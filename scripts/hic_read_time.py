#!/usr/bin/env python3
"""Thread scaling of the native `.hic` reader's packed path on the bench's synthetic chr1 @ 1 kb file (GPU box):
decode (inflate + record decode into the per-thread arenas) and fetch (copy into page-locked buffers) timed apart, for several
thread counts; three calls each, the last one reported (arenas warm)."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import ctypes          # noqa: E402
import torch           # noqa: E402
sys.path.insert(0, os.path.join(ROOT, 'tests'))
import hic_writer      # noqa: E402
from mustache_amd.hicfile import HicFile, _check     # noqa: E402
from mustache_amd.normalize import pinned_packed_alloc   # noqa: E402

path = sys.argv[1] if len(sys.argv) > 1 else "/tmp/chr1_1kb.hic"
if not os.path.exists(path):
    t0 = time.time()
    n = hic_writer.write_synthetic_hic(path, 248957, 2000, 1000, 400.0, 8000, 1, 200.0, torch.device("cuda:0"))
    print("wrote %d records, %d bytes in %.1f s" % (n, os.path.getsize(path), time.time() - t0), flush=True)
print("cpu_count", os.cpu_count(), "affinity", len(os.sched_getaffinity(0)), flush=True)
try:
    print("cgroup cpu.max:", open("/sys/fs/cgroup/cpu.max").read().strip(), flush=True)
except Exception as e:
    print("cgroup cpu.max unreadable:", e, flush=True)
for nt in (8, 16, 32, 64, 128, 256):
    h = HicFile(path)
    for rep in range(3):
        nb = ctypes.c_int64()
        t0 = time.time()
        cnt = _check(h._lib, h._lib.mst_hic_decode_intra_packed(h._h, b"chr1", 1000, b"KR", 2000, 0, nt, ctypes.byref(nb)))
        t1 = time.time()
        x, d, v, keep = pinned_packed_alloc(cnt)
        t2 = time.time()
        _check(h._lib, h._lib.mst_hic_fetch_packed(h._h, x.ctypes.data_as(ctypes.c_void_p), d.ctypes.data_as(ctypes.c_void_p),
                                                   v.ctypes.data_as(ctypes.c_void_p), cnt, nt))
        t3 = time.time()
        del x, d, v, keep
    print("threads %3d: decode %.4f s  pinned alloc %.4f s  fetch %.4f s  (%d records)" % (nt, t1 - t0, t2 - t1, t3 - t2, cnt),
          flush=True)
    h.close()

# ---- round 4: the streamed read (own inflate, row-list decode into page-locked slabs, slabs uploaded while later blocks
# inflate) next to it -- wall time until the records are in HBM; MUSTACHE_HIC_ZLIB=1 in the environment swaps zlib back in
from mustache_amd.normalize import band_from_packed, read_hic_stream_to_device   # noqa: E402
dev = torch.device("cuda:0")
for nt in (8, 16, 24, 32, 48, 64):
    h = HicFile(path)
    ts = []
    for rep in range(4):
        torch.cuda.synchronize()
        t0 = time.time()
        pc = read_hic_stream_to_device(h, "chr1", 1000, "KR", 2000, 0, dev, threads=nt, n_slabs=nt + 8)
        torch.cuda.synchronize()
        t1 = time.time()
        band = band_from_packed(pc, 2000, dev)
        torch.cuda.synchronize()
        t2 = time.time()
        ts.append((t1 - t0, t2 - t1))
        cnt = pc.count
        del pc, band
    best = min(ts[1:])
    print("streamed, threads %3d: records in HBM after %.4f s, zero fill + scatter %.4f s  (%d records; first call %.4f s)"
          % (nt, best[0], best[1], cnt, ts[0][0]), flush=True)
    h.close()
