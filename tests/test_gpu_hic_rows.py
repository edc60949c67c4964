"""mst_band_scatter_hic_rows (csrc/mst_hic_rows.hip) on the GPU: the device half of the raw `.hic` read against the host decoder
(libmustache_io.so's packed read, itself held to two independent readings of the format in test_hic_two_readings.py) -- on
everything the writer emits and on the hand-assembled corner cases of that file (empty rows, rows out of order, 16-bit wrap,
dense grids with holes, v9 mixed widths): same pixels, same float32 values, same n, through the C ABI."""
import os
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))


def _device_records(path, chrom, res, norm, dpx, size_bp=0, slab_records=512, check=False):
    import torch
    from mustache_amd.hicfile import HicFile
    from mustache_amd.normalize import band_from_packed, read_hic_stream_to_device
    dev = torch.device("cuda", 0)
    with HicFile(path) as h:
        pc = read_hic_stream_to_device(h, chrom, res, norm, dpx, size_bp, dev, threads=2, slab_records=slab_records, raw=True,
                                       keep_raw=check)
    assert pc.device_band is not None
    band = band_from_packed(pc, dpx, dev, check=check)
    d, x = torch.nonzero(band, as_tuple=True)
    v = band[d, x]
    recs = sorted(zip(x.cpu().tolist(), (x + d).cpu().tolist(), v.cpu().numpy().astype(np.float32)))
    assert pc.count == len(recs) and int(band.shape[1]) == pc.n == (max(r[1] for r in recs) + 1 if recs else 0)
    return recs


@pytest.mark.parametrize("version,float_counts,dense,short_coords,bbc", [
    (8, False, False, True, 64), (8, True, False, True, 37), (8, False, True, True, 16), (8, True, True, True, 16),
    (9, False, False, True, 64), (9, True, False, False, 50), (9, True, False, True, 23)])
def test_device_rows_equal_host_decoder_on_writer_files(tmp_path, version, float_counts, dense, short_coords, bbc):
    import test_hic_two_readings as two
    from hic_writer import write_hic
    rng = np.random.default_rng(version * 100 + bbc)
    n, res = 900, 5000
    x = rng.integers(0, n, 12000)
    y = np.minimum(x + rng.integers(0, 300, 12000), n - 1)
    key = np.unique(x * 100003 + y)
    x, y = key // 100003, key % 100003
    c = rng.integers(1, 900, len(x)).astype(np.float64) if not float_counts else \
        rng.uniform(0.25, 40, len(x)).astype(np.float32).astype(np.float64)
    norm = rng.uniform(0.4, 2.5, n + 1)
    norm[[5, 77, 899]] = np.nan
    p = str(tmp_path / "w.hic")
    write_hic(p, [("All", 7500), ("chr1", n * res), ("chrX", 12345)], {1: {res: (x, y, c)}},
              {("KR", 1, res): norm, ("VC", 1, res): np.ones(n + 1)}, version=version, block_bin_count=bbc,
              float_counts=float_counts, dense_blocks=dense, short_coords=short_coords)
    for norm_name, dpx, check in (("NONE", 310, False), ("KR", 310, True), ("KR", 120, False), ("VC", 40, False)):
        two._same(_device_records(p, "chr1", res, norm_name, dpx, check=check), two._native(p, "chr1", res, norm_name, dpx))
    # the caller's chromosome size cuts the last bins off (mustache.py:320-333), on the device as in the host decoder
    got = _device_records(p, "chr1", res, "NONE", 310, size_bp=(n - 100) * res)
    assert got == [r for r in two._native(p, "chr1", res, "NONE", 310) if r[1] < n - 100] and len(got) > 1000


@pytest.mark.parametrize("version", [7, 8, 9])
def test_device_rows_equal_host_decoder_on_hand_assembled_corner_cases(tmp_path, version):
    import test_hic_two_readings as two
    res, per_block, columns = 1000, 50000, 3
    nbins = 120000
    norm = np.linspace(0.5, 2.0, nbins)
    blocks = {
        0: two._rows(version, 37, 41, [(5, [(0, 7), (3, 2)]), (2, []), (0, [(0, 1), (1, 0), (9, 5)]), (1, [(30, 4)])], True),
        1: two._rows(version, 1000, 1000, [(0, [(0, 0.5), (1, float("nan")), (2, -3.0)]), (7, [(7, 123.25)])], False),
        2: two._rows(version, 50000, 50000, [(40000, [(40000, 6)]), (10, [(32767, 8), (32768, 9)])], True),
        3: two._grid(version, 200, 300, 4, [1, None, 3, 4, None, None, 7, 8, 9, 10], True),
        4: two._grid(version, 60000, 60010, 3, [0.25, None, 2.5, None, 4.75], False),
        5: two._grid(version, 5, 5, 1, [2, None, 4], True),
        6: two._rows(version, 0, 0, [], True)}
    if version == 9:
        blocks[7] = two._rows(version, 10, 20, [(70000, [(66000, 2.5), (5, 1.5)]), (3, [(90000, 4.0)])], False, wide_x=True, wide_y=True)
        blocks[8] = two._rows(version, 10, 20, [(9, [(70001, 3)])], True, wide_x=True, wide_y=False)
        blocks[9] = two._rows(version, 10, 20, [(80000, [(2, 3)])], True, wide_x=False, wide_y=True)
    p = str(tmp_path / "h.hic")
    two._container(p, version, nbins * res, res, per_block, columns, blocks, norm)
    # (distance limit = the band's height; the hand-numbered blocks do not sit where their numbers say, so the host side is
    # asked without a limit -- no block skipped -- and filtered here)
    dpx = 40000
    for norm_name in ("NONE", "KR"):
        want = [r for r in two._native(p, "c", res, norm_name, -1) if r[1] - r[0] <= dpx]
        assert len(want) >= 15
        import torch
        from mustache_amd.hicfile import HicFile, HicRawStream
        from mustache_amd import _lib
        from hic_rows_numpy import decode_slab
        # through the C ABI directly: slabs from the raw stream (no block skipped: max_dist -1 at open), the kernel with the
        # distance limit, the band read back
        lib = _lib.load()
        dev = torch.device("cuda", 0)
        slab_bytes, n_slabs = 4096, 4
        mem = torch.zeros(n_slabs * slab_bytes, dtype=torch.uint8).pin_memory()
        band = torch.zeros((dpx + 2, nbins), dtype=torch.float64, device=dev)
        stats = torch.zeros(4, dtype=torch.int64, device=dev)
        with HicFile(p) as h:
            st = HicRawStream(h, "c", res, norm_name, -1, mem.data_ptr(), n_slabs, slab_bytes, threads=2)
            nv, length = st.info()
            nd = None if nv is None else torch.from_numpy(nv).to(dev)
            cpu = []
            while True:
                r = st.next(50)
                if r is None:
                    continue
                if r is False:
                    break
                slab, nbytes, rows = r
                base = slab * slab_bytes
                pay, dr = mem[base:base + nbytes].to(dev), mem[base + slab_bytes - 16 * rows:base + slab_bytes].to(dev)
                _lib.check(lib.mst_band_scatter_hic_rows(pay.data_ptr(), dr.data_ptr(), rows, None if nd is None else nd.data_ptr(),
                                                         -1 if nd is None else nd.numel(), dpx, 0, nbins, dpx, band.data_ptr(),
                                                         stats.data_ptr(), 0, None))
                cpu.append(decode_slab(mem[base:base + nbytes].numpy().copy(), mem[base + slab_bytes - 16 * rows:base + slab_bytes].numpy().copy(),
                                       nv, dpx))
                torch.cuda.synchronize()
                st.release(slab)
            st.close()
        got = []
        for d0 in range(0, dpx + 2, 2000):               # (the band holds 4.8e9 samples: torch.nonzero in slices)
            sl = band[d0:d0 + 2000]
            d, x = torch.nonzero(sl, as_tuple=True)
            got += list(zip(x.cpu().tolist(), (x + d + d0).cpu().tolist(), sl[d, x].cpu().numpy().astype(np.float32)))
        two._same(sorted(got), want)
        two._same(sorted(zip(*(np.concatenate([c[i] for c in cpu]).tolist() for i in range(2)), np.concatenate([c[2] for c in cpu]))), want)
        ymax1, kept, beyond, bad = stats.cpu().tolist()
        assert ymax1 == max(r[1] for r in want) + 1 and kept == len(want) and beyond == 0 and bad == 0


def test_device_rows_random_files_equal_the_host_decoded_band(tmp_path):
    """A seeded slice of random `.hic` files (versions 8-9, row lists / dense grids, short / long coordinates, integer / float
    counts, block sizes, NaN norm entries, random slab sizes and thread counts, with and without a caller's chromosome size):
    the band of the raw streamed read (rows decoded on the device) equals the band of the packed streamed read (host decoder),
    bit for bit, with the same n and record count."""
    import torch
    import fuzz_cases
    rng = np.random.default_rng(20505)
    dev = torch.device("cuda", 0)
    total = sum(fuzz_cases.hic_rows_case(rng, str(tmp_path / ("f%d.hic" % case)), dev) for case in range(14))
    assert total > 100000
