"""INTEGRATION.md section 2 (the direct ctypes binding a maintainer would write) executed as written: the code block is
extracted from the document, its placeholders (the block `c`, the parameters of mustache(), the level table) are supplied,
and the records it produces are compared with the package's own engine."""
import ctypes
import os
import re

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_integration_md_ctypes_snippet_runs_and_matches_engine():
    import torch
    from mustache_amd.engine import ScaleSpaceEngine
    from mustache_amd.levels import LevelTable
    from mustache_amd.synth import synth_coo
    text = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    sec = text[text.index("## 2. Direct ctypes binding"):]
    code = re.search(r"```python\n(.*?)```", sec, re.S).group(1)
    n, dpx = 512, 128
    x, y, v = synth_coo(n, dpx, depth=200.0, seed=2)
    c = np.zeros((n, n))
    c[x, y] = v
    octave_values = [1.6, 3.2]
    # the snippet leaves the level table to the reader ("fill from LevelTable(...).as_struct()"): do that right after `lv = MstLevels()`
    ref_struct = LevelTable(octave_values).as_struct()
    lines = code.split("\n")
    at = [i for i, l in enumerate(lines) if l.startswith("lv = MstLevels()")]
    assert len(at) == 1
    lines.insert(at[0] + 1, "ctypes.memmove(ctypes.byref(lv), ctypes.byref(_ref_struct), ctypes.sizeof(MstLevels))")
    code = "\n".join(lines)
    code = code.replace('ctypes.CDLL("mustache_amd/libmustache_hip.so")', 'ctypes.CDLL(os.path.join(ROOT, "mustache_amd", "libmustache_hip.so"))')
    ns = dict(c=c.copy(), distance_in_px=dpx, octave_values=octave_values, _ref_struct=ref_struct, os=os, ROOT=ROOT)
    assert ctypes.sizeof(ref_struct) > 0
    exec(compile(code, "INTEGRATION.md", "exec"), ns)
    torch.cuda.synchronize()
    assert ctypes.sizeof(ns["MstLevels"]) == ctypes.sizeof(ref_struct), "the documented struct layout must match the header"
    cnt = int(ns["d_cnt"].cpu()[0])
    rec = ns["d_found"][0, :cnt].cpu().numpy()
    pix = (rec[:, 0] & 0xFFFFFFFF).astype(np.int64)
    order = np.argsort(pix)
    pvals = ns["d_p"][0, :cnt].cpu().numpy()[order]
    eng = ScaleSpaceEngine(octave_values)
    dev = torch.from_numpy(c.copy()).to(eng.device).unsqueeze(0)
    nz, nzc = eng.prologue(dev, dpx)
    found, _ = eng.sigma_loop(dev, nz, nzc, with_q=False)
    assert cnt == len(found[0]["pixel"]) > 100
    assert np.array_equal(pix[order], found[0]["pixel"].astype(np.int64))
    assert np.array_equal((rec[:, 0] >> 32)[order], found[0]["level"].astype(np.int64))
    assert np.array_equal(rec[:, 1].view(np.float64)[order], found[0]["value"])
    assert np.array_equal(pvals, found[0]["pval"])


def test_integration_md_raw_hic_snippet_runs_and_gives_the_packed_band(tmp_path):
    """INTEGRATION.md section 3's RAW `.hic` read (mst_hic_rawstream_* + mst_band_scatter_hic_rows), executed as written behind
    the section's first snippet (which opens the file) and section 2's helpers: the band it builds is the band of the host
    decoder's packed read, bit for bit."""
    import sys
    import torch
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from hic_writer import write_hic
    from mustache_amd.hicfile import HicFile, read_intra_packed
    from mustache_amd.normalize import band_from_packed
    text = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    sec2 = text[text.index("## 2. Direct ctypes binding"):]
    helpers = re.search(r"```python\n(.*?)```", sec2, re.S).group(1)
    helpers = helpers[:helpers.index("# inside the reference's mustache(")]        # lib, ck, P, S
    sec3 = text[text.index("## 3."):]
    blocks = re.findall(r"```python\n(.*?)```", sec3, re.S)
    opener = blocks[0][:blocks[0].index("px, py, pv =")]                           # io, h
    raw = [b for b in blocks if "mst_hic_rawstream_open" in b]
    assert len(raw) == 1
    n, res, dpx = 6000, 1000, 2000
    rng = np.random.default_rng(8)
    x = rng.integers(0, n - 1, 400000)
    y = np.minimum(x + rng.integers(0, 2400, 400000), n - 2)        # the last bin holds no contact: the band is trimmed
    key = np.unique(x * 100003 + y)
    x, y = key // 100003, key % 100003
    kr = rng.uniform(0.5, 2.0, n + 1)
    kr[[7, 5000]] = np.nan
    p = str(tmp_path / "sample.hic")
    write_hic(p, [("All", 1), ("chr1", n * res)], {1: {res: (x, y, rng.integers(1, 50, len(x)).astype(np.float64))}},
              {("KR", 1, res): kr}, version=8, block_bin_count=500, float_counts=False)
    fix = lambda code: code.replace('ctypes.CDLL("mustache_amd/libmustache_hip.so")', 'ctypes.CDLL(os.path.join(ROOT, "mustache_amd", "libmustache_hip.so"))') \
        .replace('ctypes.CDLL("mustache_amd/libmustache_io.so")', 'ctypes.CDLL(os.path.join(ROOT, "mustache_amd", "libmustache_io.so"))') \
        .replace('b"sample.hic"', "PATH")
    ns = dict(os=os, ROOT=ROOT, PATH=os.fsencode(p))
    for code in (helpers, opener, raw[0]):
        exec(compile(fix(code), "INTEGRATION.md", "exec"), ns)
    torch.cuda.synchronize()
    with HicFile(p) as h:
        want = band_from_packed(read_intra_packed(h, "chr1", res, "KR", dpx, 0), dpx, torch.device("cuda", 0))
    assert ns["n"] == want.shape[1] < n and torch.equal(ns["band"], want)
    kept = int(ns["stats"][1])
    assert kept == int((want != 0).sum()) > 100000 and int(ns["stats"][2]) == 0
