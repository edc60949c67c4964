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
