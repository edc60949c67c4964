"""Import the upstream reference (read-only at /root/reference) in THIS dev container only.

Test infrastructure.  Used by ``make_golden.py`` to produce the committed ``*.npz`` fixtures;
nothing that runs on the GPU box imports this file (the reference tree does not exist there).

The reference imports three third-party modules that are absent from this image at module top
(mustache.py:14-15, :24).  We provide the minimum stand-ins needed for the *module import* to succeed;
only ``multipletests(method='fdr_bh')`` is ever executed and it is restated from statsmodels'
published Benjamini-Hochberg algorithm (sort, p*m/rank, reverse cumulative minimum, clip to 1,
un-sort).  Because that dependency is absent, BH parity is pinned to this restatement only
(documented as such in DESIGN.md).
"""
import importlib.util
import os
import sys
import types

import numpy as np

REF_ROOT = "/root/reference"


def _bh(pvals, alpha=0.05, method="fdr_bh", **_kw):
    assert method == "fdr_bh"
    pvals = np.asarray(pvals, dtype=float)
    order = np.argsort(pvals)
    ps = np.take(pvals, order)
    m = len(ps)
    ecdf = np.arange(1, m + 1) / float(m)
    raw = ps / ecdf
    corr = np.minimum.accumulate(raw[::-1])[::-1]
    corr[corr > 1] = 1
    out = np.empty_like(corr)
    out[order] = corr
    return out <= alpha, out, None, None


def load_reference(name="mustache"):
    """Return the reference module ``mustache`` or ``diff_mustache`` (imported, never copied)."""
    if not os.path.isdir(REF_ROOT):
        raise RuntimeError("reference tree not present (expected only in the dev container)")
    sys.dont_write_bytecode = True
    for stub in ("hicstraw", "cooler"):
        sys.modules.setdefault(stub, types.ModuleType(stub))
    if "statsmodels" not in sys.modules:
        sm = types.ModuleType("statsmodels")
        sms = types.ModuleType("statsmodels.stats")
        smm = types.ModuleType("statsmodels.stats.multitest")
        smm.multipletests = _bh
        sm.stats = sms
        sms.multitest = smm
        sys.modules.update({"statsmodels": sm, "statsmodels.stats": sms,
                            "statsmodels.stats.multitest": smm})
    if not hasattr(np, "Inf"):
        np.Inf = np.inf  # mustache.py:234-248 uses the NumPy-1 alias
    refdir = os.path.join(REF_ROOT, "mustache")
    if refdir not in sys.path:
        sys.path.append(refdir)  # diff_mustache.py:16 does `from mustache import ...`
    import warnings
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        spec = importlib.util.spec_from_file_location(
            "_ref_" + name, os.path.join(refdir, name + ".py"))
        mod = importlib.util.module_from_spec(spec)
        if name == "mustache":
            sys.modules.setdefault("mustache", mod)
        spec.loader.exec_module(mod)
    return mod
