"""CPU oracle for the mustache scale-space hot path.  TEST INFRASTRUCTURE ONLY.

This package is a NumPy/SciPy restatement of the per-block loop-calling path of the reference
(`/root/reference/mustache/mustache.py`, v1.3.3), written from SURVEY.md section 8, each function citing
the reference lines it follows.  It exists to check the HIP path, nothing else:

  * only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import it;
  * ``mustache_amd`` (the product) never imports it and has no CPU fallback.

Pinning: the reference ships no tests or golden vectors, so the oracle is pinned against
outputs of the reference itself, imported in the dev container by ``tests/golden/make_golden.py`` and
committed as ``tests/golden/*.npz`` (``tests/test_oracle_golden.py`` replays them).  The Benjamini-Hochberg
step lives in ``statsmodels`` (unpinned in the reference's environment.yml:7-16, absent from this image); it is
restated from the published algorithm and is therefore pinned only against that restatement.
"""
from .scale_space import (gaussian_weights, level_table, blur_explicit, blur_scipy, maxfilter3_zero,
                          block_prologue, scale_space_levels, ScaleSpaceResult)
from .tail import benjamini_hochberg, block_tail
from .normalize import normalize_sparse
from .tiling import block_bounds, block_mask_size, dense_block
from .pipeline import mustache_block, regulator_coo
from .diff import diff_block

__all__ = [
    "gaussian_weights", "level_table", "blur_explicit", "blur_scipy", "maxfilter3_zero", "block_prologue",
    "scale_space_levels", "ScaleSpaceResult", "benjamini_hochberg", "block_tail", "normalize_sparse",
    "block_bounds", "block_mask_size", "dense_block", "mustache_block", "regulator_coo", "diff_block",
]
