"""Oracle: one block end to end (`mustache()`, mustache.py:697-850) and one chromosome (`regulator`, :892-937).

TEST INFRASTRUCTURE (see oracle/__init__.py).
"""
import numpy as np

from .scale_space import block_prologue, scale_space_levels
from .tail import block_tail
from .normalize import normalize_sparse
from .tiling import block_bounds, block_mask_size, dense_block, keep_loop


def mustache_block(c, start, distance_in_px, octave_values, st, pt, intra=True, blur="scipy",
                   return_intermediate=False):
    """Same result as the reference's mustache(c, chrom, chrom2, res, w, start, end, mask, dpx, octaves, st, pt);
    the arguments the reference ignores (res, pval_weights, end, mask_size) are dropped.  Mutates ``c``."""
    nz = block_prologue(c, distance_in_px, intra)
    if np.sum(nz) < 50:                                           # (:701)
        return ([], None) if return_intermediate else []
    ss = scale_space_levels(c, nz, octave_values, blur=blur)
    loops = block_tail(c, nz, ss.pval, ss.scale, start, pt, st, intra=intra)
    if return_intermediate:
        return loops, dict(nz=nz, ss=ss)
    return loops


def regulator_coo(x, y, v, res, distance_in_px, octave_values, st, pt, blur="scipy", normalize=True):
    """Normalise -> tile -> per-block loop calling -> overlap de-duplication (mustache.py:892-937, :945-960).
    Returns loops sorted by (x, y) (the reference's order is process-completion order)."""
    x = np.asarray(x, dtype=np.int64)
    y = np.asarray(y, dtype=np.int64)
    if normalize:
        normalize_sparse(x, y, v, res, distance_in_px)
    n = int(max(x.max(), y.max())) + 1
    chunk, starts, ends = block_bounds(n, distance_in_px)
    out = []
    for i in range(len(starts)):
        cc = dense_block(x, y, v, starts[i], ends[i], chunk)
        mask = block_mask_size(i, starts, ends, distance_in_px)
        for lp in mustache_block(cc, starts[i], distance_in_px, octave_values, st, pt, blur=blur):
            if keep_loop(lp[0], lp[1], starts[i], mask):
                out.append([lp[0], lp[1], lp[2], lp[3]])
    out.sort(key=lambda r: (r[0], r[1]))
    return out
