"""Oracle: overlap tiling of the band into dense square blocks (mustache.py:896-924, :948-959).

TEST INFRASTRUCTURE (see oracle/__init__.py).
"""
import numpy as np


def block_bounds(n, distance_in_px):
    """(chunk, starts, ends): chunk = max(2*dpx, 2000) (:896); one block [0, n) when n <= chunk (:899-901);
    otherwise blocks of `chunk` bins advancing by chunk-dpx, the last one right-aligned to n (:903-910)."""
    chunk = max(2 * distance_in_px, 2000)
    if n <= chunk:
        return chunk, [0], [n]
    starts, ends = [0], [chunk]
    while ends[-1] < n:
        starts.append(ends[-1] - distance_in_px)
        ends.append(starts[-1] + chunk)
    ends[-1] = n
    starts[-1] = n - chunk
    return chunk, starts, ends


def block_mask_size(i, starts, ends, distance_in_px):
    """Overlap mask of block i (:948-953): -1 for the first block, the true overlap for the last, dpx otherwise."""
    if i == 0:
        return -1
    if i == len(starts) - 1:
        return ends[i - 1] - starts[i]
    return distance_in_px


def dense_block(x, y, v, start, end, chunk):
    """COO -> dense chunk x chunk float64 block holding every entry with start <= x, y < end (:919-924)."""
    sel = np.logical_and.reduce((x >= start, x < end, y >= start, y < end))
    cc = np.zeros((chunk, chunk))
    cc[x[sel] - start, y[sel] - start] = v[sel]
    return cc


def keep_loop(lx, ly, start, mask_size):
    """De-duplication rule for loops found in the overlap (:957-959)."""
    return lx >= start + mask_size or ly >= start + mask_size
