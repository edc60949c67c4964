"""Oracle: per-diagonal z-score normalisation of the sparse contact list (mustache.py:622-686).

TEST INFRASTRUCTURE (see oracle/__init__.py).
"""
import math
import warnings

import numpy as np


def _by_diagonal(dist, dmax):
    """Entry indices grouped by diagonal, each group in original entry order (what the reference's boolean
    mask `distances == d` selects, mustache.py:633) -- a stable sort instead of dmax full-length masks."""
    order = np.argsort(dist, kind="stable")
    sd = dist[order]
    lo = np.searchsorted(sd, np.arange(dmax), side="left")
    hi = np.searchsorted(sd, np.arange(dmax), side="right")
    return [order[a:b] for a, b in zip(lo, hi)]


def normalize_sparse(x, y, v, resolution, distance_in_px):
    """In-place on ``v``; returns the (unused downstream) per-diagonal weight list like the reference."""
    x = np.asarray(x)
    y = np.asarray(y)
    n = int(max(x.max(), y.max())) + 1                      # (:623)
    weights = []
    dist = np.abs(y - x)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore", category=RuntimeWarning)
        if (n - distance_in_px) * resolution > 2000000:     # branch A (:628)
            win = int(2000000 / resolution)                 # (:631)
            box = np.ones(win)
            groups = _by_diagonal(dist, 2 + distance_in_px)
            for d in range(2 + distance_in_px):             # (:632)
                sel = groups[d]
                xs = x[sel]
                vals = np.zeros(n - d)
                vals[xs] = v[sel] + 0.001                   # (:635)
                if vals.size == 0:
                    continue
                std = np.std(v[sel])                        # population std of the raw values (:638)
                mean = np.mean(v[sel])
                if math.isnan(mean):
                    mean = 0
                if math.isnan(std):
                    std = 1
                cnt = np.convolve(vals != 0, box, mode="same")      # (:646)
                s1 = np.convolve(vals, box, mode="same")            # (:648)
                s2 = np.convolve(vals ** 2, box, mode="same")       # (:649)
                var = (s2 - s1 ** 2 / cnt) / (cnt - 1)              # (:650)
                std2 = std ** 2
                np.nan_to_num(var, copy=False, neginf=std2, posinf=std2, nan=std2)
                mu = s1 / cnt
                mu[cnt < 30] = mean                                 # (:657-658)
                var[cnt < 30] = std2
                np.nan_to_num(mu, copy=False, neginf=mean, posinf=mean, nan=mean)
                sd = np.sqrt(var)
                vals[xs] -= mu[xs]                                  # (:664-665)
                vals[xs] /= sd[xs]
                np.nan_to_num(vals, copy=False, nan=0, posinf=0, neginf=0)
                w = 1 + math.log(1 + mean, 30)                      # (:667)
                vals = vals * w
                weights.append(w)
                v[sel] = vals[xs]                                   # (:669)
        else:                                               # branch B (:671-685)
            np.nan_to_num(v, copy=False, neginf=0, posinf=0, nan=0)
            dmax = min(distance_in_px, n)
            groups = _by_diagonal(dist, dmax)
            for d in range(dmax):
                sel = groups[d]
                std = np.std(v[sel])
                mean = np.mean(v[sel])
                if math.isnan(mean):
                    mean = 0
                if math.isnan(std):
                    std = 1
                v[sel] = (v[sel] - mean) / std
                np.nan_to_num(v, copy=False, nan=0, posinf=0, neginf=0)
    return weights
