"""Host-side Gaussian level table: the arguments the reference hands to scipy.ndimage.gaussian_filter
(mustache/mustache.py:714-752), reduced to integer radii and normalised taps exactly the way SciPy does it
(scipy/ndimage/_filters.py:226-236 `_gaussian_kernel1d`, :314-316 radius = int(truncate*sigma + 0.5)).

The taps are computed here with NumPy -- the same library calls SciPy itself makes -- and handed to the HIP
kernels, so the device never evaluates exp() for them and the weights are bit-identical to the reference's on
any host."""
import math

import numpy as np

from . import _lib


class LevelTable:
    def __init__(self, octave_values, s=10):
        self.octave_values = [float(o) for o in octave_values]
        self.s = int(s)
        self.levels_per_octave = self.s + 2
        self.sigma, self.truncate, self.radius, self.taps = [], [], [], []
        for o in self.octave_values:
            for k in range(1, self.s + 3):
                # sigma_1 = o (:716); sigma_k = o * 2**((k-1)/s) (:722, :731, :748)
                sigma = o if k == 1 else o * 2 ** ((k - 1) / self.s)
                w = 2 * math.ceil(2 * sigma) + 1                    # (:717)
                t = (((w - 1) / 2) - 0.5) / sigma                   # (:718)
                r = int(t * float(sigma) + 0.5)                     # _filters.py:316
                x = np.arange(-r, r + 1)
                phi = np.exp(-0.5 / (sigma * sigma) * x ** 2)       # _filters.py:233-235
                phi = phi / phi.sum()
                self.sigma.append(sigma)
                self.truncate.append(t)
                self.radius.append(r)
                self.taps.append(phi[r:].copy())                    # centre, +1, ..., +r (bit-symmetric kernel)
        # recorded scale of tested level t (0-based): sigma_i for loop index i = 3..s+1 (mustache.py:767)
        self.tested_sigma = []
        for oi in range(len(self.octave_values)):
            for i in range(3, self.s + 2):
                self.tested_sigma.append(self.sigma[oi * self.levels_per_octave + i - 1])
        self.n_tested = len(self.tested_sigma)
        if len(self.sigma) > _lib.MST_MAX_LEVELS or self.n_tested > _lib.MST_MAX_TESTED:
            raise ValueError("too many scale-space levels for the HIP kernel (octaves=%d, s=%d)"
                             % (len(self.octave_values), self.s))
        if max(self.radius) > 28 or min(self.radius) < 1:
            raise ValueError("blur radius %d..%d outside the HIP kernel's supported range 1..28"
                             % (min(self.radius), max(self.radius)))

    def as_struct(self):
        st = _lib.MstLevels()
        st.n_octaves = len(self.octave_values)
        st.levels_per_octave = self.levels_per_octave
        for l, (r, sg, tp) in enumerate(zip(self.radius, self.sigma, self.taps)):
            st.radius[l] = r
            st.sigma[l] = sg
            for j in range(r + 1):
                st.taps[l][j] = float(tp[j])
        return st
