"""Device-side driver of the scale-space hot path: torch tensors for HBM + streams, HIP kernels through the C ABI.

PyTorch is plumbing here (device memory, streams); every number is produced by libmustache_hip.so.  A missing
library or a missing GPU is an error -- there is no CPU path in this package.
"""
import ctypes
import os

import numpy as np
import torch

from . import _lib
from .levels import LevelTable


def _ptr(t):
    return ctypes.c_void_p(t.data_ptr()) if t is not None else None


def _stream():
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


_STREAMS = {}


def device_streams(device):
    """The package's three side streams of `device`, created ONCE per process: two for the fused kernel's launches (alternating,
    so that the post-processing of one launch runs under the kernel of the next) and one for host-to-device copies of the
    streaming `.hic` read.  HIP maps streams to a handful of hardware queues in creation order; a stream per engine or per call
    makes that mapping depend on what else the process has created, and two streams that land on one queue serialise -- the
    host tail's small kernels then wait behind the next launch's fused kernel (measured: a whole-genome run 0.032 -> 0.053 s
    after another engine had created two streams).  One fixed set keeps the mapping the same in every run."""
    key = (device.type, device.index if device.index is not None else torch.cuda.current_device())
    got = _STREAMS.get(key)
    if got is None:
        got = _STREAMS[key] = tuple(torch.cuda.Stream(device) for _ in range(3))
    return got


def require_gpu():
    if not torch.cuda.is_available():
        raise RuntimeError("mustache_amd needs a ROCm GPU (MI355X/gfx950); no CPU fallback exists")
    return _lib.load()


def _off(t, n):
    """Device pointer of element n of tensor t."""
    return ctypes.c_void_p(t.data_ptr() + n * t.element_size())


class _MultiGather:
    """Block-batched forms of the tail's gathers: ONE upload of the concatenated candidates, one launch per block on the
    same stream (pointer offsets into the shared buffers), ONE download -- instead of a host round trip per block."""

    def candidate_features_multi(self, bs, pixels, halfs):
        sizes = [int(len(p)) for p in pixels]
        total = sum(sizes)
        empty = (np.zeros(0, np.uint32), np.zeros(0, np.uint32), np.zeros(0))
        if total == 0:
            return [empty for _ in bs]
        dev = self._device()
        pix = np.concatenate([np.asarray(p, dtype=np.uint32) for p in pixels])
        half = np.concatenate([np.asarray(h, dtype=np.int32) for h in halfs])
        d_pix = torch.from_numpy(pix.view(np.int32)).to(dev)
        d_half = torch.from_numpy(half).to(dev)
        cnt = torch.empty((2, total), dtype=torch.int32, device=dev)
        cval = torch.empty(total, dtype=torch.float64, device=dev)
        if not self._features_one_launch(bs, sizes, d_pix, d_half, total, cnt, cval):
            off = 0
            for b, m in zip(bs, sizes):
                if m:
                    self._features_launch(b, _off(d_pix, off), _off(d_half, off), m, _off(cnt, off), _off(cnt, total + off),
                                          _off(cval, off))
                off += m
        cnt_h = cnt.cpu().numpy().view(np.uint32)
        cval_h = cval.cpu().numpy()
        out, off = [], 0
        for m in sizes:
            out.append((cnt_h[0, off:off + m], cnt_h[1, off:off + m], cval_h[off:off + m]) if m else empty)
            off += m
        return out

    def diagonals_multi(self, bs, kss):
        sizes = [int(len(k)) for k in kss]
        total = sum(sizes)
        if total == 0:
            return [np.zeros((0, self.CH)) for _ in bs]
        dev = self._device()
        d_k = torch.from_numpy(np.concatenate([np.asarray(k, dtype=np.int32) for k in kss])).to(dev)
        out = torch.empty((total, self.CH), dtype=torch.float64, device=dev)
        off = 0
        for b, m in zip(bs, sizes):
            if m:
                self._diagonals_launch(b, _off(d_k, off), m, _off(out, off * self.CH))
            off += m
        host = self.engine._pinned("diags", (total, self.CH), torch.float64)
        host.copy_(out, non_blocking=True)
        torch.cuda.current_stream().synchronize()
        host = host.numpy()
        res, off = [], 0
        for m in sizes:
            res.append(host[off:off + m])
            off += m
        return res

    def diagonal_means_multi(self, bs, kss):
        """Per block the mean of the non-zero entries of its diagonals kss[i] (np.mean(dg[dg != 0]), mustache.py:816-820),
        computed on the device in NumPy's summation order -- bit-identical, and only one double per diagonal comes back."""
        sizes = [int(len(k)) for k in kss]
        total = sum(sizes)
        if total == 0:
            return [np.zeros(0) for _ in bs]
        dev = self._device()
        d_k = torch.from_numpy(np.concatenate([np.asarray(k, dtype=np.int32) for k in kss])).to(dev)
        out = torch.empty(total, dtype=torch.float64, device=dev)
        if not self._diag_means_one_launch(bs, sizes, d_k, out):
            off = 0
            for b, m in zip(bs, sizes):
                if m:
                    self._diag_means_launch(b, _off(d_k, off), m, _off(out, off))
                off += m
        host = out.cpu().numpy()
        res, off = [], 0
        for m in sizes:
            res.append(host[off:off + m])
            off += m
        return res

    def _diag_means_one_launch(self, bs, sizes, d_k, out):
        return False                        # overridden where all blocks share one source buffer

    def _features_one_launch(self, bs, sizes, d_pix, d_half, total, cnt, cval):
        return False                        # overridden where all blocks share one source buffer

    def cluster_representatives_multi(self, bs, qs, idxs, pt):
        """Clustering of the surviving candidates of several blocks (mustache.py:830-848) in ONE launch
        (mst_cluster_representatives): per block the record indices of the components' representatives, in the
        reference's label order.  qs[i]: q per record of block bs[i]; idxs[i]: ascending record indices of its candidates.
        Only records with q < pt can be a component's arg-min (o >= 1 everywhere else), so those are what is uploaded."""
        out = [[] for _ in bs]
        sel_pix, sel_q, sel_off, cand_pos, cand_off, back = [], [], [0], [], [0], []
        for b, q, idx in zip(bs, qs, idxs):
            rec = self.found[b]
            idx = np.asarray(idx, dtype=np.int64)
            below = np.nonzero(q < pt)[0]
            if len(idx) and not np.all(q[idx] < pt):
                raise ValueError("cluster_representatives_multi: a candidate with q >= pt")
            sel_pix.append(rec["pixel"][below].astype(np.uint32))
            sel_q.append(np.ascontiguousarray(q[below], dtype=np.float64))
            sel_off.append(sel_off[-1] + len(below))
            cand_pos.append(np.searchsorted(below, idx).astype(np.uint32))
            cand_off.append(cand_off[-1] + len(idx))
            back.append(below)
        total_c = cand_off[-1]
        if total_c == 0:
            return out
        dev = self._device()
        lib = self.engine.lib
        up = lambda a, dt: torch.from_numpy(np.ascontiguousarray(a).view(dt)).to(dev)
        d_pix = up(np.concatenate(sel_pix) if sel_off[-1] else np.zeros(1, np.uint32), np.int32)
        d_q = up(np.concatenate(sel_q) if sel_off[-1] else np.zeros(1), np.float64)
        d_soff = up(np.asarray(sel_off, dtype=np.uint32), np.int32)
        d_cpos = up(np.concatenate(cand_pos), np.int32)
        d_coff = up(np.asarray(cand_off, dtype=np.uint32), np.int32)
        d_rep = torch.empty(total_c, dtype=torch.int32, device=dev)
        d_cnt = torch.empty(len(bs), dtype=torch.int32, device=dev)
        ws_bytes = int(lib.mst_cluster_workspace_bytes(total_c))
        ws = torch.empty(ws_bytes, dtype=torch.uint8, device=dev)
        with torch.cuda.device(dev):
            _lib.check(lib.mst_cluster_representatives(_ptr(d_pix), _ptr(d_q), _ptr(d_soff), _ptr(d_cpos), _ptr(d_coff),
                                                       len(bs), int(self.CH), total_c, _ptr(d_rep), _ptr(d_cnt), _ptr(ws),
                                                       ws_bytes, _stream()))
        rep = d_rep.cpu().numpy().view(np.uint32)
        cnt = d_cnt.cpu().numpy().view(np.uint32)
        for i in range(len(bs)):
            r = rep[cand_off[i]:cand_off[i] + int(cnt[i])]
            out[i] = [int(v) for v in back[i][r]]
        return out

    def candidate_features(self, b, pixel, half):
        """(cnt1, cnt2, cval) for candidate pixels of block b (reference mustache.py:800-807, :824)."""
        return self.candidate_features_multi([b], [pixel], [half])[0]

    def diagonals(self, b, ks):
        """Rows = diagonals c[r, r+k] of block b, zero padded to CH (reference mustache.py:816-820)."""
        return self.diagonals_multi([b], [ks])[0].copy()      # the multi form hands out views of a reused pinned buffer


class BlockBatch(_MultiGather):
    """Results of the sigma loop for B blocks, plus the device buffers the tail needs.

    found[b] = dict(pixel uint32 [m] ascending, level uint32 [m] (1-based tested level), value float64 [m],
                    pval float64 [m])  on the host;  nz_count[b];  fit[b] = (loc[n_tested], scale[n_tested]).
    The tail's gathers are tiny, so dense blocks never leave the device.
    """

    def __init__(self, engine, c, nz, CH, B, nz_count, found, fit):
        self.engine, self.c, self.nz, self.CH, self.B = engine, c, nz, CH, B
        self.nz_count, self.found, self.fit = nz_count, found, fit

    def _device(self):
        return self.c.device

    def _features_launch(self, b, pix, half, m, cnt1, cnt2, cval):
        _lib.check(self.engine.lib.mst_candidate_features(_ptr(self.c), _ptr(self.nz), self.CH, b, pix, half, m, cnt1,
                                                          cnt2, cval, _stream()))

    def _diagonals_launch(self, b, ks, m, out):
        _lib.check(self.engine.lib.mst_gather_diagonals(_ptr(self.c), self.CH, b, ks, m, out, _stream()))

    def _diag_means_launch(self, b, ks, m, out):
        _lib.check(self.engine.lib.mst_diag_means(_ptr(self.c), self.CH, b, ks, m, out, _stream()))


class BandBatch(_MultiGather):
    """Same interface as BlockBatch for blocks that exist only as windows of the band (mst_scale_space_band): the tail's
    gathers read the band directly, no dense block is ever built."""

    def __init__(self, engine, band, n, dpx, starts, CH, nz_count, found, fit):
        self.engine, self.band, self.n, self.dpx, self.starts, self.CH = engine, band, int(n), int(dpx), list(starts), CH
        self.B = len(self.starts)
        self.nz_count, self.found, self.fit = nz_count, found, fit

    def _device(self):
        return self.band.device

    def _features_launch(self, b, pix, half, m, cnt1, cnt2, cval):
        _lib.check(self.engine.lib.mst_candidate_features_band(_ptr(self.band), self.n, self.dpx, int(self.starts[b]),
                                                               self.CH, pix, half, m, cnt1, cnt2, cval, _stream()))

    def _diagonals_launch(self, b, ks, m, out):
        _lib.check(self.engine.lib.mst_gather_diagonals_band(_ptr(self.band), self.n, self.dpx, int(self.starts[b]),
                                                             self.CH, ks, m, out, _stream()))

    def _features_one_launch(self, bs, sizes, d_pix, d_half, total, cnt, cval):
        starts = np.repeat(np.array([int(self.starts[b]) for b in bs], dtype=np.int64), sizes)
        d_s = torch.from_numpy(starts).to(self.band.device)
        _lib.check(self.engine.lib.mst_candidate_features_band_multi(_ptr(self.band), self.n, self.dpx, _ptr(d_s), self.CH,
                                                                     _ptr(d_pix), _ptr(d_half), int(total), _ptr(cnt),
                                                                     _off(cnt, total), _ptr(cval), _stream()))
        return True

    def _diag_means_one_launch(self, bs, sizes, d_k, out):
        starts = np.repeat(np.array([int(self.starts[b]) for b in bs], dtype=np.int64), sizes)
        d_s = torch.from_numpy(starts).to(self.band.device)
        _lib.check(self.engine.lib.mst_diag_means_band_multi(_ptr(self.band), self.n, self.dpx, _ptr(d_s), self.CH,
                                                             _ptr(d_k), int(starts.size), _ptr(out), _stream()))
        return True

    def _diag_means_launch(self, b, ks, m, out):
        _lib.check(self.engine.lib.mst_diag_means_band(_ptr(self.band), self.n, self.dpx, int(self.starts[b]), self.CH,
                                                       ks, m, out, _stream()))


class PairBandBatch(_MultiGather):
    """The two-sample caller's batch: blocks [0, P) are windows of sample 1's band, blocks [P, 2P) the same windows of
    sample 2's band (reference diff_mustache.py:671-674 builds the two dense blocks; here neither exists)."""

    def __init__(self, engine, bands, n, dpx, starts, CH, nz_count, found, fit):
        self.engine, self.bands, self.n, self.dpx, self.CH = engine, bands, int(n), int(dpx), CH
        self.starts = list(starts) + list(starts)
        self.P = len(starts)
        self.B = 2 * self.P
        self.nz_count, self.found, self.fit = nz_count, found, fit

    def _device(self):
        return self.bands[0].device

    def _band(self, b):
        return self.bands[0] if b < self.P else self.bands[1]

    def _diag_means_one_launch(self, bs, sizes, d_k, out):
        """one launch per run of consecutive blocks of the same sample"""
        starts = np.repeat(np.array([int(self.starts[b]) for b in bs], dtype=np.int64), sizes)
        d_s = torch.from_numpy(starts).to(self.bands[0].device)
        off = 0
        i = 0
        while i < len(bs):
            j = i
            while j < len(bs) and (bs[j] < self.P) == (bs[i] < self.P):
                j += 1
            m = int(sum(sizes[i:j]))
            if m:
                _lib.check(self.engine.lib.mst_diag_means_band_multi(_ptr(self._band(bs[i])), self.n, self.dpx, _off(d_s, off),
                                                                     self.CH, _off(d_k, off), m, _off(out, off), _stream()))
            off += m
            i = j
        return True

    def _features_one_launch(self, bs, sizes, d_pix, d_half, total, cnt, cval):
        """one launch per run of consecutive blocks of the same sample (callers list sample 1's blocks first, then sample 2's)"""
        starts = np.repeat(np.array([int(self.starts[b]) for b in bs], dtype=np.int64), sizes)
        d_s = torch.from_numpy(starts).to(self.bands[0].device)
        off = 0
        i = 0
        while i < len(bs):
            j = i
            while j < len(bs) and (bs[j] < self.P) == (bs[i] < self.P):
                j += 1
            m = int(sum(sizes[i:j]))
            if m:
                _lib.check(self.engine.lib.mst_candidate_features_band_multi(
                    _ptr(self._band(bs[i])), self.n, self.dpx, _off(d_s, off), self.CH, _off(d_pix, off), _off(d_half, off), m,
                    _off(cnt, off), _off(cnt, total + off), _off(cval, off), _stream()))
            off += m
            i = j
        return True

    def _features_launch(self, b, pix, half, m, cnt1, cnt2, cval):
        _lib.check(self.engine.lib.mst_candidate_features_band(_ptr(self._band(b)), self.n, self.dpx, int(self.starts[b]),
                                                               self.CH, pix, half, m, cnt1, cnt2, cval, _stream()))

    def _diagonals_launch(self, b, ks, m, out):
        _lib.check(self.engine.lib.mst_gather_diagonals_band(_ptr(self._band(b)), self.n, self.dpx, int(self.starts[b]),
                                                             self.CH, ks, m, out, _stream()))

    def _diag_means_launch(self, b, ks, m, out):
        _lib.check(self.engine.lib.mst_diag_means_band(_ptr(self._band(b)), self.n, self.dpx, int(self.starts[b]), self.CH,
                                                       ks, m, out, _stream()))


_GC_SETTLED = False


def settle_gc():
    """Once per process: move everything alive now (the modules of torch, numpy, this package: ~10^6 objects that never die) out
    of the cyclic collector's reach (`gc.freeze()`).  A chromosome's host tail makes ~2 * 10^4 short-lived containers (one
    4-element list per loop, as the reference returns them), so CPython ran a full collection every third chromosome and each
    one walked all of those objects: +27-38 ms on a 55-75 ms step, exactly periodic (scripts/pair_genome_jitter.py,
    scripts/file_leg_cpu.py; LABBOOK R5.8).  Results do not depend on it.
    A process-global, irreversible change, so the LIBRARY never makes it on its own: the command-line entry points
    (mustache.main, diff_mustache.main) and bench.py call this; a host application that imports mustache() / regulator() keeps
    its collector untouched unless it calls settle_gc() itself or exports MUSTACHE_GC_FREEZE=1 (then the first engine does).
    MUSTACHE_GC_FREEZE=0 leaves the collector alone everywhere."""
    global _GC_SETTLED
    if _GC_SETTLED or os.environ.get("MUSTACHE_GC_FREEZE", "1") == "0":
        return
    import gc
    gc.freeze()          # (no gc.collect() first: that full collection is the 0.1 s this is here to avoid; whatever cyclic garbage
                         # exists at this moment stays allocated, a few objects)
    _GC_SETTLED = True


class ScaleSpaceEngine:
    """Owns the level table and runs rows 2-7 of SURVEY.md section 8a on the GPU."""

    def __init__(self, octave_values=(1.6, 3.2), s=10, device=None):
        self.lib = require_gpu()
        self.share_tiles = True      # band source: compute the tiles two consecutive blocks have in common once (identical records)
        # several groups of blocks with host results: one launch in stages (mst_scale_space_band_stage) instead of a launch per
        # group; MUSTACHE_STAGED=0 keeps the launch-per-group form (the cross-check of tests/test_gpu_pipeline.py)
        self.staged_launches = os.environ.get("MUSTACHE_STAGED", "1") != "0"
        self.device = torch.device(device if device is not None else "cuda:%d" % torch.cuda.current_device())
        self.levels = LevelTable(octave_values, s)
        self._select_cap = 4096
        self._bh_lds_records = 1024     # LDS sort size of mst_bh_select_nowait (doubles when a launch reports MST_BH_RETRY)
        self._prefetch_guess = {}       # CH -> record columns mst_found_finish copies to the host speculatively
        self._buffers = {}              # small launch buffer sets kept for reuse (_carve)
        self._starts_arrays, self._ws_bytes = {}, {}
        self._side_streams = None
        self._lv_struct = self.levels.as_struct()
        self._found_cap = {}
        self._pin = {}
        self._pin_flip = 0
        if os.environ.get("MUSTACHE_GC_FREEZE") == "1":      # library use: opt-in only (settle_gc's docstring)
            settle_gc()

    # ---- host queries of the launch geometry (no GPU work) --------------------------------------------------------
    def band_tile_fraction(self, CH, dpx):
        """Share of a block's tiles launched with empty tiles skipped: those whose owned pixels can reach the tested band
        4 <= col - row <= dpx + 1 (mst_scale_space_band_tiles)."""
        total = ctypes.c_int32(0)
        m = self.lib.mst_scale_space_band_tiles(int(CH), int(dpx), ctypes.byref(self._lv_struct), ctypes.byref(total))
        if m < 0:
            _lib.check(m)
        return m / float(total.value)

    def band_items(self, starts, CH, dpx, skip_empty=False, share=True):
        """(workgroups one launch over the blocks at `starts` runs, tiles the blocks would run one by one, tiles computed once
        for two blocks) -- mst_scale_space_band_items, the work list the band-direct kernel is launched with."""
        st = (ctypes.c_int64 * len(starts))(*[int(a) for a in starts])
        tiles, shared = ctypes.c_int64(), ctypes.c_int64()
        m = self.lib.mst_scale_space_band_items(st, len(starts), int(CH), int(dpx), ctypes.byref(self._lv_struct),
                                                (1 if skip_empty else 0) | (0 if share else 4), ctypes.byref(tiles), ctypes.byref(shared))
        if m < 0:
            _lib.check(m)
        return int(m), int(tiles.value), int(shared.value)

    # ---- row 2: COO -> dense blocks ---------------------------------------------------------------------------
    def scatter_blocks(self, x, y, v, starts, CH):
        """x, y int64 / v float64 device tensors (upper-triangular COO, bin units) -> [B, CH, CH] float64."""
        B = len(starts)
        c = torch.empty((B, CH, CH), dtype=torch.float64, device=self.device)
        st = (ctypes.c_int64 * B)(*[int(s) for s in starts])
        with torch.cuda.device(self.device):
            _lib.check(self.lib.mst_scatter_blocks(_ptr(x), _ptr(y), _ptr(v), int(v.numel()), st, B, CH, _ptr(c),
                                                   _stream()))
        return c

    # ---- row 4 bring-up: one Gaussian level ----------------------------------------------------------------------
    def gauss_blur(self, img, taps):
        """img [B, H, W] float64 device tensor; taps = centre-first half kernel (radius = len-1)."""
        img = img.contiguous()
        B, H, W = img.shape
        out = torch.empty_like(img)
        tmp = torch.empty_like(img)
        arr = (ctypes.c_double * len(taps))(*[float(t) for t in taps])
        with torch.cuda.device(self.device):
            _lib.check(self.lib.mst_gauss_blur(_ptr(img), _ptr(out), _ptr(tmp), B, H, W, arr, len(taps) - 1, _stream()))
        return out

    # ---- rows 3-7 -------------------------------------------------------------------------------------------------
    def prologue(self, c, dpx, intra=True):
        B, CH, _ = c.shape
        nz = torch.empty((B, CH, CH), dtype=torch.uint8, device=self.device)
        nz_count = torch.empty(B, dtype=torch.int32, device=self.device)
        with torch.cuda.device(self.device):
            _lib.check(self.lib.mst_block_prologue(_ptr(c), _ptr(nz), _ptr(nz_count), B, CH, int(dpx),
                                                   1 if intra else 0, _stream()))
        return nz, nz_count

    def sigma_loop_band(self, band, n, dpx, starts, CH, **kw):
        """Rows 2-7 straight from the normalised band: blocks are cut, filled and masked inside the fused kernel.
        Returns what sigma_loop returns plus the per-block tested-pixel counts (device int32 tensor) as last element."""
        nz_count = torch.empty(len(starts), dtype=torch.int32, device=self.device)
        res = self.sigma_loop(None, None, nz_count, band_src=(band, int(n), int(dpx), [int(s) for s in starts], int(CH)),
                              **kw)
        return res + (nz_count,)       # (a record-capacity overflow re-ran the kernel into this same tensor)

    def sigma_loop(self, c, nz, nz_count, skip_empty=True, found_cap=None, download=True, timing=None, sort=True,
                   with_value=True, with_q=True, fma=False, band_src=None, select_below=None):
        """The fused kernel + p-values.  Returns host records (download=True) or the device buffers.
        `select_below=pt`: BH and the selection q < pt (mustache.py:778-797) run on the device and only those records
        come back, as dict(pixel, level, q) sorted by pixel -- all the tail ever looks at; the full found set stays in HBM.
        `timing`: optional list; receives a (start, end) torch.cuda.Event pair bracketing the mst_scale_space launch
        on the launch stream.  `band_src` = (band, n, dpx, starts, CH) selects the band-direct kernel (c, nz unused;
        nz_count is then an OUTPUT)."""
        st = self._ss_launch(c, nz, nz_count, skip_empty, found_cap, timing, fma, band_src)
        packed = download and select_below is None and not sort
        return self._ss_results(self._ss_finish(st, packed=packed), download, sort, with_value, with_q, select_below)

    def _ss_launch(self, c, nz, nz_count, skip_empty, found_cap, timing, fma, band_src, reuse=None, graph=False, out=None, band2=None):
        """Allocate the outputs and enqueue the fused kernel on the current stream (no synchronisation).  `reuse`: see _carve.
        `out`: the caller's own buffers instead (dict ws, stats, fit, count, found, pval -- e.g. one sample's rows of buffers
        that hold both samples of a two-sample call, so that ONE mst_found_finish serves both launches).
        `band2` = (second band, split): ONE launch over the blocks of two bands (the two samples of a two-sample call; band_src's
        starts list holds all B origins, blocks [split, B) read the second band)."""
        if band_src is not None:
            band, bn, bdpx, bstarts, CH = band_src
            B = len(bstarts)
            skey = tuple(bstarts)
            st_arr = self._starts_arrays.get(skey)          # repeated launches of the same blocks: no re-marshalling
            if st_arr is None:
                if len(self._starts_arrays) > 64:
                    self._starts_arrays.clear()
                st_arr = self._starts_arrays[skey] = (ctypes.c_int64 * B)(*bstarts)
        else:
            B, CH, _ = c.shape
        if found_cap is None:
            found_cap = self._found_cap.get(CH, max(4096, (CH * CH) // 32))
        lv = ctypes.byref(self._lv_struct)
        ws_bytes = self._ws_bytes.get((B, CH))
        if ws_bytes is None:
            ws_bytes = self._ws_bytes[(B, CH)] = int(self.lib.mst_scale_space_workspace_bytes(B, CH, lv))
        with torch.cuda.device(self.device):
            # the six launch buffers (kept between calls for small launches, see _carve)
            T = _lib.MST_MAX_TESTED
            ws, stats, fit, count, found, pval = (out[k] for k in ("ws", "stats", "fit", "count", "found", "pval")) \
                if out is not None else self._carve(
                (ws_bytes, torch.uint8, (ws_bytes,)), (B * T * 16, torch.float64, (B, T, 2)),
                (B * T * 16, torch.float64, (B, T, 2)), (B * 4, torch.int32, (B,)),
                (B * found_cap * 16, torch.int64, (B, found_cap, 2)),            # 16-byte records
                (B * found_cap * 8, torch.float64, (B, found_cap)), reuse=None if reuse is None else ("launch", reuse))
            ev = None
            if timing is not None:
                ev = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
                ev[0].record()
            # MST_FLAG_NO_SHARE (4): every tile once per block; default: tiles inside two consecutive blocks computed once
            # MST_FLAG_GRAPH (8): a launch that repeats with identical arguments is replayed as one hipGraph
            flags = (1 if skip_empty else 0) | (2 if fma else 0) | (0 if self.share_tiles else 4) | (8 if graph else 0)
            if band2 is not None:
                # two-sample launch: blocks [split, B) are windows of the second band (mst_scale_space_band_pair)
                _lib.check(self.lib.mst_scale_space_band_pair(_ptr(band), _ptr(band2[0]), int(band2[1]), bn, bdpx, st_arr, B, CH, lv,
                                                              _ptr(found), found_cap, _ptr(count), _ptr(stats), _ptr(nz_count),
                                                              flags, _ptr(ws), ws_bytes, _stream()))
            elif band_src is not None:
                _lib.check(self.lib.mst_scale_space_band(_ptr(band), bn, bdpx, st_arr, B, CH, lv, _ptr(found),
                                                         found_cap, _ptr(count), _ptr(stats), _ptr(nz_count), flags,
                                                         _ptr(ws), ws_bytes, _stream()))
            else:
                _lib.check(self.lib.mst_scale_space(_ptr(c), _ptr(nz), B, CH, lv, _ptr(found), found_cap,
                                                    _ptr(count), _ptr(stats), flags, _ptr(ws), ws_bytes, _stream()))
            if ev is not None:
                ev[1].record()
        return dict(args=(c, nz, nz_count, skip_empty, timing, fma, band_src), B=B, CH=CH, found_cap=found_cap, ws=ws,
                    stats=stats, fit=fit, count=count, found=found, pval=pval, ev=ev, reuse=reuse, graph=graph, band2=band2)

    def _ss_launch_staged(self, groups, band_src_all, skip_empty, found_cap, timing, fma):
        """ONE fused launch over the blocks of all `groups` (consecutive lists of block origins), enqueued on the current stream
        in one stage per group (mst_scale_space_band_stage): the work list is the whole launch's, so tiles shared by the last
        block of a group and the first block of the next are still computed once -- separate launches per group recompute them
        (0.4 ms per cut on 4000 x 4000 blocks) -- and after stage i the blocks of groups 0 .. i are final.  Returns one state
        per group in the form _ss_finish takes: views of the launch's buffers for the group's blocks, with `done` = the event
        behind the group's stage."""
        band, bn, bdpx, CH = band_src_all
        starts = [int(v) for g in groups for v in g]
        B = len(starts)
        cuts, acc = [], 0
        for g in groups[:-1]:
            acc += len(g)
            cuts.append(acc)
        st_arr = (ctypes.c_int64 * B)(*starts)
        cut_arr = (ctypes.c_int32 * max(1, len(cuts)))(*cuts)
        if found_cap is None:
            found_cap = self._found_cap.get(CH, max(4096, (CH * CH) // 32))
        lv = ctypes.byref(self._lv_struct)
        ws_bytes = self._ws_bytes.get((B, CH))
        if ws_bytes is None:
            ws_bytes = self._ws_bytes[(B, CH)] = int(self.lib.mst_scale_space_workspace_bytes(B, CH, lv))
        T = _lib.MST_MAX_TESTED
        flags = (1 if skip_empty else 0) | (2 if fma else 0) | (0 if self.share_tiles else 4)
        out = []
        with torch.cuda.device(self.device):
            ws, stats, fit, count, found, pval, nzc = self._carve(
                (ws_bytes, torch.uint8, (ws_bytes,)), (B * T * 16, torch.float64, (B, T, 2)),
                (B * T * 16, torch.float64, (B, T, 2)), (B * 4, torch.int32, (B,)),
                (B * found_cap * 16, torch.int64, (B, found_cap, 2)), (B * found_cap * 8, torch.float64, (B, found_cap)),
                (B * 4, torch.int32, (B,)), reuse=("staged", 0))
            cur = torch.cuda.current_stream(self.device)
            b0 = 0
            for gi, g in enumerate(groups):
                ev = None
                if timing is not None:
                    ev = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
                    ev[0].record()
                _lib.check(self.lib.mst_scale_space_band_stage(_ptr(band), bn, bdpx, st_arr, B, CH, lv, _ptr(found), found_cap,
                                                               _ptr(count), _ptr(stats), _ptr(nzc), flags, _ptr(ws), ws_bytes,
                                                               cut_arr, len(cuts), gi, _stream()))
                if ev is not None:
                    ev[1].record()
                b1 = b0 + len(g)
                sub_src = (band, bn, bdpx, [int(v) for v in g], CH)
                out.append(dict(args=(None, None, nzc[b0:b1], skip_empty, timing, fma, sub_src), B=b1 - b0, CH=CH,
                                found_cap=found_cap, ws=ws, stats=stats[b0:b1], fit=fit[b0:b1], count=count[b0:b1],
                                found=found[b0:b1], pval=pval[b0:b1], ev=ev, reuse=1 + gi % 2, graph=False, band2=None,
                                done=cur.record_event(), staged=True))
                b0 = b1
        return out

    def _carve(self, *parts, reuse=None):
        """Device buffers for one launch: parts = (bytes, dtype, shape).  `reuse` (a hashable key, or None): SMALL sets (< 256 MB)
        are kept and handed out again for the same key -- a launch of six 2000 x 2000 blocks is 1.75 ms of kernel, and a dozen
        allocator calls per step are 2 % of it; callers pass a key only when the buffers do not outlive the call (the results
        are host copies) and alternate the key's slot between launches in flight."""
        total = sum(int(p[0]) for p in parts)
        key = None
        if reuse is not None and total < (256 << 20):
            key = (reuse,) + tuple((int(p[0]), p[1]) for p in parts)
            hit = self._buffers.get(key)
            if hit is not None:
                return hit
        out = tuple(torch.empty(shape, dtype=dt, device=self.device) for _, dt, shape in parts)
        if key is not None:
            if len(self._buffers) > 16:
                self._buffers.clear()
            self._buffers[key] = out
        return out

    def _summary_pin(self, B):
        """Page-locked landing area of mst_found_finish's one round trip (flags, counts, tested-pixel counts, fits)."""
        need = int(self.lib.mst_found_summary_bytes(B))
        buf = self._pin.get(("summary", B))
        if buf is None or buf.numel() < need:
            buf = self._pin[("summary", B)] = torch.empty(need, dtype=torch.uint8, pin_memory=True)
        return buf

    @staticmethod
    def _parse_summary(summ, B):
        """mst_found_finish's summary block (include/mustache_hip.h) -> (found counts, tested-pixel counts, fits) host arrays"""
        h = summ.numpy()
        cw = 8 * ((B + 1) // 2)
        return (h[16:16 + 4 * B].view(np.uint32).astype(np.int64), h[16 + cw:16 + cw + 4 * B].view(np.uint32).astype(np.int64),
                h[16 + 2 * cw:16 + 2 * cw + 16 * _lib.MST_MAX_TESTED * B].view(np.float64).reshape(B, _lib.MST_MAX_TESTED, 2).copy())

    def _ss_finish(self, st, packed=False, relaunch=True):
        """p-values of the found pixels (ONE synchronisation of the launch stream: mst_found_finish brings the overflow flag, the
        record counts, the tested-pixel counts and the fits back in the same round trip); a record-capacity overflow re-runs the
        kernel.  packed=True additionally leaves the records' pixel indices / levels as narrow device arrays (st["pix"], st["lvl"])
        for a caller that downloads whole found sets."""
        nt = self.levels.n_tested
        with torch.cuda.device(self.device):
            while True:
                B, cap = st["B"], st["found_cap"]
                # whole-found-set downloads: the first `pitch` records of every block also come out as narrow, densely
                # pitched arrays and are copied to the host inside the same call; pitch = the largest count the last launch of
                # this block size saw + 5 % (the first launch of a size has no guess and takes the two-step download)
                pitch = min(cap, self._prefetch_guess.get(st["CH"], 0)) if packed else 0
                summ = self._summary_pin(B)
                dev3 = host3 = None
                if pitch > 0:
                    scratch, d_pix, d_lvl, d_pv = self._carve((summ.numel(), torch.uint8, (summ.numel(),)),
                                                              (B * pitch * 4, torch.int32, (B, pitch)),
                                                              (B * pitch, torch.uint8, (B, pitch)),
                                                              (B * pitch * 8, torch.float64, (B, pitch)),
                                                              reuse=None if st.get("reuse") is None else ("finish", st["reuse"]))
                    dev3 = (d_pix, d_lvl, d_pv)
                    self._pin_flip ^= 1
                    host3 = (self._pinned("pix", (B, pitch), torch.int32), self._pinned("lvl", (B, pitch), torch.uint8),
                             self._pinned("pv", (B, pitch), torch.float64))
                else:
                    scratch = torch.empty(summ.numel(), dtype=torch.uint8, device=self.device)
                vp = lambda t: ctypes.c_void_p(t.data_ptr()) if t is not None else None
                try:
                    _lib.check(self.lib.mst_found_finish(_ptr(st["found"]), cap, _ptr(st["count"]), _ptr(st["args"][2]),
                                                         _ptr(st["stats"]), B, nt, _ptr(st["pval"]), _ptr(st["fit"]), pitch,
                                                         *(vp(t) for t in (dev3 or (None, None, None))), _ptr(scratch),
                                                         vp(summ), *(vp(t) for t in (host3 or (None, None, None))),
                                                         8 if st.get("graph") else 0, _stream()))
                    break
                except _lib.MstOverflow:
                    if not relaunch:            # the caller owns the launches (several of them behind this one finish)
                        raise
                    c, nz, nz_count, skip_empty, timing, fma, band_src = st["args"]
                    cap = st["found_cap"] * 4   # rare: a block with an unusually dense set of local maxima
                    self._found_cap[st["CH"]] = cap
                    st = self._ss_launch(c, nz, nz_count, skip_empty, cap, timing, fma, band_src, reuse=st.get("reuse"),
                                         band2=st.get("band2"))
        if st["ev"] is not None:
            st["args"][4].append(st["ev"])      # mst_found_finish synchronised the stream: the events are complete
        st["count_h"], st["nz_h"], st["fit_h"] = self._parse_summary(summ, B)
        st["prefetched"] = None
        if packed:
            mx = int(st["count_h"].max(initial=0))
            if host3 is not None:
                # the records are on the host already when the guess held; False = tried (the staging set is already flipped)
                st["prefetched"] = host3 if mx <= pitch else False
            # next guess: 10 % above this launch's largest count, but never much below the last guess -- the launches of a run
            # differ (a genome's chromosomes, a chromosome's ends), and a guess that fails costs a second download
            self._prefetch_guess[st["CH"]] = max(mx + mx // 10 + 64, int(0.995 * self._prefetch_guess.get(st["CH"], 0)))
        return st

    def _ss_results(self, st, download, sort, with_value, with_q, select_below):
        found, pval, count, fit, found_cap = st["found"], st["pval"], st["count"], st["fit"], st["found_cap"]
        nt = self.levels.n_tested
        if not download:
            return found, pval, count, fit, found_cap
        host = (st.get("count_h"), st.get("fit_h"))
        if select_below is not None:
            return self._download_selected(found, pval, count, fit, nt, found_cap, float(select_below), host=host)
        extra = {"q": self.fdr(pval, count, found_cap)} if with_q else None
        return self._download(found, pval, count, fit, nt, sort=sort, with_value=with_value, extra=extra, host=host,
                              prefetched=st.get("prefetched"))

    def sigma_loop_band_overlapped(self, band, n, dpx, groups, CH, skip_empty=True, timing=None, fma=False, download=True,
                                   sort=True, with_value=True, with_q=True, select_below=None):
        """sigma_loop_band over several groups of blocks with copy/compute overlap: the groups' fused kernels run back
        to back on two alternating side streams, and the p-values / BH / selection / download of group i run while the
        kernel of group i + 1 is executing.  Yields, per group, what sigma_loop_band returns.
        LIFETIME of the host results: the record arrays of a group are views of two alternating page-locked staging sets
        (_pinned) -- valid until the group AFTER NEXT is fetched.  Consume each group as it is yielded (the pipeline's tail does)
        and copy what has to outlive that; `list(...)` over three or more groups leaves the first group's views showing the
        third group's bytes (scripts/staged_stress.py checks the path that way)."""
        if len(groups) == 1:
            # nothing to overlap: run on the caller's stream, without the side streams' events (a small launch -- six blocks of
            # 2000 x 2000 are 1.75 ms of kernel -- pays for every host-side call)
            # With host results (download) the launch buffers are kept between calls, so a caller that repeats the launch -- a
            # benchmark step, the same chromosome again -- presents identical arguments and the library replays it as ONE
            # hipGraph launch (MST_FLAG_GRAPH).  A graph cannot be captured on the legacy default stream: run on the first
            # side stream then.
            starts = groups[0]
            cur = torch.cuda.current_stream(self.device)
            side = None
            if download and cur.cuda_stream == 0:
                side = device_streams(self.device)[0]
                side.wait_stream(cur)                       # the band was produced on the caller's stream
            with torch.cuda.stream(side if side is not None else cur):
                if download:
                    nzc, = self._carve((len(starts) * 4, torch.int32, (len(starts),)), reuse=("nzc", 0))
                else:
                    nzc = torch.empty(len(starts), dtype=torch.int32, device=self.device)
                st = self._ss_launch(None, None, nzc, skip_empty, None, timing, fma,
                                     (band, int(n), int(dpx), [int(v) for v in starts], int(CH)),
                                     reuse=0 if download else None, graph=download)
                st2 = self._ss_finish(st, packed=download and select_below is None and not sort)
                res = self._ss_results(st2, download, sort, with_value, with_q, select_below)
            if side is not None:
                cur.wait_stream(side)           # explicit: the caller's stream is ordered behind the side stream whatever the finish did
            yield res + ((torch.from_numpy(st2["nz_h"].astype(np.uint32).view(np.int32)) if download else st2["args"][2]),)
            return
        cur = torch.cuda.current_stream(self.device)
        ready = cur.record_event()              # the band was produced on the caller's stream
        if self._side_streams is None:
            self._side_streams = list(device_streams(self.device)[:2])
        if self.staged_launches and download:
            # ONE launch in one stage per group on the first side stream; the p-values / selection / download of group i run on
            # the second one behind stage i's event, while stage i + 1 executes.  Same records as separate launches.
            ks, fs = self._side_streams
            ks.wait_event(ready)
            # (a launch's record lists, p-values and per-tile statistics are allocated for all of its blocks at once: 24 B x
            #  CH^2 / 32 per block and ~2.6 MB of statistics per 4000 x 4000 block -- 1.9 GB for chr1 at 1 kb.  Whole genomes at fine
            #  resolutions go through several staged launches of at most MUSTACHE_STAGED_GB, default 32, of such buffers each.)
            per_block = 24 * self._found_cap.get(int(CH), max(4096, (int(CH) * int(CH)) // 32)) + \
                (int(CH) // 30 + 2) * (int(CH) // 62 + 2) * (20 + 16 * self.levels.n_tested)
            budget = float(os.environ.get("MUSTACHE_STAGED_GB", "32")) * (1 << 30)
            chunks, acc = [[]], 0
            for g in groups:
                if chunks[-1] and (acc + len(g)) * per_block > budget:
                    chunks.append([])
                    acc = 0
                chunks[-1].append(g)
                acc += len(g)
            for chunk in chunks:
                cap = None
                done_groups = 0
                while True:
                    with torch.cuda.stream(ks), _lib.stage("scale-space launch"):
                        sts = self._ss_launch_staged(chunk, (band, int(n), int(dpx), int(CH)), skip_empty, cap, timing, fma)
                    try:
                        for gi in range(done_groups, len(chunk)):
                            st = sts[gi]
                            fs.wait_event(st["done"])
                            with torch.cuda.stream(fs), _lib.stage("scale-space finish"):
                                st2 = self._ss_finish(st, packed=select_below is None and not sort, relaunch=False)
                                res = self._ss_results(st2, download, sort, with_value, with_q, select_below)
                            yield res + (torch.from_numpy(st2["nz_h"].astype(np.uint32).view(np.int32)),)
                            done_groups = gi + 1
                        break
                    except _lib.MstOverflow:
                        # rare (a block with an unusually dense set of local maxima): the whole launch again with four times the
                        # record capacity; the groups already handed out stay as they are (their records were complete)
                        ks.synchronize()
                        cap = self._found_cap[int(CH)] = sts[0]["found_cap"] * 4
            cur.wait_stream(ks)
            cur.wait_stream(fs)
            return

        def finish(st):
            with torch.cuda.stream(st["stream"]):
                st2 = self._ss_finish(st, packed=download and select_below is None and not sort)
                res = self._ss_results(st2, download, sort, with_value, with_q, select_below)
            # the tested-pixel counts came back with the finish's round trip: hand them out as a HOST tensor (callers' .cpu() is
            # then free) unless the caller asked for device buffers
            nzc = torch.from_numpy(st2["nz_h"].astype(np.uint32).view(np.int32)) if download else st2["args"][2]
            return res + (nzc,)

        pending = None
        for gi, starts in enumerate(groups):
            s = self._side_streams[gi % 2]
            s.wait_event(ready)
            if pending is not None:
                s.wait_event(pending["kernel_done"])        # one fused kernel at a time
            with torch.cuda.stream(s):
                nzc = torch.empty(len(starts), dtype=torch.int32, device=self.device)
                st = self._ss_launch(None, None, nzc, skip_empty, None, timing, fma,
                                     (band, int(n), int(dpx), [int(v) for v in starts], int(CH)),
                                     reuse=(1 + gi % 2) if download else None)     # two launches in flight: two buffer sets
                st["kernel_done"] = s.record_event()
            st["stream"], st["nzc"] = s, nzc
            if pending is not None:
                yield finish(pending)
            pending = st
        if pending is not None:
            yield finish(pending)
        cur.wait_stream(self._side_streams[0])
        cur.wait_stream(self._side_streams[1])

    def fdr(self, pval, count, found_cap):
        """Benjamini-Hochberg q-values per block on the device (reference mustache.py:778); same record order as pval."""
        B = pval.shape[0]
        q = torch.empty_like(pval)
        ws_bytes = int(self.lib.mst_bh_workspace_bytes(B, found_cap))
        if ws_bytes == 0:
            raise ValueError("too many found records for one BH launch (B * capacity must fit in int32)")
        with torch.cuda.device(self.device):
            ws = torch.empty(ws_bytes, dtype=torch.uint8, device=self.device)
            _lib.check(self.lib.mst_bh_fdr(_ptr(pval), _ptr(count), B, found_cap, _ptr(q), _ptr(ws), ws_bytes, _stream()))
        return q

    def _download_selected(self, found, pval, count, fit, nt, found_cap, pt, full_sort=False, pair=None, host=None):
        """BH-FDR and the selection q < pt on the device; only the selected records come back.  Default: mst_bh_select
        (sorts only the records that can be selected -- same selected set, bit-identical q); full_sort=True runs mst_bh_fdr
        over all records and then mst_select_below (kept as the cross-check).  pair = (ppair [2P, found_cap], P), two-sample
        path: the selected records also carry `pair`, `value` and `v_other` (mst_pair_gather)."""
        B = count.shape[0]
        cap = self._select_cap
        ws_bytes = int(self.lib.mst_bh_workspace_bytes(B, found_cap))
        if ws_bytes == 0:
            raise ValueError("too many found records for one BH launch (B * capacity must fit in int32)")
        with torch.cuda.device(self.device):
            q = self.fdr(pval, count, found_cap) if full_sort else None
            ws = None if full_sort else torch.empty(ws_bytes, dtype=torch.uint8, device=self.device)
            while True:
                pix = torch.empty((B, cap), dtype=torch.int32, device=self.device)
                lvl = torch.empty((B, cap), dtype=torch.int32, device=self.device)
                qs = torch.empty((B, cap), dtype=torch.float64, device=self.device)
                n_sel = torch.empty(B, dtype=torch.int32, device=self.device)
                if full_sort:
                    _lib.check(self.lib.mst_select_below(_ptr(found), _ptr(q), _ptr(count), B, found_cap, pt, cap,
                                                         _ptr(pix), _ptr(lvl), _ptr(qs), _ptr(n_sel), _stream()))
                else:
                    # without the library's own look at the subset sizes (mst_bh_select_nowait: the in-LDS sort for every
                    # block): the counts are read right below anyway, and a block whose subset does not fit it says so there
                    idx = torch.empty((B, cap), dtype=torch.int32, device=self.device) if pair is not None else None
                    args = (_ptr(found), _ptr(pval), _ptr(count), B, found_cap, pt, cap, _ptr(pix), _ptr(lvl), _ptr(qs),
                            None if idx is None else _ptr(idx), _ptr(n_sel), _ptr(ws), ws_bytes, _stream())
                    # the sort's LDS request: 1024 records (14 KB) until a launch of this engine needed more -- this kernel runs
                    # next to the following group's fused kernel, which leaves little LDS free
                    _lib.check(self.lib.mst_bh_select_nowait(*(args[:12] + (self._bh_lds_records,) + args[12:])))
                n_h = n_sel.cpu().numpy().view(np.uint32).astype(np.int64)
                if not full_sort and (n_h == 0xFFFFFFFF).any():       # MST_BH_RETRY: this launch through the synchronising form
                    self._bh_lds_records = min(4096, self._bh_lds_records * 2)
                    if idx is not None:
                        _lib.check(self.lib.mst_bh_select_records(*args))
                    else:
                        _lib.check(self.lib.mst_bh_select(*(args[:10] + args[11:])))
                    n_h = n_sel.cpu().numpy().view(np.uint32).astype(np.int64)
                if n_h.max(initial=0) <= cap:
                    break
                cap = self._select_cap = int(n_h.max()) * 2         # rare: re-run with room for every selected record
            mx = int(n_h.max(initial=0))
            extra_h = {}
            self._pin_flip ^= 1
            if pair is not None:
                ppair, P = pair
                g = torch.empty((3, B, cap), dtype=torch.float64, device=self.device)
                _lib.check(self.lib.mst_pair_gather(_ptr(found), found_cap, _ptr(count), _ptr(ppair), int(P), _ptr(idx),
                                                    _ptr(pix), _ptr(n_sel), cap, mx, _ptr(g[0]), _ptr(g[1]), _ptr(g[2]),
                                                    _stream()))
                g_p = self._pinned("sel_pair", (3, B, max(mx, 1)), torch.float64)       # lands with the batch below
                g_p.copy_(g[:, :, :max(mx, 1)], non_blocking=True)
                g_h = g_p.numpy()
                extra_h = {"pair": g_h[0], "value": g_h[1], "v_other": g_h[2]}
            # the three record arrays in ONE round trip (page-locked staging, one synchronisation); the fits came back with
            # mst_found_finish already when the caller has them
            w = max(mx, 1)
            pix_p = self._pinned("sel_pix", (B, w), torch.int32)
            lvl_p = self._pinned("sel_lvl", (B, w), torch.int32)
            q_p = self._pinned("sel_q", (B, w), torch.float64)
            pix_p.copy_(pix[:, :w], non_blocking=True)
            lvl_p.copy_(lvl[:, :w], non_blocking=True)
            q_p.copy_(qs[:, :w], non_blocking=True)
            fit_p = None
            if host is None or host[1] is None:
                fit_p = self._pinned("sel_fit", tuple(fit.shape), torch.float64)
                fit_p.copy_(fit, non_blocking=True)
            torch.cuda.current_stream().synchronize()
            pix_h = pix_p.numpy().view(np.uint32)
            lvl_h = lvl_p.numpy().view(np.uint32)
            q_h = q_p.numpy()
            fit_h = fit_p.numpy() if fit_p is not None else host[1]
        out, fits = [], []
        for b in range(B):
            m = int(n_h[b])
            order = np.argsort(pix_h[b, :m], kind="stable")        # the kernel appends in arbitrary order; pixels are unique
            rec = {"pixel": pix_h[b, :m][order], "level": lvl_h[b, :m][order], "q": q_h[b, :m][order]}
            for name, arr in extra_h.items():
                rec[name] = arr[b, :m][order]
            out.append(rec)
            fits.append((fit_h[b, :nt, 0].copy(), fit_h[b, :nt, 1].copy()))
        return out, fits

    def _pinned(self, key, shape, dtype):
        """Page-locked host staging buffers (D2H at PCIe rate).  Two sets alternate, so the arrays handed out by one
        download stay valid until the second-next download -- long enough for the per-batch tail that consumes them."""
        need = int(np.prod(shape))
        slot = (key, self._pin_flip)
        buf = self._pin.get(slot)
        if buf is None or buf.numel() < need or buf.dtype != dtype:
            buf = self._pin[slot] = torch.empty(max(need, 1), dtype=dtype, pin_memory=True)
        return buf[:need].view(*shape)

    def _download(self, found, pval, count, fit, nt, sort=True, extra=None, with_value=True, host=None, prefetched=None):
        """Found records -> host.  The kernel appends records per workgroup, so their order inside a block is
        arbitrary; with sort=True they are ordered by pixel index on the device first (row-major = the reference's nz
        order, which the tail's look-ups rely on).  The returned arrays are views into pinned staging memory (see
        _pinned); with_value=False leaves the winning DoG values on the device (only the two-sample path needs them)."""
        if prefetched and not sort and not extra and not with_value and host is not None:
            # mst_found_finish already copied the records (its guess of the largest count held): nothing left to fetch
            cnt, fit_h = host
            pix_n = prefetched[0].numpy().view(np.uint32)
            lvl_n, pv_n = prefetched[1].numpy(), prefetched[2].numpy()
            out, fits = [], []
            for b in range(len(cnt)):
                m = int(cnt[b])
                out.append(dict(pixel=pix_n[b, :m], level=lvl_n[b, :m], pval=pv_n[b, :m]))
                fits.append((fit_h[b, :nt, 0].copy(), fit_h[b, :nt, 1].copy()))
            return out, fits
        if prefetched is None:
            self._pin_flip ^= 1
        if host is not None and host[0] is not None:       # counts and fits came back with mst_found_finish's round trip
            cnt, fit_h = host
            cnt_d = count.to(torch.int64) if sort else None
        else:
            cnt_d = count.to(torch.int64)
            cnt = cnt_d.cpu().numpy()
            fit_h = fit.cpu().numpy()
        B = len(cnt)
        mx = int(cnt.max()) if B else 0
        out, fits = [], []
        extra_h = {}
        if mx > 0:
            rec = found[:, :mx]
            word = rec[..., 0]
            pv = pval[:, :mx]
            if sort:
                pix = word & 0xFFFFFFFF
                valid = torch.arange(mx, device=found.device)[None, :] < cnt_d[:, None]
                order = torch.argsort(torch.where(valid, pix, torch.full_like(pix, 1 << 40)), dim=1)
                rec = torch.gather(rec, 1, order[..., None].expand(-1, -1, 2))
                word = rec[..., 0]
                pv = torch.gather(pv, 1, order)
            for name, t in (extra or {}).items():      # further per-record float64 arrays, same order as the records
                t = t[:, :mx]
                extra_h[name] = (torch.gather(t, 1, order) if sort else t).cpu().numpy()
            pix_h = self._pinned("pix", (B, mx), torch.int32)
            lvl_h = self._pinned("lvl", (B, mx), torch.uint8)
            pv_h = self._pinned("pv", (B, mx), torch.float64)
            pix_h.copy_((word & 0xFFFFFFFF).to(torch.int32), non_blocking=True)
            lvl_h.copy_((word >> 32).to(torch.uint8), non_blocking=True)
            pv_h.copy_(pv, non_blocking=True)
            if with_value:
                val_h = self._pinned("val", (B, mx), torch.int64)
                val_h.copy_(rec[..., 1], non_blocking=True)
            torch.cuda.current_stream().synchronize()
            pix_n = pix_h.numpy().view(np.uint32)
            lvl_n = lvl_h.numpy()
            pv_n = pv_h.numpy()
            val_n = val_h.numpy().view(np.float64) if with_value else None
        for b in range(B):
            m = int(cnt[b])
            if m:
                d = dict(pixel=pix_n[b, :m], level=lvl_n[b, :m], pval=pv_n[b, :m])
                if with_value:
                    d["value"] = val_n[b, :m]
                for name, arr in extra_h.items():
                    d[name] = arr[b, :m]
            else:
                d = dict(pixel=np.zeros(0, np.uint32), level=np.zeros(0, np.uint8), pval=np.zeros(0))
                if with_value:
                    d["value"] = np.zeros(0)
                for name in (extra or {}):
                    d[name] = np.zeros(0)
            out.append(d)
            fits.append((fit_h[b, :nt, 0].copy(), fit_h[b, :nt, 1].copy()))
        return out, fits

    # ---- two-sample additions (reference diff_mustache.py:262-276, :371-385) --------------------------------------------
    def pair_pvalues(self, c, nz, found, found_cap, count):
        """c / nz: [2P, CH, CH] filled blocks and masks, sample 1 in [0, P), sample 2 in [P, 2P); found / count: the
        device records of the 2P-block sigma loop.  Returns ppair [2P, found_cap] (device)."""
        P2, CH, _ = c.shape
        P = P2 // 2
        lt = self.levels
        n_oct, lpo, tpo = len(lt.octave_values), lt.levels_per_octave, lt.s - 1
        dev = self.device
        with torch.cuda.device(dev):
            cd = torch.empty((P, CH, CH), dtype=torch.float64, device=dev)
            nzb = torch.empty((P, CH, CH), dtype=torch.uint8, device=dev)
            nzbc = torch.empty(P, dtype=torch.int32, device=dev)
            _lib.check(self.lib.mst_diff_image(_ptr(c[:P]), _ptr(c[P:]), _ptr(nz[:P]), _ptr(nz[P:]), P, CH, _ptr(cd),
                                               _ptr(nzb), _ptr(nzbc), _stream()))
            g2 = torch.empty((n_oct, P, CH, CH), dtype=torch.float64, device=dev)
            g3 = torch.empty((n_oct, P, CH, CH), dtype=torch.float64, device=dev)
            fit = torch.empty((n_oct, P, 2), dtype=torch.float64, device=dev)
            ws = torch.empty(2048 * P, dtype=torch.uint8, device=dev)
            for o in range(n_oct):
                # the reference's Lc of the difference image: G(sigma_2) - G(sigma_3) of the octave (diff_mustache.py:315-336)
                g2[o] = self.gauss_blur(cd, lt.taps[o * lpo + 1])
                g3[o] = self.gauss_blur(cd, lt.taps[o * lpo + 2])
                _lib.check(self.lib.mst_masked_normfit(_ptr(g2[o]), _ptr(g3[o]), _ptr(nzb), _ptr(nzbc), P, CH * CH,
                                                       _ptr(fit[o]), _ptr(ws), ws.numel(), _stream()))
            ppair = torch.empty((P2, found_cap), dtype=torch.float64, device=dev)
            for off in (0, P):
                _lib.check(self.lib.mst_pair_pvalues(_ptr(found), found_cap, _ptr(count), _ptr(g2), _ptr(g3), _ptr(fit),
                                                     P, CH, n_oct, tpo, off, _ptr(ppair), _stream()))
        return ppair, fit

    def run_band_pairs(self, bands, n, dpx, starts, CH, skip_empty=True, select_below=None):
        """Both samples' sigma loops straight from their bands + the pair p-values: PairBandBatch over 2P blocks whose
        records carry `pair` and `q`.  select_below = pt (what the per-chromosome driver passes): BH, the selection q < pt and
        the differential test's look-ups happen on the device and only the selected records come back, each with `pair`,
        `value` and `v_other` (the partner sample's winning value at that pixel, NaN if it did not find it); without it the
        whole found sets are downloaded, sorted by pixel (the cross-check form)."""
        P = len(starts)
        lt = self.levels
        n_oct, tpo = len(lt.octave_values), lt.s - 1
        lv = ctypes.byref(self._lv_struct)
        T, nt = _lib.MST_MAX_TESTED, lt.n_tested
        starts_i = [int(v) for v in starts]
        st_arr = (ctypes.c_int64 * P)(*starts_i)
        cap = self._found_cap.get(CH, max(4096, (CH * CH) // 32))
        ws_bytes = self._ws_bytes.get((2 * P, CH))
        if ws_bytes is None:
            ws_bytes = self._ws_bytes[(2 * P, CH)] = int(self.lib.mst_scale_space_workspace_bytes(2 * P, CH, lv))
        # A SMALL call (its buffers are kept between calls, _carve) repeats with identical arguments when the caller repeats it: the
        # two fused launches are then replayed as hipGraphs (MST_FLAG_GRAPH: uploads, counter zeroing, kernel, reduction in one
        # launch each, no dispatch gaps -- ~40 us of idle device before each kernel otherwise).  A graph cannot be captured on the
        # legacy default stream: such a call runs on the first side stream.
        small = 2 * P * cap * 24 + ws_bytes < (200 << 20)
        cur = torch.cuda.current_stream(self.device)
        side = None
        if small and cur.cuda_stream == 0:
            side = device_streams(self.device)[0]
            side.wait_stream(cur)                           # the bands were produced on the caller's stream
        with torch.cuda.device(self.device), torch.cuda.stream(side if side is not None else cur):
            dog = None
            while True:
                # ONE set of record buffers for both samples (sample 1 in rows [0, P), sample 2 in [P, 2P)): the two fused
                # launches write their halves, one mst_found_finish (one synchronisation) serves both, and nothing has to be
                # concatenated afterwards.  Small sets are kept between calls (_carve).
                found, pval, count, stats, fit, nzc, ws = self._carve(
                    (2 * P * cap * 16, torch.int64, (2 * P, cap, 2)), (2 * P * cap * 8, torch.float64, (2 * P, cap)),
                    (2 * P * 4, torch.int32, (2 * P,)), (2 * P * T * 16, torch.float64, (2 * P, T, 2)),
                    (2 * P * T * 16, torch.float64, (2 * P, T, 2)), (2 * P * 4, torch.int32, (2 * P,)),
                    (ws_bytes, torch.uint8, (ws_bytes,)), reuse=("pairs", 0))
                # ... filled by ONE fused launch over the 2P blocks (mst_scale_space_band_pair: sample 2's blocks read its own band)
                self._ss_launch(None, None, nzc, skip_empty, cap, None, False, (bands[0], int(n), int(dpx), starts_i + starts_i, int(CH)),
                                out=dict(ws=ws, stats=stats, fit=fit, count=count, found=found, pval=pval), graph=small,
                                band2=(bands[1], P))
                if dog is None:
                    # the difference kernel needs the bands only: queued behind the sigma loops, before anything is waited for
                    dog = torch.empty((n_oct, P, CH, CH), dtype=torch.float64, device=self.device)
                    nfit = torch.empty((n_oct, P, 2), dtype=torch.float64, device=self.device)
                    mcount = torch.empty(P, dtype=torch.int32, device=self.device)
                    dws_bytes = int(self.lib.mst_diff_dog_workspace_bytes(P, CH, lv))
                    dws = torch.empty(dws_bytes, dtype=torch.uint8, device=self.device)
                    _lib.check(self.lib.mst_diff_dog_band(_ptr(bands[0]), _ptr(bands[1]), int(n), int(dpx), st_arr, P, CH, lv,
                                                          _ptr(dog), _ptr(nfit), _ptr(mcount), _ptr(dws), dws_bytes, _stream()))
                st = dict(args=(None, None, nzc, skip_empty, None, False, None), B=2 * P, CH=CH, found_cap=cap, stats=stats,
                          fit=fit, count=count, found=found, pval=pval, ev=None, reuse=None, graph=False)
                try:
                    if select_below is not None:
                        # the per-chromosome driver's form: everything behind the kernels is queued at once and waited for ONCE
                        done = self._pairs_one_wait(st, P, dog, nfit, n_oct, tpo, float(select_below))
                        if done is not None:
                            recs, fits, nz_h, norm_fit = done
                            batch = PairBandBatch(self, bands, n, dpx, starts, CH, nz_h, recs, fits)
                            batch.norm_fit = norm_fit
                            return batch
                    st = self._ss_finish(st, relaunch=False)
                    break
                except _lib.MstOverflow:        # rare: a block with an unusually dense set of local maxima -- both samples again
                    cap = cap * 4
                    self._found_cap[CH] = cap
            ppair = torch.empty((2 * P, cap), dtype=torch.float64, device=self.device)
            for off in (0, P):
                _lib.check(self.lib.mst_pair_pvalues_dog(_ptr(found), cap, _ptr(count), _ptr(dog), _ptr(nfit), P, CH, n_oct, tpo,
                                                         off, _ptr(ppair), _stream()))
            nfit_p = self._pinned("pair_nfit", tuple(nfit.shape), torch.float64)      # lands with the downloads' synchronisation
            nfit_p.copy_(nfit, non_blocking=True)
            host = (st["count_h"], st["fit_h"])
            if select_below is not None:
                recs, fits = self._download_selected(found, pval, count, fit, nt, cap, float(select_below), pair=(ppair, P),
                                                     host=host)
            else:
                recs, fits = self._download(found, pval, count, fit, nt, sort=True,
                                            extra={"pair": ppair, "q": self.fdr(pval, count, cap)}, host=host)
            torch.cuda.current_stream().synchronize()
            norm_fit = nfit_p.numpy().copy()
        batch = PairBandBatch(self, bands, n, dpx, starts, CH, st["nz_h"], recs, fits)
        batch.norm_fit = norm_fit
        return batch

    def _pairs_one_wait(self, st, P, dog, nfit, n_oct, tpo, pt):
        """Behind the two sigma loops and the difference kernel of a two-sample launch: p-values (mst_found_finish, no wait), pair
        p-values, BH + selection q < pt (mst_bh_select_nowait), the differential test's look-ups for the selected records
        (mst_pair_gather) and the downloads -- all queued back to back, ONE synchronisation, then the checks that used to cost a
        round trip each (record capacity: MstOverflow to the caller; selection capacity / oversized BH subset: None, the caller
        takes the step-by-step path).  A call on six block pairs of 2000 x 2000 is 1.4 ms of kernels: three more waits of
        ~45 us each were 10 % of it.  Returns (records, fits, tested-pixel counts, norm.fit) or None."""
        B, cap, CH = st["B"], st["found_cap"], st["CH"]
        found, pval, count, fit, stats, nzc = st["found"], st["pval"], st["count"], st["fit"], st["stats"], st["args"][2]
        nt = self.levels.n_tested
        sel = self._pair_sel_cap = getattr(self, "_pair_sel_cap", 256)
        ws_bytes = int(self.lib.mst_bh_workspace_bytes(B, cap))
        if ws_bytes == 0:
            return None
        summ = self._summary_pin(B)
        # everything that goes back to the host lies in ONE device buffer {q, pair / value / other, norm.fit | pixel, level, count}
        # and comes back in ONE copy (six small copies were 45 us of device time and 0.4 ms of host time)
        nf = int(nfit.numel())
        n8, n4 = B * sel + 3 * B * sel + nf, 2 * B * sel + B
        scratch, ppair, idx, ws, blob = self._carve(
            (summ.numel(), torch.uint8, (summ.numel(),)), (B * cap * 8, torch.float64, (B, cap)), (B * sel * 4, torch.int32, (B, sel)),
            (ws_bytes, torch.uint8, (ws_bytes,)), (8 * n8 + 4 * n4, torch.uint8, (8 * n8 + 4 * n4,)), reuse=("pairs-tail", 0))
        f8, i4 = blob[:8 * n8].view(torch.float64), blob[8 * n8:].view(torch.int32)
        qs, g, nfit_d = f8[:B * sel].view(B, sel), f8[B * sel:4 * B * sel].view(3, B, sel), f8[4 * B * sel:].view(nfit.shape)
        pix, lvl, n_sel = i4[:B * sel].view(B, sel), i4[B * sel:2 * B * sel].view(B, sel), i4[2 * B * sel:]
        none3 = (None, None, None)
        _lib.check(self.lib.mst_found_finish(_ptr(found), cap, _ptr(count), _ptr(nzc), _ptr(stats), B, nt, _ptr(pval), _ptr(fit), 0,
                                             *none3, _ptr(scratch), ctypes.c_void_p(summ.data_ptr()), *none3, 16, _stream()))
        for off in (0, P):
            _lib.check(self.lib.mst_pair_pvalues_dog(_ptr(found), cap, _ptr(count), _ptr(dog), _ptr(nfit), P, CH, n_oct, tpo, off,
                                                     _ptr(ppair), _stream()))
        _lib.check(self.lib.mst_bh_select_nowait(_ptr(found), _ptr(pval), _ptr(count), B, cap, pt, sel, _ptr(pix), _ptr(lvl), _ptr(qs),
                                                 _ptr(idx), _ptr(n_sel), self._bh_lds_records, _ptr(ws), ws_bytes, _stream()))
        _lib.check(self.lib.mst_pair_gather(_ptr(found), cap, _ptr(count), _ptr(ppair), int(P), _ptr(idx), _ptr(pix), _ptr(n_sel), sel,
                                            sel, _ptr(g[0]), _ptr(g[1]), _ptr(g[2]), _stream()))
        nfit_d.copy_(nfit)
        self._pin_flip ^= 1
        hblob = self._pinned("pw_blob", (int(blob.numel()),), torch.uint8)
        hblob.copy_(blob, non_blocking=True)
        torch.cuda.current_stream().synchronize()
        h8, h4 = hblob[:8 * n8].view(torch.float64).numpy(), hblob[8 * n8:].view(torch.int32).numpy()
        host = [h4[2 * B * sel:], h4[:B * sel].reshape(B, sel), h4[B * sel:2 * B * sel].reshape(B, sel), h8[:B * sel].reshape(B, sel),
                h8[B * sel:4 * B * sel].reshape(3, B, sel), h8[4 * B * sel:].reshape(tuple(nfit.shape))]
        if int(summ.numpy()[:4].view(np.int32)[0]):       # overflow / non-finite flags: the library words the error
            _lib.check(self.lib.mst_found_summary_status(ctypes.c_void_p(summ.data_ptr()), cap))   # MstOverflow: relaunch, larger lists
        n_h = host[0].view(np.uint32).astype(np.int64)
        mx = int(n_h.max(initial=0))
        if mx > sel:                                     # (MST_BH_RETRY = 0xFFFFFFFF included)
            if mx != 0xFFFFFFFF:
                self._pair_sel_cap = mx * 2              # room for every selected record from the next call on
            else:
                self._bh_lds_records = min(4096, self._bh_lds_records * 2)
            return None
        _, nz_h, fit_h = self._parse_summary(summ, B)
        # the kernel appends in arbitrary order: ONE sort by (block, pixel) over the live slots (pixels are unique inside a block),
        # then every block's arrays are slices of the sorted ones
        grid = self._slot_grid.get((B, sel)) if hasattr(self, "_slot_grid") else None
        if grid is None:
            self._slot_grid = {(B, sel): (np.arange(sel, dtype=np.int64)[None, :], np.arange(B, dtype=np.int64)[:, None] << 32)}
            grid = self._slot_grid[(B, sel)]
        live = np.flatnonzero(grid[0] < n_h[:, None])                      # ascending: by block, then slot
        pixf = host[1].view(np.uint32).reshape(-1)
        flat = live[np.argsort(((live // sel) << 32) | pixf[live], kind="stable")]
        g_h = host[4].reshape(3, -1)
        cols = (("pixel", pixf[flat]), ("level", host[2].view(np.uint32).reshape(-1)[flat]),
                ("q", host[3].reshape(-1)[flat]), ("pair", g_h[0][flat]), ("value", g_h[1][flat]), ("v_other", g_h[2][flat]))
        fit_c = fit_h[:, :nt, :]
        recs, fits, e = [], [], 0
        for b, m in enumerate(n_h.tolist()):
            recs.append({k: v[e:e + m] for k, v in cols})
            fits.append((fit_c[b, :, 0], fit_c[b, :, 1]))
            e += m
        return recs, fits, nz_h, host[5].copy()

    def run_band_pairs_overlapped(self, bands, n, dpx, groups, CH, skip_empty=True, select_below=None):
        """run_band_pairs over several groups of block pairs with the device work of group i + 1 queued BEFORE the results of
        group i are collected: the groups' kernels (both samples' sigma loops + the difference kernel) run back to back on two
        alternating side streams while the caller's host tail of the previous group -- and its small gathers on the caller's
        stream -- proceed.  Yields one PairBandBatch per group, identical to run_band_pairs(group)."""
        if len(groups) <= 1 or select_below is None:
            for starts in groups:
                yield self.run_band_pairs(bands, n, dpx, starts, CH, skip_empty=skip_empty, select_below=select_below)
            return
        cur = torch.cuda.current_stream(self.device)
        ready = cur.record_event()
        streams = device_streams(self.device)[:2]
        lt = self.levels
        n_oct, tpo = len(lt.octave_values), lt.s - 1
        lv = ctypes.byref(self._lv_struct)

        def launch(gi, starts):
            s = streams[gi % 2]
            s.wait_event(ready)
            if gi:
                s.wait_event(launch.prev_done)                   # one group's kernels at a time
            P = len(starts)
            with torch.cuda.stream(s), torch.cuda.device(self.device):
                cap = self._found_cap.get(CH, max(4096, (CH * CH) // 32))
                nzc = torch.empty(2 * P, dtype=torch.int32, device=self.device)
                st2 = [int(v) for v in starts]
                # both samples' blocks in ONE fused launch (rows [0, P) sample 1, [P, 2P) sample 2: mst_scale_space_band_pair)
                sst = self._ss_launch(None, None, nzc, skip_empty, cap, None, False, (bands[0], int(n), int(dpx), st2 + st2, int(CH)),
                                      band2=(bands[1], P))
                # the difference kernel needs the bands only: queue it behind the sigma loops right away
                st_arr = (ctypes.c_int64 * P)(*[int(v) for v in starts])
                dog = torch.empty((n_oct, P, CH, CH), dtype=torch.float64, device=self.device)
                nfit = torch.empty((n_oct, P, 2), dtype=torch.float64, device=self.device)
                mcount = torch.empty(P, dtype=torch.int32, device=self.device)
                ws_bytes = int(self.lib.mst_diff_dog_workspace_bytes(P, CH, lv))
                ws = torch.empty(ws_bytes, dtype=torch.uint8, device=self.device)
                _lib.check(self.lib.mst_diff_dog_band(_ptr(bands[0]), _ptr(bands[1]), int(n), int(dpx), st_arr, P, CH, lv,
                                                      _ptr(dog), _ptr(nfit), _ptr(mcount), _ptr(ws), ws_bytes, _stream()))
                launch.prev_done = s.record_event()
            return dict(stream=s, starts=starts, cap=cap, nzc=nzc, sst=sst, dog=dog, nfit=nfit, keep=(ws, mcount, st_arr))

        def collect(g):
            P = len(g["starts"])
            with torch.cuda.stream(g["stream"]), torch.cuda.device(self.device):
                st = self._ss_finish(g["sst"])
                if st["found_cap"] != g["cap"]:
                    # a record-capacity overflow re-ran the launch with more room: redo this group the plain way
                    return self.run_band_pairs(bands, n, dpx, g["starts"], CH, skip_empty=skip_empty, select_below=select_below)
                cap = g["cap"]
                found, pval, count, fit = st["found"], st["pval"], st["count"], st["fit"]
                ppair = torch.empty((2 * P, cap), dtype=torch.float64, device=self.device)
                for off in (0, P):
                    _lib.check(self.lib.mst_pair_pvalues_dog(_ptr(found), cap, _ptr(count), _ptr(g["dog"]), _ptr(g["nfit"]), P, CH,
                                                             n_oct, tpo, off, _ptr(ppair), _stream()))
                recs, fits = self._download_selected(found, pval, count, fit, self.levels.n_tested, cap, float(select_below),
                                                     pair=(ppair, P))
                nz_h = st["nz_h"]
                norm_fit = g["nfit"].cpu().numpy()
            batch = PairBandBatch(self, bands, n, dpx, g["starts"], CH, nz_h, recs, fits)
            batch.norm_fit = norm_fit
            return batch

        pending = None
        for gi, starts in enumerate(groups):
            g = launch(gi, starts)
            if pending is not None:
                yield collect(pending)
            pending = g
        yield collect(pending)
        cur.wait_stream(streams[0])
        cur.wait_stream(streams[1])

    def run_block_pairs(self, c, dpx, intra=True, skip_empty=True):
        """c: [2P, CH, CH] raw blocks (sample 1 first, then sample 2), mutated in place.  BlockBatch over all 2P blocks
        whose records also carry `pair` (the differential p-value)."""
        nz, nz_count = self.prologue(c, dpx, intra)
        found, pval, count, fit, cap = self.sigma_loop(c, nz, nz_count, skip_empty=skip_empty, download=False)
        ppair, nfit = self.pair_pvalues(c, nz, found, cap, count)
        recs, fits = self._download(found, pval, count, fit, self.levels.n_tested, sort=True,
                                    extra={"pair": ppair, "q": self.fdr(pval, count, cap)})
        B, CH, _ = c.shape
        batch = BlockBatch(self, c, nz, CH, B, nz_count.cpu().numpy().view(np.uint32).astype(np.int64), recs, fits)
        batch.norm_fit = nfit.cpu().numpy()
        return batch

    def run_blocks(self, c, dpx, intra=True, skip_empty=True):
        """c: [B, CH, CH] float64 device tensor holding raw (normalised, un-filled) blocks; mutated in place
        like the reference mutates its block (mustache.py:703-706)."""
        nz, nz_count = self.prologue(c, dpx, intra)
        found, fits = self.sigma_loop(c, nz, nz_count, skip_empty=skip_empty)
        B, CH, _ = c.shape
        return BlockBatch(self, c, nz, CH, B, nz_count.cpu().numpy().view(np.uint32).astype(np.int64), found, fits)
