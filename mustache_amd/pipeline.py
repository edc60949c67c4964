"""Per-chromosome pipeline on one GPU (or one rank's share of it): the body of the reference's regulator()
(mustache/mustache.py:892-937) with every block resident in HBM.

  COO (host) -> device -> band -> normalize_band -> [batches of blocks] blocks_from_band -> fused sigma loop
  -> found records -> host tail per block -> overlap mask -> (gather across ranks)
"""
import ctypes
import time

import numpy as np
import torch

from . import _lib
from .engine import ScaleSpaceEngine, BlockBatch, BandBatch, _ptr, _stream
from .normalize import band_from_coo, band_from_host_coo, normalize_band
from .sharding import shard_blocks, gather_loops, world
from .tail import batch_tail


def split_groups(items, per, px_per_item=16e6):
    """items in groups of `per` (the stages of one fused launch, engine.sigma_loop_band_overlapped): the finish and the host tail
    of a group run under the next group's kernel, only the LAST group's are exposed -- so a last group of six or more blocks
    and at least ~100 Mpix gives all but 6 % of its items (at least two: their kernel has to cover the finish of the group in
    front) to a group of their own (a stage boundary costs 0.08 ms, an exposed 4000 x 4000 block ~0.13 ms).  Smaller launches
    stay whole: they are latency-bound and replayed as one graph."""
    groups = [items[i:i + per] for i in range(0, len(items), per)]
    if groups and len(groups[-1]) >= 6 and len(groups[-1]) * px_per_item >= 96e6:
        last = groups.pop()
        t = max(2, int(round(0.06 * len(last))))
        groups += [last[:-t], last[-t:]]
    return groups


# what the last run_band / run_layout of this process measured: blocks, Mpix, seconds in the scale-space stage and in the tail,
# the fused kernel's own time (HIP events) -- the command line's verbose summary prints it (mustache.main, `-v`)
LAST_RUN = {}


def run_summary(t=None):
    """one line for the verbose output: blocks, Mpix, stage seconds, fused-kernel milliseconds, Mpix/s"""
    t = LAST_RUN if t is None else t
    if not t or not t.get("mpix"):
        return ""
    dev, tail = t.get("scale_space_s", 0.0), t.get("tail_s", 0.0)
    rate = t["mpix"] / max(dev + tail, 1e-9)
    return ("  [gpu] %d block(s) of %d x %d = %.1f Mpix: scale-space %.3f s (fused launches %.1f ms by HIP events, a first launch with its work-list set-up), tail %.3f s -> %.0f Mpix/s"
            % (t.get("blocks", 0), t.get("chunk", 0), t.get("chunk", 0), t["mpix"], dev, t.get("kernel_ms", 0.0), tail, rate))


def block_tiling(n, distance_in_px):
    from .mustache import block_tiling as _bt
    return _bt(n, distance_in_px)


def block_mask_size(i, start, end, overlap):
    from .mustache import block_mask_size as _bm
    return _bm(i, start, end, overlap)


class GenomeLayout:
    """Several chromosomes of one run side by side in ONE band, so that all their blocks go through the same launches
    (BASELINE configs 3 and 5; the reference walks the chromosomes one after the other, mustache.py:1057-1067, and a 5 kb
    chromosome is only 5-31 blocks -- too few to fill the chip).  Chromosome c owns the columns [off[c], off[c] + slot[c]) of
    the band, slot[c] >= max(n_c, CHUNK) rounded up to 64 columns and zero beyond n_c: a block that starts at off[c] + s reads
    exactly what it reads from the chromosome's own band at s -- pixels past the chromosome's end are 0 = "no contact" in
    both (the kernels' `start + column < n` test against the total width never has to know the chromosome), and no block
    reaches into the next slot.  Nothing in the kernels or the C ABI changes; block statistics, found sets and loops are
    those of the per-chromosome run, bit for bit."""

    def __init__(self, ns, dpx):
        self.ns, self.dpx = [int(n) for n in ns], int(dpx)
        self.CH = max(2 * self.dpx, 2000)
        self.slot = [-(-max(n, self.CH) // 64) * 64 for n in self.ns]
        self.off = [0]
        for w in self.slot[:-1]:
            self.off.append(self.off[-1] + w)
        self.N = self.off[-1] + self.slot[-1] if self.ns else 0
        self.tiling = [block_tiling(n, self.dpx) for n in self.ns]
        # global block list in chromosome order: (chromosome, block index, local start, global start)
        self.blocks = [(c, i, t[1][i], self.off[c] + t[1][i]) for c, t in enumerate(self.tiling) for i in range(len(t[1]))]

    def band(self, bands, device, consume=False):
        """bands[c]: [dpx + 2, n_c] normalised band of chromosome c (device) -> the [dpx + 2, N] genome band.
        consume=True: bands[c] is set to None as soon as it has been copied, so that (when the caller holds no other
        reference) the chromosome bands are released one by one while the genome band fills -- the peak is the genome
        band + the bands not yet copied, not twice the genome."""
        g = torch.zeros((self.dpx + 2, self.N), dtype=torch.float64, device=device)
        for c, (n, off) in enumerate(zip(self.ns, self.off)):
            g[:, off:off + n] = bands[c].view(self.dpx + 2, -1)[:, :n]
            if consume:
                bands[c] = None
        return g


def genome_batch_budget(device=None):
    """Bytes of normalised chromosome bands a whole-genome run may hold before it flushes them through run_genome.
    MUSTACHE_GENOME_BATCH_GB, when set, is taken as is; otherwise 35 % of the device memory that is free right now, capped
    at 64 GiB: at a flush the genome band (as large as the held bands together) is allocated while the held bands are
    released one by one (GenomeLayout.band(consume=True)), on top of the found-record buffers and workspaces of the launches
    and the reader's next chromosome being normalised (two more bands) -- 2 x 35 % leaves room for those on any device."""
    import os
    env = os.environ.get("MUSTACHE_GENOME_BATCH_GB")
    if env:
        return int(float(env) * (1 << 30))
    free, _total = torch.cuda.mem_get_info(device)
    return int(min(64 << 30, 0.35 * free))


class ChromosomePipeline:
    def __init__(self, octave_values=(1.6, 3.2), device=None, max_batch_bytes=48 << 30):
        self.engine = ScaleSpaceEngine(octave_values, device=device)
        self.device = self.engine.device
        self.max_batch_bytes = max_batch_bytes
        self.overlap_blocks = 16        # blocks per fused-kernel launch on the band path (copy / host-tail overlap)

    # ---- device stages --------------------------------------------------------------------------------------------
    def blocks_from_band(self, band, n, dpx, starts, CH):
        B = len(starts)
        c = torch.empty((B, CH, CH), dtype=torch.float64, device=self.device)
        nz = torch.empty((B, CH, CH), dtype=torch.uint8, device=self.device)
        nzc = torch.empty(B, dtype=torch.int32, device=self.device)
        st = (ctypes.c_int64 * B)(*[int(s) for s in starts])
        with torch.cuda.device(self.device):
            _lib.check(self.engine.lib.mst_blocks_from_band(_ptr(band), int(n), int(dpx), st, B, CH, _ptr(c),
                                                            _ptr(nz), _ptr(nzc), _stream()))
        return c, nz, nzc

    def batches(self, idx, CH, dense=True):
        per_block = (CH * CH * 9 if dense else 0) + max(4096, CH * CH // 32) * 48
        bs = max(1, int(self.max_batch_bytes // per_block))
        return [idx[i:i + bs] for i in range(0, len(idx), bs)]

    def run_band(self, band, n, dpx, st, pt, skip_empty=True, distributed=True, timings=None, shard=None,
                 dense=False):
        """band: normalised band on the device.  Returns this chromosome's loops (all ranks, after the gather).
        `shard=(rank, world_size)` runs one rank's share without a process group (no gather) -- used by tests."""
        CH, start, end = block_tiling(n, dpx)
        rank, ws = world() if distributed else (0, 1)
        if shard is not None:
            rank, ws, distributed = shard[0], shard[1], False
        mine = shard_blocks(len(start), rank, ws)
        loops = []
        t_dev = t_tail = 0.0

        def tail(batch, group, starts_g):
            with _lib.stage("tail"):
                tails = batch_tail(batch, list(range(len(group))), starts_g, pt, st, intra=True)
            for j, i in enumerate(group):
                mask = block_mask_size(i, start, end, dpx)
                for lp in tails[j]:
                    if lp[0] >= start[i] + mask or lp[1] >= start[i] + mask:      # mustache.py:957-959
                        loops.append([lp[0], lp[1], lp[2], lp[3]])

        if dense:           # materialise [B, CH, CH] blocks first (what mustache() gets from its caller) -- cross-check path
            for group in self.batches(mine, CH, dense):
                t0 = time.time()
                starts_g = [start[i] for i in group]
                c, nz, nzc = self.blocks_from_band(band, n, dpx, starts_g, CH)
                found, fits = self.engine.sigma_loop(c, nz, nzc, skip_empty=skip_empty, with_value=False,
                                                     select_below=pt)
                batch = BlockBatch(self.engine, c, nz, CH, len(group),
                                   nzc.cpu().numpy().view(np.uint32).astype(np.int64), found, fits)
                t1 = time.time()
                tail(batch, group, starts_g)
                t_dev += t1 - t0
                t_tail += time.time() - t1
                del c, nz, batch
        else:
            # default: blocks are windows of the band, cut inside the fused kernel.  The blocks go through it in groups of
            # `overlap_blocks`; BH + selection + download AND the host tail of one group run under the kernel of the next
            groups = split_groups(mine, self.overlap_blocks, float(CH) * CH)
            starts = [[start[i] for i in g] for g in groups]
            t0 = time.time()
            kev = []                               # (start, end) events of the fused kernel's launches / stages
            for group, starts_g, (found, fits, nzc) in zip(groups, starts, self.engine.sigma_loop_band_overlapped(
                    band, n, dpx, starts, CH, skip_empty=skip_empty, with_value=False, select_below=pt, timing=kev)):
                t1 = time.time()
                batch = BandBatch(self.engine, band, n, dpx, starts_g, CH,
                                  nzc.cpu().numpy().view(np.uint32).astype(np.int64), found, fits)
                tail(batch, group, starts_g)
                t_tail += time.time() - t1
                del batch
            t_dev = time.time() - t0 - t_tail
            LAST_RUN.update(kernel_ms=sum(a.elapsed_time(b) for a, b in kev))
        LAST_RUN.update(scale_space_s=t_dev, tail_s=t_tail, blocks=len(mine), chunk=CH, mpix=len(mine) * CH * CH / 1e6)
        if timings is not None:
            timings.update(LAST_RUN)
        return gather_loops(loops, device=self.device) if (distributed and ws > 1) else loops

    def blocks_per_launch(self, CH):
        """Blocks per fused-kernel launch on the band path: 16 of 4000 x 4000, 64 of 2000 x 2000 (>= 256 Mpix per launch)."""
        return max(self.overlap_blocks, (1 << 28) // (CH * CH))

    def run_genome(self, bands, ns, dpx, st, pt, skip_empty=True, timings=None):
        """Whole-genome batched form of run_band (single process / one rank's own chromosomes): bands[c] is the normalised
        band of chromosome c.  All blocks of all chromosomes go through the same groups of launches, BH / selection /
        download and host tail of one group under the kernel of the next.  Returns the list of loops per chromosome
        (chromosome coordinates), identical to run_band on each chromosome alone."""
        lay = GenomeLayout(ns, dpx)
        # `bands` is consumed (entries set to None as they are copied): a caller that drops its own references beforehand
        # keeps the peak at the genome band + the bands not yet copied
        return self.run_layout(lay, lay.band(bands, self.device, consume=True), st, pt, skip_empty=skip_empty, timings=timings)

    def run_layout(self, lay, gband, st, pt, skip_empty=True, timings=None):
        """run_genome's body on a prepared layout + genome band."""
        CH, dpx, ns = lay.CH, lay.dpx, lay.ns
        per = self.blocks_per_launch(CH)
        groups = split_groups(lay.blocks, per, float(CH) * CH)
        loops = [[] for _ in ns]
        t0 = time.time()
        t_tail = 0.0
        kev = []
        for group, (found, fits, nzc) in zip(groups, self.engine.sigma_loop_band_overlapped(
                gband, lay.N, dpx, [[g[3] for g in grp] for grp in groups], CH, skip_empty=skip_empty, with_value=False,
                select_below=pt, timing=kev)):
            t1 = time.time()
            batch = BandBatch(self.engine, gband, lay.N, dpx, [g[3] for g in group], CH,
                              nzc.cpu().numpy().view(np.uint32).astype(np.int64), found, fits)
            with _lib.stage("tail"):
                tails = batch_tail(batch, list(range(len(group))), [g[2] for g in group], pt, st, intra=True)
            for j, (c, i, s_loc, _) in enumerate(group):
                _, start, end = lay.tiling[c]
                mask = block_mask_size(i, start, end, dpx)
                for lp in tails[j]:
                    if lp[0] >= s_loc + mask or lp[1] >= s_loc + mask:            # mustache.py:957-959
                        loops[c].append([lp[0], lp[1], lp[2], lp[3]])
            t_tail += time.time() - t1
            del batch
        LAST_RUN.update(scale_space_s=time.time() - t0 - t_tail, tail_s=t_tail, blocks=len(lay.blocks), chunk=CH,
                        mpix=len(lay.blocks) * CH * CH / 1e6, launches=len(groups),
                        kernel_ms=sum(a.elapsed_time(b) for a, b in kev))
        if timings is not None:
            timings.update(LAST_RUN)
        return loops

    def normalized_band_packed(self, pc, dpx, normalized=False):
        """hicfile.PackedContacts of one chromosome -> (normalised band on the device, n)."""
        from .normalize import band_from_packed
        with _lib.stage("band from records"):
            band = band_from_packed(pc, dpx, self.device)       # pc.n_parts > 1: the ranks exchange their shares in here
        n = int(band.shape[1])                              # = max(binY) + 1 over ALL shares (mustache.py:894)
        if not normalized and n > 0:
            with _lib.stage("normalise"):
                band, _, _ = normalize_band(band, n, dpx, pc.res)
        return band, n

    def run_packed(self, pc, dpx, st, pt, verbose=False, skip_empty=True, distributed=True, timings=None):
        """run() for the native `.hic` reader's packed records."""
        t0 = time.time()
        if verbose:
            print("Normalizing contact map...")
        band, n = self.normalized_band_packed(pc, dpx)
        torch.cuda.synchronize(self.device)
        t1 = time.time()
        if n == 0:                                          # no rank read a record (every rank sees the same n)
            print('There is no contact in this chromosome to work on.')
            return []
        if verbose:
            print("Loop calling...")
        loops = self.run_band(band, n, dpx, st, pt, skip_empty=skip_empty, distributed=distributed, timings=timings)
        if timings is not None:
            timings["normalize_s"] = t1 - t0
        return loops

    def normalized_band(self, x, y, v, res, dpx, normalized=False):
        """Host COO of one chromosome -> (normalised band on the device, n)   (mustache.py:894-895)."""
        x = np.ascontiguousarray(np.asarray(x), dtype=np.int64)
        y = np.ascontiguousarray(np.asarray(y), dtype=np.int64)
        v = np.ascontiguousarray(np.asarray(v), dtype=np.float64)
        n = int(max(x.max(), y.max())) + 1
        with _lib.stage("band from records"):
            band = band_from_host_coo(x, y, v, n, dpx, self.device)
        if not normalized:
            with _lib.stage("normalise"):
                band, _, _ = normalize_band(band, n, dpx, res)
        return band, n

    def run(self, x, y, v, res, dpx, st, pt, normalized=False, verbose=False, skip_empty=True, distributed=True,
            timings=None):
        """x, y, v: the reference's host COO.  Every rank reads the same chromosome; normalisation is replicated
        (cheap on the GPU), blocks are sharded."""
        x = np.ascontiguousarray(np.asarray(x), dtype=np.int64)
        y = np.ascontiguousarray(np.asarray(y), dtype=np.int64)
        v = np.ascontiguousarray(np.asarray(v), dtype=np.float64)
        n = int(max(x.max(), y.max())) + 1                                       # mustache.py:894
        t0 = time.time()
        band = band_from_host_coo(x, y, v, n, dpx, self.device)
        if not normalized:
            if verbose:
                print("Normalizing contact map...")
            band, _, _ = normalize_band(band, n, dpx, res)
        torch.cuda.synchronize(self.device)
        t1 = time.time()
        if verbose:
            print("Loop calling...")
        loops = self.run_band(band, n, dpx, st, pt, skip_empty=skip_empty, distributed=distributed, timings=timings)
        if timings is not None:
            timings["normalize_s"] = t1 - t0
        return loops
