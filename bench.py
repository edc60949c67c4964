#!/usr/bin/env python3
"""bench.py -- scale-space Mpix/s of the HIP hot path on synthetic banded contact maps (BASELINE.json metric).

    python bench.py --gpus 1 --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N --steps K --warmup W

Workload (config.workload): the synthetic HFFc6-like chr1 at 1 kb of SURVEY.md section 8d -- n = 248,957 bins, distance
limit 2,000 bins, 124 overlapping dense blocks of 4000 x 4000 float64 (1,984 Mpix).  One "step" = one pass of rows 2-7
of SURVEY.md section 8a over ALL blocks of the chromosome: normalised band resident in HBM -> fused kernel (blocks cut out
of the band, filled and masked while the tile is staged; sigma-stack / DoG / 3x3 max / sieve / level statistics) ->
p-values of the found pixels -> compacted found records on the host.  The step is ONE launch in two or three stages
(Workload.step): the records of one stage's blocks are finished and downloaded under the next stage's kernel.

N ranks = STRONG scaling, the reference's own partition (one process per block of ONE chromosome, mustache.py:913-937):
the 124 blocks are split into N contiguous ranges, no data-path collective; the step time is the MAX over ranks between
two barriers and `value` = 1,984 Mpix / that time, whole job, at every N.  (`--scaling weak` = one whole chromosome per
rank, the genome partition; the mode that is not the headline is timed in the same run and reported as `other_scaling`.)

On the same JSON line: `roofline` (the fused kernel, HIP events on the launch stream, against the FP64 vector pipe WITHOUT
FMA -- bit-exactness with SciPy forbids contraction: 1152 flops per pixel over 39.3 TFLOP/s; `hbm_model` = SURVEY 8d's
592 B / pixel model beside the PMC traffic), `cpu_baseline` (the CPU oracle: one block on 1 core, `p4` = 8 blocks in 4 processes
as the reference's default -p 4, and the first / middle / last block's whole found sets checked against the HIP path),
`band_skip` (empty tiles skipped: the product mode) and `tile_sharing` / `no_share` (separate speed-ups), `chr21_5kb` /
`diff_chr21_5kb`, `end_to_end` / `end_to_end_from_file`, `ranks` (at N = 1 with a labelled single-GPU projection of the strong
split: every rank's block range of N = 2, 4, 8 timed alone on this GPU).  `--extra` adds the side legs of scripts/bench_extra.py.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

BYTES_PER_PIXEL = 592.0          # level-streaming algorithmic traffic, fp64
HBM_PEAK_GBS = 8000.0            # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
FP64_PEAK_TFLOPS = 78.6          # FMA-counted vector FP64 peak (256 CU x 4 SIMD x 16 lanes x 2 x 2.4 GHz)
FLOPS_PER_PIXEL = 1152.0         # 24 blurs x 2 axes x (1 + 3 r) non-fusable flops, sum of r = 184
PEAK_TF = FP64_PEAK_TFLOPS / 2   # add / mul rate without FMA
OVERLAP = int(os.environ.get("MST_BENCH_OVERLAP", "3"))   # most stages per step (copy/compute overlap), see Workload.step


def executed_flops_per_pixel():
    """FP64 blur operations the kernel really issues per block pixel: 22 distinct blurs (the two levels that repeat between
    the octaves are computed once), the V pass over the 2r halo columns of its 64-column region, and the 1-pixel ring of
    the 32 x 64 region around the 30 x 62 pixels a workgroup owns (mst_scale_space.hip, Tile<32, 64, 14>)."""
    from mustache_amd.levels import LevelTable
    lt = LevelTable((1.6, 3.2), 10)
    radii = [int(r) for r in lt.radius]
    lpo = lt.levels_per_octave
    distinct = radii[:lpo] + radii[lpo + 2:]                       # octave 2 starts from the state octave 1 left behind
    per_region_px = sum((1 + 3 * r) * ((64 + 2 * r) / 64.0 + 1.0) for r in distinct)      # V pass + H pass
    return per_region_px * (32 * 64) / (30.0 * 62.0)


def stage_cuts(nb, CH, overlap=3, shares=None):
    """Where a step's nb blocks are cut into the stages of its one launch: [0, ..., nb].  Small launches (< 4 blocks or < 120 Mpix:
    six blocks of 2000 x 2000) stay whole and are replayed as a graph; otherwise the LAST stage is 6 % of the blocks, at least two
    beyond 16 blocks (its kernel has to cover the download of the stage before it), with two equal stages in front from 24 blocks up.
    `shares` ("0.8,0.2": MST_BENCH_SHARES, scripts/share_split_time.py) overrides the rule."""
    if shares:
        cuts = [0]
        for f in [float(x) for x in shares.split(",")][:-1]:
            cuts.append(min(nb - 1, max(cuts[-1] + 1, int(round(cuts[-1] + f * nb)))))
        return cuts + [nb]
    last = max(1 if nb <= 16 else 2, int(round(0.06 * nb)))    # (16 blocks: [15, 1] 14.81 ms, [14, 2] 14.86, measured)
    if nb < 4 or nb * CH * CH < 120e6 or overlap < 2:
        return [0, nb]
    if nb < 24 or overlap < 3:
        return [0, nb - last, nb]
    return [0, (nb - last + 1) // 2, nb - last, nb]


def work_items(w, skip_empty, share=True):
    """(workgroups launched, tiles the blocks would run one by one, tiles computed once for two blocks, workgroups per
    launch) over the launches of one step of workload w -- asked of the library (engine.band_items)."""
    if w.pipe.engine.staged_launches:      # the groups are stages of ONE work list: sharing crosses them
        blocks = [w.start[i] for g in w.groups for i in g]
        p = w.pipe.engine.band_items(blocks, w.CH, w.dpx, skip_empty, share)
        # a stage's workgroups = the items of its blocks: the list of the first k blocks holds exactly the whole list's items
        # of those blocks (an item shared with block k is listed once either way)
        ends, acc = [], 0
        for g in w.groups:
            acc += len(g)
            ends.append(acc)
        pre = [0] + [w.pipe.engine.band_items(blocks[:e], w.CH, w.dpx, skip_empty, share)[0] for e in ends[:-1]] + [p[0]]
        return p[0], p[1], p[2], [b - a for a, b in zip(pre[:-1], pre[1:])]
    per = [w.pipe.engine.band_items([w.start[i] for i in g], w.CH, w.dpx, skip_empty, share) for g in w.groups]
    return sum(p[0] for p in per), sum(p[1] for p in per), sum(p[2] for p in per), [p[0] for p in per]


def share_projection(w, one_rank_ms, steps=3):
    """While no multi-GPU node is at hand: the step of EVERY rank's contiguous block range at N = 2, 4, 8, each timed alone on
    this GPU exactly like the headline step (the strong split has no data-path collective, so a rank's step IS its range's
    step); projected step at N = the slowest range.  A single-GPU share timing, not a multi-GPU run: xGMI, the host's PCIe
    sharing and clock differences between GPUs are not in it."""
    import torch
    keep = (w.rank, w.world, w.scaling)
    proj, eff, per_rank = {}, {}, {}
    try:
        for n in (2, 4, 8):
            times = []
            for r in range(n):
                w.rank, w.world = r, n
                w.set_scaling("strong")
                for _ in range(2):
                    w.step(False)
                torch.cuda.synchronize()
                t0 = time.time()
                for _ in range(steps):
                    w.step(False)
                torch.cuda.synchronize()
                times.append((time.time() - t0) / steps * 1e3)
            proj[str(n)], per_rank[str(n)] = round(max(times), 3), [round(t, 3) for t in times]
            eff[str(n)] = round(one_rank_ms / (n * max(times)), 4)
    finally:
        w.rank, w.world = keep[0], keep[1]
        w.set_scaling(keep[2])
        w.kernel_ms.clear()
    return {"projected_step_ms_at": proj, "projected_efficiency_at": eff, "projected_step_ms_per_rank": per_rank,
            "projected_n_x_share_over_one_rank_at": {k: round(int(k) * v / one_rank_ms, 4) for k, v in proj.items()},
            "kind": "single-GPU share timing, not a multi-GPU run",
            "projection_note": "every rank's block range of the strong split timed alone on this GPU, %d steps each; projected step = "
                               "the slowest range; efficiency = one-rank step / (N x projected step)" % steps}


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--no-cpu", action="store_true", help="skip the CPU baseline leg")
    ap.add_argument("--no-file", action="store_true", help="skip the end-to-end-from-a-.hic-file leg")
    ap.add_argument("--core", action="store_true", help="no file or CPU legs (what the PMC passes of scripts/profile_bench.sh run)")
    ap.add_argument("--extra", action="store_true", help="add the side legs of scripts/bench_extra.py (N = 1 only)")
    ap.add_argument("--small", action="store_true", help="debug: 12 blocks instead of 124")
    ap.add_argument("--scaling", choices=("strong", "weak"), default="strong",
                    help="N > 1: strong (default, the metric) = ONE chromosome's blocks in contiguous ranges over the ranks; weak "
                         "= one whole chromosome per rank.  The other mode is timed after the headline (`other_scaling`)")
    a = ap.parse_args()
    a.no_cpu, a.no_file = a.no_cpu or a.core, a.no_file or a.core
    return a


def make_band(n, dpx, depth, nloops, seed, res, device, reps=8):
    """Synthetic chromosome -> normalised band on `device` (input preparation, not timed)."""
    import torch
    from mustache_amd.synth import band_counts
    from mustache_amd.normalize import normalize_band
    cols = 16384
    raw = torch.empty((dpx + 2, n), dtype=torch.float64, device=device)
    for i0 in range(0, n, cols):                      # generated in column slabs to bound temporaries
        i1 = min(n, i0 + cols)
        raw[:, i0:i1] = band_counts(n, dpx, depth, nloops, seed, i0=i0, i1=i1, device=device)
    band, _, _ = normalize_band(raw, n, dpx, res)      # untimed: code objects, and the allocator's first 4 GB block
    ms = []
    # steady state: HIP events around mst_normalize_band (both kernels); the clocks and the TLB settle over ~4 calls, so 3
    # more untimed calls precede the median of 5
    for it in range(reps):
        del band
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        band, _, _ = normalize_band(raw, n, dpx, res)
        e1.record()
        torch.cuda.synchronize()
        if it >= 3 or reps < 4:
            ms.append(e0.elapsed_time(e1))
    return band, sorted(ms)[len(ms) // 2] * 1e-3


class Workload:
    def __init__(self, name, n, dpx, res, depth, nloops, seed, device, rank, world, scaling="strong"):
        from mustache_amd.pipeline import ChromosomePipeline, block_tiling
        self.name, self.n, self.dpx, self.res = name, n, dpx, res
        self.pipe = ChromosomePipeline((1.6, 3.2), device=device)
        self.band, self.normalize_s = make_band(n, dpx, depth, nloops, seed, res, device)
        self.CH, self.start, self.end = block_tiling(n, dpx)
        self.rank, self.world = rank, world
        self.set_scaling(scaling)
        self.kernel_ms = []

    def set_scaling(self, scaling):
        """strong: the blocks of ONE chromosome in contiguous ranges over the ranks (the metric); weak: every rank runs ALL
        blocks of its own copy of the chromosome (N chromosomes per step).  Identical at one rank."""
        from mustache_amd.sharding import shard_blocks
        self.scaling = scaling
        nb = len(self.start)
        if scaling == "weak":
            self.mine = list(range(nb))
            self.total_mpix = self.world * nb * self.CH * self.CH / 1e6
        else:
            self.mine = shard_blocks(nb, self.rank, self.world)
            self.total_mpix = nb * self.CH * self.CH / 1e6

    def step(self, skip_empty=False, download=True, fma=False):
        """rows 2-7 for this rank's blocks; returns the found records per group of blocks.  The blocks go through the fused
        kernel as ONE launch in two or three STAGES (mst_scale_space_band_stage: one work list, so tiles shared by consecutive
        blocks are computed once across the stages too): the p-values and the pinned download of one stage's blocks run on a
        second stream under the next stage's kernel.  Only the LAST stage's download is exposed, so it is the small one: 6 % of
        the blocks, at least two (its kernel must still cover the download of the stage before it); measured splits:
        scripts/share_split_time.py, LABBOOK R6.2."""
        key = (tuple(self.mine), OVERLAP, os.environ.get("MST_BENCH_SHARES"))
        if getattr(self, "_groups_for", None) != key:
            groups = []
            for batch in self.pipe.batches(self.mine, self.CH, dense=False):
                cuts = stage_cuts(len(batch), self.CH, OVERLAP, os.environ.get("MST_BENCH_SHARES"))
                groups += [batch[a:b] for a, b in zip(cuts[:-1], cuts[1:]) if b > a]
            self.groups = groups                             # the split of the blocks into stages is fixed between steps
            self._groups_for = key
            self._group_starts = [[self.start[i] for i in g] for g in groups]
        # blocks are windows of the band: cut, filled (mustache.py:703-706) and masked (:699) inside the fused kernel
        return list(self.pipe.engine.sigma_loop_band_overlapped(
            self.band, self.n, self.dpx, self._group_starts, self.CH, skip_empty=skip_empty,
            download=download, timing=self.kernel_ms, sort=False, with_value=False, with_q=False, fma=fma))


class Ctx:
    """what the legs share: the workload, the ranks, and the timed loop of the contract (barrier + synchronize on both sides,
    MAX over ranks)"""

    def __init__(self, w, args, device, rank, world, grouped, backend):
        self.w, self.args, self.device, self.rank, self.world, self.grouped, self.backend = w, args, device, rank, world, grouped, backend

    def barrier(self):
        import torch
        if self.grouped:
            import torch.distributed as dist
            dist.barrier()
        torch.cuda.synchronize()

    def timed(self, skip_empty, steps, warmup, fma=False):
        """-> (job seconds = max over ranks, this rank's kernel launch times [ms], last step's records, every rank's own
        seconds up to its stream-synchronised end, before the closing barrier)"""
        import torch
        w = self.w
        for _ in range(warmup):
            w.step(skip_empty, fma=fma)
        w.kernel_ms.clear()
        self.barrier()
        t0 = time.time()
        for _ in range(steps):
            last = w.step(skip_empty, fma=fma)
        torch.cuda.synchronize()
        own = time.time() - t0
        self.barrier()
        dt = time.time() - t0
        owns = [own]
        if self.grouped:
            import torch.distributed as dist
            t = torch.tensor([dt, own], dtype=torch.float64, device=self.device if self.backend == "nccl" else "cpu")
            allt = [torch.zeros_like(t) for _ in range(self.world)]
            dist.all_gather(allt, t)
            dt, owns = max(float(a[0]) for a in allt), [float(a[1]) for a in allt]
        return dt, [a.elapsed_time(b) for a, b in w.kernel_ms], last, owns


def file_leg(cx, keep=200.0, on_band=None):
    """file -> loops for a config-4-shaped `.hic` (chr1 at 1 kb), stage by stage, on all ranks: the native
    reader inflates on host threads into page-locked slabs that go to the device while later blocks inflate; with N ranks
    rank r inflates share r of the blocks, the shares are exchanged (sharding.all_gather_packed: RCCL over xGMI from device
    memory), every rank builds and normalises the same band, runs its contiguous range of blocks, and the loops are
    gathered.  Times are rank 0's wall clock between two barriers; pass 1 pays for fresh pages, passes 2-6 are the steady
    state of a whole-genome run.  `total_s` and the stage times beside it are the MEDIAN steady pass (`best_total_s` beside)."""
    import shutil
    import tempfile
    import torch
    import torch.distributed as dist
    from mustache_amd.hicfile import HicFile
    from mustache_amd.normalize import band_from_packed, normalize_band, read_hic_stream_to_device
    w, rank, world = cx.w, cx.rank, cx.world
    tmp = path = None
    nrec = size = 0
    try:
        if rank == 0:
            tmp = tempfile.mkdtemp(prefix="mst_bench_")
            path = os.path.join(tmp, "chr1_1kb.hic")
            sys.path.insert(0, os.path.join(ROOT, "tests"))
            from hic_writer import write_synthetic_hic          # test infrastructure; untimed
            nrec = write_synthetic_hic(path, w.n, w.dpx, w.res, 400.0, 8000 if w.n > 100000 else 800, 1, keep, cx.device)
            size = os.path.getsize(path)
        if cx.grouped:
            box = [path]
            dist.broadcast_object_list(box, src=0)          # one node: /tmp is shared by the ranks
            path = box[0]
        h = HicFile(path)
        passes = []
        for rep in range(6):
            cx.barrier()
            t = [time.time()]
            cpu0 = time.process_time()             # user + system seconds of all threads
            pc = read_hic_stream_to_device(h, "chr1", w.res, "KR", w.dpx, 0, cx.device, part=(rank, world))   # records in HBM
            t.append(time.time())
            cpu1 = time.process_time()
            band = band_from_packed(pc, w.dpx, cx.device)      # zero fill + scatter; world > 1: the shares are exchanged in here
            n = int(band.shape[1])
            torch.cuda.synchronize()
            t.append(time.time())
            nb, _, _ = normalize_band(band, n, w.dpx, w.res)
            torch.cuda.synchronize()
            t.append(time.time())
            loops = w.pipe.run_band(nb, n, w.dpx, 0.88, 0.1, timings={}, distributed=cx.grouped)   # blocks sharded, loops gathered
            cx.barrier()
            t.append(time.time())
            reads, recs = [t[1] - t[0]], [len(pc)]
            if cx.grouped:
                box = [None] * world
                dist.all_gather_object(box, (t[1] - t[0], len(pc)))
                reads, recs = [b[0] for b in box], [b[1] for b in box]
            passes.append({"total_s": round(t[4] - t[0], 4), "reader_s": round(t[1] - t[0], 4),
                           "band_scatter_s": round(t[2] - t[1], 4), "normalize_s": round(t[3] - t[2], 4),
                           "kernels_and_tail_s": round(t[4] - t[3], 4),
                           "reader_plus_upload_s": round(t[2] - t[0], 4), "gpu_step_s": round(t[4] - t[2], 4),
                           "reader_cpu_s": round(cpu1 - cpu0, 3),
                           "loops": len(loops), "records": int(sum(recs)), "n": n, "hic_blocks_total": pc.blocks_total,
                           "read_s_per_rank": [round(r, 4) for r in reads], "records_per_rank": [int(r) for r in recs]})
            if rep == 5 and on_band is not None:
                on_band(nb, n)
            del pc, band, nb
        h.close()
        steady = sorted(passes[1:], key=lambda p: p["total_s"])
        out = dict(steady[len(steady) // 2])
        out.update({"best_total_s": steady[0]["total_s"], "first_pass_total_s": passes[0]["total_s"], "ranks": world,
                    "steady_passes": {k: [p[k] for p in passes[1:]] for k in
                                      ("total_s", "reader_s", "reader_plus_upload_s", "gpu_step_s", "reader_cpu_s")},
                    "passes_reader_plus_upload_within_gpu_step": sum(p["reader_plus_upload_s"] <= p["gpu_step_s"] for p in passes[1:]),
                    "file": {"format": ".hic v8, float counts, 1000-bin blocks, zlib level 1, KR vector of ones",
                             "records_written": int(nrec), "bytes": int(size),
                             "pixel_kept_with_probability": "min(1, %g / (d + 1))" % keep},
                    "host_threads": os.cpu_count(),
                    "note": "open file -> final loop list on %d GPU(s): reader_s ends with the records in HBM, band_scatter_s = zero "
                            "fill + scatter (+ the exchange between ranks), then normalisation, fused kernels on this rank's blocks "
                            "+ device BH / selection / clustering + host tail (product mode) + the gather of the loops" % world})
        return out
    finally:
        if tmp:                     # rank 0 only; every rank is past the closing barrier of the last pass by now
            shutil.rmtree(tmp, ignore_errors=True)


def _dense_raw_block(w, block_index):
    import numpy as np
    s = w.start[block_index]
    CH, dpx = w.CH, w.dpx
    slab = w.band[:, s:s + CH].cpu().numpy()
    c = np.zeros((CH, CH))
    r = np.arange(CH)
    for d in range(dpx + 2):
        L = CH - d
        c[r[:L], r[:L] + d] = slab[d, :L]
    return c


def _oracle_block(args):
    """one block through the oracle's rows 3-7 (the checker and the CPU baseline; also runs in bench_extra's worker processes)"""
    c, dpx, want_set = args
    import numpy as np
    import oracle
    t0 = time.time()
    nz = oracle.block_prologue(c, dpx)
    ss = oracle.scale_space_levels(c, nz, [1.6, 3.2], blur="scipy")
    dt = time.time() - t0
    # (after the clock) the found set in the GPU records' terms: row-major pixel index, 1-based tested level, vAll, p-value
    hit = ss.pval != 2
    if not want_set:
        return dt, int(hit.sum()), int(nz.sum()), None, t0
    pix = np.flatnonzero(nz.ravel())[hit].astype(np.uint32)
    return dt, int(hit.sum()), int(nz.sum()), (pix, ss.level[hit].astype(np.uint32), ss.best[hit], ss.pval[hit]), t0


def _compare_block(w, bi, cset):
    """block bi of the workload through the HIP path -- launched with a neighbour, so that it RECEIVES the tiles it shares with the
    block before it and / or GIVES those it shares with the block after it -- against the oracle's found set of the same block:
    (pixels, levels and values identical, largest relative p-value difference, GPU found count)"""
    import numpy as np
    nb = len(w.start)
    idx = [i for i in (bi - 1, bi, bi + 1) if 0 <= i < nb]
    g = w.pipe.engine.sigma_loop_band(w.band, w.n, w.dpx, [w.start[i] for i in idx], w.CH, skip_empty=False, with_q=False)[0][idx.index(bi)]
    cpix, clvl, cval, cp = cset
    same = (len(g["pixel"]) == len(cpix) and np.array_equal(g["pixel"], cpix) and np.array_equal(g["level"], clvl)
            and np.array_equal(g["value"], cval))
    p_err = float(np.max(np.abs(g["pval"] - cp) / np.maximum(cp, 1e-300))) if same and len(cpix) else None
    return bool(same), p_err, len(g["pixel"])


def cpu_baseline(w, value):
    """The oracle (reference arithmetic: SciPy gaussian_filter / maximum_filter / expm1), rows 3-7, beside the GPU step:
    (a) the middle block in this process, nothing else running: the 1-core timing sample;
    (b) 8 blocks in 4 worker processes at once -- the reference's default `-p 4` (mustache.py:146), what BASELINE.md's
        ">= 50 x" is stated against -- timed from the first worker's start to the last worker's end (process start-up and the
        hand-over of the blocks are outside), scaled linearly to the chromosome.
    Three of those blocks are also CHECKED against the HIP path at full size: the first (mask_size -1), the middle one and the
    last (right-aligned, mask_size > the distance limit, mustache.py:909-910, :948-953) -- whole found sets, levels and
    values with array_equal, p-values to 1e-9."""
    import multiprocessing as mp
    nb = len(w.start)
    bi = nb // 2
    cpu_s, cpu_found, cpu_nz, cset, _ = _oracle_block((_dense_raw_block(w, bi), w.dpx, True))
    checks = {bi: _compare_block(w, bi, cset)}
    edge = [0, nb - 1] if nb >= 5 else []
    inner = [i for i in (nb // 8, nb // 4, 3 * nb // 8, 5 * nb // 8, 3 * nb // 4, 7 * nb // 8) if i not in (0, bi, nb - 1)]
    sub = edge + sorted(set(inner))[:8 - len(edge)]
    p4_error = None
    try:
        with mp.get_context("spawn").Pool(4) as pool:
            res = pool.map(_oracle_block, [(_dense_raw_block(w, i), w.dpx, i in edge) for i in sub], chunksize=1)
    except Exception as e:        # a box that cannot start worker processes still gets its three-block check (inline, 1 core)
        p4_error = repr(e)
        sub = list(edge)
        res = [_oracle_block((_dense_raw_block(w, i), w.dpx, True)) for i in sub]
    wall4 = (max(r[4] + r[0] for r in res) - min(r[4] for r in res)) if res else float("nan")
    for i, r in zip(sub, res):
        if i in edge:
            checks[i] = _compare_block(w, i, r[3])
    compared = sorted(checks)
    perr = [checks[i][1] for i in compared if checks[i][1] is not None]
    one = w.CH * w.CH / 1e6 / cpu_s
    p4 = one if p4_error else len(sub) * w.CH * w.CH / 1e6 / wall4
    cpu = {"value": round(one, 4), "unit": "Mpix/s", "cores": 1, "kind": "port",
           "sample": "block %d of the same workload (one 4000x4000 block, %.1f s), rows 3-7 of the oracle = the reference's "
                     "SciPy calls, single process.  NOT in this baseline: the reference's normalize_sparse (row 1) and its "
                     "tail (rows 8-9); the GPU side of the ratio (`value`) is rows 2-7" % (bi, cpu_s),
           "found_pixels_cpu": cpu_found, "found_pixels_gpu": checks[bi][2],
           "blocks_compared": compared,
           "found_set_pixels_levels_values_identical": all(checks[i][0] for i in compared),
           "found_pixels_per_block_compared": [checks[i][2] for i in compared],
           "pvalue_max_rel_err": max(perr) if perr else None,
           "p4": {"value": round(p4, 4), "unit": "Mpix/s", "procs": 1 if p4_error else 4, "blocks": len(sub), "block_indices": sub,
                  "scaled": "rate of %d of the %d blocks, taken as the chromosome's" % (len(sub), nb),
                  "wall_s": round(wall4, 2), "core_s_per_block": round(sum(r[0] for r in res) / max(1, len(res)), 2),
                  "error": p4_error,
                  "sample": "the reference's default -p 4 (mustache.py:146): %d blocks in 4 worker processes, two rounds; first "
                            "worker start -> last worker end, same rows 3-7 per block as the 1-core leg" % len(sub)},
           "cpu_model": _cpu_model(), "host_cores": os.cpu_count()}
    return cpu, round(value / cpu["value"], 1), round(value / cpu["p4"]["value"], 1)


def _cpu_model():
    try:
        return next(l.split(":", 1)[1].strip() for l in open("/proc/cpuinfo") if l.startswith("model name"))
    except Exception:
        return "unknown"


def _free_port():
    import socket
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        return sk.getsockname()[1]


def self_launch(args):
    """`python bench.py --gpus N` with N > 1 and no torchrun environment (WORLD_SIZE unset): start the N ranks ourselves
    (torch.distributed.run on 127.0.0.1, a port that is free right now) and hand their exit code on."""
    import subprocess
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus), "--master-addr",
           "127.0.0.1", "--master-port", str(_free_port()), os.path.abspath(__file__)] + sys.argv[1:]
    print("bench.py: --gpus %d without a torchrun environment: launching `%s`" % (args.gpus, " ".join(cmd)), file=sys.stderr,
          flush=True)
    raise SystemExit(subprocess.call(cmd))


def pmc_traffic(px_computed):
    """HBM bytes per launch from the latest committed PMC session (profiles/rNN_pmc_traffic.json: FETCH_SIZE x 2 on gfx950 +
    WRITE_SIZE of the fused kernel, its own rocprofv3 --pmc pass), scaled to this launch's computed pixels."""
    import glob
    import re
    cands = sorted(f for f in glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_traffic.json"))
                   if re.fullmatch(r"r\d+_pmc_traffic\.json", os.path.basename(f)))
    try:
        pj = json.load(open(cands[-1]))
        bpp = pj.get("bytes_per_computed_pixel", pj["bytes_per_pixel"])
        return round(bpp * px_computed), "profiles/%s (%s; launch form: %s), %.2f B per computed pixel" % (
            os.path.basename(cands[-1]), pj.get("command", "?"), pj.get("launch_form", "dense, MST_FLAG_NO_SHARE"), bpp)
    except Exception:
        return None, None


def small_shape_legs(device):
    """Second half of the metric's name: chr21 @ 5 kb on 1 GPU (6 blocks of 2000 x 2000, one launch), same timed region; and
    the two-sample call (diff_mustache.py:260-569) on the same shape.  Each call ends with its results on
    the host: wall time per call, median of many (a 1.8 ms step is at the mercy of single host hiccups in a mean)."""
    import torch
    from mustache_amd.diff_mustache import _pairs_from_filled
    w5 = Workload("chr21@5kb synthetic", 9630, 400, 5000, 300.0, 300, 0, device, 0, 1)
    band_b, _ = make_band(9630, 400, 260.0, 300, 7, 5000, device)

    def per_call_ms(fn, reps, warm=3):
        ts = []
        for i in range(warm + reps):
            t0 = time.time()
            fn()
            if i >= warm:
                ts.append((time.time() - t0) * 1e3)
        torch.cuda.synchronize()
        ts.sort()
        return {"median": round(ts[len(ts) // 2], 3), "min": round(ts[0], 3), "p90": round(ts[int(0.9 * len(ts))], 3), "calls": len(ts)}

    one = per_call_ms(lambda: w5.step(False), 40)
    two = per_call_ms(lambda: _pairs_from_filled(w5.pipe.engine, w5.pipe, [w5.band, band_b], w5.n, w5.dpx, w5.start, w5.CH, pt=0.1), 30)
    pairs_s = w5.total_mpix * 1e6 / (two["median"] * 1e-3)
    # per pixel pair: both sigma loops (2 x 1152 flops on the launched share of the tiles) + the difference image's G_2, G_3
    # in both octaves on every tile (radii 4, 4, 7, 8: 146 flops)
    frac_tiles = w5.pipe.engine.band_tile_fraction(w5.CH, w5.dpx)
    tf = pairs_s * (2 * FLOPS_PER_PIXEL * frac_tiles + 146.0) / 1e12
    return ({"value": round(w5.total_mpix / (one["median"] * 1e-3), 1), "unit": "Mpix/s", "blocks": len(w5.start),
             "chunk": w5.CH, "ms_per_step": one},
            {"value": round(pairs_s / 1e6, 1), "unit": "Mpix-pairs/s", "block_pairs": len(w5.start), "chunk": w5.CH,
             "ms_per_call": two,
             "roofline": {"bound": "fp64_valu", "launched_tile_fraction": round(frac_tiles, 4), "achieved": round(tf, 3),
                          "peak": PEAK_TF, "unit": "TFLOP/s", "frac": round(tf / PEAK_TF, 4),
                          "note": "whole two-sample call, wall clock (both sigma loops with tile lists, mst_diff_dog_band, pair "
                                  "p-values, BH + selection + partner look-ups on the device, selected records to the host)"}})


def main():
    args = parse()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        self_launch(args)
    import torch
    import torch.distributed as dist
    from mustache_amd.sharding import shard_blocks
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:      # the launcher's environment is what the processes really are: say so and go on
        print("bench.py: --gpus %d but WORLD_SIZE=%d: running with %d rank(s)" % (args.gpus, world, world), file=sys.stderr, flush=True)
    # test hooks (the N > 1 control flow on a single-GPU box): a gloo process group and all ranks on GPU 0
    backend = os.environ.get("MST_BENCH_BACKEND", "nccl")
    if os.environ.get("MST_BENCH_ONE_DEVICE"):
        local = 0
    torch.cuda.set_device(local)
    device = torch.device("cuda", local)
    from mustache_amd.engine import settle_gc
    settle_gc()      # as the command-line entry points do (the library itself never freezes the collector)
    # MST_BENCH_FORCE_DIST=1: a process group even with one rank, so that the RCCL calls of the N > 1 path run on a 1-GPU box
    force_dist = bool(os.environ.get("MST_BENCH_FORCE_DIST")) and world == 1
    grouped = world > 1 or force_dist
    if grouped:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if "MASTER_PORT" not in os.environ:
            if world > 1:
                raise SystemExit("WORLD_SIZE=%d without MASTER_PORT: launch with torch.distributed.run, or run `python "
                                 "bench.py --gpus N` without any torchrun variable and bench.py starts the ranks itself" % world)
            os.environ["MASTER_PORT"] = str(_free_port())          # 1-rank group (MST_BENCH_FORCE_DIST): any free port
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=device)   # RCCL over xGMI
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)

    n = 248957 if not args.small else 4000 + 11 * 2000
    w = Workload("chr1@1kb synthetic (n=%d, dpx=2000, %s blocks of 4000x4000 fp64)", n, 2000, 1000, 400.0,
                 8000 if not args.small else 800, 1, device, rank, world, scaling=args.scaling)
    nb = len(w.start)
    w.name = w.name % (n, nb)
    if world > 1 and args.scaling == "weak":
        w.name = "%d x %s, one chromosome per rank" % (world, w.name)
    cx = Ctx(w, args, device, rank, world, grouped, backend)
    w.step(False)          # set-up, untimed: first-touch of the pinned staging buffers and the allocator's block cache

    dt, kms, last, owns = cx.timed(False, args.steps, args.warmup)
    ms_per_step = dt / args.steps * 1e3
    value = w.total_mpix / (dt / args.steps)

    # roofline of the dominant kernel (this rank's launches): algorithmic flops per launch / event-timed duration
    launches_per_step = max(1, len(kms) // args.steps)
    k_ms = sum(kms) / len(kms)
    # every rank's own view on stderr, so that a partial failure of an N > 1 run can be told from the log
    print("\nRANK_FRAGMENT " + json.dumps({"rank": rank, "world": world, "device": local, "backend": backend if grouped else None,
                                         "blocks": len(w.mine), "own_ms_per_step": round(owns[rank if grouped else 0] / args.steps * 1e3, 3),
                                         "kernel_ms_mean": round(k_ms, 3), "job_ms_per_step": round(ms_per_step, 3)}),
          file=sys.stderr, flush=True)
    px_per_launch = len(w.mine) * w.CH * w.CH / launches_per_step
    # Work really executed: a tile inside two overlapping blocks of a launch is computed ONCE, so the workgroups cover work_items
    # / tiles of the blocks' pixels.  The roofline prices those only (skipped work is a separate speed-up).
    wi = work_items(w, False)
    px_computed = px_per_launch * wi[0] / float(wi[1])
    achieved_tf = px_computed * FLOPS_PER_PIXEL / (k_ms * 1e-3) / 1e12
    credited_tf = px_per_launch * FLOPS_PER_PIXEL / (k_ms * 1e-3) / 1e12
    achieved_gbs = px_per_launch * BYTES_PER_PIXEL / (k_ms * 1e-3) / 1e9
    exec_fpp = executed_flops_per_pixel()
    traffic, traffic_source = pmc_traffic(px_computed)
    roof = {"bound": "fp64_valu", "achieved": round(achieved_tf, 3), "peak": PEAK_TF, "unit": "TFLOP/s",
            "frac": round(achieved_tf / PEAK_TF, 4), "traffic": traffic, "traffic_source": traffic_source,
            "kernel": "scale_space_kernel<Tile<32,64,14>, band>", "kernel_ms": round(k_ms, 3),
            "launches_per_step": launches_per_step, "kernel_ms_per_step": round(k_ms * launches_per_step, 3),
            "pixels_per_launch": int(px_per_launch), "computed_pixels_per_launch": int(round(px_computed)),
            "flops_per_pixel": FLOPS_PER_PIXEL, "executed_flops_per_pixel": round(exec_fpp, 1),
            "executed_frac": round(px_computed * exec_fpp / (k_ms * 1e-3) / 1e12 / PEAK_TF, 4),
            "work_items": wi[0], "tiles": wi[1], "shared_tiles": wi[2], "work_items_per_launch": wi[3],
            "block_pixel_view": {"achieved": round(credited_tf, 3), "frac": round(credited_tf / PEAK_TF, 4),
                                 "note": "every BLOCK pixel credited (a shared tile counted twice): the view of `value`, not efficiency"},
            "note": "peak = FP64 add/mul rate without FMA (78.6 / 2 at the 2.4 GHz spec clock; contraction would break bit-identity "
                    "with SciPy).  achieved = 1152 blur flops x the pixels of the workgroups that RAN (pixels_per_launch x work_items "
                    "/ tiles) / launch time (HIP events on the launch stream); executed_* adds halo columns + ring, drops the two "
                    "repeated levels; max / sieve / statistics are not counted",
            "hbm_model": {"bound": "hbm", "bytes_per_pixel_model": BYTES_PER_PIXEL, "achieved_equivalent": round(achieved_gbs, 1),
                          "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac_of_model_roofline": round(achieved_gbs / HBM_PEAK_GBS, 4),
                          "note": "SURVEY 8d's level-streaming model: the rate the kernel WOULD need if it wrote and re-read every "
                                  "level -- it holds all levels on chip instead, see traffic"}}

    # the same step with empty tiles skipped (the product mode: identical results, a separate speed-up, never `value`)
    hs = max(1, args.steps // 2)
    frac = w.pipe.engine.band_tile_fraction(w.CH, w.dpx)
    dt_s, kms_s, _, _ = cx.timed(True, hs, 1)
    wis = work_items(w, True)
    bs_credit = len(w.mine) * w.CH * w.CH * frac * FLOPS_PER_PIXEL / (sum(kms_s) / hs * 1e-3) / 1e12
    band_skip = {"value": round(w.total_mpix / (dt_s / hs), 1), "unit": "Mpix/s", "speedup": round((dt / args.steps) / (dt_s / hs), 3),
                 "kernel_ms_per_step": round(sum(kms_s) / hs, 3), "launched_tile_fraction": round(frac, 4),
                 "roofline": {"bound": "fp64_valu", "peak": PEAK_TF, "unit": "TFLOP/s",
                              "achieved": round(bs_credit * wis[0] / wis[1], 3), "frac": round(bs_credit * wis[0] / wis[1] / PEAK_TF, 4),
                              "work_items": wis[0], "tiles": wis[1], "shared_tiles": wis[2], "work_items_per_launch": wis[3],
                              "block_pixel_view": {"achieved": round(bs_credit, 3), "frac": round(bs_credit / PEAK_TF, 4)},
                              "note": "1152 flops per pixel of the band tiles the workgroups really computed over the kernel time"},
                 "note": "only the tiles that can reach the tested band are launched; `value` / `roofline` are always the dense run"}

    # both again with every tile computed once PER BLOCK (MST_FLAG_NO_SHARE): identical records; the kernel's own rate
    w.pipe.engine.share_tiles = False
    dt_n, kms_n, _, _ = cx.timed(False, hs, 1)
    dt_ns, kms_ns, _, _ = cx.timed(True, hs, 1)
    w.pipe.engine.share_tiles = True
    kn, kns = sum(kms_n) / hs, sum(kms_ns) / hs
    tf_n = len(w.mine) * w.CH * w.CH * FLOPS_PER_PIXEL / (kn * 1e-3) / 1e12
    tf_ns = len(w.mine) * w.CH * w.CH * frac * FLOPS_PER_PIXEL / (kns * 1e-3) / 1e12
    no_share = {"value": round(w.total_mpix / (dt_n / hs), 1), "unit": "Mpix/s", "kernel_ms_per_step": round(kn, 3),
                "roofline": {"bound": "fp64_valu", "achieved": round(tf_n, 3), "peak": PEAK_TF, "unit": "TFLOP/s", "frac": round(tf_n / PEAK_TF, 4)},
                "band_skip": {"value": round(w.total_mpix / (dt_ns / hs), 1), "unit": "Mpix/s", "kernel_ms_per_step": round(kns, 3),
                              "roofline": {"bound": "fp64_valu", "achieved": round(tf_ns, 3), "peak": PEAK_TF, "unit": "TFLOP/s",
                                           "frac": round(tf_ns / PEAK_TF, 4)}},
                "work_items_per_launch": {"dense": work_items(w, False, share=False)[3], "band_skip": work_items(w, True, share=False)[3]},
                "note": "MST_FLAG_NO_SHARE: every workgroup's flops are algorithmic flops of one block -- the kernel's own efficiency"}
    tile_sharing = {"speedup": round(value / no_share["value"], 4), "kernel_speedup": round(kn / (k_ms * launches_per_step), 4),
                    "band_skip_speedup": round(band_skip["value"] / no_share["band_skip"]["value"], 4),
                    "work_items": wi[0], "tiles": wi[1], "shared_tiles": wi[2],
                    "note": "blocks overlap by half their edge at 1 kb (mustache.py:899-908); a tile inside two blocks of a launch is "
                            "computed once, its records / statistics delivered to both.  `value` is WITH sharing, no_share without"}

    strong = args.scaling == "strong"
    ranges = [shard_blocks(nb, r, world) for r in range(world)]
    per_rank = [len(r) for r in ranges] if strong else [nb] * world
    out = {"metric": "scale-space Mpix/s (sigma-stack+local-max)", "value": round(value, 1), "unit": "Mpix/s",
           "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms_per_step, 3),
           "higher_is_better": True, "scaling": args.scaling, "vs_baseline": None, "dtype": "f64", "data": "synthetic",
           "config": {"workload": w.name, "blocks": nb, "chunk": w.CH, "distance_px": w.dpx,
                      "megapixels_per_step": round(w.total_mpix, 1), "partition": "blocks" if strong else "chromosomes",
                      "sharding": ("the %d blocks of ONE chromosome in contiguous ranges over %d rank(s): %s" % (
                          nb, world, ", ".join("rank %d = [%d, %d)" % (r, b[0], b[-1] + 1) for r, b in enumerate(ranges) if b)))
                      if strong else ("one whole chromosome (%d blocks) per rank, %d rank(s)" % (nb, world)),
                      "timed_region": "normalised band in HBM -> fused kernel (blocks cut, filled, masked in-kernel; sigma loop, sieve, "
                                      "level statistics) -> p-values -> found records of every block on the host; one launch in %d "
                                      "stages per step, the download of one stage's blocks under the kernel of the next" % len(w.groups)},
           "ranks": {"ms_per_step_max": round(max(owns) / args.steps * 1e3, 3), "ms_per_step_min": round(min(owns) / args.steps * 1e3, 3),
                     "blocks_per_rank_max": max(per_rank), "blocks_per_rank_min": min(per_rank),
                     "imbalance_bound": round(max(per_rank) * world / float(sum(per_rank)), 4),
                     "efficiency_bound": round(sum(per_rank) / float(world * max(per_rank)), 4),
                     "efficiency_bound_at": {str(k): round(nb / (k * -(-nb // k)), 4) for k in (1, 2, 4, 8)},
                     "note": "contiguous split: the slowest rank carries ceil(blocks / ranks) blocks -> efficiency_bound_at"},
           "roofline": roof, "tile_sharing": tile_sharing, "band_skip": band_skip, "no_share": no_share,
           "normalize_ms_untimed": round(w.normalize_s * 1e3, 2)}

    if world > 1:
        # the other partitioning, timed after the headline (every rank takes part): same step, same timed region, max over ranks
        other = "weak" if strong else "strong"
        w.set_scaling(other)
        dt_o, _, _, owns_o = cx.timed(False, args.steps, 2)
        out["other_scaling"] = {"scaling": other, "value": round(w.total_mpix / (dt_o / args.steps), 1), "unit": "Mpix/s",
                                "ms_per_step": round(dt_o / args.steps * 1e3, 3), "megapixels_per_step": round(w.total_mpix, 1),
                                "blocks_on_this_rank": len(w.mine),
                                "ms_per_step_max": round(max(owns_o) / args.steps * 1e3, 3),
                                "ms_per_step_min": round(min(owns_o) / args.steps * 1e3, 3),
                                "note": "weak = a whole chromosome per rank (a genome run); strong = ONE chromosome's blocks over the ranks"}
        w.set_scaling(args.scaling)

    solo = rank == 0 and world == 1
    if solo:
        out["ranks"].update(share_projection(w, ms_per_step))
    if solo:
        # informational: the whole per-chromosome run from the normalised band (rows 2-9, product mode) -- NOT part of `value`
        w.pipe.run_band(w.band, w.n, w.dpx, 0.88, 0.1, distributed=False)      # first call: staging buffers, allocator
        runs = []
        for _ in range(3):
            tm = {}
            torch.cuda.synchronize()
            t0 = time.time()
            loops = w.pipe.run_band(w.band, w.n, w.dpx, 0.88, 0.1, timings=tm, distributed=False)
            torch.cuda.synchronize()
            runs.append((time.time() - t0, tm.get("tail_s", 0.0)))
        runs.sort()
        out["end_to_end"] = {"rows_2_to_9_s": round(runs[1][0], 3), "normalize_s": round(w.normalize_s, 3),
                             "tail_s": round(runs[1][1], 3), "loops": len(loops), "all_runs_s": [round(r[0], 3) for r in runs],
                             "note": "synthetic chr1@1kb from the normalised band to the final loop list, 1 GPU (median of 3)"}
        out["chr21_5kb"], out["diff_chr21_5kb"] = small_shape_legs(device)
    extra = None
    if args.extra and solo:
        extra = __import__("scripts.bench_extra", fromlist=["Extra"]).Extra(sys.modules[__name__], cx, out, value, no_share)
        extra.before_file_leg()
    if not args.no_file:
        # every rank takes part (N > 1: each inflates its share of the file, the shares are exchanged); rank 0 reports
        out["end_to_end_from_file"] = fl = file_leg(cx, on_band=extra.sparse_step if extra else None)
        out["ranks"]["read_s"] = fl["read_s_per_rank"]
    if solo and not args.no_cpu:
        out["cpu_baseline"], out["speedup_vs_cpu_1core"], out["speedup_vs_cpu_p4"] = cpu_baseline(w, value)
        if extra:
            extra.cpu_pools()
    if rank == 0:
        print(json.dumps(out))
    if grouped:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
