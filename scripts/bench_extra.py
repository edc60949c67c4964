#!/usr/bin/env python3
"""Side legs of bench.py (`python bench.py --extra`, one GPU): everything that is not the driver's line -- FMA mode, the COO
scatter of row 2, normalisation against HBM, a whole hg19-shaped genome at 5 kb (one and two samples), the wide-radius and
large-window instantiations, the thinned band of the file leg through the headline step, and the -p 4 / per-node CPU legs.
bench.py constructs `Extra` and calls its three hooks; every leg adds one key to the same JSON line."""
import os
import time

# hg19 chromosome lengths (chr1..22, X, Y), bp: the shape of BASELINE configs 3 and 5 (whole genome at 5 kb)
HG19 = [249250621, 243199373, 198022430, 191154276, 180915260, 171115067, 159138663, 146364022, 141213431, 135534747,
        135006516, 133851895, 115169878, 107349540, 102531392, 90354753, 81195210, 78077248, 59128983, 63025520, 48129895,
        51304566, 155270560, 59373566]


def genome_workload(B, name, res, dpx, depth, seed, device, sizes=HG19, two_samples=False):
    """All chromosomes of a synthetic hg19-shaped genome at `res` in ONE band (pipeline.GenomeLayout): the blocks of every
    chromosome go through the same launches.  Same step() as the single-chromosome workload (B = the bench module)."""
    import torch
    from mustache_amd.pipeline import ChromosomePipeline, GenomeLayout

    class GenomeWorkload(B.Workload):
        def __init__(self):
            self.name, self.dpx, self.res = name, dpx, res
            self.pipe = ChromosomePipeline((1.6, 3.2), device=device)
            ns = [-(-s // res) for s in sizes]
            self.layout = lay = GenomeLayout(ns, dpx)
            bands = [[], []]
            self.normalize_s = 0.0
            for c, n in enumerate(ns):
                for smp in range(2 if two_samples else 1):
                    b, t = B.make_band(n, dpx, depth * (1.0 if smp == 0 else 0.87), max(30, n // 30), seed + 100 * smp + c, res,
                                       device, reps=1)
                    bands[smp].append(b)
                    self.normalize_s += t
            self.band = lay.band(bands[0], device)
            self.band2 = lay.band(bands[1], device) if two_samples else None
            del bands
            torch.cuda.empty_cache()
            self.n, self.CH = lay.N, lay.CH
            self.start = [g[3] for g in lay.blocks]
            self.end = None
            self.mine = list(range(len(self.start)))
            self.total_mpix = len(self.start) * self.CH * self.CH / 1e6
            self.kernel_ms = []

    return GenomeWorkload()


def _pool(B, w, block_indices, procs):
    """`procs` worker processes over the given blocks (the reference's default parallelism is -p 4, mustache.py:146)."""
    import multiprocessing as mp
    blocks = [(B._dense_raw_block(w, i), w.dpx, False) for i in block_indices]
    t0 = time.time()
    with mp.get_context("spawn").Pool(procs) as pool:
        pool.map(B._oracle_block, blocks, chunksize=1)
    return time.time() - t0


class Extra:
    def __init__(self, B, cx, out, value, no_share):
        self.B, self.cx, self.out, self.value, self.no_share = B, cx, out, value, no_share

    def sparse_step(self, nb, n, steps=3):
        """The fused step on the THINNED band the file leg read (~30 % of the band's pixels tested instead of ~67 %): the same
        timed region as `value`, dense and tile list -- the kernel's time does not depend on how many pixels are tested."""
        import copy
        import torch
        w = self.cx.w
        if n != w.n:
            return
        w2 = copy.copy(w)
        w2.band, w2.kernel_ms = nb, []
        res = {"tested_share_of_band": round(float((nb[:w.dpx + 1] > 0).sum().item()) / float((w.dpx + 1) * n), 4)}
        for key, skip in (("dense", False), ("band_skip", True)):
            w2.step(skip_empty=skip)
            torch.cuda.synchronize()
            w2.kernel_ms.clear()
            t0 = time.time()
            for _ in range(steps):
                w2.step(skip_empty=skip)
            torch.cuda.synchronize()
            dt = (time.time() - t0) / steps
            res[key] = {"value": round(w.total_mpix / dt, 1), "unit": "Mpix/s", "ms_per_step": round(dt * 1e3, 2),
                        "kernel_ms_per_step": round(sum(a.elapsed_time(b) for a, b in w2.kernel_ms) / steps, 2)}
        self.out["sparse_1kb"] = res

    def cpu_pools(self):
        """one process per core, capped (MST_BENCH_CPU_PROCS, default 16; 0 switches the leg off)"""
        B, w, out, value = self.B, self.cx.w, self.out, self.value
        bi = len(w.start) // 2          # (the -p 4 leg is on the default line: cpu_baseline.p4, speedup_vs_cpu_p4)
        P = min(int(os.environ.get("MST_BENCH_CPU_PROCS", "16")), len(w.start), os.cpu_count() or 1)
        if P > 0:
            wallp = _pool(B, w, [(bi + j) % len(w.start) for j in range(P)], P)
            out["cpu_baseline_node"] = {"value": round(P * w.CH * w.CH / 1e6 / wallp, 3), "unit": "Mpix/s", "cores": P,
                                        "kind": "port", "host_cores": os.cpu_count(),
                                        "sample": "%d blocks of the same workload in %d worker processes at once, wall %.1f s "
                                        "incl. process start-up" % (P, P, wallp)}
            out["speedup_vs_cpu_node"] = round(value / out["cpu_baseline_node"]["value"], 1)

    def before_file_leg(self):
        import torch
        B, cx, out, value, no_share = self.B, self.cx, self.out, self.value, self.no_share
        w, args, device = cx.w, cx.args, cx.device
        FLOPS_PER_PIXEL, peak_tf, HBM_PEAK_GBS = B.FLOPS_PER_PIXEL, B.PEAK_TF, B.HBM_PEAK_GBS
        hs = max(1, args.steps // 2)
        # opt-in relaxed arithmetic (fused multiply-add per tap pair): DoG no longer bit-identical (~1e-16 relative, north_star
        # allows 1e-5), found set unchanged on every case tested.  Never the headline.
        dt_f, kms_f, _, _ = cx.timed(False, hs, 1, fma=True)
        out["fma_mode"] = {"value": round(w.total_mpix / (dt_f / hs), 1), "unit": "Mpix/s",
                           "kernel_ms_per_step": round(sum(kms_f) / hs, 3),
                           "note": "MST_FLAG_FMA, dense; not bit-exact DoG, therefore never the headline value"}
        # row 1 of SURVEY 8a: 16 B per band sample (8 read + 8 written) over mst_normalize_band (HIP events, median)
        gbs = 16.0 * (w.dpx + 2) * w.n / w.normalize_s / 1e9
        out["normalize_roofline"] = {"bound": "hbm", "unit": "GB/s", "peak": HBM_PEAK_GBS, "achieved": round(gbs, 1),
                                     "frac": round(gbs / HBM_PEAK_GBS, 4)}
        dt = w.total_mpix / value * args.steps
        # SURVEY 8d defines the metric's timed region from "normalised COO resident on device", i.e. including row 2's scatter
        # (mustache.py:919-924, `cc[xc, yc] = vc`).  Here normalisation runs on the band, so the step above starts one stage
        # later; this leg times that stage for the same chromosome -- the normalised band's non-zero samples as an int64 /
        # int64 / float64 COO on the device -> mst_band_from_coo (zero fill + scatter) -- and adds it to the step
        from mustache_amd.normalize import band_from_coo
        xs, ys, vs = [], [], []
        cols = 1 << 15
        for i0 in range(0, w.n, cols):
            sl = w.band[:, i0:i0 + cols]
            dd, cc = torch.nonzero(sl, as_tuple=True)
            xs.append(cc + i0)
            ys.append(cc + i0 + dd)
            vs.append(sl[dd, cc])
            del dd, cc, sl
        qx, qy, qv = torch.cat(xs), torch.cat(ys), torch.cat(vs)
        del xs, ys, vs
        sc_ms = []
        for it in range(4):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            rebuilt = band_from_coo(qx, qy, qv, w.n, w.dpx)
            e1.record()
            torch.cuda.synchronize()
            sc_ms.append(e0.elapsed_time(e1))
            if it == 0:
                same_band = bool(torch.equal(rebuilt, w.band))
            del rebuilt
        sc = sorted(sc_ms[1:])[1] * 1e-3
        out["row2_scatter"] = {"records": int(qv.numel()), "ms": round(sc * 1e3, 3), "band_rebuilt_identical": same_band,
                               "GB/s": round((24.0 * qv.numel() + 8.0 * w.band.numel()) / sc / 1e9, 1),
                               "value_from_coo": round(w.total_mpix / (dt / args.steps + sc), 1), "unit": "Mpix/s",
                               "note": "mst_band_from_coo on the chromosome's normalised COO (int64 x, int64 y, float64 v on the "
                                       "device; 24 B read per record + the 8 B/sample zero fill of the band), median of 3, HIP "
                                       "events; value_from_coo = megapixels / (step + this) = the metric with SURVEY 8d's timed "
                                       "region 'normalised COO resident on device -> found records on host'"}
        del qx, qy, qv
        torch.cuda.empty_cache()
        # BASELINE configs 3 and 5: a whole hg19-shaped genome at 5 kb (24 chromosomes, ~390 blocks of 2000 x 2000) with all
        # chromosomes side by side in one band (pipeline.GenomeLayout) -- every launch carries blocks of many chromosomes
        wg = genome_workload(B, "hg19-shaped genome @5kb synthetic", 5000, 400, 300.0, 1000, device, two_samples=True)
        gsteps = 3
        for _ in range(2):
            wg.step(False)
        wg.kernel_ms.clear()
        torch.cuda.synchronize()
        t0 = time.time()
        for _ in range(gsteps):
            wg.step(False)
        torch.cuda.synchronize()
        g_dt = (time.time() - t0) / gsteps
        g_kms = sum(a.elapsed_time(b) for a, b in wg.kernel_ms) / gsteps
        wg.step(True)
        wg.kernel_ms.clear()
        torch.cuda.synchronize()
        t0 = time.time()
        for _ in range(gsteps):
            wg.step(True)
        torch.cuda.synchronize()
        g_dt_s = (time.time() - t0) / gsteps
        g_kms_s = sum(a.elapsed_time(b) for a, b in wg.kernel_ms) / gsteps
        g_frac = wg.pipe.engine.band_tile_fraction(wg.CH, wg.dpx)
        g_tf = wg.total_mpix * 1e6 * FLOPS_PER_PIXEL / (g_kms * 1e-3) / 1e12
        g_tf_s = wg.total_mpix * 1e6 * g_frac * FLOPS_PER_PIXEL / (g_kms_s * 1e-3) / 1e12
        wg.pipe.run_layout(wg.layout, wg.band, 0.88, 0.1)
        gruns = []
        for _ in range(3):
            tmg = {}
            torch.cuda.synchronize()
            t0 = time.time()
            gl = wg.pipe.run_layout(wg.layout, wg.band, 0.88, 0.1, timings=tmg)
            torch.cuda.synchronize()
            gruns.append((time.time() - t0, tmg))
        gruns.sort(key=lambda r: r[0])
        g_e2e, tmg = gruns[1]
        out["genome_5kb"] = {"value": round(wg.total_mpix / g_dt, 1), "unit": "Mpix/s", "chromosomes": len(HG19),
                             "blocks": len(wg.start), "chunk": wg.CH, "band_columns": wg.n,
                             "megapixels_per_step": round(wg.total_mpix, 1), "ms_per_step": round(g_dt * 1e3, 3),
                             "vs_chr1_1kb_value": round(wg.total_mpix / g_dt / value, 4),
                             "vs_chr1_1kb_no_share_value": round(wg.total_mpix / g_dt / no_share["value"], 4),
                             "roofline": {"bound": "fp64_valu", "achieved": round(g_tf, 3), "peak": peak_tf, "unit": "TFLOP/s",
                                          "frac": round(g_tf / peak_tf, 4), "kernel_ms_per_step": round(g_kms, 3)},
                             "band_skip": {"value": round(wg.total_mpix / g_dt_s, 1), "unit": "Mpix/s",
                                           "launched_tile_fraction": round(g_frac, 4),
                                           "roofline": {"bound": "fp64_valu", "achieved": round(g_tf_s, 3), "peak": peak_tf,
                                                        "unit": "TFLOP/s", "frac": round(g_tf_s / peak_tf, 4),
                                                        "kernel_ms_per_step": round(g_kms_s, 3)}},
                             "end_to_end": {"rows_2_to_9_s": round(g_e2e, 3), "tail_s": round(tmg.get("tail_s", 0.0), 3),
                                            "launches": tmg.get("launches"), "loops": sum(len(o) for o in gl),
                                            "normalize_s_all_chromosomes": round(wg.normalize_s, 4)},
                             "note": "same timed region as `value` (dense step: fused kernel, p-values, found records to the "
                                     "host), all chromosomes' blocks batched into the same launches; band_skip / end_to_end = "
                                     "the product mode (tile lists; + BH, selection, filters, clustering, overlap masks).  At 5 kb "
                                     "blocks of 2000 overlap by 400 bins only, so tile sharing saves ~4 % here against ~22-30 % at "
                                     "1 kb: vs_chr1_1kb_no_share_value is the like-for-like ratio (launch-boundness), "
                                     "vs_chr1_1kb_value includes the 1 kb run's sharing"}
        # two-sample whole genome (config 5): every block pair of every chromosome in ONE run_band_pairs call
        eng = wg.pipe.engine
        for _ in range(2):
            eng.run_band_pairs([wg.band, wg.band2], wg.n, wg.dpx, wg.start, wg.CH, select_below=0.1)
        torch.cuda.synchronize()
        t0 = time.time()
        for _ in range(gsteps):
            eng.run_band_pairs([wg.band, wg.band2], wg.n, wg.dpx, wg.start, wg.CH, select_below=0.1)
        torch.cuda.synchronize()
        gp_s = wg.total_mpix * 1e6 / ((time.time() - t0) / gsteps)
        from mustache_amd.diff_mustache import run_pair_layout
        run_pair_layout(wg.pipe, wg.layout, [wg.band, wg.band2], 0.88, 0.1, 0.1)
        pruns = []
        for _ in range(3):
            torch.cuda.synchronize()
            t0 = time.time()
            rows = run_pair_layout(wg.pipe, wg.layout, [wg.band, wg.band2], 0.88, 0.1, 0.1)
            torch.cuda.synchronize()
            pruns.append(time.time() - t0)
        gp_e2e = sorted(pruns)[1]
        gp_flops = 2 * FLOPS_PER_PIXEL * g_frac + 146.0
        out["diff_genome_5kb"] = {"value": round(gp_s / 1e6, 1), "unit": "Mpix-pairs/s", "block_pairs": len(wg.start),
                                  "chunk": wg.CH, "chromosomes": len(HG19),
                                  "end_to_end": {"rows_2_to_9_both_samples_s": round(gp_e2e, 3),
                                                 "tagged_rows": sum(len(o) for o in rows),
                                                 "note": "run_pair_layout: the device part above + the batched host tail "
                                                         "(filters, device clustering, differential test, overlap masks)"},
                                  "roofline": {"bound": "fp64_valu", "flops_per_pixel_pair_launched": round(gp_flops, 1),
                                               "launched_tile_fraction": round(g_frac, 4),
                                               "achieved": round(gp_s * gp_flops / 1e12, 3), "peak": peak_tf,
                                               "unit": "TFLOP/s", "frac": round(gp_s * gp_flops / 1e12 / peak_tf, 4)},
                                  "note": "two-sample caller over the whole genome in one call (both sigma loops with tile "
                                          "lists, mst_diff_dog_band over all tiles, pair p-values, BH + selection + partner "
                                          "look-ups on the device, selected records to the host), wall clock"}
        del wg, eng
        torch.cuda.empty_cache()
        # instantiations outside the headline configuration, so that their cost is on the line: the wide-radius tile that
        # serves -sz / -oc (blur radius 15..28) and the normalisation kernel for windows beyond 8400 bins (< 238 bp)
        from mustache_amd.pipeline import ChromosomePipeline, block_tiling
        from mustache_amd.normalize import normalize_band
        from mustache_amd.synth import band_counts
        var = {}
        nv = 4000 + 5 * 2000
        bandv, _ = B.make_band(nv, 2000, 400.0, 500, 3, 1000, device, reps=1)
        CHv, startv, _ = block_tiling(nv, 2000)
        for label, octs in (("octaves_3.2_6.4", (3.2, 6.4)), ("octaves_1.6_3.2_6.4", (1.6, 3.2, 6.4))):
            eng = ChromosomePipeline(octs, device=device).engine
            r = {}
            for mode, skip in (("dense", False), ("band_skip", True)):
                tms = []
                for it in range(3):
                    tm = []
                    eng.sigma_loop_band(bandv, nv, 2000, startv, CHv, skip_empty=skip, download=False, timing=tm)
                    torch.cuda.synchronize()
                    tms.append(tm[0][0].elapsed_time(tm[0][1]))
                r[mode] = round(len(startv) * CHv * CHv / 1e6 / (sorted(tms)[1] * 1e-3), 1)
            var[label] = dict(r, unit="Mpix/s", kernel="scale_space_kernel<Tile<32,64,28,4>, band> (512 threads, one workgroup per CU)", blocks=len(startv),
                              chunk=CHv, max_radius=int(max(eng.levels.radius)))
            del eng
        del bandv
        nw, resw = 60000, 222                                   # window int(2e6 / 222) = 9009 bins
        raww = band_counts(nw, 2000, 30.0, 100, 5, device=device)
        tms = []
        for it in range(3):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            normalize_band(raww, nw, 2000, resw)
            e1.record()
            torch.cuda.synchronize()
            tms.append(e0.elapsed_time(e1))
        var["normalize_window_9009"] = {"ms": round(sorted(tms)[1], 3), "samples": nw * 2002,
                                        "GB/s_on_16B_per_sample": round(16.0 * nw * 2002 / (sorted(tms)[1] * 1e-3) / 1e9, 1),
                                        "kernel": "normalize_local_kernel<32, samples only, prefix sums of the block sums> (windows "
                                                  "8401-16384 bins = resolutions below ~238 bp; no scratch)"}
        del raww
        out["variants"] = var
