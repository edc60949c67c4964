#!/usr/bin/env python3
"""Drop-in host side for the reference's two-sample caller (ay-lab/mustache v1.3.3, mustache/diff_mustache.py).

    diff_mustache(c1, c2, chromosome, chromosome2, res, start, end, mask_size, distance_in_px, octave_values,
                  st, pt, pt2)                           <- diff_mustache.py:260-569
    process_block(...)                                   <- diff_mustache.py:694-717
    regulator(f1, f2, ...)                               <- diff_mustache.py:572-690
    main()                                               <- diff_mustache.py:720-906   (.loop1/.diffloop1/.loop2/.diffloop2)

Per chromosome both samples' sigma loops run band-direct (mst_scale_space_band on each sample's band) and the difference
image is formed, blurred and scored inside mst_diff_dog_band -- no dense block of either sample, no difference image and no
blurred level of it reaches HBM.  diff_mustache() itself receives dense blocks (the reference's seam) and runs them through
mst_scale_space / mst_diff_image / mst_gauss_blur / mst_masked_normfit / mst_pair_pvalues; the tests hold the two routes equal.  Behaviour that looks odd but is the
reference's is kept and marked: the difference DoG is the octave's D_2 for every tested level (:336 vs :363), the
bias of sample 1 is never applied by main() (:824-827), and -d is clamped to 2000 * res (:770-778).
"""
import argparse
import math
import os
import sys
import time

import numpy as np

from .mustache import (_engine, block_tiling, block_mask_size, parseBP, read_pd)
from .tail import (fdr_candidates_multi, diag_mean_filter_multi, cluster_representatives, loops_from_reps,
                   _features_multi)


def _pair_tails(batch, pairs, pt, pt2, st, intra):
    """diff_mustache.py:428-569 for several block pairs at once: pairs = [(b1, b2, start)] (block of sample 1, block of
    sample 2, coordinate offset).  Every stage -- BH candidates + sparsity windows, diagonal means, clustering, the
    tested-in-the-other-sample look-up -- is ONE device round trip for all the pairs instead of one per block.  Returns one
    (loops1, diff_loops1, loops2, diff_loops2) tuple per pair, identical to the pair-by-pair evaluation."""
    empty = ([], [], [], [])
    out = [empty for _ in pairs]
    live = [k for k, (b1, b2, _) in enumerate(pairs)
            if batch.nz_count[b1] >= 50 and batch.nz_count[b2] >= 50                      # (:266)
            and batch.nz_count[b1] >= 10000 and batch.nz_count[b2] >= 10000]              # (:430)
    if not live:
        return out
    L = len(live)
    blocks = [pairs[k][0] for k in live] + [pairs[k][1] for k in live]      # sample 1's blocks, then sample 2's (one launch each)
    cand = fdr_candidates_multi(batch, blocks, pt, st)                                     # (:432-505)
    state = {}
    for j, k in enumerate(live):
        (q1, i1, c1), (q2, i2, c2) = cand[j], cand[L + j]
        if i1.size and i2.size:                                                            # (:507)
            state[k] = [q1, i1, c1, q2, i2, c2]
    if intra and state:                                                                    # (:516-529)
        ks = list(state)
        # the reference filters sample 1 first and returns if nothing is left, then sample 2: evaluating both is the same
        flt = diag_mean_filter_multi(batch, [pairs[k][0] for k in ks] + [pairs[k][1] for k in ks],
                                     [state[k][1] for k in ks] + [state[k][4] for k in ks],
                                     [state[k][2] for k in ks] + [state[k][5] for k in ks])
        for j, k in enumerate(ks):
            state[k][1], state[k][4] = flt[j], flt[len(ks) + j]
            if state[k][1].size == 0 or state[k][4].size == 0:
                del state[k]
    if not state:
        return out
    ks = list(state)
    M = len(ks)
    cl_blocks = [pairs[k][0] for k in ks] + [pairs[k][1] for k in ks]
    cl_q = [state[k][0] for k in ks] + [state[k][3] for k in ks]
    cl_idx = [state[k][1] for k in ks] + [state[k][4] for k in ks]
    multi = getattr(batch, "cluster_representatives_multi", None)                          # (:531-561)
    reps_all = multi(cl_blocks, cl_q, cl_idx, pt) if multi is not None else \
        [cluster_representatives(batch, b, q, idx) for b, q, idx in zip(cl_blocks, cl_q, cl_idx)]
    # is the representative's pixel tested in the OTHER sample?  one gather for all pairs
    other_blocks = [pairs[k][1] for k in ks] + [pairs[k][0] for k in ks]
    rep_pix = [batch.found[b]["pixel"][reps] if reps else np.zeros(0, np.uint32) for b, reps in zip(cl_blocks, reps_all)]
    feats = _features_multi(batch, other_blocks, rep_pix, [np.zeros(len(p), np.int64) for p in rep_pix])
    for j, k in enumerate(ks):
        b1, b2, start = pairs[k]
        res4 = []
        for s_, (b, bo) in enumerate(((b1, b2), (b2, b1))):
            e = s_ * M + j
            q, reps, pix, nz_other = cl_q[e], reps_all[e], rep_pix[e], feats[e][0]
            loops = loops_from_reps(batch, b, q, reps, start)
            rec, other = batch.found[b], batch.found[bo]
            # differential subset (:567-568): pair < pt2 and v_self > v_other, where v = 1 off-nz, vAll on nz (0 if not found)
            looked_up = "v_other" in rec       # selected-only records (engine.run_band_pairs(select_below=pt)): device look-up
            ridx = np.asarray(reps, dtype=np.int64)
            if looked_up:
                v_other = rec["v_other"][ridx].astype(np.float64)
                missing = np.isnan(v_other)                                # the other sample did not find this pixel
            else:
                opix = other["pixel"].astype(np.int64)
                pos = np.searchsorted(opix, pix.astype(np.int64))
                pos_c = np.minimum(pos, max(len(opix) - 1, 0))
                hit = (pos < len(opix)) & (opix[pos_c] == pix.astype(np.int64)) if len(opix) else np.zeros(len(pix), bool)
                v_other = np.where(hit, other["value"][pos_c] if len(opix) else 0.0, 0.0).astype(np.float64)
                missing = ~hit
            v_other = np.where(missing, np.where(np.asarray(nz_other) != 0, 0.0, 1.0), v_other)
            keep = (rec["pair"][ridx] < pt2) & (rec["value"][ridx] > v_other) if len(ridx) else np.zeros(0, bool)
            diff = [loops[i] for i in np.nonzero(keep)[0]]
            res4.extend([loops, diff])
        out[k] = tuple(res4)
    return out


def _pair_tail(batch, b1, b2, start, pt, pt2, st, intra):
    """diff_mustache.py:428-569 on the records of blocks b1 (sample 1) and b2 (sample 2)."""
    return _pair_tails(batch, [(b1, b2, start)], pt, pt2, st, intra)[0]


def diff_mustache(c1, c2, chromosome, chromosome2, res, start, end, mask_size, distance_in_px, octave_values, st, pt,
                  pt2):
    """(loops1, diff_loops1, loops2, diff_loops2) for one dense block pair; c1 and c2 are filled in place like the
    reference does (:268-273)."""
    import torch
    eng = _engine(octave_values)
    c1, c2 = np.asarray(c1), np.asarray(c2)
    if c1.shape != c2.shape or c1.ndim != 2 or c1.shape[0] != c1.shape[1] or c1.dtype != np.float64 or c2.dtype != np.float64:
        raise ValueError("diff_mustache(): c1 and c2 must be square float64 arrays of the same shape")
    intra = chromosome == chromosome2
    dev = torch.empty((2,) + c1.shape, dtype=torch.float64, device=eng.device)
    dev[0].copy_(torch.from_numpy(np.ascontiguousarray(c1)))
    dev[1].copy_(torch.from_numpy(np.ascontiguousarray(c2)))
    batch = eng.run_block_pairs(dev, distance_in_px, intra=intra)
    if batch.nz_count[0] >= 50 and batch.nz_count[1] >= 50:               # the reference returns before filling (:266-273)
        from .mustache import fill_like_reference                          # the same fills the device copies hold, on the host
        fill_like_reference(c1, distance_in_px, intra)
        fill_like_reference(c2, distance_in_px, intra)
    return _pair_tail(batch, 0, 1, start, pt, pt2, st, intra)


def process_block(i, start, end, overlap_size, cc1, cc2, chromosome, chromosome2, res, distance_in_px, octave_values, o,
                  st, pt, pt2):
    """diff_mustache.py:694-717: tags 1..4 = loops1, diff1, loops2, diff2, after the overlap mask."""
    mask_size = block_mask_size(i, start, end, overlap_size)
    res4 = diff_mustache(cc1, cc2, chromosome, chromosome2, res, start[i], end[i], mask_size, distance_in_px,
                         octave_values, st, pt, pt2)
    _append_tagged(o, res4, start[i], mask_size)


def _append_tagged(o, res4, start_i, mask_size):
    for tag, loops in enumerate(res4, start=1):
        for loop in loops:
            if loop[0] >= start_i + mask_size or loop[1] >= start_i + mask_size:
                o.append([loop[0], loop[1], loop[2], loop[3], tag])


def normalized_pair_bands(pipe, coo1, coo2, res, distance_in_px):
    """Both samples' normalised bands over the common width n = max(n1, n2) (diff_mustache.py:628-635): each sample is
    normalised with ITS OWN n, blocks are cut with the common one."""
    import torch
    from .normalize import band_from_host_coo, normalize_band
    dev = pipe.device
    from .hicfile import PackedContacts
    from .normalize import band_from_packed
    ns, coos = [], []
    for coo in (coo1, coo2):
        if isinstance(coo, PackedContacts):            # a `.hic` sample read by the native reader (records possibly on the device)
            ns.append(int(coo.n))
            coos.append(coo)
            continue
        x, y, v = coo
        x = np.ascontiguousarray(np.asarray(x), dtype=np.int64)
        y = np.ascontiguousarray(np.asarray(y), dtype=np.int64)
        v = np.ascontiguousarray(np.asarray(v), dtype=np.float64)
        ns.append(int(max(x.max(), y.max())) + 1)                          # (:630-631)
        coos.append((x, y, v))
    n = max(ns)                                                            # (:632)
    dbands = []
    for coo, n_s in zip(coos, ns):
        if isinstance(coo, PackedContacts):
            band = band_from_packed(coo, distance_in_px, dev)
        else:
            band = band_from_host_coo(coo[0], coo[1], coo[2], n_s, distance_in_px, dev)
        band, _, _ = normalize_band(band, n_s, distance_in_px, res)       # (:634-635 -> mustache.py:623)
        if n_s < n:
            pad = torch.zeros((distance_in_px + 2, n), dtype=torch.float64, device=dev)
            pad[:, :n_s] = band
            band = pad
        dbands.append(band)
    return dbands, n


def run_pair_genome(pipe, pairs, distance_in_px, st, pt, pt2):
    """pairs: [(dbands, n)] per chromosome (normalized_pair_bands).  ALL block pairs of all chromosomes go through the same
    launches: the two samples' bands are laid side by side per chromosome (pipeline.GenomeLayout) and every group of block
    pairs is one engine.run_band_pairs call.  Returns the tagged rows [x, y, fdr, sigma, tag] per chromosome, identical to
    call_diff_loops_coo on each chromosome alone."""
    from .pipeline import GenomeLayout
    lay = GenomeLayout([n for _, n in pairs], distance_in_px)
    # `pairs` is consumed: each sample's chromosome bands are released one by one as they are copied into that sample's
    # genome band (the caller holds no other reference), so the peak is the genome bands + the bands not yet copied
    per_sample = [[d[s_] for d, _ in pairs] for s_ in (0, 1)]
    del pairs[:]
    gbands = [lay.band(per_sample[s_], pipe.device, consume=True) for s_ in (0, 1)]
    return run_pair_layout(pipe, lay, gbands, st, pt, pt2)


def run_pair_layout(pipe, lay, gbands, st, pt, pt2):
    """run_pair_genome's body on a prepared layout + the two samples' genome bands."""
    eng = pipe.engine
    CH, distance_in_px, pairs = lay.CH, lay.dpx, lay.ns
    # per block pair in HBM: D_2 of the difference image for every octave + the two samples' record buffers
    per_pair = len(eng.levels.octave_values) * CH * CH * 8 + 2 * max(4096, CH * CH // 32) * 48
    # groups of block pairs: two are in flight at a time (the device work of group i + 1 runs under the host tail of group i),
    # each at least ~256 Mpix per sample so that its launches fill the chip
    bs = max(1, min(int(pipe.max_batch_bytes // (2 * per_pair)), max(pipe.blocks_per_launch(CH), -(-len(lay.blocks) // 4))))
    out = [[] for _ in pairs]
    groups = [lay.blocks[g0:g0 + bs] for g0 in range(0, len(lay.blocks), bs)]
    for grp, batch in zip(groups, eng.run_band_pairs_overlapped(gbands, lay.N, distance_in_px,
                                                                [[g[3] for g in grp] for grp in groups], CH, select_below=pt)):
        P = len(grp)
        tails = _pair_tails(batch, [(j, P + j, g[2]) for j, g in enumerate(grp)], pt, pt2, st, True)
        for j, (c, i, s_loc, _) in enumerate(grp):
            _, start, end = lay.tiling[c]
            mask = block_mask_size(i, start, end, distance_in_px)
            _append_tagged(out[c], tails[j], s_loc, mask)
        del batch
    return out


def call_diff_loops_coo(coo1, coo2, res, distance_in_px, octave_values, st, pt, pt2, verbose=True):
    """regulator's body after the readers (diff_mustache.py:628-685): normalise both samples on the GPU, cut the same
    tiling out of both bands, run all block pairs."""
    from .pipeline import ChromosomePipeline
    pipe = ChromosomePipeline(octave_values)
    dbands, n = normalized_pair_bands(pipe, coo1, coo2, res, distance_in_px)
    if verbose:
        print("Loop calling...")
    return run_pair_genome(pipe, [(dbands, n)], distance_in_px, st, pt, pt2)[0]


def _pairs_from_filled(eng, pipe, dbands, n, dpx, starts, CH, dense=False, pt=None):
    """Sigma loops of both samples + pair p-values for the block pairs that start at `starts`.  Default: band-direct -- both
    samples' tiles are cut out of their bands inside the kernels (engine.run_band_pairs), no dense block, difference image or
    blurred level of it is ever materialised.  dense=True: the reference's own data flow (filled dense blocks of both
    samples, difference image, its two blurs per octave) -- kept as the cross-check path."""
    if not dense:
        return eng.run_band_pairs(dbands, n, dpx, starts, CH, select_below=pt)
    import torch
    from .engine import BlockBatch
    c1, nz1, cnt1 = pipe.blocks_from_band(dbands[0], n, dpx, starts, CH)
    c2, nz2, cnt2 = pipe.blocks_from_band(dbands[1], n, dpx, starts, CH)
    c = torch.cat([c1, c2])
    nz = torch.cat([nz1, nz2])
    nzc = torch.cat([cnt1, cnt2])
    del c1, c2, nz1, nz2
    found, pval, count, fit, cap = eng.sigma_loop(c, nz, nzc, download=False)
    ppair, nfit = eng.pair_pvalues(c, nz, found, cap, count)
    recs, fits = eng._download(found, pval, count, fit, eng.levels.n_tested, sort=True,
                               extra={"pair": ppair, "q": eng.fdr(pval, count, cap)})
    batch = BlockBatch(eng, c, nz, CH, c.shape[0], nzc.cpu().numpy().view(np.uint32).astype(np.int64), recs, fits)
    batch.norm_fit = nfit.cpu().numpy()
    return batch


def read_pair(f1, f2, norm_method, CHRM_SIZE, res, distance_in_bp, bias1, bias2, chromosome, chromosome2, verbose=True):
    """The reading half of regulator() (diff_mustache.py:591-626): -> (coo1, coo2, res) or None when a sample is empty."""
    if not chromosome2 or chromosome2 == 'n':
        chromosome2 = chromosome
    if chromosome != chromosome2:
        raise NotImplementedError("inter-chromosomal mode is non-functional in the reference (diff_mustache.py:687-690)")
    if verbose:
        print("Reading contact map...")
    coos = []
    for f, bias in ((f1, bias1), (f2, bias2)):
        if f.endswith(".hic"):
            from .readers import hic_backend, read_hic_file, read_hic_packed
            if hic_backend() == "native":
                # the packed / streamed form the single-sample CLI uses: same record set as read_hic_file (pinned on the
                # reference's, tests/test_readers_ref.py) without the int64 / float64 triple; None = no contact
                coo = read_hic_packed(f, norm_method, CHRM_SIZE, distance_in_bp, chromosome, res)
            else:
                coo = read_hic_file(f, norm_method, CHRM_SIZE, distance_in_bp, chromosome, chromosome2, res)
        elif f.endswith(".cool"):
            from .readers import read_cooler
            x, y, v, r2 = read_cooler(f, distance_in_bp, chromosome, chromosome2, norm_method)
            if coos and r2 != res:
                raise ValueError('Both contact maps should have the same resolution.')
            res, coo = r2, (x, y, v)
        elif f.endswith(".mcool"):
            from .readers import read_mcooler
            coo = read_mcooler(f, distance_in_bp, chromosome, chromosome2, res, norm_method)
        else:
            coo = read_pd(f, distance_in_bp, bias, chromosome, res)
        coos.append(coo)
    from .hicfile import PackedContacts
    empty = lambda c: c is None or (len(c) == 0 if isinstance(c, PackedContacts) else len(c[2]) == 0)
    if empty(coos[0]) or empty(coos[1]):
        return None
    return coos[0], coos[1], res


def regulator(f1, f2, norm_method, CHRM_SIZE, outdir, bed1="", bed2="", res=5000, sigma0=1.6, s=10, pt=0.1, pt2=0.1,
              st=0.88, octaves=2, verbose=True, nprocesses=4, distance_filter=2000000, bias1=False, bias2=False,
              chromosome='n', chromosome2=None):
    """Two-sample loop calling for one chromosome (diff_mustache.py:572-690); returns [x, y, fdr, sigma, tag] rows."""
    octave_values = [sigma0 * (2 ** i) for i in range(octaves)]
    got = read_pair(f1, f2, norm_method, CHRM_SIZE, res, distance_filter, bias1, bias2, chromosome, chromosome2, verbose)
    if got is None:
        return []
    coo1, coo2, res = got
    if verbose:
        print("Normalizing contact map...")
    distance_in_px = int(math.ceil(distance_filter // res))
    return call_diff_loops_coo(coo1, coo2, res, distance_in_px, octave_values, st, pt, pt2, verbose=verbose)


def parse_args(args):
    p = argparse.ArgumentParser(description="Check the help flag")
    p.add_argument("-f1", "--file1", dest="f_path1", help="REQUIRED: Contact map 1", required=False)
    p.add_argument("-f2", "--file2", dest="f_path2", help="REQUIRED: Contact map 2", required=False)
    p.add_argument("-d", "--distance", dest="distFilter", help="Maximum distance (in bp) between loop loci", required=False)
    p.add_argument("-o", "--outfile", dest="outdir", help="REQUIRED: prefix of the four output files", required=True)
    p.add_argument("-r", "--resolution", dest="resolution", help="REQUIRED: resolution of the contact maps", required=True)
    p.add_argument("-bed1", "--bed1", dest="bed1", default="", required=False)
    p.add_argument("-m1", "--matrix1", dest="mat1", default="", required=False)
    p.add_argument("-bed2", "--bed2", dest="bed2", default="", required=False)
    p.add_argument("-m2", "--matrix2", dest="mat2", default="", required=False)
    p.add_argument("-b1", "--biases1", dest="biasfile1", required=False)
    p.add_argument("-b2", "--biases2", dest="biasfile2", required=False)
    p.add_argument("-cz", "--chromosomeSize", default="", dest="chrSize_file", required=False)
    p.add_argument("-norm", "--normalization", default=False, dest="norm_method", required=False)
    p.add_argument("-st", "--sparsityThreshold", dest="st", type=float, default=0.88, required=False)
    p.add_argument("-pt", "--pThreshold", dest="pt", type=float, default=0.2, required=False)
    p.add_argument("-pt2", "--pThreshold2", dest="pt2", type=float, default=0.1, required=False)
    p.add_argument("-sz", "--sigmaZero", dest="s_z", type=float, default=1.6, required=False)
    p.add_argument("-oc", "--octaves", dest="octaves", default=2, type=int, required=False)
    p.add_argument("-i", "--iterations", dest="s", default=10, type=int, required=False)
    p.add_argument("-p", "--processes", dest="nprocesses", default=4, type=int, required=False)
    p.add_argument("-ch", "--chromosome", dest="chromosome", nargs='+', default='n', required=False)
    p.add_argument("-ch2", "--chromosome2", dest="chromosome2", nargs='+', default='n', required=False)
    p.add_argument("-v", "--verbose", dest="verbose", type=bool, default=True, required=False)
    return p.parse_args(args)


def resolve_distance_filter(dist_arg, res):
    """diff_mustache.py:759-778 (note the 2000*res upper clamp, unlike mustache.py)."""
    d = parseBP(dist_arg)
    if not d:
        if 200 * res >= 2000000:
            return 200 * res
        if 2000 * res <= 2000000:
            return 2000 * res
        return 2000000
    if d < 200 * res:
        return 200 * res
    if d > 2000 * res:
        return 2000 * res
    if d > 2000000:
        return 2000000
    return d


HEADER = "BIN1_CHR\tBIN1_START\tBIN1_END\tBIN2_CHROMOSOME\tBIN2_START\tBIN2_END\tFDR\tDETECTION_SCALE\n"
SUFFIX = {1: ".loop1", 2: ".diffloop1", 3: ".loop2", 4: ".diffloop2"}


def main(argv=None):
    t0 = time.time()
    args = parse_args(sys.argv[1:] if argv is None else argv)
    f1, f2 = args.f_path1, args.f_path2
    if args.bed1 and args.mat1:
        f1 = args.mat1
    if args.bed2 and args.mat2:
        f2 = args.mat2
    if not f1 or not f2 or not os.path.exists(f1) or not os.path.exists(f2):
        print("Error: Couldn't find the specified contact files")
        return
    res = parseBP(args.resolution)
    if not res:
        print("Error: Invalid resolution")
        return
    if not args.chromosome or args.chromosome == 'n':
        if f1.endswith((".cool", ".mcool", ".hic")):
            from .readers import list_chromosomes
            chr_list = list_chromosomes(f1, res)
        else:
            print("Error: Please enter the chromosome name.")
            return
    else:
        chr_list = list(args.chromosome)
    chr_list2 = list(args.chromosome2) if isinstance(args.chromosome2, list) else list(chr_list)
    if len(chr_list) != len(chr_list2):
        print("Error: the same number of chromosome1 and chromosome2 should be provided.")
        return
    distFilter = resolve_distance_filter(args.distFilter, res)
    if args.biasfile1 and not os.path.exists(args.biasfile1):
        print("Error: Couldn't find the specified bias file1")
        return
    if args.biasfile2 and not os.path.exists(args.biasfile2):
        print("Error: Couldn't find the specified bias file2")
        return
    biasf1 = False            # reference quirk (:824-827): -b1 is checked for existence but never passed on
    biasf2 = args.biasfile2 if args.biasfile2 else False
    pairs = list(zip(chr_list, chr_list2))

    # multi-GPU (`torchrun -m mustache_amd.diff_mustache ...`): chromosomes are dealt to the ranks largest first, each rank
    # reads and runs its own, ONE gather of (chromosome, x, y, fdr, sigma, list tag) records, rank 0 writes the four files
    from .sharding import assign_chromosomes, gather_records, init_from_env
    rank, world_size = init_from_env()
    mine = list(range(len(pairs)))
    if world_size > 1:
        from .readers import chromosome_sizes
        sizes = chromosome_sizes(f1, res)
        weights = [sizes.get(str(c), sizes.get("chr" + str(c).replace("chr", ""), 1)) for c, _ in pairs]
        owner = assign_chromosomes(weights, world_size)
        mine = [i for i in range(len(pairs)) if owner[i] == rank]

    def write(i, o):
        chromosome, chromosome2 = pairs[i]
        if i == 0:
            for suf in SUFFIX.values():
                with open(args.outdir + suf, 'w') as fh:
                    fh.write(HEADER)
        counts = {1: 0, 2: 0, 3: 0, 4: 0}
        files = {t: open(args.outdir + suf, 'a') for t, suf in SUFFIX.items()}
        try:
            from .mustache import _scalar_text           # the reference's str() of NumPy scalars, same text (mustache.write_loops)
            c1, c2, rs = str(chromosome), str(chromosome2), int(res)
            for r in o:
                counts[r[4]] += 1
                x, y = int(r[0]), int(r[1])
                files[r[4]].write("%s\t%d\t%d\t%s\t%d\t%d\t%s\t%s\n" % (c1, x * rs, (x + 1) * rs, c2, y * rs, (y + 1) * rs,
                                                                       _scalar_text(r[2]), _scalar_text(r[3])))
        finally:
            for fh in files.values():
                fh.close()
        return counts

    results = {}

    def emit(i, o):
        nonlocal t0
        chromosome = pairs[i][0]
        if world_size > 1:
            results[i] = o
            counts = {t: sum(1 for r in o if r[4] == t) for t in (1, 2, 3, 4)}
        else:
            counts = write(i, o)
        print(f"({counts[1]},{counts[3]}) loops and ({counts[2]},{counts[4]}) differential-loops found in "
              f"chrmosome={chromosome} for detection-fdr<{args.pt} and difference-fdr<{args.pt2} in {time.time() - t0:.2f}sec")
        t0 = time.time()

    # Several chromosomes on this rank (the whole-genome run of BASELINE config 5): both samples' normalised bands are
    # collected in HBM and the block pairs of ALL chromosomes go through the same launches (run_pair_genome); the reference
    # runs chromosome after chromosome (diff_mustache.py:858-906).  Same rows, in the same chromosome order.
    batched = len(mine) > 1
    genome_budget = None                 # bytes of held bands (both samples); pipeline.genome_batch_budget at the first band
    held, held_bytes, pipe = [], 0, None

    def flush():
        nonlocal held, held_bytes
        if held:
            idx, dpx_h = [h[0] for h in held], held[0][3]
            prs = [(list(h[1]), h[2]) for h in held]
            held, held_bytes = [], 0             # `prs` holds the only references: run_pair_genome releases them as it copies
            rows = run_pair_genome(pipe, prs, dpx_h, args.st, args.pt, args.pt2)
            for i, o in zip(idx, rows):
                emit(i, o)
        held, held_bytes = [], 0

    for i in mine:
        chromosome, chromosome2 = pairs[i]
        if not batched:
            emit(i, regulator(f1, f2, args.norm_method, False, args.outdir, bed1=args.bed1, bed2=args.bed2, res=res,
                              sigma0=args.s_z, s=args.s, verbose=args.verbose, pt=args.pt, pt2=args.pt2, st=args.st,
                              distance_filter=distFilter, nprocesses=args.nprocesses, bias1=biasf1, bias2=biasf2,
                              chromosome=chromosome, chromosome2=chromosome2, octaves=args.octaves))
            continue
        got = read_pair(f1, f2, args.norm_method, False, res, distFilter, biasf1, biasf2, chromosome, chromosome2,
                        args.verbose)
        if got is None:
            flush()                               # keeps the output in chromosome order
            emit(i, [])
            continue
        if pipe is None:
            from .pipeline import ChromosomePipeline
            pipe = ChromosomePipeline([args.s_z * (2 ** o_) for o_ in range(args.octaves)])
            from .engine import settle_gc
            settle_gc()                           # the command-line process only; the library leaves the collector alone
        coo1, coo2, res_c = got
        dpx = int(math.ceil(distFilter // res_c))
        if args.verbose:
            print("Normalizing contact map...")
        dbands, n = normalized_pair_bands(pipe, coo1, coo2, res_c, dpx)
        nbytes = sum(b.numel() * 8 for b in dbands)
        if genome_budget is None:
            from .pipeline import genome_batch_budget
            genome_budget = genome_batch_budget(pipe.device)
        if held and (held[0][3] != dpx or held_bytes + nbytes > genome_budget):
            flush()
        if nbytes > genome_budget:               # this chromosome alone is over the budget: run it by itself, no second copy
            flush()
            alone = [(list(dbands), n)]
            del dbands
            emit(i, run_pair_genome(pipe, alone, dpx, args.st, args.pt, args.pt2)[0])
            continue
        held.append((i, dbands, n, dpx))
        held_bytes += nbytes
    flush()
    if world_size > 1:
        rec = np.array([[i, float(r[0]), float(r[1]), float(r[2]), float(r[3]), float(r[4])]
                        for i, o in results.items() for r in o], dtype=np.float64).reshape(-1, 6)
        parts = gather_records(rec)
        if rank == 0:
            allrec = np.concatenate(parts)
            for i in range(len(pairs)):
                rows = allrec[allrec[:, 0] == i]
                write(i, [[np.int64(a), np.int64(b), np.float64(q), np.float64(sg), int(t)] for _, a, b, q, sg, t in rows])


if __name__ == '__main__':
    main()
