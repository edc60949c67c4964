#!/usr/bin/env python3
"""Drop-in host side for the reference's per-chromosome run (ay-lab/mustache v1.3.3, mustache/mustache.py).

Same function names, argument meaning and return types as the reference for everything on the hot path:

    mustache(c, chromosome, chromosome2, res, pval_weights, start, end, mask_size, distance_in_px,
             octave_values, st, pt)                      <- mustache.py:697-850
    process_block(...)                                   <- mustache.py:945-960
    normalize_sparse(x, y, v, resolution, distance_in_px) <- mustache.py:622-686
    regulator(f, norm_method, CHRM_SIZE, outdir, ...)    <- mustache.py:853-942
    read_pd / read_bias / get_sep / is_chr / parseBP / parse_args / main  (text input, CLI)

but the arithmetic runs in HIP kernels on an MI355X (mustache_amd/csrc, C ABI in include/mustache_hip.h).
Where the reference forks one OS process per block, this host keeps all blocks of a chromosome resident in HBM
and launches each kernel once over the whole batch.
"""
import argparse
import math
import os
import sys
import time
from collections import defaultdict

import numpy as np

from .tail import block_tail

_ENGINES = {}


def _engine(octave_values):
    from .engine import ScaleSpaceEngine
    key = tuple(float(o) for o in octave_values)
    eng = _ENGINES.get(key)
    if eng is None:
        eng = _ENGINES[key] = ScaleSpaceEngine(key)
    return eng


# --------------------------------------------------------------------------------------------------------------
# per-block entry point (reference mustache.py:697-850)
# --------------------------------------------------------------------------------------------------------------
def mustache(c, chromosome, chromosome2, res, pval_weights, start, end, mask_size, distance_in_px, octave_values,
             st, pt):
    """Loops of one dense block.  `c` is a square float64 array; like the reference, it is modified in place
    (diagonals <= 4 and > distance_in_px are set to 2).  `res`, `pval_weights`, `end` and `mask_size` are
    accepted and ignored, exactly as in the reference body.  Returns [[x+start, y+start, fdr, sigma], ...]."""
    import torch
    eng = _engine(octave_values)
    c = np.asarray(c)
    if c.ndim != 2 or c.shape[0] != c.shape[1] or c.dtype != np.float64:
        raise ValueError("mustache(): c must be a square float64 array")
    from .engine import BlockBatch
    intra = chromosome == chromosome2
    n = c.shape[0]
    dev = torch.from_numpy(np.ascontiguousarray(c)).to(eng.device).unsqueeze(0)
    nz, nzc = eng.prologue(dev, distance_in_px, intra)
    # BH and the selection q < pt on the device, only the selected records come back (what the per-chromosome driver does)
    found, fits = eng.sigma_loop(dev, nz, nzc, with_value=False, select_below=pt)
    batch = BlockBatch(eng, dev, nz, n, 1, nzc.cpu().numpy().view(np.uint32).astype(np.int64), found, fits)
    if int(batch.nz_count[0]) < 50:
        return []                            # mustache.py:701-702 returns before the fills of :703-706: `c` stays untouched
    # the caller's block gets the fills of mustache.py:703-706 written on the host -- the same values the device copy holds
    # (tests/test_gpu_block.py), without 128 MB coming back over PCIe per 4000 x 4000 block
    fill_like_reference(c, distance_in_px, intra)
    return block_tail(batch, 0, start, pt, st, intra=intra)


def fill_like_reference(c, distance_in_px, intra):
    """The in-place fills of mustache.py:703-706 on the caller's host block: 2 on and below diagonal 4 and, within a
    chromosome, from diagonal distance_in_px + 1 outwards."""
    n = c.shape[0]
    if c.strides[1] == 8 and c.strides[0] % 8 == 0 and c.strides[0] >= 8 * n and c.flags.writeable:
        from . import hicfile                       # rows of contiguous doubles: the threaded form in libmustache_io.so
        lib = hicfile.load()
        hicfile._check(lib, lib.mst_host_fill_block(c.ctypes.data, n, c.strides[0] // 8, int(distance_in_px), 1 if intra else 0, 8))
        return
    for r in range(n):                              # any other layout (a transposed view, ...): NumPy row slices
        c[r, :min(n, r + 5)] = 2.0
        if intra:
            c[r, r + distance_in_px + 1:] = 2.0


def process_block(i, start, end, overlap_size, cc, chromosome, chromosome2, res, pval_weights, distance_in_px,
                  octave_values, o, st, pt):
    """reference mustache.py:945-960: run one block and append the loops that survive the overlap mask to `o`."""
    mask_size = block_mask_size(i, start, end, overlap_size)
    loops = mustache(cc, chromosome, chromosome2, res, pval_weights, start[i], end[i], mask_size, distance_in_px,
                     octave_values, st, pt)
    for loop in loops:
        if loop[0] >= start[i] + mask_size or loop[1] >= start[i] + mask_size:
            o.append([loop[0], loop[1], loop[2], loop[3]])


# --------------------------------------------------------------------------------------------------------------
# tiling (reference mustache.py:896-910, :948-953)
# --------------------------------------------------------------------------------------------------------------
def block_tiling(n, distance_in_px):
    """(CHUNK_SIZE, start[], end[]) exactly as regulator computes them."""
    chunk = max(2 * distance_in_px, 2000)
    if n <= chunk:
        return chunk, [0], [n]
    start, end = [0], [chunk]
    while end[-1] < n:
        start.append(end[-1] - distance_in_px)
        end.append(start[-1] + chunk)
    end[-1] = n
    start[-1] = end[-1] - chunk
    return chunk, start, end


def block_mask_size(i, start, end, overlap_size):
    if i == 0:
        return -1
    if i == len(start) - 1:
        return end[i - 1] - start[i]
    return overlap_size


# --------------------------------------------------------------------------------------------------------------
# text readers (reference mustache.py:191-297) -- host I/O, pandas like the reference
# --------------------------------------------------------------------------------------------------------------
def parseBP(s):
    """'5kb' / '1mb' / '5000' -> base pairs; False when unparsable (reference mustache.py:29-49)."""
    if not s:
        return False
    if s.isnumeric():
        return int(s)
    s = s.lower()
    for unit, mult in (("kb", 1000), ("mb", 1000000)):
        if unit in s:
            head = s.split(unit)[0]
            return int(head) * mult if head.isnumeric() else False
    return False


def is_chr(s, c):
    return str(c).replace('chr', '') == str(s).replace('chr', '')


def get_sep(f):
    """Guess the column separator from the first line (reference mustache.py:199-215)."""
    with open(f) as fh:
        for line in fh:
            if "\t" in line:
                return '\t'
            if " " in line.strip():
                return ' '
            if "," in line:
                return ','
            if len(line.split(' ')) == 1:
                return ' '
            break
    raise FileNotFoundError


def read_bias(f, chromosome, res):
    """bin -> bias factor; NaN or < 0.2 become +inf so the contact is dropped (reference mustache.py:218-251)."""
    d = defaultdict(lambda: 1.0)
    if not f:
        return False
    sep = get_sep(f)
    with open(f) as fh:
        for pos, line in enumerate(fh):
            cols = line.strip().split(sep)
            if len(cols) == 3:
                if is_chr(cols[0], chromosome):
                    val = float(cols[2])
                    d[float(cols[1]) // res] = val if (not np.isnan(val) and val >= 0.2) else np.inf
            elif len(cols) == 1:
                val = float(cols[0])
                d[pos] = val if (not np.isnan(val) and val >= 0.2) else np.inf
    return d


def _parse_contacts_native(f, sep, chromosome):
    """(n_cols, pos1, pos2, count) through libmustache_io.so's text parser (bit for bit what pandas' C parser yields, see
    include/mustache_io.h), or None when the file needs pandas itself / MUSTACHE_TEXT_BACKEND=pandas asks for it."""
    if os.environ.get("MUSTACHE_TEXT_BACKEND", "native").lower() == "pandas":
        return None
    from .hicfile import HicError, read_text_contacts
    try:
        return read_text_contacts(f, sep, chromosome)
    except HicError as e:
        if e.code == -3:                    # a construct the native parser does not cover: let pandas decide
            return None
        raise


def read_pd(f, distance_in_bp, bias, chromosome, res):
    """3-column (pos1 pos2 count) or 5-column (chr1 pos1 chr2 pos2 count) text -> upper-triangular COO in bin
    units, counts divided by bias[x]*bias[y], non-positive rows dropped (reference mustache.py:254-297)."""
    sep = get_sep(f)
    native = _parse_contacts_native(f, sep, chromosome)
    if native is not None:
        ncols, p1, p2, cnt = native                                        # read_csv + dropna (+ the is_chr filters)
        if ncols == 5 and len(cnt) == 0:
            print('Could\'t read any interaction for this chromosome!')
            return
        keep = np.abs(p1 - p2) <= ((distance_in_bp / res + 1) * res)       # (:267, :283)
        a = np.floor_divide(p1[keep], res)                                 # df[1] //= res
        b = np.floor_divide(p2[keep], res)
        cnt = cnt[keep]
        bias = read_bias(bias, chromosome, res)
        if bias:
            cnt = np.divide(cnt, np.vectorize(bias.get)(a, 1)) if len(a) else cnt
            cnt = np.divide(cnt, np.vectorize(bias.get)(b, 1)) if len(b) else cnt
        pos = cnt > 0
        a, b, cnt = a[pos].astype(np.int64), b[pos].astype(np.int64), cnt[pos]
        return np.minimum(a, b), np.maximum(a, b), cnt
    import pandas as pd
    df = pd.read_csv(f, sep=sep, header=None)
    df.dropna(inplace=True)
    if df.shape[1] == 5:
        df = df[np.vectorize(is_chr)(df[0], chromosome)]
        if df.shape[0] == 0:
            print('Could\'t read any interaction for this chromosome!')
            return
        df = df[np.vectorize(is_chr)(df[2], chromosome)]
        a, b, cnt = 1, 3, 4
    elif df.shape[1] == 3:
        a, b, cnt = 0, 1, 2
    else:
        raise ValueError("contact text file must have 3 or 5 columns")
    df = df.loc[np.abs(df[a] - df[b]) <= ((distance_in_bp / res + 1) * res), :].copy()
    df[a] //= res
    df[b] //= res
    bias = read_bias(bias, chromosome, res)
    if bias:
        df[cnt] = np.divide(df[cnt], np.vectorize(bias.get)(df[a], 1))
        df[cnt] = np.divide(df[cnt], np.vectorize(bias.get)(df[b], 1))
    df = df.loc[df[cnt] > 0, :]
    x = np.min(df.loc[:, [a, b]], axis=1)
    y = np.max(df.loc[:, [a, b]], axis=1)
    return x, y, np.array(df[cnt])


# --------------------------------------------------------------------------------------------------------------
# normalisation (reference mustache.py:622-686) -- on the GPU, diagonal-major band layout
# --------------------------------------------------------------------------------------------------------------
def normalize_sparse(x, y, v, resolution, distance_in_px):
    """Per-diagonal sliding-window z-score, in place on `v` (host arrays in, host array out), computed on the GPU.
    Returns the per-diagonal weight list like the reference (unused downstream)."""
    from .normalize import normalize_sparse_device
    return normalize_sparse_device(x, y, v, resolution, distance_in_px)


# --------------------------------------------------------------------------------------------------------------
# per-chromosome driver (reference mustache.py:853-942)
# --------------------------------------------------------------------------------------------------------------
def call_loops_coo(x, y, v, res, distance_in_px, octave_values, st, pt, chromosome='n', chromosome2=None,
                   verbose=True, normalized=False, timings=None, distributed=True):
    """regulator's body after the reader: normalise, tile, run every block, de-duplicate the overlaps.
    x, y, v are host arrays (the reference's COO).  Returns the reference's list of [x, y, fdr, sigma]."""
    from .pipeline import ChromosomePipeline
    pipe = ChromosomePipeline(octave_values)
    return pipe.run(x, y, v, res, distance_in_px, st, pt, normalized=normalized, verbose=verbose, timings=timings,
                    distributed=distributed)


def _check_pair(f, chromosome, chromosome2):
    if not chromosome2 or chromosome2 == 'n':
        chromosome2 = chromosome
    if (chromosome != chromosome2) and not (('.hic' in f) or ('.cool' in f) or ('.mcool' in f)):
        print("Interchromosomal analysis is only supported for .hic and .cool input formats.")
        raise FileNotFoundError
    if chromosome != chromosome2:
        raise NotImplementedError("inter-chromosomal mode is non-functional in the reference (mustache.py:939-942)")
    return chromosome2


def read_contacts(f, norm_method, CHRM_SIZE, res, distance_filter, bias, chromosome, chromosome2, verbose=True,
                  packed=False, part=(0, 1), device=None):
    """The reading half of regulator() (reference mustache.py:866-889): host I/O only, so main() can fetch the next
    chromosome while the GPU works on the current one.  Returns (x, y, v, res) or None when nothing was read.
    packed=True: a `.hic` file read by the native reader comes back as hicfile.PackedContacts (12 bytes per record, what
    the GPU loader takes) instead of the int64 / float64 triple."""
    chromosome2 = _check_pair(f, chromosome, chromosome2)
    distance_in_bp = distance_filter
    if verbose:
        print("Reading contact map...")
    if f.endswith(".hic") and packed:
        from .readers import hic_backend, read_hic_packed
        if hic_backend() == "native":
            # part = (rank, ranks): several GPUs on ONE chromosome -- each rank inflates its share of the blocks only
            return read_hic_packed(f, norm_method, CHRM_SIZE, distance_in_bp, chromosome, res, part=part, device=device)
    if f.endswith(".hic"):
        from .readers import read_hic_file
        x, y, v = read_hic_file(f, norm_method, CHRM_SIZE, distance_in_bp, chromosome, chromosome2, res)
    elif f.endswith(".cool"):
        from .readers import read_cooler
        x, y, v, res = read_cooler(f, distance_in_bp, chromosome, chromosome2, norm_method)
    elif f.endswith(".mcool"):
        from .readers import read_mcooler
        x, y, v = read_mcooler(f, distance_in_bp, chromosome, chromosome2, res, norm_method)
    else:
        r = read_pd(f, distance_in_bp, bias, chromosome, res)
        if r is None:
            return None
        x, y, v = r
    if len(v) == 0:
        return None
    return np.asarray(x), np.asarray(y), np.asarray(v, dtype=np.float64), res


def regulator(f, norm_method, CHRM_SIZE, outdir, bed="", res=5000, sigma0=1.6, s=10, pt=0.1, st=0.88, octaves=2,
              verbose=True, nprocesses=4, distance_filter=2000000, bias=False, chromosome='n', chromosome2=None,
              contacts=None, shard_blocks=True):
    """Loop calling for one chromosome (reference mustache.py:853-942).  `s` is accepted and ignored like in the
    reference (s = 10 is hard-wired at :711); `nprocesses` is ignored: all blocks run as one GPU batch.
    `contacts` (not in the reference): what read_contacts() returned for this chromosome, when the caller read ahead;
    `shard_blocks=False`: in a multi-GPU job this rank runs the whole chromosome alone (whole-genome sharding by chromosome)."""
    chromosome2 = _check_pair(f, chromosome, chromosome2)
    octave_values = [sigma0 * (2 ** i) for i in range(octaves)]
    if contacts is None:
        contacts = read_contacts(f, norm_method, CHRM_SIZE, res, distance_filter, bias, chromosome, chromosome2, verbose)
        if contacts is None:
            return []
    from .hicfile import PackedContacts
    if isinstance(contacts, PackedContacts):
        from .pipeline import ChromosomePipeline
        distance_in_px = int(math.ceil(distance_filter // contacts.res))
        return ChromosomePipeline(octave_values).run_packed(contacts, distance_in_px, st, pt, verbose=verbose,
                                                            distributed=shard_blocks)
    x, y, v, res = contacts
    distance_in_px = int(math.ceil(distance_filter // res))
    return call_loops_coo(x, y, v, res, distance_in_px, octave_values, st, pt, chromosome, chromosome2, verbose=verbose,
                          distributed=shard_blocks)


# --------------------------------------------------------------------------------------------------------------
# CLI (reference mustache.py:52-178, :963-1111): same flags and defaults
# --------------------------------------------------------------------------------------------------------------
def parse_args(args):
    p = argparse.ArgumentParser(description="Check the help flag")
    p.add_argument("-f", "--file", dest="f_path", help="REQUIRED: Contact map", required=False)
    p.add_argument("-d", "--distance", dest="distFilter",
                   help="REQUIRED: Maximum distance (in bp) allowed between loop loci", required=False)
    p.add_argument("-o", "--outfile", dest="outdir", help="REQUIRED: Name of the output file.", required=True)
    p.add_argument("-r", "--resolution", dest="resolution", help="REQUIRED: Resolution used for the contact maps",
                   required=True)
    p.add_argument("-bed", "--bed", dest="bed", help="BED file for HiC-Pro type input", default="", required=False)
    p.add_argument("-m", "--matrix", dest="mat", help="MATRIX file for HiC-Pro type input", default="",
                   required=False)
    p.add_argument("-b", "--biases", dest="biasfile",
                   help="RECOMMENDED: biases calculated by ICE or KR norm for each locus", required=False)
    p.add_argument("-cz", "--chromosomeSize", default="", dest="chrSize_file",
                   help="RECOMMENDED: .hic corresponding chromosome size file.", required=False)
    p.add_argument("-norm", "--normalization", default=False, dest="norm_method",
                   help="RECOMMENDED: Hi-C normalization method (KR, VC,...).", required=False)
    p.add_argument("-st", "--sparsityThreshold", dest="st", type=float, default=0.88,
                   help="OPTIONAL: sparsity threshold, default 0.88 (relax for sparse data, e.g. 0.8).")
    p.add_argument("-pt", "--pThreshold", dest="pt", type=float, default=0.2,
                   help="OPTIONAL: FDR threshold for the output. Default is 0.2")
    p.add_argument("-sz", "--sigmaZero", dest="s_z", type=float, default=1.6,
                   help="OPTIONAL: sigma0 of the scale space. DEFAULT is 1.6.")
    p.add_argument("-oc", "--octaves", dest="octaves", default=2, type=int, help="OPTIONAL: octave count. DEFAULT 2.")
    p.add_argument("-i", "--iterations", dest="s", default=10, type=int,
                   help="OPTIONAL: accepted for compatibility; the reference ignores it as well.")
    p.add_argument("-p", "--processes", dest="nprocesses", default=4, type=int,
                   help="OPTIONAL: accepted for compatibility; blocks are batched on the GPU instead.")
    p.add_argument("-ch", "--chromosome", dest="chromosome", nargs='+', default='n', required=False,
                   help="REQUIRED: chromosome(s) to run on. Optional for cooler files.")
    p.add_argument("-ch2", "--chromosome2", dest="chromosome2", nargs='+', default='n', required=False,
                   help="Optional: second chromosome (inter-chromosomal mode is non-functional upstream).")
    p.add_argument("-v", "--verbose", dest="verbose", type=bool, default=True, help="OPTIONAL: verbosity")
    return p.parse_args(args)


def resolve_distance_filter(dist_arg, res, quiet=False):
    """reference mustache.py:996-1015: default and clamps of -d."""
    say = (lambda *a: None) if quiet else print
    distFilter = parseBP(dist_arg)
    if not distFilter:
        if 200 * res >= 2000000:
            distFilter = 200 * res
            say("The distance limit is set to {}bp".format(200 * res))
        elif 2000 * res <= 2000000:
            distFilter = 2000 * res
            say("The distance limit is set to {}bp".format(2000 * res))
        else:
            distFilter = 2000000
            say("The distance limit is set to 2Mbp")
    elif distFilter < 200 * res:
        say("The distance limit is set to {}bp".format(200 * res))
        distFilter = 200 * res
    elif distFilter > 10000 * res:
        say("The distance limit is set to {}bp".format(10000 * res))
        distFilter = 10000 * res
    elif distFilter > 10000000:
        distFilter = 10000000
        say("The distance limit is set to 10Mbp")
    return distFilter


def write_loops(path, chromosome, chromosome2, res, loops, first):
    """reference mustache.py:1081-1103: header once, then one TSV row per loop."""
    if first:
        with open(path, 'w') as out_file:
            out_file.write("BIN1_CHR\tBIN1_START\tBIN1_END\tBIN2_CHROMOSOME\tBIN2_START\tBIN2_END\tFDR\tDETECTION_SCALE\n")
    # the reference writes str() of NumPy scalars; Python ints and repr() of Python floats give the same text (held by
    # tests/test_host_logic.py) without NumPy scalar arithmetic and str() per field: 4 x faster on 13 000 rows
    c1, c2, res = str(chromosome), str(chromosome2), int(res)
    loops = list(loops)
    if loops and all(isinstance(v, (int, np.integer)) for v in loops[0][:2]) and \
            all(isinstance(v, (float, np.float64)) for v in loops[0][2:4]):
        n = len(loops)
        try:
            xs = np.fromiter((lp[0] for lp in loops), dtype=np.int64, count=n).tolist()
            ys = np.fromiter((lp[1] for lp in loops), dtype=np.int64, count=n).tolist()
            qs = np.fromiter((lp[2] for lp in loops), dtype=np.float64, count=n).tolist()
            ss = np.fromiter((lp[3] for lp in loops), dtype=np.float64, count=n).tolist()
            rows = ["%s\t%d\t%d\t%s\t%d\t%d\t%r\t%r\n" % (c1, x * res, (x + 1) * res, c2, y * res, (y + 1) * res, q, sg)
                    for x, y, q, sg in zip(xs, ys, qs, ss)]
        except (TypeError, ValueError):
            rows = None
    else:
        rows = None
    if rows is None:                      # rows of other types: field by field, as the reference spells it
        rows = [c1 + '\t' + str(lp[0] * res) + '\t' + str((lp[0] + 1) * res) + '\t' + c2 + '\t' + str(lp[1] * res) + '\t' +
                str((lp[1] + 1) * res) + '\t' + _scalar_text(lp[2]) + '\t' + _scalar_text(lp[3]) + '\n' for lp in loops]
    with open(path, 'a') as out_file:
        out_file.write("".join(rows))


def _scalar_text(v):
    """str(v) for what a loop row holds: np.float64 / float -> repr(float(v)) (the same text), anything else -> str(v)."""
    return repr(float(v)) if isinstance(v, (float, np.float64)) else str(v)      # str(np.float32) is NOT repr(float(v)): left to str()


def main(argv=None):
    start_time = time.time()
    args = parse_args(sys.argv[1:] if argv is None else argv)
    from .sharding import init_from_env
    rank, _world = init_from_env()      # multi-GPU: blocks of each chromosome are sharded, rank 0 writes the TSV
    print("\n")
    f = args.f_path
    if args.bed and args.mat:
        f = args.mat
    if not f or not os.path.exists(f):
        print("Error: Couldn't find the specified contact files")
        return
    res = parseBP(args.resolution)
    if not res:
        print("Error: Invalid resolution")
        return
    if not args.chromosome or args.chromosome == 'n':
        if f.endswith(".cool") or f.endswith(".mcool") or f.endswith(".hic"):
            from .readers import list_chromosomes
            chr_list = list_chromosomes(f, res)
        else:
            print("Error: Please enter the chromosome name.")
            return
    else:
        chr_list = list(args.chromosome)
    if (args.chromosome2 and args.chromosome2 != 'n') and (len(chr_list) != len(args.chromosome2)):
        print("Error: the same number of chromosome1 and chromosome2 should be provided.")
        return
    chr_list2 = list(args.chromosome2) if isinstance(args.chromosome2, list) else list(chr_list)
    distFilter = resolve_distance_filter(args.distFilter, res)

    chrSize_in_bp = False
    if args.chrSize_file:
        import pandas as pd
        csz = pd.read_csv(args.chrSize_file, header=None, sep='\t')
        chrSize_in_bp = {"chr" + str(csz.iloc[i, 0]).replace('chr', ''): csz.iloc[i, 1] for i in range(csz.shape[0])}

    if args.biasfile and not os.path.exists(args.biasfile):
        print("Error: Couldn't find specified bias file")
        return
    biasf = args.biasfile if args.biasfile else False
    pairs = list(zip(chr_list, chr_list2))

    try:                                     # the reader thread below must upload to THIS thread's device, not to GPU 0
        import torch
        my_device = torch.device("cuda", torch.cuda.current_device()) if torch.cuda.is_available() else None
    except ImportError:
        my_device = None

    def fetch(i):
        chromosome, chromosome2 = pairs[i]
        CHRM_SIZE = chrSize_in_bp["chr" + str(chromosome).replace('chr', '')] if chrSize_in_bp else False
        try:
            from ._lib import stage
            with stage("read %s" % chromosome):
                return read_contacts(f, args.norm_method, CHRM_SIZE, res, distFilter, biasf, chromosome, chromosome2,
                                     verbose=args.verbose, packed=True, part=(0, 1) if by_chromosome else (rank, _world),
                                     device=my_device)
        except BaseException as e:          # re-raised in the main thread, at this chromosome's turn
            return e

    # Multi-GPU: a single chromosome (or fewer chromosomes than GPUs) is sharded by BLOCKS, every rank reading the same
    # input; a whole-genome run is sharded by CHROMOSOME (largest first, sharding.assign_chromosomes), each rank reading and
    # running only its own -- small chromosomes would otherwise be cut into launches of a few blocks per GPU.
    by_chromosome = _world > 1 and len(pairs) >= _world
    mine = list(range(len(pairs)))
    if by_chromosome:
        from .readers import chromosome_sizes
        from .sharding import assign_chromosomes
        sizes = chromosome_sizes(f, res)
        weights = [sizes.get(str(c), sizes.get("chr" + str(c).replace("chr", ""), 1)) for c, _ in pairs]
        owner = assign_chromosomes(weights, _world)
        mine = [i for i in range(len(pairs)) if owner[i] == rank]

    # the next chromosome is read (host I/O; the native .hic reader and pandas release the GIL) while the GPU works on
    # the current one -- the reference reads and computes strictly in turn (mustache.py:1057-1080)
    from concurrent.futures import ThreadPoolExecutor
    results = {}

    def emit(i, o):
        nonlocal start_time
        chromosome, chromosome2 = pairs[i]
        print("{0} loops found for chrmosome={1}, fdr<{2} in {3}sec".format(
            len(o), chromosome, args.pt, "%.2f" % (time.time() - start_time)))
        if args.verbose:
            # beside the reference's line (mustache.py:1075-1076): what the GPU stage did -- blocks, Mpix, the fused kernel's time
            from .pipeline import LAST_RUN, run_summary
            line = run_summary()
            if line:
                print(line)
                LAST_RUN.clear()
        if by_chromosome:
            results[i] = o
        elif rank == 0 and (i == 0 or o):
            write_loops(args.outdir, chromosome, chromosome2, res, o, first=(i == 0))
        start_time = time.time()

    # Several chromosomes on this rank (a whole-genome run): their normalised bands are collected in HBM and ALL their
    # blocks go through one sequence of launches (pipeline.run_genome) -- same loops as chromosome by chromosome, without
    # the launch-bound tail of 5-31 blocks per chromosome.  `genome_budget` bounds the bands held at once.
    batched = len(mine) > 1 and (_world == 1 or by_chromosome)
    genome_budget = None                 # bytes; from the device's free memory at the first band (pipeline.genome_batch_budget)
    held, held_bytes, pipe = [], 0, None

    def flush():
        nonlocal held, held_bytes
        if held:
            dpx = held[0][3]
            idx, bands, ns = [h[0] for h in held], [h[1] for h in held], [h[2] for h in held]
            held, held_bytes = [], 0             # `bands` is now the only reference: run_genome releases them as it copies
            loops = pipe.run_genome(bands, ns, dpx, args.st, args.pt)
            for i, o in zip(idx, loops):
                emit(i, o)
        held, held_bytes = [], 0

    with ThreadPoolExecutor(max_workers=1) as pool:
        ahead = pool.submit(fetch, mine[0]) if mine else None
        for k, i in enumerate(mine):
            chromosome, chromosome2 = pairs[i]
            contacts = ahead.result()
            ahead = pool.submit(fetch, mine[k + 1]) if k + 1 < len(mine) else None
            if isinstance(contacts, BaseException):
                raise contacts
            if batched:
                if contacts is None:
                    flush()                       # keeps the output in chromosome order
                    emit(i, [])
                    continue
                if pipe is None:
                    from .pipeline import ChromosomePipeline
                    pipe = ChromosomePipeline([args.s_z * (2 ** o_) for o_ in range(args.octaves)])
                    from .engine import settle_gc
                    settle_gc()                   # the command-line process only; the library leaves the collector alone
                _check_pair(f, chromosome, chromosome2)
                from .hicfile import PackedContacts
                is_packed = isinstance(contacts, PackedContacts)
                res_c = contacts.res if is_packed else contacts[3]
                dpx = int(math.ceil(distFilter // res_c))
                if args.verbose:
                    print("Normalizing contact map...")
                if is_packed:
                    band, n = pipe.normalized_band_packed(contacts, dpx)
                else:
                    band, n = pipe.normalized_band(contacts[0], contacts[1], contacts[2], res_c, dpx)
                del contacts
                if genome_budget is None:
                    from .pipeline import genome_batch_budget
                    genome_budget = genome_batch_budget(pipe.device)
                if held and (held[0][3] != dpx or held_bytes + band.numel() * 8 > genome_budget):
                    flush()
                if band.numel() * 8 > genome_budget:
                    # one chromosome alone is over the budget (a second copy of its band would not fit beside it): the
                    # per-chromosome form, which runs straight from this band
                    flush()
                    emit(i, pipe.run_band(band, n, dpx, args.st, args.pt, distributed=False))
                    del band
                    continue
                held.append((i, band, n, dpx))
                held_bytes += band.numel() * 8
                continue
            if contacts is None:
                o = []
            else:
                o = regulator(f, args.norm_method, False, args.outdir, bed=args.bed,
                              res=getattr(contacts, "res", None) or contacts[3], sigma0=args.s_z,
                              s=args.s, verbose=args.verbose, pt=args.pt, st=args.st, distance_filter=distFilter,
                              nprocesses=args.nprocesses, bias=biasf, chromosome=chromosome, chromosome2=chromosome2,
                              octaves=args.octaves, contacts=contacts, shard_blocks=not by_chromosome)
            emit(i, o)
        flush()
    if by_chromosome:
        # one gather of (chromosome index, x, y, fdr, sigma) records; rank 0 writes the chromosomes in their order
        from .sharding import gather_records
        rec = np.array([[i, float(a), float(b), float(q), float(sg)] for i, o in results.items() for a, b, q, sg in o],
                       dtype=np.float64).reshape(-1, 5)
        parts = gather_records(rec)
        if rank == 0:
            allrec = np.concatenate(parts) if parts else np.zeros((0, 5))
            for i, (chromosome, chromosome2) in enumerate(pairs):
                rows = allrec[allrec[:, 0] == i]
                o = [[np.int64(a), np.int64(b), np.float64(q), np.float64(sg)] for _, a, b, q, sg in rows]
                if i == 0 or o:
                    write_loops(args.outdir, chromosome, chromosome2, res, o, first=(i == 0))


if __name__ == '__main__':
    main()
