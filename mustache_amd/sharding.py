"""Multi-GPU sharding of the per-chromosome run: one process per GPU, a contiguous range of blocks each, ONE gather of the
candidate-loop records at the end (reference: one multiprocessing.Process per block + a Manager().list(),
mustache/mustache.py:913-937).  Blocks share nothing, so there is no data-path collective; the gather payload is a
few hundred records per rank.  Backend "nccl" is RCCL over xGMI on ROCm; "gloo" is used by the CPU tests."""
import numpy as np
import torch
import torch.distributed as dist


def world():
    if dist.is_available() and dist.is_initialized():
        return dist.get_rank(), dist.get_world_size()
    return 0, 1


def shard_blocks(nblocks, rank, world_size):
    """Indices of the blocks rank `rank` owns: a contiguous range, sizes differing by at most one.  Contiguous, not
    round-robin, because consecutive blocks overlap by half their edge and the fused kernel computes the tiles two
    consecutive blocks of a launch have in common only once (mst_scale_space_band): a rank's neighbours must be its own."""
    q, rem = divmod(int(nblocks), int(world_size))
    lo = rank * q + min(rank, rem)
    return list(range(lo, lo + q + (1 if rank < rem else 0)))


def assign_chromosomes(weights, world_size):
    """Whole-genome runs: which rank owns which chromosome.  Longest-processing-time-first: chromosomes by decreasing
    weight (size in bins; the work is ~ proportional to it), each to the least loaded rank so far; ties go to the lower
    rank / earlier chromosome, so every rank computes the same table.  Returns owner[i] for chromosome i."""
    load = [0.0] * world_size
    owner = [0] * len(weights)
    for i in sorted(range(len(weights)), key=lambda k: (-float(weights[k]), k)):
        r = min(range(world_size), key=lambda k: (load[k], k))
        owner[i] = r
        load[r] += float(weights[i])
    return owner


def gather_records(rec, device=None, group=None, force=False):
    """all_gather of a float64 [m, w] array per rank (m differs per rank) -> list of the ranks' arrays, on every rank.
    `force=True` runs the two collectives even in a 1-rank group (the RCCL smoke test on single-GPU boxes)."""
    rank, ws = world()
    rec = np.ascontiguousarray(rec, dtype=np.float64)
    if ws == 1 and not (force and dist.is_available() and dist.is_initialized()):
        return [rec]
    if device is None:
        device = torch.device("cuda", torch.cuda.current_device()) if dist.get_backend(group) == "nccl" else "cpu"
    width = rec.shape[1]
    cnt = torch.tensor([len(rec)], dtype=torch.int64, device=device)
    counts = [torch.zeros_like(cnt) for _ in range(ws)]
    dist.all_gather(counts, cnt, group=group)
    counts = [int(c.item()) for c in counts]
    pad = torch.zeros((max(max(counts), 1), width), dtype=torch.float64, device=device)
    if len(rec):
        pad[:len(rec)] = torch.from_numpy(rec).to(device)
    parts = [torch.zeros_like(pad) for _ in range(ws)]
    dist.all_gather(parts, pad, group=group)
    return [parts[r][:counts[r]].cpu().numpy() for r in range(ws)]


def all_gather_packed(x, dist_, v, n, device, group=None):
    """One process per GPU, each rank holds the packed records of ITS share of a chromosome's `.hic` blocks (host arrays:
    x int32, dist_ int32 or uint16, v float32; `n` = its max(binY) + 1): every rank receives all shares.  Returns
    (parts, n_all): parts[r] = (x, dist, v, count) device tensors of rank r's records, n_all = max over the ranks.
    Two collectives: all_gather of (count, n), then all_gather of the padded [bytes] record block -- over RCCL / xGMI from
    device memory with backend nccl (each rank uploads only its own share across PCIe), over gloo from host memory in the
    CPU tests.  The record SET every rank ends up with is the same, so the band, its normalisation and everything
    downstream are identical on every rank and identical to the 1-rank run."""
    import numpy as np
    rank, ws = world()
    on_dev = dist.get_backend(group) == "nccl"
    where = device if on_dev else "cpu"
    cnt = int(len(v))
    meta = torch.tensor([cnt, int(n)], dtype=torch.int64, device=where)
    metas = [torch.zeros_like(meta) for _ in range(ws)]
    dist.all_gather(metas, meta, group=group)
    counts = [int(m[0].item()) for m in metas]
    n_all = max(int(m[1].item()) for m in metas)
    as_t = lambda a: a if torch.is_tensor(a) else torch.from_numpy(np.ascontiguousarray(a))     # host arrays or tensors anywhere
    x, dist_, v = as_t(x), as_t(dist_), as_t(v)
    dbytes = dist_.element_size()
    mx = max(max(counts), 1)
    seg = [(-(-4 * mx // 16) * 16), (-(-dbytes * mx // 16) * 16), (-(-4 * mx // 16) * 16)]     # 16-byte aligned segments
    block = torch.zeros(sum(seg), dtype=torch.uint8, device=where)
    off = 0
    for arr, sz in zip((x, dist_, v), seg):
        if cnt:
            src = arr.contiguous().view(torch.uint8).reshape(-1)
            block[off:off + src.numel()].copy_(src, non_blocking=on_dev)
        off += sz
    got = [torch.empty_like(block) for _ in range(ws)]
    dist.all_gather(got, block, group=group)
    parts = []
    ddt = torch.int32 if dbytes == 4 else torch.uint16
    for r in range(ws):
        g = got[r] if on_dev else got[r].to(device)
        xs = g[0:seg[0]].view(torch.int32)[:counts[r]]
        ds = g[seg[0]:seg[0] + seg[1]].view(ddt)[:counts[r]]
        vs = g[seg[0] + seg[1]:].view(torch.float32)[:counts[r]]
        parts.append((xs, ds, vs, counts[r]))
    return parts, n_all


def all_gather_raw(parts, device, group=None):
    """The raw form of all_gather_packed (normalize.read_hic_stream_to_device, `.hic` v7-9): every rank holds the RAW slabs of
    its share of a chromosome's blocks -- parts = [(payload uint8 device tensor, row directory uint8 device tensor (16 bytes
    per row, include/mustache_hicrow.h), rows)] -- and receives every rank's: (rank, slabs) with slabs[r] = rank r's list in
    the same form (device tensors; slabs[rank] = `parts` itself).  Two collectives: all_gather of (slab count, blob bytes),
    then all_gather of one padded byte blob per rank {int64 [slabs][2] (payload bytes, rows), then per slab its payload padded
    to 16 bytes and its directory} -- 6 bytes per record over RCCL / xGMI from device memory (gloo: from host memory)."""
    rank, ws = world()
    on_dev = dist.get_backend(group) == "nccl"
    where = device if on_dev else "cpu"
    pad16 = lambda b: -(-int(b) // 16) * 16
    k = len(parts)
    sizes = [(int(p[0].numel()), int(p[2])) for p in parts]
    nbytes = 16 * k + sum(pad16(b) + 16 * r for b, r in sizes)
    meta = torch.tensor([k, nbytes], dtype=torch.int64, device=where)
    metas = [torch.zeros_like(meta) for _ in range(ws)]
    dist.all_gather(metas, meta, group=group)
    ks = [int(m[0].item()) for m in metas]
    mx = max(max(int(m[1].item()) for m in metas), 16)
    blob = torch.zeros(mx, dtype=torch.uint8, device=where)
    if k:
        blob[:16 * k].copy_(torch.tensor(sizes, dtype=torch.int64).view(torch.uint8).reshape(-1), non_blocking=on_dev)
    off = 16 * k
    for (pay, dr, rows), (b, r) in zip(parts, sizes):
        blob[off:off + b].copy_(pay.reshape(-1), non_blocking=on_dev)
        off += pad16(b)
        blob[off:off + 16 * r].copy_(dr.reshape(-1)[:16 * r], non_blocking=on_dev)
        off += 16 * r
    got = [torch.empty_like(blob) for _ in range(ws)]
    dist.all_gather(got, blob, group=group)
    out = []
    for r in range(ws):
        if r == rank:
            out.append(list(parts))
            continue
        g = got[r] if on_dev else got[r].to(device)
        table = g[:16 * ks[r]].cpu().view(torch.int64).reshape(-1, 2).tolist() if ks[r] else []
        off, lst = 16 * ks[r], []
        for b, rows in table:
            pay = g[off:off + b]
            off += pad16(b)
            lst.append((pay, g[off:off + 16 * rows], int(rows)))
            off += 16 * rows
        out.append(lst)
    return rank, out


def gather_loops(loops, device=None, group=None):
    """All ranks pass their list of [x, y, fdr, sigma]; every rank gets the concatenation in rank order
    (rank 0 writes the TSV).  Two collectives: all_gather of the counts, all_gather of the padded records."""
    rank, ws = world()
    if ws == 1:
        return list(loops)
    rec = np.zeros((len(loops), 4), dtype=np.float64)
    for i, lp in enumerate(loops):
        rec[i] = (float(lp[0]), float(lp[1]), float(lp[2]), float(lp[3]))   # bin indices < 2^53: exact
    out = []
    for arr in gather_records(rec, device, group):
        out.extend([[np.int64(a), np.int64(b), np.float64(q), np.float64(s)] for a, b, q, s in arr])
    return out


def init_from_env():
    """`python -m torch.distributed.run --nproc-per-node N -m mustache_amd ...`: one process per GPU.  Joins the process
    group described by the torchrun environment (RANK / WORLD_SIZE / LOCAL_RANK / MASTER_*) with backend nccl (= RCCL) and
    binds this process to its GPU.  A plain single-process run (no WORLD_SIZE) is left untouched.  Returns (rank, world)."""
    import os
    ws = int(os.environ.get("WORLD_SIZE", "1"))
    if ws <= 1 or (dist.is_available() and dist.is_initialized()):
        return world()
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if os.environ.get("MUSTACHE_ONE_DEVICE"):            # test hook: all ranks on GPU 0 (single-GPU boxes)
        local = 0
    torch.cuda.set_device(local)
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    if os.environ.get("MUSTACHE_DIST_BACKEND", "nccl") == "gloo":      # test hook, pairs with MUSTACHE_ONE_DEVICE
        dist.init_process_group("gloo", rank=int(os.environ["RANK"]), world_size=ws)
    else:
        dist.init_process_group("nccl", rank=int(os.environ["RANK"]), world_size=ws,
                                device_id=torch.device("cuda", local))
    return world()
