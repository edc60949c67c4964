"""Multi-rank path on CPU: 2 processes, gloo backend (the GPU job uses the same code with backend nccl = RCCL)."""
import os
import socket

import numpy as np
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, ws, port, nblocks, q):
    import torch.distributed as dist
    from mustache_amd.sharding import shard_blocks, gather_loops, world
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=ws)
    assert world() == (rank, ws)
    mine = shard_blocks(nblocks, rank, ws)
    # every block "finds" (block id) loops with recognisable coordinates; rank 1 also tests the empty-rank case
    loops = []
    if not (rank == 1 and nblocks == 1):
        for b in mine:
            for k in range(b % 3 + 1):
                loops.append([np.int64(1000 * b + k), np.int64(1000 * b + k + 7), np.float64(0.01 * (k + 1)),
                              np.float64(1.6 * 2 ** (0.1 * (b % 9 + 2)))])
    allv = gather_loops(loops, device="cpu")
    q.put((rank, mine, [[float(v) for v in lp] for lp in allv]))
    dist.barrier()
    dist.destroy_process_group()


def _run(nblocks, ws=2):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, ws, port, nblocks, q)) for r in range(ws)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in range(ws)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    return sorted(res)


def test_shard_and_gather_two_ranks():
    nblocks = 7
    res = _run(nblocks)
    owned = sorted(b for _, mine, _ in res for b in mine)
    assert owned == list(range(nblocks)), "every block exactly once"
    assert res[0][2] == res[1][2], "all ranks see the same gathered list"
    # equals the single-rank result as a set
    expect = []
    for b in range(nblocks):
        for k in range(b % 3 + 1):
            expect.append([1000.0 * b + k, 1000.0 * b + k + 7, 0.01 * (k + 1), 1.6 * 2 ** (0.1 * (b % 9 + 2))])
    assert sorted(res[0][2]) == sorted(expect)


def test_gather_with_an_empty_rank():
    res = _run(1)
    assert res[0][1] == [0] and res[1][1] == []
    assert len(res[0][2]) == 1 and res[0][2] == res[1][2]


def test_assign_chromosomes_lpt():
    from mustache_amd.sharding import assign_chromosomes
    sizes = [248, 242, 198, 190, 181, 170, 159, 145, 138, 133, 135, 133, 114, 107, 101, 90, 83, 80, 58, 64, 46, 50, 156]
    for ws in (1, 2, 4, 8):
        owner = assign_chromosomes(sizes, ws)
        assert len(owner) == len(sizes) and set(owner) <= set(range(ws))
        load = [sum(s for s, o in zip(sizes, owner) if o == r) for r in range(ws)]
        assert max(load) <= sum(sizes) / ws + max(sizes) * (1 - 1 / ws) + 1e-9       # the LPT bound
        assert max(load) - min(load) <= max(sizes)
        assert owner == assign_chromosomes(sizes, ws), "deterministic: every rank computes the same table"
    assert assign_chromosomes([5, 5, 5], 3) == [0, 1, 2]
    assert assign_chromosomes([], 4) == []


def _records_worker(rank, ws, port, q):
    import numpy as np
    import torch.distributed as dist
    from mustache_amd.sharding import gather_records
    dist.init_process_group("gloo", init_method="tcp://127.0.0.1:%d" % port, rank=rank, world_size=ws)
    rec = np.array([[rank, 10.0 * rank + k, 0.5 ** k, 7, 1e-300] for k in range(rank * 2)], dtype=np.float64).reshape(-1, 5)
    parts = gather_records(rec)
    q.put((rank, [p.tolist() for p in parts]))
    dist.barrier()
    dist.destroy_process_group()


def test_gather_records_five_columns_with_an_empty_rank():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_records_worker, args=(r, 3, port, q)) for r in range(3)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in range(3))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert res[0][1] == res[1][1] == res[2][1]
    parts = res[0][1]
    assert [len(p) for p in parts] == [0, 2, 4] and parts[2][3] == [2.0, 23.0, 0.125, 7.0, 1e-300]


def _packed_worker(rank, ws, port, q, dist_dtype):
    import torch
    import torch.distributed as dist
    from mustache_amd.sharding import all_gather_packed
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=ws)
    rng = np.random.default_rng(100 + rank)
    m = [1000, 0, 2477][rank]                       # unequal shares, one of them empty
    x = rng.integers(0, 5000, m).astype(np.int32)
    d = rng.integers(0, 300, m).astype(dist_dtype)
    v = rng.uniform(0.5, 9.0, m).astype(np.float32)
    parts, n_all = all_gather_packed(x, d, v, int((x.astype(np.int64) + d).max()) + 1 if m else 0, torch.device("cpu"))
    out = [(p[0].numpy().copy(), p[1].numpy().astype(np.int64), p[2].numpy().copy(), p[3]) for p in parts]
    q.put((rank, n_all, out, (x, d.astype(np.int64), v)))
    dist.barrier()
    dist.destroy_process_group()


def test_all_gather_packed_three_ranks_uint16_and_int32():
    """sharding.all_gather_packed: every rank ends up with every rank's share (record for record), the common n is the
    maximum -- with an empty share in the middle, for both distance widths."""
    for dt in (np.int32, np.uint16):
        ctx = mp.get_context("spawn")
        q = ctx.Queue()
        port = _free_port()
        procs = [ctx.Process(target=_packed_worker, args=(r, 3, port, q, dt)) for r in range(3)]
        for p in procs:
            p.start()
        res = sorted([q.get(timeout=120) for _ in range(3)], key=lambda t: t[0])
        for p in procs:
            p.join(timeout=60)
            assert p.exitcode == 0
        own = [r[3] for r in res]
        n_want = max(int((o[0].astype(np.int64) + o[1]).max()) + 1 for o in own if len(o[0]))
        for rank, n_all, parts, _ in res:
            assert n_all == n_want
            assert [p[3] for p in parts] == [1000, 0, 2477]
            for (px, pd, pv, cnt), (ox, od, ov) in zip(parts, own):
                assert np.array_equal(px, ox) and np.array_equal(pd, od) and np.array_equal(pv, ov)


def _raw_worker(rank, ws, port, q):
    import torch
    import torch.distributed as dist
    from mustache_amd.sharding import all_gather_raw
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=ws)
    rng = np.random.default_rng(300 + rank)
    parts = []
    for k in range([3, 0, 2][rank]):                 # slabs per rank: unequal, one rank without any
        rows = int(rng.integers(1, 40))
        nbytes = 2 * int(rng.integers(1, 3000))      # even, not a multiple of 16 in general
        parts.append((torch.from_numpy(rng.integers(0, 256, nbytes).astype(np.uint8)),
                      torch.from_numpy(rng.integers(0, 256, 16 * rows).astype(np.uint8)), rows))
    me, got = all_gather_raw(parts, torch.device("cpu"))
    q.put((rank, me, [[(p.numpy().copy(), d.numpy().copy(), r) for p, d, r in lst] for lst in got]))
    dist.barrier()
    dist.destroy_process_group()


def test_all_gather_raw_three_ranks():
    """sharding.all_gather_raw (the exchange of the raw `.hic` slabs, N > 1 ranks on one chromosome): every rank receives every
    rank's slabs byte for byte -- payloads of odd sizes, directories, row counts; a rank without slabs takes part."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_raw_worker, args=(r, 3, port, q)) for r in range(3)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=120) for _ in range(3)], key=lambda t: t[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    own = [r[2][r[0]] for r in res]
    assert [len(o) for o in own] == [3, 0, 2]
    for rank, me, got in res:
        assert me == rank and [len(g) for g in got] == [3, 0, 2]
        for r in range(3):
            for (gp, gd, gr), (op, od, orr) in zip(got[r], own[r]):
                assert gr == orr and np.array_equal(gp, op) and np.array_equal(gd, od)
