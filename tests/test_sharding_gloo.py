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
