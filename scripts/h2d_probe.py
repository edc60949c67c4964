"""Host -> device copy rate out of a page-locked slab pool in the slab sizes of the `.hic` read (one copy per slab, copy stream)."""
import time, torch
dev = torch.device("cuda:0")
cap = 1 << 20
nsl = 72
pool = torch.empty(nsl * cap * 10, dtype=torch.uint8, pin_memory=True)
pool.fill_(1)
side = torch.cuda.Stream()
def run(mode, nslabs=125, streams=1):
    sts = [torch.cuda.Stream() for _ in range(streams)]
    torch.cuda.synchronize()
    t0 = time.time()
    keep = []
    for i in range(nslabs):
        base = (i % nsl) * cap * 10
        s = sts[i % streams]
        with torch.cuda.stream(s):
            if mode == "three":
                hx = pool[base:base + 4 * cap].view(torch.int32)
                hv = pool[base + 4 * cap:base + 8 * cap].view(torch.float32)
                hd = pool[base + 8 * cap:base + 10 * cap].view(torch.uint16)
                keep.append(tuple(t.to(dev, non_blocking=True) for t in (hx, hd, hv)))
            else:
                keep.append(pool[base:base + 10 * cap].to(dev, non_blocking=True))
            s.record_event()
    torch.cuda.synchronize()
    dt = time.time() - t0
    return dt, nslabs * cap * 10 / dt / 1e9
for mode in ("three", "one"):
    for streams in (1, 2, 4):
        for rep in range(3):
            dt, gbs = run(mode, streams=streams)
        print("%s copies per slab, %d stream(s): %.4f s  %.1f GB/s" % (mode, streams, dt, gbs), flush=True)
# one big copy
big = torch.empty(125 * cap * 10, dtype=torch.uint8, pin_memory=True)
for rep in range(3):
    torch.cuda.synchronize(); t0 = time.time(); d = big.to(dev, non_blocking=True); torch.cuda.synchronize(); dt = time.time() - t0
print("one 1.3 GB copy: %.4f s %.1f GB/s" % (dt, big.numel() / dt / 1e9))
