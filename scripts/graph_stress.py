#!/usr/bin/env python3
"""Diagnostic (GPU box): are replayed graphs (MST_FLAG_GRAPH: the fused launch and mst_found_finish) reliable under the conditions
of a long sweep -- several call signatures alternating, more signatures than cache entries (graphs are evicted, destroyed and
captured again), the finish's scratch buffer holding garbage before every call?  Per iteration: pick one of `nconf` configurations
(own buffers, own scratch), launch the fused kernel, poison the scratch (0xFF, same stream), run mst_found_finish, compare the
summary (flags word, record counts, tested-pixel counts) with the first plain pass of that configuration.
argv: iters [graph 0/1] [nconf]"""
import ctypes
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np      # noqa: E402
import torch            # noqa: E402
from mustache_amd.engine import ScaleSpaceEngine, device_streams, _ptr, _stream      # noqa: E402

iters = int(sys.argv[1]) if len(sys.argv) > 1 else 500
graph = int(sys.argv[2]) if len(sys.argv) > 2 else 1
nconf = int(sys.argv[3]) if len(sys.argv) > 3 else 9
dev = torch.device("cuda:0")
eng = ScaleSpaceEngine(device=dev)
CH = 2000
g = torch.Generator(device="cpu").manual_seed(5)
side = device_streams(dev)[0]
confs = []
for k in range(nconf):
    n = 2100 + 400 * k
    dpx = 70 + 30 * (k % 4)
    band = (torch.rand(dpx, n, generator=g, dtype=torch.float64) * (torch.rand(dpx, n, generator=g) < 0.3)).to(dev)
    B = 1 + k % 3
    starts = [0] if B == 1 else [int(round(i * (n - CH) / (B - 1))) for i in range(B)]
    confs.append(dict(n=n, dpx=dpx, band=band, starts=starts, B=B))
rng = np.random.default_rng(3)
bad = {}
nt = eng.levels.n_tested
with torch.cuda.stream(side):
    for it in range(iters + nconf):
        k = it if it < nconf else int(rng.integers(nconf))
        c = confs[k]
        use_graph = bool(graph) and it >= nconf
        if "nzc" not in c:
            c["nzc"] = torch.empty(c["B"], dtype=torch.int32, device=dev)
            c["summ"] = torch.empty(int(eng.lib.mst_found_summary_bytes(c["B"])), dtype=torch.uint8, pin_memory=True)
            c["scratch"] = torch.empty(c["summ"].numel(), dtype=torch.uint8, device=dev)
        st = eng._ss_launch(None, None, c["nzc"], True, None, None, False, (c["band"], c["n"], c["dpx"], c["starts"], CH),
                            reuse=("stress", k), graph=use_graph)
        c["scratch"].fill_(0xFF)
        c["summ"].fill_(0x55)
        B, cap = st["B"], st["found_cap"]
        rc = eng.lib.mst_found_finish(_ptr(st["found"]), cap, _ptr(st["count"]), _ptr(c["nzc"]), _ptr(st["stats"]), B, nt,
                                      _ptr(st["pval"]), _ptr(st["fit"]), 0, None, None, None, _ptr(c["scratch"]),
                                      ctypes.c_void_p(c["summ"].data_ptr()), None, None, None, 8 if use_graph else 0, _stream())
        h = c["summ"].numpy()
        cw = 8 * ((B + 1) // 2)
        got = (int(h[:4].view(np.int32)[0]), tuple(h[16:16 + 4 * B].view(np.uint32)), tuple(h[16 + cw:16 + cw + 4 * B].view(np.uint32)))
        if it < nconf:
            c["want"] = got
            assert rc == 0 and got[0] == 0, (rc, got)
        elif rc != 0 or got != c["want"]:
            key = (rc, "flags %#x" % got[0] if got[0] != c["want"][0] else "", "counts" if got[1] != c["want"][1] else "",
                   "nz" if got[2] != c["want"][2] else "")
            bad[key] = bad.get(key, 0) + 1
print("iters %d graph %d nconf %d: failures %s" % (iters, graph, nconf, bad or "none"), flush=True)
