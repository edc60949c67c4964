#!/usr/bin/env python3
"""A/B harness for builds of libmustache_hip.so (GPU box).

    python scripts/exp_variants.py scratch_libs/a.so scratch_libs/b.so ...      # one subprocess per library
    MUSTACHE_HIP_LIB=... python scripts/exp_variants.py --one                   # what each subprocess runs

Per library: the fused kernel on 12 blocks of the chr1 @ 1 kb shape (4000 x 4000, distance limit 2000), dense and
with empty tiles skipped, timed with events on the launch stream (median of `reps`), plus an order-independent checksum
of the found records (count, pixel sum, level sum, bit pattern sum of the DoG values) and of the level statistics --
builds that claim bit-exactness must print the same checksums as the reference build.
"""
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def one(reps=5, blocks=12):
    import numpy as np
    import torch
    from mustache_amd.pipeline import ChromosomePipeline, block_tiling
    from mustache_amd.synth import band_counts
    from mustache_amd.normalize import normalize_band
    dev = torch.device("cuda:0")
    dpx, res = 2000, 1000
    n = 4000 + (blocks - 1) * 2000
    raw = band_counts(n, dpx, 400.0, 800, 1, device=dev)
    band, _, _ = normalize_band(raw, n, dpx, res)
    del raw
    octs = tuple(float(o) for o in os.environ.get("EXP_OCTAVES", "1.6,3.2").split(","))      # e.g. 3.2,6.4 -> the wide-radius tile
    pipe = ChromosomePipeline(octs, device=dev)
    CH, start, end = block_tiling(n, dpx)
    eng = pipe.engine
    eng.share_tiles = os.environ.get("EXP_SHARE", "1") != "0"       # EXP_SHARE=0: MST_FLAG_NO_SHARE (every tile once per block)
    out = {"lib": os.environ.get("MUSTACHE_HIP_LIB", "default"), "blocks": len(start), "octaves": list(octs)}
    modes = os.environ.get("EXP_MODES", "dense,skip").split(",")
    for mode, skip in (("dense", False), ("skip", True)):
        if mode not in modes:
            continue
        ms = []
        for it in range(reps + 1):
            tm = []
            found, pval, count, fit, cap, nzc = eng.sigma_loop_band(band, n, dpx, start, CH, skip_empty=skip,
                                                                    download=False, timing=tm)
            torch.cuda.synchronize()
            if it:
                ms.append(tm[0][0].elapsed_time(tm[0][1]))
        ms.sort()
        cnt = count.cpu().numpy().view(np.uint32).astype(np.int64)
        rec = found.cpu().numpy()
        pix = lvl = val = 0
        for b in range(len(cnt)):
            w = rec[b, :cnt[b], 0]
            pix += int((w & 0xFFFFFFFF).sum())
            lvl += int((w >> 32).sum())
            val += int(rec[b, :cnt[b], 1].astype(np.uint64).sum(dtype=np.uint64))
        fsum = float(np.nansum(fit.cpu().numpy()[:, :eng.levels.n_tested, :]))
        out[mode] = {"ms": round(ms[len(ms) // 2], 3), "min_ms": round(ms[0], 3),
                     "gpix_s": round(len(start) * CH * CH / 1e9 / (ms[len(ms) // 2] * 1e-3), 3),
                     "check": "%d/%d/%d/%d/%r/%d" % (int(cnt.sum()), pix, lvl, val % (1 << 61), fsum,
                                                      int(nzc.cpu().numpy().view(np.uint32).sum()))}
    print("EXP " + json.dumps(out), flush=True)


def main():
    if "--one" in sys.argv:
        one(blocks=int(os.environ.get("EXP_BLOCKS", "12")))       # EXP_BLOCKS=1: a launch's fixed cost shows
        return
    libs = [a for a in sys.argv[1:] if not a.startswith("--")]
    for spec in libs:                      # path[@NAME=VALUE[;NAME=VALUE...]]  (extra environment for PROFILE builds)
        lib, _, extra = spec.partition("@")
        env = dict(os.environ)
        if lib != "default":
            env["MUSTACHE_HIP_LIB"] = os.path.abspath(lib)
        for kv in filter(None, extra.split(";")):           # NAME=VALUE pairs separated by ';' (values may hold commas)
            k, _, v = kv.partition("=")
            env[k] = v
        t0 = time.time()
        p = subprocess.run([sys.executable, os.path.abspath(__file__), "--one"], env=env, capture_output=True, text=True)
        lines = [l for l in p.stdout.splitlines() if l.startswith("EXP ")]
        if lines:
            print(lines[-1] + "  # %s %.0f s" % (extra, time.time() - t0), flush=True)
            continue
        else:
            print("EXP-FAIL %s rc=%d\n%s" % (spec, p.returncode, (p.stdout + p.stderr)[-1500:]), flush=True)


if __name__ == "__main__":
    main()
