#!/usr/bin/env python3
"""Condense a gpurun_out/prof_<tag>/ directory (scripts/profile_bench.sh) into <tag>_summary.md, <tag>_pmc_traffic.json (HBM
bytes per pixel of the fused kernel, corrected as MI355X_MICROARCH.md prescribes) and <tag>_kernel_stats.csv, written next to the
raw data; copy the three into profiles/.  Register / LDS figures come from the code object (scripts/kernel_resources.py), not from
the trace's granule-encoded VGPR_Count column."""
import csv
import collections
import json
import os
import sys

tag = sys.argv[1]
src = sys.argv[2] if len(sys.argv) > 2 else os.path.join("gpurun_out", "prof_" + tag)
out_md = os.path.join(src, tag + "_summary.md")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "scripts"))
lines = ["# rocprofv3 summary `%s`" % tag, "",
         "Command: `rocprofv3 --kernel-trace --stats -- python bench.py --steps 3 --warmup 1 --no-cpu --no-file` "
         "(scripts/profile_bench.sh); PMC passes: `rocprofv3 --pmc <counters> --kernel-trace -- python bench.py --small "
         "--steps 1 --warmup 0 --core` with `MST_BENCH_OVERLAP=1` (12 blocks of 4000x4000 in ONE launch), one pass per counter "
         "group, all in the same session as the trace.  The traced run uses bench.py's default: ONE launch per step in three stages (mst_scale_space_band_stage: three kernel invocations over one work list), the finish of a stage under the next stage's kernel.", ""]

# ---- kernel stats ------------------------------------------------------------------------------------------------
st = list(csv.DictReader(open(os.path.join(src, "trace", "trace_kernel_stats.csv"))))
lines += ["## Kernel time (--stats), top 12", "", "| kernel | calls | total ms | avg ms | % |", "|---|---|---|---|---|"]
for r in st[:12]:
    name = r["Name"].replace("(anonymous namespace)::", "").replace("void ", "")
    name = name[:90]
    lines.append("| `%s` | %s | %.3f | %.3f | %s |" % (name, r["Calls"], float(r["TotalDurationNs"]) / 1e6,
                                                     float(r["AverageNs"]) / 1e6, r["Percentage"]))
tr = list(csv.DictReader(open(os.path.join(src, "trace", "trace_kernel_trace.csv"))))
ss = [r for r in tr if "scale_space_kernel" in r["Kernel_Name"] and ", true>," not in r["Kernel_Name"]]   # exact mode only
# the fused kernel's launches are 1-D grids of work items (tiles; a tile shared by two blocks is one item): label them with
# the per-launch item counts bench.py prints on its JSON line (same run: bench_trace.log)
bench_line = {}
try:
    for l in open(os.path.join(src, "bench_trace.log")):
        if l.startswith("{"):
            bench_line = json.loads(l)
except Exception:
    pass
labels = {}
def _lab(counts, name):
    for c in counts or []:
        labels.setdefault((int(c) + 7) // 8 * 8, []).append(name)
rf = bench_line.get("roofline", {})
_lab(rf.get("work_items_per_launch"), "chr1@1kb dense step (tiles shared)")
_lab(bench_line.get("band_skip", {}).get("roofline", {}).get("work_items_per_launch"), "chr1@1kb tile-list step (tiles shared)")
ns = bench_line.get("no_share", {}).get("work_items_per_launch", {})
_lab(ns.get("dense"), "chr1@1kb dense step, MST_FLAG_NO_SHARE")
_lab(ns.get("band_skip"), "chr1@1kb tile-list step, MST_FLAG_NO_SHARE")
groups = collections.defaultdict(list)
for r in ss:
    groups[int(r["Grid_Size_X"]) // int(r["Workgroup_Size_X"])].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6)
lines += ["", "## Fused kernel `scale_space_kernel<Tile<32,64,14>, band>` (exact arithmetic) per launch size", "",
          "| workgroups (work items, padded to 8) | launches | mean ms | first durations ms | what (from the bench line of the same run) |",
          "|---|---|---|---|---|"]
for gx, d in sorted(groups.items(), reverse=True):
    lines.append("| %d | %d | %.3f | %s | %s |" % (gx, len(d), sum(d) / len(d), ", ".join("%.3f" % x for x in d[:6]),
                                                  "; ".join(sorted(set(labels.get(gx, ["other workload (chr21, genome, variants, end to end)"]))))))
def _step(counts):
    """kernel ms of one step = sum over its launches of the MEDIAN duration of that launch size; None if a size is missing"""
    tot = 0.0
    for c in counts or []:
        d = groups.get((int(c) + 7) // 8 * 8)
        if not d:
            return None
        tot += sorted(d)[len(d) // 2]          # median: the first launches of a process run at a higher clock (cool chip)
    return tot if counts else None
px_step = 124 * 4000 * 4000
lines += ["", "Kernel time of one chr1 @ 1 kb step (124 blocks, 1984 Mpix) from the trace (median duration per launch size), and what it is in the no-FMA FP64 roofline "
          "(1152 algorithmic flops per pixel, 39.3 TFLOP/s):", ""]
frac_band = bench_line.get("band_skip", {}).get("launched_tile_fraction")
for name, counts, px in (("dense, tiles shared (`value`)", rf.get("work_items_per_launch"), px_step),
                         ("tile list, tiles shared (`band_skip`, the product mode)",
                          bench_line.get("band_skip", {}).get("roofline", {}).get("work_items_per_launch"),
                          px_step * frac_band if frac_band else None),
                         ("dense, MST_FLAG_NO_SHARE (`no_share`, the kernel's own figure)", ns.get("dense"), px_step),
                         ("tile list, MST_FLAG_NO_SHARE", ns.get("band_skip"), px_step * frac_band if frac_band else None)):
    ms = _step(counts)
    if ms and px:
        tf = px * 1152.0 / (ms * 1e-3) / 1e12
        # work really executed: a tile shared by two blocks is computed once (work items / tiles of the same bench line)
        src_r = rf if "dense" in name else bench_line.get("band_skip", {}).get("roofline", {})
        ratio = 1.0
        if "tiles shared" in name and src_r.get("tiles"):
            ratio = src_r["work_items"] / float(src_r["tiles"])
        lines.append("* %s: **%.2f ms** -> on the work EXECUTED (%.1f %% of the block pixels%s) %.2f TFLOP/s = **%.3f** "
                     "[bench.py's `roofline.frac` view]; crediting every block pixel %.2f TFLOP/s = %.3f"
                     % (name, ms, 100 * ratio * (frac_band if px != px_step else 1.0),
                        ": the tiles that can reach the band, each once" if px != px_step else "", tf * ratio, tf * ratio / 39.3,
                        tf, tf / 39.3))
r0 = ss[0]
try:
    import kernel_resources
    res = [k for k in kernel_resources.kernels(os.path.join(ROOT, "mustache_amd", "libmustache_hip.so"))]
    lines += ["", "## Registers / LDS / spills from the code object (`scripts/kernel_resources.py`)", "",
              "| kernel | VGPR | AGPR | SGPR | SGPR spills | VGPR spills | scratch B/lane |", "|---|---|---|---|---|---|---|"]
    for k in res:
        nm = kernel_resources.demangle(k.get("name", "?")).replace("(anonymous namespace)::", "").replace("void ", "")
        if not any(t in nm for t in ("scale_space_kernel", "normalize_walk", "diag_stats", "diff_dog", "normalize_prefix_kernel<13>")):
            continue
        lines.append("| `%s` | %d | %d | %d | %d | %d | %d |" % (nm.split("(")[0][:80], k.get("vgpr_count", -1), k.get("agpr_count", -1),
                                                              k.get("sgpr_count", -1), k.get("sgpr_spill_count", -1),
                                                              k.get("vgpr_spill_count", -1), k.get("private_segment_fixed_size", -1)))
    lines += ["", "(The trace's `VGPR_Count` column reads %s for the fused kernel: it is an allocation-granule field, not the register "
              "count.)  Workgroup %s threads, dynamic LDS %s B." % (r0["VGPR_Count"], r0["Workgroup_Size_X"], r0.get("LDS_Block_Size", "?")), ""]
except Exception as e:                                      # llvm tools missing: say so instead of printing a wrong number
    lines += ["", "(code-object metadata unavailable: %r)" % (e,), ""]

# ---- PMC -----------------------------------------------------------------------------------------------------------
def pmc(dirname, kernel_sub):
    p = os.path.join(src, dirname, "pmc_counter_collection.csv")
    if not os.path.exists(p):
        return {}
    rows = list(csv.DictReader(open(p)))
    per = collections.defaultdict(dict)
    for r in rows:
        if kernel_sub in r["Kernel_Name"]:
            per[int(r["Dispatch_Id"])][r["Counter_Name"]] = float(r["Counter_Value"])
            per[int(r["Dispatch_Id"])]["_grid"] = int(r["Grid_Size"])
    return per

traffic = {}
# the PMC runs' own bench line says how many work items the launch that `value` times has (dense, tiles shared)
pmc_line = {}
try:
    for l in open(os.path.join(src, "bench_pmc_FETCH_SIZE.log")):
        if l.startswith("{"):
            pmc_line = json.loads(l)
except Exception:
    pass
prf = pmc_line.get("roofline", {})
shared_items = (prf.get("work_items_per_launch") or [None])[0]


def counters_of(grid_threads):
    """counters of the first fused-kernel dispatch with that many threads (None: the largest dispatch), over all PMC passes"""
    out = {}
    for d in sorted(os.listdir(src)):
        if d.startswith("pmc_"):
            per = pmc(d, "scale_space_kernel")
            if not per:
                continue
            want = grid_threads if grid_threads is not None else max(v["_grid"] for v in per.values())
            hits = [k for k, v in per.items() if v["_grid"] == want]
            if not hits:
                continue
            for k, v in per[min(hits)].items():
                if k != "_grid":
                    out[k] = v
    return out


lines += ["## PMC, the largest dispatch of the fused kernel on 12 blocks (192 Mpix): the dense launch with MST_FLAG_NO_SHARE "
          "(every tile once per block, 104 520 workgroups) -- the form comparable with rounds 1 and 2", ""]
allc = counters_of(None)
shared = counters_of(((shared_items + 7) // 8 * 8) * 256) if shared_items else {}
lines += ["| counter | value |", "|---|---|"] + ["| %s | %.6g |" % kv for kv in sorted(allc.items())]
px = 12 * 4000 * 4000
if "FETCH_SIZE" in allc and "WRITE_SIZE" in allc:
    # FETCH_SIZE/WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE under-reports wide coalesced streams by 2x
    # (MI355X_MICROARCH.md, HBM section) -> doubled; WRITE_SIZE taken as is (uncalibrated).
    fetch_b = allc["FETCH_SIZE"] * 1024 * 2
    write_b = allc["WRITE_SIZE"] * 1024
    traffic = {"bytes_per_launch_12_blocks": fetch_b + write_b, "fetch_bytes_corrected": fetch_b, "write_bytes": write_b,
               "bytes_per_pixel": (fetch_b + write_b) / px,
               "bytes_per_launch": (fetch_b + write_b) / px * 124 * 4000 * 4000,
               "note": "FETCH_SIZE x2 (gfx950 correction), KiB units; measured on 12 blocks, scaled per pixel to the "
                       "124-block launch bench.py times"}
    if "FETCH_SIZE" in shared and "WRITE_SIZE" in shared and prf.get("tiles"):
        # the launch form bench.py times for `value`: dense, a tile inside two overlapping blocks computed once
        sb = shared["FETCH_SIZE"] * 1024 * 2 + shared["WRITE_SIZE"] * 1024
        computed_px = px * prf["work_items"] / float(prf["tiles"])
        traffic.update({"launch_form": "dense, tiles shared (the launch `value` times): %d work items for %d tiles of 12 blocks"
                                       % (prf["work_items"], prf["tiles"]),
                        "bytes_per_launch_12_blocks_shared": sb, "bytes_per_computed_pixel": sb / computed_px,
                        "bytes_per_block_pixel_shared": sb / px})
        lines += ["", "The launch form bench.py times for `value` (dense, tiles of overlapping blocks computed once: %d work items "
                  "instead of %d): fetch %.1f MB (x2-corrected) + write %.1f MB = **%.2f B per computed pixel** (%.2f B per block "
                  "pixel)." % (prf["work_items"], prf["tiles"], shared["FETCH_SIZE"] * 2048 / 1e6, shared["WRITE_SIZE"] * 1024 / 1e6,
                               sb / computed_px, sb / px)]
    lines += ["", "HBM traffic of the launch: fetch %.1f MB (x2-corrected) + write %.1f MB = **%.2f B/pixel** "
              "(algorithmic level-streaming model: 592 B/pixel.  With the dense-block source the input alone is 8 "
              "B/pixel + 1 B/pixel of mask; with the band source -- the default since r01e -- only the in-band part of "
              "the band is read, the constant regions are synthesised in the kernel)." % (fetch_b / 1e6, write_b / 1e6, (fetch_b + write_b) / px)]
if "SQ_WAVE_CYCLES" in allc:
    wc = allc["SQ_WAVE_CYCLES"]
    lines += ["", "Wave-cycle split: ACTIVE_INST_ANY %.1f %%, of which VALU %.1f %% of wave cycles; WAIT_INST_ANY %.1f %%; "
              "WAIT_ANY %.1f %%.  VALU instructions per pixel: %.0f."
              % (100 * allc.get("SQ_ACTIVE_INST_ANY", 0) / wc, 100 * allc.get("SQ_ACTIVE_INST_VALU", 0) / wc,
                 100 * allc.get("SQ_WAIT_INST_ANY", 0) / wc, 100 * allc.get("SQ_WAIT_ANY", 0) / wc,
                 allc.get("SQ_INSTS_VALU", 0) * 64 / px)]
if "SQ_LDS_BANK_CONFLICT" in allc and allc.get("SQ_LDS_IDX_ACTIVE"):
    lines += ["LDS: bank-conflict cycles / LDS active cycles = %.1f %%."
              % (100 * allc["SQ_LDS_BANK_CONFLICT"] / allc["SQ_LDS_IDX_ACTIVE"])]
# ---- the same counters for each launch form of the PMC runs' 12-block workload (the forms differ in their workgroup counts)
try:
    forms = [("dense, tiles shared (the launch `value` times)", shared_items),
             ("tile list, tiles shared (product mode)", (pmc_line.get("band_skip", {}).get("roofline", {}).get("work_items_per_launch") or [None])[0]),
             ("dense, every tile once per block (`no_share`)", (pmc_line.get("no_share", {}).get("work_items_per_launch", {}).get("dense") or [None])[0]),
             ("tile list, every tile once per block", (pmc_line.get("no_share", {}).get("work_items_per_launch", {}).get("band_skip") or [None])[0])]
    lines += ["", "## The counters by launch form (same PMC runs; a form is recognised by its workgroup count; per COMPUTED pixel = "
              "per owned pixel of a workgroup that ran, 30 x 62 each)", "",
              "| launch form | workgroups | VALU instr / px | LDS instr / px | VALU busy (wave-cycles -> pipe at 2 waves/SIMD) | WAIT_ANY | "
              "LDS conflict share | HBM B / px |", "|---|---|---|---|---|---|---|---|"]
    by_form = {}
    for label, items in forms:
        if not items:
            continue
        c = counters_of(((items + 7) // 8 * 8) * 256)
        if not c or "SQ_WAVE_CYCLES" not in c:
            continue
        pxc = items * 30.0 * 62.0
        wc = c["SQ_WAVE_CYCLES"]
        row = {"workgroups": items, "valu_per_px": c.get("SQ_INSTS_VALU", 0) * 64 / pxc, "lds_per_px": c.get("SQ_INSTS_LDS", 0) * 64 / pxc,
               "valu_busy": c.get("SQ_ACTIVE_INST_VALU", 0) / wc, "wait_any": c.get("SQ_WAIT_ANY", 0) / wc,
               "lds_conflict": c.get("SQ_LDS_BANK_CONFLICT", 0) / c["SQ_LDS_IDX_ACTIVE"] if c.get("SQ_LDS_IDX_ACTIVE") else None,
               "hbm_b_per_px": (c["FETCH_SIZE"] * 2048 + c["WRITE_SIZE"] * 1024) / pxc if "FETCH_SIZE" in c and "WRITE_SIZE" in c else None}
        by_form[label] = row
        lines.append("| %s | %d | %.0f | %.0f | %.1f %% -> %.1f %% | %.1f %% | %s | %s |" % (
            label, items, row["valu_per_px"], row["lds_per_px"], 100 * row["valu_busy"], 200 * row["valu_busy"], 100 * row["wait_any"],
            "-" if row["lds_conflict"] is None else "%.1f %%" % (100 * row["lds_conflict"]),
            "-" if row["hbm_b_per_px"] is None else "%.2f" % row["hbm_b_per_px"]))
    traffic["by_launch_form"] = by_form
except Exception as e:
    lines += ["", "(per-form counter table unavailable: %r)" % (e,)]
# ---- the other kernels of the path: durations from the trace, HBM bytes from the PMC passes ------------------------------
def pmc_all(kernel_sub):
    """{counter: [values per dispatch, dispatch order]} over all PMC passes for kernels whose name holds kernel_sub."""
    out = collections.defaultdict(list)
    for d in sorted(os.listdir(src)):
        if d.startswith("pmc_"):
            per = pmc(d, kernel_sub)
            for disp in sorted(per):
                for k, v in per[disp].items():
                    if k != "_grid":
                        out[k].append((per[disp]["_grid"], v))
    return out

def durations(kernel_sub):
    return [(int(r["Grid_Size_X"]) * int(r["Grid_Size_Y"]), (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6)
            for r in tr if kernel_sub in r["Kernel_Name"]]

lines += ["", "## Normalisation (row 1) and two-sample kernels", "",
          "Trace durations are from the full bench run (normalisation of the chr1 @ 1 kb band: n = 248,957, 2,002 diagonals, "
          "5.0e8 samples; two-sample path on the chr21 @ 5 kb shape: 6 block pairs of 2000 x 2000).  PMC bytes are from the "
          "--small runs (n = 26,000: 5.2e7 band samples), FETCH_SIZE x2 + WRITE_SIZE of the largest dispatch of each kernel.", "",
          "| kernel | launches | largest-launch ms (trace) | PMC fetch MB (x2) | PMC write MB | note |", "|---|---|---|---|---|---|"]
extra = {}
for sub, note in (("diag_stats_kernel", "one read of the band (8 B / sample)"),
                  ("normalize_walk_kernel", "8 B read (+ priming block, + the centre samples from cache) + 8 B written per sample"),
                  ("diff_dog_kernel", "two bands in, D_2 per octave out (16 B / pixel pair written)"),
                  ("pair_pvalue_dog_kernel", "gather of D_2 at the found pixels")):
    du = durations(sub)
    if not du:
        continue
    big = max(g for g, _ in du)
    dd = [t for g, t in du if g == big]
    pm = pmc_all(sub)
    fe = max((v for g, v in pm.get("FETCH_SIZE", [(0, 0.0)])), default=0.0) * 1024 * 2 / 1e6
    wr = max((v for g, v in pm.get("WRITE_SIZE", [(0, 0.0)])), default=0.0) * 1024 / 1e6
    extra[sub] = {"launches": len(du), "largest_ms_mean": sum(dd) / len(dd), "pmc_fetch_mb_small": fe, "pmc_write_mb_small": wr}
    lines.append("| `%s` | %d | %.3f (x%d) | %.1f | %.1f | %s |" % (sub, len(du), sum(dd) / len(dd), len(dd), fe, wr, note))
if "diag_stats_kernel" in extra and "normalize_walk_kernel" in extra:
    tot = extra["diag_stats_kernel"]["largest_ms_mean"] + extra["normalize_walk_kernel"]["largest_ms_mean"]
    samples = 248957 * 2002
    lines += ["", "mst_normalize_band on chr1 @ 1 kb = %.3f ms (statistics) + %.3f ms (windows) = **%.3f ms** for %.2e samples: "
              "%.0f GB/s of the 16 B / sample model = %.3f of 8 TB/s." % (extra["diag_stats_kernel"]["largest_ms_mean"],
              extra["normalize_walk_kernel"]["largest_ms_mean"], tot, samples, 16.0 * samples / tot / 1e6,
              16.0 * samples / tot / 1e6 / 8000.0)]
open(out_md, "w").write("\n".join(lines) + "\n")
if traffic:
    traffic["command"] = "MST_BENCH_OVERLAP=1 rocprofv3 --pmc FETCH_SIZE|WRITE_SIZE --kernel-trace -- python bench.py --small --steps 1 --warmup 0 --no-cpu"
    traffic["other_kernels"] = extra
    json.dump(traffic, open(os.path.join(src, tag + "_pmc_traffic.json"), "w"), indent=1)
# keep the raw stats csv next to the summary
import shutil
shutil.copy(os.path.join(src, "trace", "trace_kernel_stats.csv"), os.path.join(src, tag + "_kernel_stats.csv"))
print("\n".join(lines))
