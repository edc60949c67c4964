#!/usr/bin/env python3
"""Condense a gpurun_out/prof_<tag>/ directory (scripts/profile_bench.sh) into profiles/<tag>_summary.md and
profiles/pmc_traffic.json (HBM bytes per launch of the fused kernel, corrected as MI355X_MICROARCH.md prescribes)."""
import csv
import collections
import json
import os
import sys

tag = sys.argv[1]
src = os.path.join("gpurun_out", "prof_" + tag)
out_md = os.path.join("profiles", tag + "_summary.md")
lines = ["# rocprofv3 summary `%s`" % tag, "",
         "Command: `rocprofv3 --kernel-trace --stats -- python bench.py --steps 3 --warmup 1 --no-cpu` "
         "(scripts/profile_bench.sh); PMC passes: `rocprofv3 --pmc <counters> --kernel-trace -- python bench.py --small "
         "--steps 1 --warmup 0 --no-cpu` with `MST_BENCH_OVERLAP=1` (12 blocks of 4000x4000 in ONE launch), one pass per counter "
         "group.  The traced run uses bench.py's default of 4 launches per step (31 blocks each) on alternating streams.", ""]

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
groups = collections.defaultdict(list)
for r in ss:
    groups[(int(r["Grid_Size_X"]) // int(r["Workgroup_Size_X"]), int(r["Grid_Size_Y"]))].append(
        (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6)
lines += ["", "## Fused kernel `scale_space_kernel<Tile<32,64,14>, band>` (exact arithmetic) per launch shape", "",
          "| tiles/block (padded) | blocks | launches | durations ms |", "|---|---|---|---|"]
for (gx, gy), d in sorted(groups.items(), reverse=True):
    lines.append("| %d | %d | %d | %s |" % (gx, gy, len(d), ", ".join("%.3f" % x for x in d[:8])))
big = max(groups, key=lambda k: k[0] * k[1])
dense = [x for x in groups[big] if x > 0.8 * max(groups[big])]
lines += ["", "Dense %d-block launches: mean **%.3f ms** over %d launches; the shorter launches of the same shape are the "
          "`band_skip` runs." % (big[1], sum(dense) / len(dense), len(dense))]
# bench.py splits a step into several launches of unequal size (the last one smallest); roofline.kernel_ms is the mean
# over ALL dense launches of the timed steps = (sum over the shapes of the workload's tile count) / (number of launches)
gx_big = big[0]
all_dense, blocks_dense = [], 0
for (gx, gy), d in groups.items():
    if gx == gx_big and gy * 3 >= big[1]:        # the step's launches; smaller ones are other workloads / ablations
        dd = [x for x in d if x > 0.8 * max(d)]
        all_dense += dd
        blocks_dense += gy * len(dd)
if all_dense:
    lines += ["", "All dense launches of the workload's tile shape (%d launches, %d blocks in total): mean **%.3f ms per launch** "
              "(what `roofline.kernel_ms` averages), **%.4f ms per block** -> %.1f ms per 124-block step."
              % (len(all_dense), blocks_dense, sum(all_dense) / len(all_dense), sum(all_dense) / blocks_dense,
                 124 * sum(all_dense) / blocks_dense)]
r0 = ss[0]
lines += ["", "VGPR_Count %s, Accum_VGPR_Count %s, SGPR_Count %s, Scratch_Size %s B/lane, workgroup %s threads."
          % (r0["VGPR_Count"], r0["Accum_VGPR_Count"], r0["SGPR_Count"], r0["Scratch_Size"], r0["Workgroup_Size_X"]), ""]

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
lines += ["## PMC, first dense launch of the fused kernel on 12 blocks (192 Mpix)", ""]
allc = {}
for d in sorted(os.listdir(src)):
    if d.startswith("pmc_"):
        per = pmc(d, "scale_space_kernel")
        if not per:
            continue
        big = max(v["_grid"] for v in per.values())
        first = min(k for k, v in per.items() if v["_grid"] == big)
        for k, v in per[first].items():
            if k != "_grid":
                allc[k] = v
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
os.makedirs("profiles", exist_ok=True)
open(out_md, "w").write("\n".join(lines) + "\n")
if traffic:
    json.dump(traffic, open(os.path.join("profiles", "pmc_traffic.json"), "w"), indent=1)
# keep the raw stats csv next to the summary
import shutil
shutil.copy(os.path.join(src, "trace", "trace_kernel_stats.csv"), os.path.join("profiles", tag + "_kernel_stats.csv"))
print("\n".join(lines))
