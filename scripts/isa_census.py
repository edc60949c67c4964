#!/usr/bin/env python3
"""Static instruction census of the fused kernel's level loop (no GPU needed): what one wave issues per Gaussian level, by
category and by phase, for every blur radius of the default level table -- the audit of the PMC figure "VALU instructions per
computed pixel" (profiles/r05g_pmc_traffic.json: 2370 on band tiles, of which 1335 are the blur's executed flops).

    python scripts/isa_census.py [-DMST_...=...] [other_revision.hip] > profiles/r06_isa_census.txt

How: hipcc -S (device only, the product flags of mustache_amd/csrc/Makefile) on mst_scale_space.hip; the band-source kernel of
the default tile (Tile<32,64,14,8,1,false,false>, BAND = true) is cut into basic blocks with LLVM's own loop annotations
("in Loop: Header=... Depth=2" = the level loop).  Inside the loop a block that holds v_mul_f64 belongs to one case of the
radius switch: [axis-0 main items] -> [axis-0 leftover pieces, executed by the first 16 r threads only] -> s_barrier ->
[axis-1 pass]; the radius of a case follows from its multiplies (8 (r + 1) in the main block).  Everything else in the loop is
common to all radii: tap fetch, dispatch, DoG, edge strip, second barrier, 3 x 3 maximum, sieve, statistics, state moves.
Blocks of the branchy sieve form that only run when some lane passes a test are counted as executed (an upper bound)."""
import collections
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "mustache_amd", "csrc", "mst_scale_space.hip")
FLAGS = ("-O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -fno-fast-math -Wno-unused-function -fno-honor-nans "
         "-mno-amdgpu-ieee --cuda-device-only -S").split()
KERNEL = re.compile(r"^_ZN\S*scale_space_kernelINS_4TileILi32ELi64ELi14ELi8ELi1ELb0ELb0EEELb1EEE\S*:")

CATS = ["fp64 add/mul", "fp64 max/min", "fp64 compare", "v_mov_b64", "v_mov_b32", "DPP move", "v_cndmask", "int / bit VALU",
        "LDS read", "LDS write", "SALU", "s_waitcnt / s_nop", "branch", "barrier", "scalar load", "other"]
VALU = CATS[:8]


def category(op, text):
    if "dpp" in op or " row_" in text or " wave_sh" in text or "row_bcast" in text:
        return "DPP move"
    if op.startswith(("v_add_f64", "v_mul_f64", "v_fma_f64")):
        return "fp64 add/mul"
    if op.startswith(("v_max_f64", "v_min_f64")):
        return "fp64 max/min"
    if op.startswith("v_cmp") and "f64" in op:
        return "fp64 compare"
    if op.startswith("v_mov_b64"):
        return "v_mov_b64"
    if op.startswith(("v_mov_b32", "v_accvgpr")):
        return "v_mov_b32"
    if op.startswith("v_cndmask"):
        return "v_cndmask"
    if op.startswith("v_"):
        return "int / bit VALU"
    if op.startswith("ds_read") or op.startswith("ds_load"):
        return "LDS read"
    if op.startswith("ds_"):
        return "LDS write"
    if op in ("s_waitcnt", "s_nop"):
        return "s_waitcnt / s_nop"
    if op.startswith(("s_cbranch", "s_branch", "s_setpc", "s_swappc")):
        return "branch"
    if op == "s_barrier":
        return "barrier"
    if op.startswith(("s_load", "s_buffer_load")):
        return "scalar load"
    if op.startswith("s_"):
        return "SALU"
    return "other"


def kernel_text(defines, src=SRC):
    with tempfile.TemporaryDirectory() as td:
        out = os.path.join(td, "k.s")
        subprocess.run(["/opt/rocm/bin/hipcc"] + FLAGS + list(defines) + ["-I" + os.path.dirname(SRC), "-I" + os.path.join(ROOT, "include"),
                        src, "-o", out], check=True, capture_output=True)
        lines = open(out).read().split("\n")
    i0 = next(i for i, l in enumerate(lines) if KERNEL.match(l))
    i1 = next(i for i in range(i0, len(lines)) if lines[i].startswith(".Lfunc_end"))
    return lines[i0 + 1:i1]


def blocks(lines):
    """[(label, loop annotation, [(op, text)])] in layout order"""
    out, cur = [], ["entry", "", []]
    for l in lines:
        m = re.match(r"^(\.LBB\d+_\d+):\s*(;.*)?$", l) or re.match(r"^; %bb\.(\d+):\s*(;.*)?$", l)
        if m:
            out.append(tuple(cur))
            cur = [m.group(1), m.group(2) or "", []]
            continue
        t = l.strip()
        if not t or t.startswith((";", ".")):
            if t.startswith(";") and "Loop" in t and not cur[2]:
                cur[1] += " " + t
            continue
        cur[2].append((t.split()[0], t))
    out.append(tuple(cur))
    return out


def count(ins):
    c = collections.Counter()
    for op, text in ins:
        c[category(op, text)] += 1
    return c


def main():
    defines = [a for a in sys.argv[1:] if a.startswith("-D")]
    srcs = [a for a in sys.argv[1:] if not a.startswith("-")]         # another revision of the file (git show REV:... > /tmp/x.hip)
    bl = blocks(kernel_text(defines, srcs[0] if srcs else SRC))
    # the level loop: the depth-2 loop that holds barriers
    heads = collections.Counter()
    for lab, ann, ins in bl:
        m = re.search(r"Header=(BB\d+_\d+) Depth=2", ann)
        if m and any(op == "s_barrier" for op, _ in ins):
            heads[m.group(1)] += 1
    head = heads.most_common(1)[0][0]
    in_loop = [b for b in bl if re.search(r"(Header=%s Depth=2|Parent Loop %s)" % (head, head), b[1]) or b[0] == "." + head[0:0] + "L" + head]
    cases, common = {}, collections.Counter()
    seq = [(lab, count(ins), ins) for lab, ann, ins in in_loop]
    i = 0
    pend = []
    while i < len(seq):
        lab, c, ins = seq[i]
        muls = sum(1 for op, _ in ins if op.startswith("v_mul_f64"))
        if muls == 0:
            common.update(c)
            i += 1
            continue
        pend.append((lab, c, ins, muls))
        i += 1
    # case blocks come in layout order main, pieces, (barrier) axis-1: split the axis-1 block at its barrier
    k = 0
    while k < len(pend):
        lab, c, ins, muls = pend[k]
        r = muls // 8 - 1
        main_c = c
        pieces_c, h_c = collections.Counter(), collections.Counter()
        k += 1
        if k < len(pend) and pend[k][3] == 4 * (r + 1):
            pieces_c = pend[k][1]
            k += 1
        if k < len(pend) and pend[k][3] == 8 * (r + 1):
            h_c = pend[k][1]
            k += 1
        else:       # no separate pieces block found: the axis-1 pass may share a block -- report as is
            pass
        cases[r] = (main_c, pieces_c, h_c)
    radii = [4, 4, 4, 4, 5, 5, 5, 6, 6, 6, 7, 7, 8, 8, 9, 10, 10, 11, 12, 12, 13, 14]      # the 22 distinct blurs of octaves (1.6, 3.2)
    print("static census of one wave's level loop, fused kernel (default tile, band source)%s" % (("  [" + " ".join(defines) + "]") if defines else ""))
    print("blocks in the level loop: %d; radius cases found: %s" % (len(in_loop), sorted(cases)))
    print()
    print("per thread and level, common part (all radii): tap fetch, dispatch, DoG, edge strip, maxima, sieve, statistics, state moves")
    for cat in CATS:
        if common[cat]:
            print("  %-20s %5d" % (cat, common[cat]))
    print("  %-20s %5d" % ("VALU total", sum(common[c] for c in VALU)))
    print()
    print("per thread and level, radius cases: axis-0 main | axis-0 leftover pieces (first 16 r threads) | axis-1")
    print("  r   blur flops (main, pieces, axis-1)   other VALU (main, pieces, axis-1)   LDS reads   LDS writes")
    tot_blur = tot_other = tot_common = 0.0
    for r in sorted(cases):
        m, p, h = cases[r]
        oth = lambda c: sum(c[x] for x in VALU) - c["fp64 add/mul"]
        print("  %2d  %5d %5d %5d                   %5d %5d %5d                  %4d %4d %4d   %4d %4d %4d"
              % (r, m["fp64 add/mul"], p["fp64 add/mul"], h["fp64 add/mul"], oth(m), oth(p), oth(h), m["LDS read"], p["LDS read"],
                 h["LDS read"], m["LDS write"], p["LDS write"], h["LDS write"]))
    print()
    # weighted per computed pixel: 256 threads per 30 x 62 owned pixels; pieces run on 16 r of the 256 threads
    px = 30 * 62
    for r in radii:
        if r not in cases:
            continue
        m, p, h = cases[r]
        share = min(1.0, 16.0 * r / 256.0)
        blur = m["fp64 add/mul"] + h["fp64 add/mul"] + share * p["fp64 add/mul"]
        other = sum((m[x] + h[x] + share * p[x]) for x in VALU) - blur
        tot_blur += blur
        tot_other += other
        tot_common += sum(common[x] for x in VALU)
    # the DoG's 8 subtractions sit in the common part and are fp64 add/mul there
    f = 256.0 / px
    print("per COMPUTED pixel over the 22 blurs of octaves (1.6, 3.2) (x 256 threads / %d owned pixels):" % px)
    print("  blur arithmetic (v_add/mul_f64 in the passes)        %7.1f" % (tot_blur * f))
    print("  other VALU inside the passes (addresses, moves)      %7.1f" % (tot_other * f))
    print("  common part VALU (DoG, maxima, sieve, stats, moves)  %7.1f   (all 22 levels counted with the full sieve: 18 run it, 4 only the maxima)"
          % (tot_common * f))
    print("  level loop total                                     %7.1f   (+ staging and epilogue outside the loop)"
          % ((tot_blur + tot_other + tot_common) * f))
    outside = collections.Counter()
    for lab, ann, ins in bl:
        if (lab, ann, ins) not in in_loop:
            outside.update(count(ins))
    print("  outside the loop, static VALU (staging loops run several times)  %d per thread = %.1f per pixel if run once"
          % (sum(outside[c] for c in VALU), sum(outside[c] for c in VALU) * f))


if __name__ == "__main__":
    main()
