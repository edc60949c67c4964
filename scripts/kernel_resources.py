#!/usr/bin/env python3
"""Register / LDS / spill figures of the kernels in a built library or object, read from the code object's own metadata
(the AMDGPU note record: .vgpr_count, .agpr_count, .sgpr_count, .*_spill_count, .group_segment_fixed_size).

    python scripts/kernel_resources.py mustache_amd/libmustache_hip.so [name-substring]

The rocprofv3 kernel trace's VGPR_Count column is a granule-encoded field and must not be read as a register count;
scripts/summarize_profile.py takes its numbers from here."""
import os
import re
import subprocess
import sys
import tempfile

LLVM = "/opt/rocm/lib/llvm/bin"


def code_objects(path):
    """Yield paths of the gfx950 code objects bundled in `path` (.so / .o with a .hip_fatbin section)."""
    td = tempfile.mkdtemp(prefix="kres")
    fat = os.path.join(td, "fat.bin")
    subprocess.run([LLVM + "/llvm-objcopy", "--dump-section", ".hip_fatbin=" + fat, path, os.path.join(td, "copy")],
                   check=True, capture_output=True)
    blob = open(fat, "rb").read()
    # a library holds one bundle per translation unit, back to back (each starts with the magic string)
    starts = [m.start() for m in re.finditer(rb"__CLANG_OFFLOAD_BUNDLE__", blob)]
    for i, a in enumerate(starts):
        b = starts[i + 1] if i + 1 < len(starts) else len(blob)
        part = os.path.join(td, "b%d.bin" % i)
        open(part, "wb").write(blob[a:b])
        out = os.path.join(td, "b%d.co" % i)
        r = subprocess.run([LLVM + "/clang-offload-bundler", "--unbundle", "--type=o", "--input=" + part,
                            "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", "--output=" + out], capture_output=True)
        if r.returncode == 0 and os.path.exists(out) and os.path.getsize(out) > 0:
            yield out


def kernels(path):
    res = []
    for co in code_objects(path):
        txt = subprocess.run([LLVM + "/llvm-readelf", "--notes", co], capture_output=True, text=True).stdout
        cur = {}
        for line in txt.splitlines():
            m = re.match(r"\s*-?\s*\.(\w+):\s*(.*)", line)
            if not m:
                continue
            k, v = m.group(1), m.group(2).strip()
            if k == "agpr_count" and cur.get("name"):
                res.append(cur)
                cur = {}
            if k in ("agpr_count", "vgpr_count", "sgpr_count", "vgpr_spill_count", "sgpr_spill_count",
                     "group_segment_fixed_size", "private_segment_fixed_size", "max_flat_workgroup_size"):
                cur[k] = int(v)
            elif k == "name":
                cur["name"] = v
        if cur.get("name"):
            res.append(cur)
    return res


def demangle(name):
    for tool in (LLVM + "/llvm-cxxfilt", "llvm-cxxfilt", "c++filt"):
        try:
            out = subprocess.run([tool, name], capture_output=True, text=True).stdout.strip()
            if out and out != name:
                return out
        except Exception:
            pass
    # no demangler on this box: keep the readable middle of the Itanium name
    import re
    m = re.search(r"N_1\d+(\w+?)I", name) or re.search(r"N_1\d+(\w+?)E", name)
    tail = re.sub(r"^.*?(INS_4Tile|ILi)", r"\1", name) if "Tile" in name or "ILi" in name else ""
    return (m.group(1) if m else name) + ("<" + tail[:60] + ">" if tail else "")


if __name__ == "__main__":
    flt = sys.argv[2] if len(sys.argv) > 2 else ""
    for k in kernels(sys.argv[1]):
        nm = demangle(k.get("name", "?"))
        if flt and flt not in nm:
            continue
        print("%-100s vgpr %3d agpr %3d sgpr %3d spill v%d s%d lds %6d scratch %d" % (
            nm[:100], k.get("vgpr_count", -1), k.get("agpr_count", -1), k.get("sgpr_count", -1),
            k.get("vgpr_spill_count", -1), k.get("sgpr_spill_count", -1), k.get("group_segment_fixed_size", -1),
            k.get("private_segment_fixed_size", -1)))
