#!/usr/bin/env python3
"""Register / LDS / spill numbers of the kernels in an ALREADY BUILT object of libsvo_hip (no compiler run, no GPU): the
gfx950 code object is taken out of the object file's .hip_fatbin section and its metadata notes are read.

  scripts/kernel_resources_built.py svo_pro_universal_amd/csrc/sparse_align.o [regex]

Importable: kernels(path) -> {demangled name: dict(vgpr, vgpr_spill, sgpr, sgpr_spill, lds, scratch)}."""
import os
import re
import subprocess
import sys
import tempfile

LLVM = "/opt/rocm/lib/llvm/bin"


def kernels(obj_path):
    out = {}
    with tempfile.TemporaryDirectory() as tmp:
        fat, co = os.path.join(tmp, "fat.bin"), os.path.join(tmp, "k.co")
        subprocess.check_call([os.path.join(LLVM, "llvm-objcopy"), "--dump-section", ".hip_fatbin=" + fat, obj_path, os.path.join(tmp, "copy.o")])
        subprocess.check_call([os.path.join(LLVM, "clang-offload-bundler"), "--unbundle", "--type=o", "--input=" + fat,
                               "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", "--output=" + co])
        notes = subprocess.check_output([os.path.join(LLVM, "llvm-readelf"), "--notes", co], text=True)
    cur = {}
    keys = {".name": "name", ".vgpr_count": "vgpr", ".vgpr_spill_count": "vgpr_spill", ".sgpr_count": "sgpr",
            ".sgpr_spill_count": "sgpr_spill", ".group_segment_fixed_size": "lds", ".private_segment_fixed_size": "scratch"}
    blocks = re.split(r"\n\s+- \.agpr_count:|\n\s+- \.args:", notes)
    for b in blocks:
        cur = {}
        for line in b.splitlines():
            m = re.match(r"\s*(\.[a-z_]+):\s+(\S+)\s*$", line)
            if m and m.group(1) in keys and keys[m.group(1)] not in cur:
                cur[keys[m.group(1)]] = m.group(2)
        if "name" in cur and "vgpr" in cur:
            name = cur.pop("name")
            try:
                name = subprocess.check_output(["c++filt", name], text=True).strip()
            except Exception:
                pass
            name = re.sub(r"\(.*$", "", name.replace("svoh::", "").replace("void ", ""))
            out[name] = {k: int(v) for k, v in cur.items()}
    return out


if __name__ == "__main__":
    rx = re.compile(sys.argv[2] if len(sys.argv) > 2 else ".")
    for name, d in sorted(kernels(sys.argv[1]).items(), key=lambda kv: kv[1].get("vgpr_spill", 0)):
        if rx.search(name):
            print("%-64s vgpr %3d  vspill %3d  sgpr %3d  sspill %3d  lds %6d  scratch %5d" % (
                name[:64], d.get("vgpr", -1), d.get("vgpr_spill", -1), d.get("sgpr", -1), d.get("sgpr_spill", -1), d.get("lds", -1), d.get("scratch", -1)))
