#!/usr/bin/env python3
"""Per basic block of ONE kernel: what its instructions are -- fp64 FMA / MUL / ADD, conversions, byte unpacking, LDS and
global reads, scratch, scalar, moves, waits -- for the fp64-heavy blocks (the pixel loops).  Cross-compiled, no GPU needed.

  scripts/isa_block_census.py svo_pro_universal_amd/csrc/sparse_align.hip sparse_align_kernelILi8ELi256ELb0ELb0ELb0E ["-DFLAG=1 ..."]

The answer to "where do the instructions of a patch-iteration go" (VERDICT r05 next #3): rows of the table below times the
trip counts noted by the kernel's structure give the instruction budget of a pass."""
import collections
import re
import subprocess
import sys
import tempfile

src, pat = sys.argv[1], sys.argv[2]
extra = sys.argv[3].split() if len(sys.argv) > 3 else []
flags = ["-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950"] + extra
if any(src.endswith(x) for x in ("klt.hip", "matcher.hip", "detector.hip")):
    flags.append("-ffp-contract=off")
with tempfile.TemporaryDirectory() as tmp:
    subprocess.check_call(["/opt/rocm/bin/hipcc"] + flags + ["--cuda-device-only", "-S", src, "-o", tmp + "/k.s"], stderr=subprocess.DEVNULL)
    lines = open(tmp + "/k.s").read().splitlines()
start = next(i for i, l in enumerate(lines) if re.match(r"^_Z\w*" + re.escape(pat) + r"\w*:", l))
end = next(i for i in range(start, len(lines)) if "s_endpgm" in lines[i])


def kind(t):
    op = t.split()[0]
    if op.startswith("v_fma_f64") or op.startswith("v_fmac_f64"): return "fma_f64"
    if op.startswith("v_mul_f64"): return "mul_f64"
    if op.startswith("v_add_f64"): return "add_f64"
    if "f64" in op and op.startswith("v_cvt"): return "cvt_f64"
    if "f64" in op: return "other_f64"
    if op.startswith("v_cvt"): return "cvt_other"
    if op.startswith(("v_bfe", "v_lshr", "v_lshl", "v_and", "v_or", "v_perm", "v_alignb", "v_alignbit")): return "unpack/bit"
    if op.startswith("ds_"): return "lds"
    if op.startswith(("global_", "buffer_", "flat_")): return "vmem"
    if op.startswith("scratch_"): return "scratch"
    if op.startswith("s_waitcnt"): return "waitcnt"
    if op.startswith("s_"): return "scalar"
    if op.startswith(("v_mov", "v_accvgpr", "v_readlane", "v_readfirstlane", "v_writelane")): return "move"
    if op.startswith(("v_add", "v_sub", "v_mad", "v_mul_lo", "v_mul_u", "v_mul_i", "v_ashr", "v_cmp", "v_cndmask", "v_min", "v_max")): return "int/other valu"
    return "other"


blocks, cur = [], None
for l in lines[start:end]:
    m = re.match(r"^(\.LBB\d+_\d+):(.*)", l)
    if m:
        cur = dict(name=m.group(1), depth=0, c=collections.Counter())
        d = re.search(r"Depth=(\d+)", l)
        if d:
            cur["depth"] = int(d.group(1))
        blocks.append(cur)
        continue
    if cur is None:
        continue
    t = l.strip()
    if not t or t.startswith(";") or t.startswith("."):
        continue
    cur["c"][kind(t)] += 1
cols = ["fma_f64", "mul_f64", "add_f64", "cvt_f64", "other_f64", "unpack/bit", "int/other valu", "move", "lds", "vmem", "scratch", "scalar", "waitcnt", "cvt_other", "other"]
print("kernel %s  (%s)" % (pat, " ".join(extra) or "default flags"))
print("%-12s %5s %6s | " % ("block", "depth", "total") + " ".join("%9s" % c[:9] for c in cols))
tot = collections.Counter()
for b in blocks:
    f64 = sum(b["c"][k] for k in ("fma_f64", "mul_f64", "add_f64", "other_f64"))
    for k, v in b["c"].items():
        tot[k] += v
    if f64 < 40:
        continue
    n = sum(b["c"].values())
    print("%-12s %5d %6d | " % (b["name"], b["depth"], n) + " ".join("%9d" % b["c"][c] for c in cols))
print("%-12s %5s %6d | " % ("whole kernel", "", sum(tot.values())) + " ".join("%9d" % tot[c] for c in cols))
