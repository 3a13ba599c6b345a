#!/bin/bash
# fp64-heavy basic blocks of one kernel and the scratch (spill) instructions inside them -- cross-compiled, no GPU needed.
#   scripts/isa_loop_census.sh svo_pro_universal_amd/csrc/sparse_align.hip 'sparse_align_kernelILi8ELi256ELb0ELb0ELb0E'
set -e
SRC=$1; PAT=$2
TMP=$(mktemp -d)
FLAGS="-O3 -std=c++17 -fPIC --offload-arch=gfx950"
case "$SRC" in *klt.hip|*matcher.hip|*detector.hip) FLAGS="$FLAGS -ffp-contract=off";; esac
/opt/rocm/bin/hipcc $FLAGS --cuda-device-only -S "$SRC" -o $TMP/k.s 2>/dev/null
python3 - "$TMP/k.s" "$PAT" <<'PY'
import re, sys
lines = open(sys.argv[1]).read().splitlines()
start = next(i for i, l in enumerate(lines) if re.match(r'^_Z\w*' + re.escape(sys.argv[2]) + r'\w*:', l))
end = next(i for i in range(start, len(lines)) if 's_endpgm' in lines[i])
blocks, cur = [], None
for l in lines[start:end]:
    m = re.match(r'^(\.LBB\d+_\d+):(.*)', l)
    if m:
        cur = dict(name=m.group(1), depth=0, n=0, f64=0, scratch=0)
        d = re.search(r'Depth=(\d+)', l)
        if d: cur['depth'] = int(d.group(1))
        blocks.append(cur); continue
    if cur is None: continue
    t = l.strip()
    if not t or t.startswith(';'): continue
    cur['n'] += 1
    cur['f64'] += 'f64' in t
    cur['scratch'] += t.startswith('scratch_')
big = [b for b in blocks if b['f64'] >= 40]
for b in big: print('%-12s loop depth %d  %4d instructions  %4d fp64  %d scratch' % (b['name'], b['depth'], b['n'], b['f64'], b['scratch']))
print('scratch instructions: %d in the kernel, %d in its fp64-heavy blocks (the pixel loops are the depth-7 blocks)' % (sum(b['scratch'] for b in blocks), sum(b['scratch'] for b in big)))
PY
rm -rf $TMP
