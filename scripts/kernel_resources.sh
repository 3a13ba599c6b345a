#!/bin/bash
# register / LDS / spill numbers of every kernel of one .hip file (cross-compiled for gfx950, no GPU needed)
#   scripts/kernel_resources.sh svo_pro_universal_amd/csrc/sparse_align.hip [pattern] [extra hipcc flags...]
set -e
SRC=$1; PAT=${2:-.}; shift; shift || true
TMP=$(mktemp -d)
FLAGS="-O3 -std=c++17 -fPIC --offload-arch=gfx950"
case "$SRC" in *klt.hip|*matcher.hip|*detector.hip) FLAGS="$FLAGS -ffp-contract=off";; esac
/opt/rocm/bin/hipcc $FLAGS "$@" --cuda-device-only -c "$SRC" -o $TMP/b.o
/opt/rocm/lib/llvm/bin/clang-offload-bundler --unbundle --type=o --input=$TMP/b.o --targets=hipv4-amdgcn-amd-amdhsa--gfx950 --output=$TMP/k.co
/opt/rocm/lib/llvm/bin/llvm-readelf --notes $TMP/k.co | grep -E "^\s+\.name:|\.vgpr_count|vgpr_spill|\.sgpr_count|sgpr_spill|group_segment_fixed|private_segment_fixed" \
  | paste - - - - - - - | sed 's/ \+/ /g' \
  | sed -E 's/\.group_segment_fixed_size:/lds/; s/\.private_segment_fixed_size:/scratch/; s/\.sgpr_count:/sgpr/; s/\.sgpr_spill_count:/sspill/; s/\.vgpr_count:/vgpr/; s/\.vgpr_spill_count:/vspill/; s/\.name://' \
  | grep -E "$PAT" | while read -r line; do n=$(echo "$line" | grep -oE "_Z[A-Za-z0-9_]+" | head -1); d=$(c++filt "$n" | sed "s/svoh:://g; s/^void //; s/(.*//" | cut -c1-80); echo "$line" | sed "s|$n|$d|"; done
rm -rf $TMP
