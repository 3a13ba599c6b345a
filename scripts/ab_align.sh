#!/bin/bash
# kernel time of the two headline alignment launches for several builds of the library on ONE box, interleaved twice
# usage: scripts/ab_align.sh <variant> ...   (variant "main" = the tree's library, else build/libsvo_hip_<variant>.so)
for rep in 1 2; do for v in "$@"; do for p in 4 8; do
  lib=""; [ "$v" != main ] && lib="build/libsvo_hip_$v.so"
  SVOH_LIB=$lib python bench.py --patch $p --no-cpu-baseline --no-secondary --steps 20 2>/dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('$v P=$p kernel_ms %.4f  min %.4f max %.4f' % (d.get('kernel_ms'), d.get('kernel_ms_min'), d.get('kernel_ms_max')))"
done; done; done
