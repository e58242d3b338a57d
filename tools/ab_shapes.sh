#!/bin/bash
# A/B two builds of the library over a list of shapes (developer tool).
#   bash tools/ab_shapes.sh libA.so libB.so ["U V S C D" ...]      default: the shapes that run at one wave per SIMD
A="$1"; B="$2"; shift 2
if [ $# -eq 0 ]; then
  set -- "1146 720 100 3 120" "1024 256 64 3 64" "1024 256 56 3 64" "1024 512 201 1 64" "1024 512 256 1 64" "1024 512 160 1 64" "1024 512 48 3 64"
fi
for L in "$A" "$B"; do
  echo "== $L"
  for SH in "$@"; do
    RSLF_LIBRARY=$L python tools/quick_bench.py $SH -2 4 2>/dev/null
  done
done
