#!/bin/bash
# A/B two builds of the library over the shapes that run at one wave per SIMD (developer tool).
#   bash tools/ab_shapes.sh libA.so libB.so
for L in "$1" "$2"; do
  echo "== $L"
  for SH in "1146 720 100 3 120" "1024 256 64 3 64" "1024 256 56 3 64" "1024 512 201 1 64" "1024 512 256 1 64" "1024 512 160 1 64" "1024 512 48 3 64"; do
    RSLF_LIBRARY=$L python tools/quick_bench.py $SH -2 4
  done
done
