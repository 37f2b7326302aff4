#!/bin/bash
# measurement only: time of the 3x3 kernel with parts of its loop removed (wrong results)
for a in 0 1 2 4 8 3 6 12 9 11 15; do
  echo "== YV4_H16_ABLATE=$a"
  YV4_H16_ABLATE=$a python tools/conv_bench.py --dtype bf16 --tiles 4 --filter "256->256 k3s1" --reps 7 2>&1 | grep "k3s1"
  YV4_H16_ABLATE=$a python tools/conv_bench.py --dtype bf16 --tiles 4 --filter "128->128 k3s1" --reps 7 2>&1 | grep "k3s1"
done
