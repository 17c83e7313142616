#!/bin/bash
# heads timing under POGS_DENSE_WGRAD_WIDE = 0 / 1, twice
set -e
mkdir -p gpurun_out/r3
for w in 0 1 0 1; do
  echo "POGS_DENSE_WGRAD_WIDE=$w"
  POGS_DENSE_WGRAD_WIDE=$w timeout -k 10 300 python3 scripts/bench_heads.py 2>&1 | grep "forward + backward"
done
