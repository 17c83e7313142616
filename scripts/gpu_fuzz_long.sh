#!/bin/bash
# randomised parity sweep, two fresh seeds (progress lines keep the run visible)
set -e
mkdir -p gpurun_out/r3/final
timeout -k 10 560 python3 -u scripts/fuzz_parity.py 2500 31 > gpurun_out/r3/final/fuzz_31.log 2>&1 || { tail -30 gpurun_out/r3/final/fuzz_31.log; exit 1; }
tail -1 gpurun_out/r3/final/fuzz_31.log
timeout -k 10 560 python3 -u scripts/fuzz_parity.py 2500 47 > gpurun_out/r3/final/fuzz_47.log 2>&1 || { tail -30 gpurun_out/r3/final/fuzz_47.log; exit 1; }
tail -1 gpurun_out/r3/final/fuzz_47.log
