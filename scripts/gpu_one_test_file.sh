#!/bin/bash
# one test file (argument) on the GPU box; log under gpurun_out/r3
set -e
mkdir -p gpurun_out/r3
timeout -k 10 600 python -u -m pytest $1 -m gpu -x -q --timeout 180 --timeout-method thread > gpurun_out/r3/one_file.log 2>&1 || { tail -40 gpurun_out/r3/one_file.log; exit 1; }
tail -3 gpurun_out/r3/one_file.log
