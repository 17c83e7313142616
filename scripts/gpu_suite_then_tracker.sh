#!/bin/bash
# whole GPU suite, then the tracker's default loop
set -e
mkdir -p gpurun_out/r3
timeout -k 10 1000 python -u -m pytest tests -m gpu -x -q --timeout 180 --timeout-method thread > gpurun_out/r3/gpu_suite.log 2>&1 || { tail -60 gpurun_out/r3/gpu_suite.log; exit 1; }
tail -3 gpurun_out/r3/gpu_suite.log
FRAMES=6 timeout -k 10 200 python3 scripts/profile_tracker.py 2>&1 | grep ms/frame
