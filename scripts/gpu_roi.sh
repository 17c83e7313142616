#!/bin/bash
set -e
mkdir -p gpurun_out/r3
timeout -k 10 300 python -u -m pytest tests/test_tracking_loss.py tests/test_full_size.py -m gpu -x -q --timeout 180 --timeout-method thread -k "crop" > gpurun_out/r3/crop_test.log 2>&1 || { grep -E "^E|assert" gpurun_out/r3/crop_test.log | head -20; exit 1; }
tail -2 gpurun_out/r3/crop_test.log
timeout -k 10 300 python3 scripts/roi_stages.py 2>&1 | grep -v amdgpu.ids
