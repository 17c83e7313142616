#!/bin/bash
# tracker default loop under tuning variables given as arguments (one run each), e.g. "POGS_RAST_BWD_MFMA=5"
set -e
mkdir -p gpurun_out/r3
echo "default:"; timeout -k 10 200 python3 scripts/profile_tracker.py 2>&1 | grep ms/frame
for v in "$@"; do
  echo "$v:"; env $v timeout -k 10 200 python3 scripts/profile_tracker.py 2>&1 | grep ms/frame
done
echo "default again:"; timeout -k 10 200 python3 scripts/profile_tracker.py 2>&1 | grep ms/frame
