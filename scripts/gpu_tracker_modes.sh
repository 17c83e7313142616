#!/bin/bash
set -e
for m in "" "FUSE=1" "NORGB=1"; do echo "mode: $m"; env $m timeout -k 10 200 python3 scripts/tracker_stages.py 2>&1 | grep -v amdgpu; done
