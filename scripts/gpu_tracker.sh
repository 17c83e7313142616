#!/bin/bash
# tracker: its tests, then the cfg3 bench line
set -e
mkdir -p gpurun_out/r3
timeout -k 10 600 python -u -m pytest tests -m gpu -x -q --timeout 180 --timeout-method thread -k "tracking or binning or depth_order or pose_step or rigid" > gpurun_out/r3/tracker_tests.log 2>&1 || { tail -40 gpurun_out/r3/tracker_tests.log; exit 1; }
tail -3 gpurun_out/r3/tracker_tests.log
timeout -k 10 500 python bench.py --config cfg3 > gpurun_out/r3/bench_cfg3.json 2> gpurun_out/r3/bench_cfg3.err || { tail -20 gpurun_out/r3/bench_cfg3.err; exit 1; }
python - <<'PY'
import json
d = json.load(open("gpurun_out/r3/bench_cfg3.json"))
c = d["config"]
print(d["value"], {k: v for k, v in c.items() if k.startswith("ms_") or k.startswith("loss")})
print(c["stages_ms_per_iteration"])
print(d["roofline"])
PY
