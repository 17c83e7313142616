#!/bin/bash
# shared depth order: its tests, the projection tests, then the POGS three-call step with and without sharing
set -e
mkdir -p gpurun_out/r3
timeout -k 10 400 python -u -m pytest tests/test_gpu_parity.py -m gpu -x -q --timeout 120 --timeout-method thread -k "depth_order or projection or golden or rasterization" > gpurun_out/r3/depth_order_tests.log 2>&1 || { tail -30 gpurun_out/r3/depth_order_tests.log; exit 1; }
tail -3 gpurun_out/r3/depth_order_tests.log
timeout -k 10 300 python bench.py --config pogs_step > gpurun_out/r3/bench_pogs_step.json 2> gpurun_out/r3/bench_pogs_step.err || { tail -20 gpurun_out/r3/bench_pogs_step.err; exit 1; }
python - <<'PY'
import json
d = json.load(open("gpurun_out/r3/bench_pogs_step.json"))
print(d["value"], d["ms_per_step"], d["config"]["sum_of_calls_alone_ms"], "shared:", d["config"]["ms_per_step_shared_depth_order"])
for c in d["config"]["calls"]:
    print(c["call"], c["ms_alone"], c["stages_ms"])
PY
