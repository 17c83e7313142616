#!/bin/bash
# the whole GPU suite, then the POGS three-call step; logs under gpurun_out/r3
set -e
mkdir -p gpurun_out/r3
timeout -k 10 1000 python -u -m pytest tests -m gpu -x -q --timeout 180 --timeout-method thread > gpurun_out/r3/gpu_suite.log 2>&1 || { tail -40 gpurun_out/r3/gpu_suite.log; exit 1; }
tail -3 gpurun_out/r3/gpu_suite.log
timeout -k 10 300 python bench.py --config pogs_step > gpurun_out/r3/bench_pogs_step.json 2> gpurun_out/r3/bench_pogs_step.err || { tail -20 gpurun_out/r3/bench_pogs_step.err; exit 1; }
python - <<'PY'
import json
d = json.load(open("gpurun_out/r3/bench_pogs_step.json"))
print(d["value"], d["ms_per_step"], d["config"]["sum_of_calls_alone_ms"], "shared:", d["config"]["ms_per_step_shared_depth_order"])
for c in d["config"]["calls"]:
    print(c["call"], c["ms_alone"], c["stages_ms"])
PY
timeout -k 10 300 python bench.py --no-cpu-baseline --no-variants > gpurun_out/r3/bench_default_quick.json 2> gpurun_out/r3/bench_default_quick.err || { tail -20 gpurun_out/r3/bench_default_quick.err; exit 1; }
python -c "
import json; d=json.load(open('gpurun_out/r3/bench_default_quick.json')); print(d['value'], d['ms_per_step'])"
