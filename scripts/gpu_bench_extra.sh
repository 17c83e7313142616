#!/bin/bash
# bench.py's POGS three-call step and the like-for-like single-GPU optimiser step; results under gpurun_out/r3
set -e
mkdir -p gpurun_out/r3
timeout -k 10 300 python bench.py --config pogs_step > gpurun_out/r3/bench_pogs_step.json 2> gpurun_out/r3/bench_pogs_step.err || { tail -20 gpurun_out/r3/bench_pogs_step.err; exit 1; }
python - <<'PY'
import json
d = json.load(open("gpurun_out/r3/bench_pogs_step.json"))
print(d["value"], d["ms_per_step"], d["config"]["sum_of_calls_alone_ms"])
for c in d["config"]["calls"]:
    print(c["call"], c["ms_alone"], c["n_isects_composited"], c["kernels"], c["stages_ms"])
print(d["roofline"])
PY
timeout -k 10 300 python bench.py --no-cpu-baseline --with-optimizer --no-variants > gpurun_out/r3/bench_opt.json 2> gpurun_out/r3/bench_opt.err || { tail -20 gpurun_out/r3/bench_opt.err; exit 1; }
python - <<'PY'
import json
d = json.load(open("gpurun_out/r3/bench_opt.json"))
print(d["value"], d["ms_per_step"], d["config"]["parallelism"])
print(d["stages_ms_per_step"])
PY
