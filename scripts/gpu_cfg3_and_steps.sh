#!/bin/bash
# cfg3 bench line, the POGS three-call step and the default line (quick), after a planner change
set -e
mkdir -p gpurun_out/r3
timeout -k 10 500 python bench.py --config cfg3 > gpurun_out/r3/bench_cfg3.json 2> gpurun_out/r3/bench_cfg3.err || { tail -20 gpurun_out/r3/bench_cfg3.err; exit 1; }
python - <<'PY'
import json
d = json.load(open("gpurun_out/r3/bench_cfg3.json"))
c = d["config"]
print(d["value"], {k: v for k, v in c.items() if k.startswith("ms_") or k.startswith("loss")})
print(c["stages_ms_per_iteration"])
PY
timeout -k 10 300 python bench.py --config pogs_step --steps 40 > gpurun_out/r3/bench_pogs_step.json 2> gpurun_out/r3/bench_pogs_step.err || { tail -20 gpurun_out/r3/bench_pogs_step.err; exit 1; }
python - <<'PY'
import json
d = json.load(open("gpurun_out/r3/bench_pogs_step.json"))
c = d["config"]
print(d["ms_per_step"], "sum alone", c["sum_of_calls_alone_ms"], "shared:", c["ms_per_step_shared_depth_order"])
for k in c["calls"]:
    print(k["call"], k["ms_alone"], k["kernels"])
PY
timeout -k 10 300 python bench.py --config cfg1 --no-cpu-baseline --no-variants > gpurun_out/r3/bench_cfg1.json 2> gpurun_out/r3/bench_cfg1.err || { tail -20 gpurun_out/r3/bench_cfg1.err; exit 1; }
python -c "
import json; d=json.load(open('gpurun_out/r3/bench_cfg1.json')); print('cfg1', d['value'], d['ms_per_step'])"
