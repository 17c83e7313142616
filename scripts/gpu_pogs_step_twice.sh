#!/bin/bash
# the POGS three-call step, twice in a row (box-to-box and run-to-run spread), with the host's enqueue time
set -e
mkdir -p gpurun_out/r3
for i in 1 2; do
timeout -k 10 300 python bench.py --config pogs_step --steps 40 > gpurun_out/r3/bench_pogs_step_$i.json 2> gpurun_out/r3/bench_pogs_step.err || { tail -20 gpurun_out/r3/bench_pogs_step.err; exit 1; }
python - $i <<'PY'
import json, sys
d = json.load(open("gpurun_out/r3/bench_pogs_step_%s.json" % sys.argv[1]))
c = d["config"]
print(d["ms_per_step"], "sum alone", c["sum_of_calls_alone_ms"], "shared:", c["ms_per_step_shared_depth_order"], "host", c["host_enqueue_ms_per_step"])
PY
done
