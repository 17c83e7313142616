#!/bin/bash
# host cost of a step: the same call sequences on a scene so small that the GPU work is a fraction of the host's
set -e
mkdir -p gpurun_out/r3
timeout -k 10 300 python bench.py --config pogs_step --steps 200 --gaussians 100000 --width 192 --height 108 > gpurun_out/r3/host_pogs_step.json 2> gpurun_out/r3/host.err || { tail -20 gpurun_out/r3/host.err; exit 1; }
python - <<'PY'
import json
d = json.load(open("gpurun_out/r3/host_pogs_step.json"))
c = d["config"]
print("pogs_step small:", d["ms_per_step"], "sum alone", c["sum_of_calls_alone_ms"], "shared:", c["ms_per_step_shared_depth_order"], "host", c["host_enqueue_ms_per_step"])
for k in c["calls"]:
    print(k["call"], k["ms_alone"], sum(k["stages_ms"].values()))
PY
timeout -k 10 300 python bench.py --no-cpu-baseline --no-variants --steps 200 --gaussians 100000 --width 192 --height 108 > gpurun_out/r3/host_cfg2.json 2> gpurun_out/r3/host.err || { tail -20 gpurun_out/r3/host.err; exit 1; }
python -c "
import json; d=json.load(open('gpurun_out/r3/host_cfg2.json')); print('cfg2 small:', d['ms_per_step'], d.get('stages_ms_per_step'))"
