#!/bin/bash
# the round-end sequence: whole GPU suite, smoke(), default bench line; then a soak of the default step and the tracker
set -e
mkdir -p gpurun_out/r3
timeout -k 10 1000 python -u -m pytest tests -m gpu -x -q --timeout 180 --timeout-method thread > gpurun_out/r3/gpu_suite.log 2>&1 || { tail -60 gpurun_out/r3/gpu_suite.log; exit 1; }
tail -2 gpurun_out/r3/gpu_suite.log
timeout -k 10 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
timeout -k 10 400 python bench.py > gpurun_out/r3/bench_check.json 2> gpurun_out/r3/bench_check.err || { tail gpurun_out/r3/bench_check.err; exit 1; }
python -c "
import json; d=json.load(open('gpurun_out/r3/bench_check.json')); print('default', d['value'], d['ms_per_step'])"
timeout -k 10 400 python bench.py --steps 1200 --warmup 5 --no-cpu-baseline --no-variants > gpurun_out/r3/bench_soak.json 2> gpurun_out/r3/bench_soak.err || { tail gpurun_out/r3/bench_soak.err; exit 1; }
python -c "
import json; d=json.load(open('gpurun_out/r3/bench_soak.json')); print('soak 1200 steps', d['value'], d['ms_per_step'])"
FRAMES=20 timeout -k 10 300 python3 scripts/profile_tracker.py 2>&1 | grep ms/frame
