#!/bin/bash
set -e
mkdir -p gpurun_out/r3/final
timeout -k 10 500 python3 bench.py --config cfg3 > gpurun_out/r3/final/bench_cfg3.json 2> gpurun_out/r3/final/bench_cfg3.err || { tail gpurun_out/r3/final/bench_cfg3.err; exit 1; }
python3 -c "
import json; d=json.load(open('gpurun_out/r3/final/bench_cfg3.json')); c=d['config']; print('cfg3', d['value'], {k: round(v,1) for k, v in c.items() if k.startswith('ms_')}); print(c['stages_ms_per_iteration'])"
