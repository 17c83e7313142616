#!/bin/bash
set -e
mkdir -p gpurun_out/r3
timeout -k 10 600 python -u -m pytest tests/test_gpu_parity.py tests/test_full_size.py -m gpu -x -q --timeout 180 --timeout-method thread -k "frozen_wide or tracking or raster_backward" > gpurun_out/r3/sparse.log 2>&1 || { grep -E "^E|assert|FAILED" gpurun_out/r3/sparse.log | head -20; exit 1; }
tail -2 gpurun_out/r3/sparse.log
timeout -k 10 500 python3 bench.py --config cfg3 > gpurun_out/r3/bench_cfg3.json 2> gpurun_out/r3/bench_cfg3.err || { tail gpurun_out/r3/bench_cfg3.err; exit 1; }
python3 -c "
import json; d=json.load(open('gpurun_out/r3/bench_cfg3.json')); c=d['config']; print('cfg3', d['value'], {k: round(v,1) for k, v in c.items() if k.startswith('ms_')})"
