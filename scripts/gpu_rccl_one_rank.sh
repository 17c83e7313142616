#!/bin/bash
# the N > 1 path of bench.py with ONE rank on the nccl (RCCL) backend: process group on the device, barriers, the
# exchange's reduce-scatter / all-gather (library form), all-to-all (direct form) and the all-reduce exchange
set -e
mkdir -p gpurun_out/r3
export POGS_BENCH_FORCE_DIST=1 POGS_EXCHANGE_SINGLE_RANK_COLLECTIVES=1
for form in library direct allreduce; do
  extra=""; unset POGS_EXCHANGE_DIRECT
  if [ $form = direct ]; then export POGS_EXCHANGE_DIRECT=1; fi
  if [ $form = allreduce ]; then extra="--exchange allreduce"; fi
  timeout -k 10 240 python bench.py --gpus 1 --steps 10 --warmup 3 $extra > gpurun_out/r3/rccl_one_rank_$form.json 2> gpurun_out/r3/rccl_one_rank_$form.err || { tail -30 gpurun_out/r3/rccl_one_rank_$form.err; exit 1; }
  python - $form <<'PY'
import json, sys
d = json.load(open("gpurun_out/r3/rccl_one_rank_%s.json" % sys.argv[1]))
print(sys.argv[1], d["value"], d["ms_per_step"], d["config"]["parallelism"])
PY
done
