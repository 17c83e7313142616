# bench.py on the other workload shapes (single GPU share of cfg4 / cfg5, the plumbing shape cfg1); one JSON line each
set -e
mkdir -p gpurun_out/r3
for c in cfg1 cfg4 cfg5; do
  timeout -k 10 400 python bench.py --config $c --no-cpu-baseline --no-variants > gpurun_out/r3/bench_$c.json 2> gpurun_out/r3/bench_$c.err || { tail -20 gpurun_out/r3/bench_$c.err; exit 1; }
  python - $c <<'PY'
import json, sys
d = json.load(open("gpurun_out/r3/bench_%s.json" % sys.argv[1]))
print(sys.argv[1], round(d["value"], 1), d["unit"], round(d["ms_per_step"], 3), "ms", d["roofline"]["kernel"], round(d["roofline"]["frac"], 3), d["stages_ms_per_step"])
PY
done
