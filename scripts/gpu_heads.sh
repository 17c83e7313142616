#!/bin/bash
# heads: tests, timing against torch, kernel trace of the fused path
set -e
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r3/heads_${1:-run}
mkdir -p $O
timeout -k 10 300 python -u -m pytest tests/test_heads.py -m gpu -x -q --timeout 120 --timeout-method thread > $O/tests.log 2>&1 || { tail -30 $O/tests.log; exit 1; }
tail -2 $O/tests.log
timeout -k 10 300 python3 scripts/bench_heads.py > $O/bench.log 2>&1 || { tail $O/bench.log; exit 1; }
cat $O/bench.log
timeout -s KILL 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt -- python3 scripts/bench_heads.py --reps 3 > $O/kt.log 2>&1 || { tail $O/kt.log; exit 1; }
python3 - $O <<'PY'
import csv, glob, sys
o = sys.argv[1]
f = glob.glob(o + "/kt/*/*kernel_stats.csv")[0]
rows = list(csv.DictReader(open(f)))
for r in rows[:14]:
    if "pr::" in r["Name"]:
        print("%-90s %6s %9.1f us" % (r["Name"][:90], r["Calls"], float(r["AverageNs"]) / 1e3))
PY
