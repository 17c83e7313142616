# Kernel trace of bench.py --config pogs_step: per-kernel totals of the three renders of a POGS iteration.
set -e
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r3/pogs_step_trace
mkdir -p $O
timeout -s KILL 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt -- python3 bench.py --config pogs_step --steps 4 --warmup 2 --precondition-s 0 > $O/kt.log 2>&1 || { tail $O/kt.log; exit 1; }
python3 - $O <<'PY'
import csv, glob, sys
o = sys.argv[1]
rows = list(csv.DictReader(open(glob.glob(o + "/kt/*/*kernel_stats.csv")[0])))
for r in rows[:24]:
    print("%-100s %5s %9.1f us %6.2f%%" % (r["Name"][:100], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["Percentage"])))
PY
