# Kernel trace of the tracking loop's default configuration (scripts/profile_tracker.py): per-kernel totals and launches
# per iteration.
set -e
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r3/cfg3_${1:-trace}
mkdir -p $O
timeout -k 10 200 python3 scripts/profile_tracker.py > $O/plain.log 2>&1 || { tail $O/plain.log; exit 1; }
cat $O/plain.log
timeout -s KILL 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt -- python3 scripts/profile_tracker.py > $O/kt.log 2>&1 || { tail $O/kt.log; exit 1; }
python3 - $O <<'PY'
import csv, glob, sys
o = sys.argv[1]
f = glob.glob(o + "/kt/*/*kernel_stats.csv")[0]
rows = list(csv.DictReader(open(f)))
calls = sum(int(r["Calls"]) for r in rows)
tot = sum(float(r["TotalDurationNs"]) for r in rows)
# 1 warm-up frame + 3 frames of 50 iterations (+ set-up)
print("kernel launches %d, GPU kernel time %.1f ms; per iteration (200 iterations): %.1f launches, %.3f ms" % (calls, tot / 1e6, calls / 200, tot / 1e6 / 200))
for r in rows[:60]:
    print("%-100s %6s %9.1f us %6.2f%%" % (r["Name"][:100], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["Percentage"])))
import shutil
shutil.copy(f, o + "/kernel_stats.csv")
PY
