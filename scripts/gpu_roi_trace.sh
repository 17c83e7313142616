set -e
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r3/roi_trace
mkdir -p $O
timeout -s KILL 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt -- python3 scripts/roi_stages.py > $O/kt.log 2>&1 || { tail $O/kt.log; exit 1; }
python3 - $O <<'PY'
import csv, glob, sys
o = sys.argv[1]
f = glob.glob(o + "/kt/*/*kernel_stats.csv")[0]
rows = list(csv.DictReader(open(f)))
calls = sum(int(r["Calls"]) for r in rows)
tot = sum(float(r["TotalDurationNs"]) for r in rows)
ours = sum(float(r["TotalDurationNs"]) for r in rows if "pr::" in r["Name"] or r["Name"].startswith(("tile_", "publish", "isect")))
print("launches/iter %.1f kernel ms/iter %.3f ours %.3f" % (calls / 250, tot / 1e6 / 250, ours / 1e6 / 250))
for r in rows[:45]:
    print("%-100s %6s %9.1f us %6.2f%%" % (r["Name"][:100], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["Percentage"])))
PY
