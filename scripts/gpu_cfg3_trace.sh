# Kernel trace of the tracking loop (bench.py --config cfg3): per-kernel totals and launches per iteration.
set -e
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r3/cfg3_${1:-trace}
mkdir -p $O
timeout -k 10 300 python3 bench.py --config cfg3 --steps 3 --warmup 1 --no-cpu-baseline > $O/bench.json 2> $O/bench.err || { tail $O/bench.err; exit 1; }
cat $O/bench.json
timeout -s KILL 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt -- python3 bench.py --config cfg3 --steps 1 --warmup 1 --no-cpu-baseline > $O/kt.log 2>&1 || { tail $O/kt.log; exit 1; }
python3 - $O <<'PY'
import csv, glob, sys
o = sys.argv[1]
f = glob.glob(o + "/kt/*/*kernel_stats.csv")[0]
rows = list(csv.DictReader(open(f)))
calls = sum(int(r["Calls"]) for r in rows)
tot = sum(float(r["TotalDurationNs"]) for r in rows)
# 1 warm-up frame + 1 timed frame, each in two variants (transform fused / own pass): 4 frames of 50 iterations (+ set-up)
print("kernel launches %d, GPU kernel time %.1f ms; per iteration (200 iterations): %.1f launches, %.3f ms" % (calls, tot / 1e6, calls / 200, tot / 1e6 / 200))
for r in rows[:32]:
    print("%-86s %6s %9.1f us %6.2f%%" % (r["Name"][:86], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["Percentage"])))
PY
