set -e
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/prof2
mkdir -p $O
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt -- python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-variants --views 1 > $O/kt.log 2>&1 || { tail $O/kt.log; exit 1; }
python3 - <<'PY'
import csv,glob
f=glob.glob('/root/repo/gpurun_out/prof2/kt/*/*_kernel_stats.csv')[0]
for r in list(csv.DictReader(open(f)))[:14]:
    print(f"{r['Name'][:80]:80s} calls={r['Calls']:>5s} avg_us={float(r['AverageNs'])/1e3:9.1f}")
PY
