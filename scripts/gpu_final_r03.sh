# Final measurements of the tree: default bench line (with CPU baseline and variants), its kernel trace, the POGS
# three-call step, the tracker (bench line + kernel trace of the default loop), and a fuzz slice with a fresh seed.
set -e
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r3/final
mkdir -p $O
timeout -k 10 400 python3 bench.py > $O/bench_default.json 2> $O/bench_default.err || { tail $O/bench_default.err; exit 1; }
python3 -c "
import json; d=json.load(open('$O/bench_default.json')); print('default', d['value'], d['ms_per_step'], d['roofline']['frac'], d['cpu_baseline']['value'])"
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt_cfg2 -- python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-variants > $O/kt_cfg2.log 2>&1 || { tail $O/kt_cfg2.log; exit 1; }
cp $O/kt_cfg2/*/*kernel_stats.csv $O/cfg2_kernel_stats.csv
echo "cfg2 trace done"
timeout -k 10 300 python3 bench.py --config pogs_step --steps 40 > $O/bench_pogs_step.json 2> $O/bench_pogs_step.err || { tail $O/bench_pogs_step.err; exit 1; }
python3 -c "
import json; d=json.load(open('$O/bench_pogs_step.json')); c=d['config']; print('pogs_step', d['ms_per_step'], c['ms_per_step_shared_depth_order'], [k['ms_alone'] for k in c['calls']])"
timeout -k 10 500 python3 bench.py --config cfg3 > $O/bench_cfg3.json 2> $O/bench_cfg3.err || { tail $O/bench_cfg3.err; exit 1; }
python3 -c "
import json; d=json.load(open('$O/bench_cfg3.json')); c=d['config']; print('cfg3', d['value'], {k: round(v,1) for k, v in c.items() if k.startswith('ms_')}); print(c['stages_ms_per_iteration'])"
timeout -s KILL 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt_cfg3 -- python3 scripts/profile_tracker.py > $O/kt_cfg3.log 2>&1 || { tail $O/kt_cfg3.log; exit 1; }
cp $O/kt_cfg3/*/*kernel_stats.csv $O/cfg3_kernel_stats.csv
echo "cfg3 trace done"
timeout -k 10 280 python3 bench.py --with-optimizer --no-cpu-baseline --no-variants > $O/bench_with_optimizer.json 2> $O/bench_opt.err || { tail $O/bench_opt.err; exit 1; }
python3 -c "
import json; d=json.load(open('$O/bench_with_optimizer.json')); print('with optimizer', d['value'], d['ms_per_step'])"
timeout -k 10 500 python3 -u scripts/fuzz_parity.py 1500 23 > $O/fuzz.log 2>&1 || { tail -30 $O/fuzz.log; exit 1; }
tail -1 $O/fuzz.log
