set -e
cd $GRAFT_REPO_ROOT
for pc in 0.5 0.5; do
timeout -k 10 300 python -u bench.py --steps 20 --warmup 5 --no-cpu-baseline --precondition-s $pc | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('precondition $pc', d['ms_per_step'], {k:round(v['ms_per_step'],3) for k,v in d['variants'].items()}, d['roofline']['frac'], d['roofline']['fp32'])"
done
