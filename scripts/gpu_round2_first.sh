set -e
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 900 python -u -m pytest tests/test_gpu_parity.py -x -q -m gpu --timeout 300 --timeout-method thread > gpurun_out/t1_parity.log 2>&1 || { tail -40 gpurun_out/t1_parity.log; exit 1; }
tail -3 gpurun_out/t1_parity.log
timeout -k 10 300 python -u bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/b1_default.json 2> gpurun_out/b1_default.err || { tail -20 gpurun_out/b1_default.err; exit 1; }
cat gpurun_out/b1_default.json
POGS_RAST_FWD_MFMA=1 timeout -k 10 300 python -u bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-variants --views 1 > gpurun_out/b1_mfma_1view.json 2> gpurun_out/b1_mfma.err || { tail -20 gpurun_out/b1_mfma.err; exit 1; }
cat gpurun_out/b1_mfma_1view.json
