set -e
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 300 python -u -m pytest tests/wip_no_alloc.py -x -q -m gpu --timeout 120 --timeout-method thread > gpurun_out/t2_noalloc.log 2>&1 || { tail -40 gpurun_out/t2_noalloc.log; exit 1; }
tail -3 gpurun_out/t2_noalloc.log
timeout -k 10 600 python -u -m pytest tests/test_host_glue.py tests/test_full_size.py tests/test_gpu_parity.py::test_two_threads_on_two_streams_render_independently tests/test_gpu_parity.py::test_binning_with_lists_sized_from_a_guess -x -q -m gpu --timeout 300 --timeout-method thread > gpurun_out/t2_new.log 2>&1 || { tail -60 gpurun_out/t2_new.log; exit 1; }
tail -3 gpurun_out/t2_new.log
timeout -k 10 300 python -u bench.py --config cfg3 --steps 3 --warmup 1 > gpurun_out/b2_cfg3.json 2> gpurun_out/b2_cfg3.err || { tail -20 gpurun_out/b2_cfg3.err; exit 1; }
cat gpurun_out/b2_cfg3.json
timeout -k 10 300 python -u bench.py --config cfg5 --steps 5 --warmup 2 --no-cpu-baseline --no-variants > gpurun_out/b2_cfg5.json 2> gpurun_out/b2_cfg5.err || { tail -20 gpurun_out/b2_cfg5.err; exit 1; }
cat gpurun_out/b2_cfg5.json
