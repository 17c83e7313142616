# The driver's round-end GPU tier, rehearsed: the whole -m gpu suite in one process, then smoke().
set -e
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 1100 python -u -m pytest tests -x -q -m gpu --timeout 300 --timeout-method thread > gpurun_out/full_suite.log 2>&1 || { tail -60 gpurun_out/full_suite.log; exit 1; }
tail -3 gpurun_out/full_suite.log
timeout -k 10 120 python -u -m pytest tests/test_zz_no_alloc.py -x -q -m gpu --timeout 120 --timeout-method thread 2>&1 | tail -2
timeout -k 10 200 python -u -c "import __graft_entry__ as g; g.smoke()"
