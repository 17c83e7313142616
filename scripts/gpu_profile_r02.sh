# Round-2 profile collection on the GPU box: counter list, diagnostic pair counts, kernel trace, PMC passes.
set -e
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/prof
mkdir -p $O
rocprofv3 -L > $O/counters.txt 2>&1 || true
POGS_RAST_LIB=$GRAFT_REPO_ROOT/pogs_amd/lib_diag/libpogs_rast.so timeout -k 10 200 python -u scripts/diag_bwd.py 1000000 1920 1080 32 16 fused > $O/diag_bwd_cfg2.json 2> $O/diag.err || { tail $O/diag.err; exit 1; }
cat $O/diag_bwd_cfg2.json
B="python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-variants --views 1"
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt -- python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-variants > $O/kt.log 2>&1 || { tail $O/kt.log; exit 1; }
timeout -s KILL 120 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY --output-format csv -d $O/pmc_sq1 -- $B > $O/pmc_sq1.log 2>&1 || { tail $O/pmc_sq1.log; exit 1; }
timeout -s KILL 120 rocprofv3 --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_LDS --output-format csv -d $O/pmc_sq2 -- $B > $O/pmc_sq2.log 2>&1 || { tail $O/pmc_sq2.log; exit 1; }
timeout -s KILL 120 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch -- $B > $O/pmc_fetch.log 2>&1 || { tail $O/pmc_fetch.log; exit 1; }
timeout -s KILL 120 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pmc_write -- $B > $O/pmc_write.log 2>&1 || { tail $O/pmc_write.log; exit 1; }
find $O -name "*.csv" | head -30
