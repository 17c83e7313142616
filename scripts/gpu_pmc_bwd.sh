# Counter passes over scripts/run_steps.py for the backward kernel the environment selects.
# usage: gpu_pmc_bwd.sh <tag> [kernel-name regex, default raster_bwd]   (POGS_RAST_* variables are inherited)
set -e
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r3/pmc_$1
mkdir -p $O
B="python3 scripts/run_steps.py 3"
timeout -s KILL 120 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt -- $B > $O/kt.log 2>&1 || { tail $O/kt.log; exit 1; }
timeout -s KILL 120 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY --output-format csv -d $O/sq1 -- $B > $O/sq1.log 2>&1 || { tail $O/sq1.log; exit 1; }
timeout -s KILL 120 rocprofv3 --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_LDS --output-format csv -d $O/sq2 -- $B > $O/sq2.log 2>&1 || { tail $O/sq2.log; exit 1; }
timeout -s KILL 120 rocprofv3 --pmc SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_VMEM SQ_IFETCH SQ_WAIT_INST_ANY SQ_INSTS_BRANCH --output-format csv -d $O/sq3 -- $B > $O/sq3.log 2>&1 || { tail -5 $O/sq3.log; }
timeout -s KILL 120 rocprofv3 --pmc GRBM_GUI_ACTIVE --output-format csv -d $O/grbm -- $B > $O/grbm.log 2>&1 || { tail -5 $O/grbm.log; }
python3 scripts/pmc_any.py "${2:-raster_bwd}" $O/sq1 $O/sq2 $O/sq3 $O/grbm > $O/summary.json
cat $O/summary.json
grep -h "raster_bwd\|raster_fwd" $O/kt/*/*kernel_stats.csv | cut -c1-60,200-400 | head
