# Counter passes over scripts/heads_once.py for the dense kernels.  usage: gpu_pmc_heads.sh <tag>
set -e
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r3/pmc_heads_$1
mkdir -p $O
B="python3 scripts/heads_once.py"
timeout -s KILL 120 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY --output-format csv -d $O/sq1 -- $B > $O/sq1.log 2>&1 || { tail $O/sq1.log; exit 1; }
timeout -s KILL 120 rocprofv3 --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_LDS --output-format csv -d $O/sq2 -- $B > $O/sq2.log 2>&1 || { tail $O/sq2.log; exit 1; }
timeout -s KILL 120 rocprofv3 --pmc SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_VMEM SQ_IFETCH SQ_WAIT_INST_ANY SQ_INSTS_BRANCH --output-format csv -d $O/sq3 -- $B > $O/sq3.log 2>&1 || { tail -5 $O/sq3.log; }
timeout -s KILL 120 rocprofv3 --pmc GRBM_GUI_ACTIVE --output-format csv -d $O/grbm -- $B > $O/grbm.log 2>&1 || { tail -5 $O/grbm.log; }
timeout -s KILL 120 rocprofv3 --pmc TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_TA_TCP_STATE_READ_sum --output-format csv -d $O/tcp -- $B > $O/tcp.log 2>&1 || { tail -5 $O/tcp.log; }
python3 scripts/pmc_any.py "dense_" $O/sq1 $O/sq2 $O/sq3 $O/grbm $O/tcp > $O/summary.json
cat $O/summary.json
