# Rehearsal of bench.py's N > 1 path on the one GPU of the box: two ranks share the device, gloo moves the buffers
# through the host (RCCL refuses two ranks on one GPU), so the numbers mean nothing -- the code path is what runs.
set -e
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export POGS_BENCH_SHARE_GPU=1 POGS_BENCH_BACKEND=gloo MASTER_ADDR=127.0.0.1
for ex in sharded allreduce; do
  timeout -k 10 400 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 2 --steps 3 --warmup 1 --gaussians 200000 --width 960 --height 540 --exchange $ex > gpurun_out/rehearse_$ex.json 2> gpurun_out/rehearse_$ex.err || { tail -30 gpurun_out/rehearse_$ex.err; exit 1; }
  cat gpurun_out/rehearse_$ex.json
done
