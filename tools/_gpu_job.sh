cd $GRAFT_REPO_ROOT
mkdir -p /tmp/dpo
OCTMAE_DP_OUT=/tmp/dpo OCTMAE_DP_BACKEND=rccl_one_gpu NCCL_DEBUG=WARN timeout 120 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29621 tests/dp_worker.py 2>&1 | tail -15
cat /tmp/dpo/result.json 2>/dev/null
