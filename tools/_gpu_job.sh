cd $GRAFT_REPO_ROOT
timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29631 bench.py --gpus 2 --steps 2 --warmup 1 --global-batch 16 --gloo-one-gpu 2>&1 | tail -4 | cut -c1-1500
