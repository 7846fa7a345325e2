cd $GRAFT_REPO_ROOT
export EAVSR_DIST_BACKEND=gloo
timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 2 --steps 2 --warmup 1 --clips 2 --no-cpu-baseline 2>&1 | tail -2 | cut -c1-300
# the training step (eager; a graphed step with two ranks sharing one device interleaves pathologically)
timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29512 bench.py --gpus 2 --mode train --steps 2 --warmup 1 2>&1 | tail -1 | cut -c1-420
