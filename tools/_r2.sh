cd /root/repo
IMMUNOSTRUCT_DIST_BACKEND=gloo IMMUNOSTRUCT_FORCE_DEVICE=0 timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29521 bench.py --gpus 2 --steps 10 --warmup 3 2>&1 | tail -3 | cut -c1-1500
