cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_dp_on_device.py tests/test_hip_backward.py tests/test_abi_exports.py -m gpu -q -x -p no:cacheprovider 2>&1 | tail -15
ANERF_BENCH_BACKEND=gloo python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 2 --workload train --steps 5 --cpu-rays 0 2>/dev/null | tail -1 | cut -c1-300
