cd $GRAFT_REPO_ROOT
B=$GRAFT_REPO_ROOT/bench.py
echo "== nccl world-1, train, overlap path"
ANERF_BENCH_FORCE_DIST=1 timeout 300 python $B --workload train --cpu-rays 0 --steps 10 2>gpurun_out/r02r_err.txt | python -c "import sys,json; r=json.loads(sys.stdin.read()); print('train3072 nccl1', round(r['ms_per_step'],3), r['config']['parallelism'])"
tail -3 gpurun_out/r02r_err.txt
ANERF_BENCH_FORCE_DIST=1 timeout 300 python $B --workload train_mixamo --cpu-rays 0 --steps 10 --opt-pose-step 3 2>/dev/null | python -c "import sys,json; r=json.loads(sys.stdin.read()); print('mixamo nccl1', round(r['ms_per_step'],3))"
ANERF_BENCH_FORCE_DIST=1 timeout 300 python $B --cpu-rays 0 --extra off --steps 3 2>/dev/null | python -c "import sys,json; r=json.loads(sys.stdin.read()); print('render nccl1', round(r['ms_per_step'],3))"
echo "== gloo 2 ranks"
for wl in train render64; do ANERF_BENCH_BACKEND=gloo timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29512 bench.py --gpus 2 --workload $wl --steps 3 --cpu-rays 0 2>/dev/null | tail -1 | python -c "import sys,json; r=json.loads(sys.stdin.read()); print(r['config']['workload'][:40], r['n_gpus'], round(r['ms_per_step'],3), round(r['value']))"; done
