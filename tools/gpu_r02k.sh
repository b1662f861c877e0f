cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests -m gpu -q -x -p no:cacheprovider > gpurun_out/r02k_pytest.log 2>&1
echo "pytest rc=$?" >> gpurun_out/r02k_pytest.log
tail -3 gpurun_out/r02k_pytest.log
B=$GRAFT_REPO_ROOT/bench.py
KT_LINES=6 tools/kt.sh r02k_train3072 -- python $B --workload train --cpu-rays 0 --steps 10 2>&1 | tail -5
python $B --workload train --cpu-rays 0 --steps 20 2>/dev/null | python -c "import sys,json; r=json.loads(sys.stdin.read()); print('train3072', round(r['ms_per_step'],3), round(r['roofline']['frac'],4))"
python $B --workload train --n-rand 384 --cpu-rays 0 --steps 40 2>/dev/null | python -c "import sys,json; r=json.loads(sys.stdin.read()); print('train384', round(r['ms_per_step'],3), round(r['roofline']['frac'],4))"
python $B --workload train_mixamo --cpu-rays 0 --steps 20 2>/dev/null | python -c "import sys,json; r=json.loads(sys.stdin.read()); print('mixamo', round(r['ms_per_step'],3), round(r['roofline']['frac'],4))"
