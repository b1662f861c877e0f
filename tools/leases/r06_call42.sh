#!/bin/bash
# round 6, call 42: k_reduce_dw2 with eighteen partials in flight + 32-bit index arithmetic -- same bits?  how long?
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $O
export TMPDIR=/tmp
{
echo "== gradient digests, library before the change"; ANERF_LIB=$GRAFT_REPO_ROOT/tools/exp/libanerf_before_reduce.so python tools/diag/grad_digest.py 2>&1 | grep rays
echo "== gradient digests, library after the change";  python tools/diag/grad_digest.py 2>&1 | grep rays
for lib in before after; do
  for wl in "train 1" "train_mixamo 20"; do
    set -- $wl
    L=""; [ $lib = before ] && L=$GRAFT_REPO_ROOT/tools/exp/libanerf_before_reduce.so
    rm -rf /tmp/prof_$lib
    ( [ -n "$L" ] && export ANERF_LIB=$L; cd /tmp && rocprofv3 --kernel-trace --stats -d /tmp/prof_$lib -o t -- python $GRAFT_REPO_ROOT/bench.py --workload $1 --n-rand 384 --opt-pose-step $2 --steps 60 --warmup 5 --extra off --cpu-rays 0 --graph on > /tmp/prof_$lib.out 2>&1 )
    echo "== $lib, $1 384 rays: $(grep -o '"ms_per_step": [0-9.]*' /tmp/prof_$lib.out | head -1)"
    f=$(find /tmp/prof_$lib -name "*kernel_stats.csv" | head -1)
    python - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows:
    if "reduce_dw" in r["Name"] or "k_gemm_tn" in r["Name"] or "k_adam" in r["Name"]:
        print(f"   {r['Name'][:60]:60s} calls {r['Calls']:>5s}  avg {float(r['AverageNs'])/1e3:8.2f} us")
PY
  done
done
} > $O/r06_reduce_dw_18_ab.txt 2>&1
cat $O/r06_reduce_dw_18_ab.txt
