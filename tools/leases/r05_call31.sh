#!/bin/bash
# round 5, call 31: kernel trace of the end-to-end loop at 3072 rays (where do the 1.2 ms beyond the bench's step go?)
ROOT=$GRAFT_REPO_ROOT; O=$ROOT/gpurun_out/r05_call31; mkdir -p $O /tmp/prof
cd /tmp && export TMPDIR=/tmp
cfg="--subject spheres --n-kps 8 --n-cams 6 --hw 128 --n-rand 3072 --n-sample-images 24 --iters 400"
rocprofv3 --kernel-trace --memory-copy-trace --stats -d /tmp/prof/e2e3072 -- python $ROOT/tools/train_synthetic.py $cfg > /tmp/prof/e2e3072.log 2>/tmp/prof/e2e3072.err
python $ROOT/tools/rocprof_summary.py /tmp/prof/e2e3072_kt.txt /tmp/prof/e2e3072 > /dev/null 2>&1
{ echo "# rocprofv3 --kernel-trace --memory-copy-trace --stats -- python tools/train_synthetic.py $cfg"; grep "^{" /tmp/prof/e2e3072.log | python -c "
import sys,json
d=json.loads(sys.stdin.read()); print('# it/s under the profiler: %.1f (%.2f ms per iteration)' % (d['it_per_s'], 1e3/d['it_per_s']))"
  head -32 /tmp/prof/e2e3072_kt.txt
  python - <<'PY'
import glob,sqlite3
f=glob.glob("/tmp/prof/e2e3072/**/*_results.db",recursive=True)[0]
con=sqlite3.connect(f)
tabs=[r[0] for r in con.execute("select name from sqlite_master where type in ('table','view')")]
mc=[t for t in tabs if 'memory_cop' in t.lower()]
print("# memory-copy tables:", mc[:6])
for t in mc[:3]:
    try:
        cols=[r[1] for r in con.execute(f"pragma table_info({t})")]
        print("#", t, cols)
    except Exception as e: print(e)
PY
} > $O/e2e_3072_kernel_stats.txt 2>&1
head -40 $O/e2e_3072_kernel_stats.txt | cut -c1-160
