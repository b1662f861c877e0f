#!/bin/bash
# round 5, call 27: where the end-to-end loop's time goes -- rocprofv3 kernel trace of tools/train_synthetic.py (600 iterations, graphs on)
ROOT=$GRAFT_REPO_ROOT; O=$ROOT/gpurun_out/r05_call27; mkdir -p $O /tmp/prof
cd /tmp && export TMPDIR=/tmp
for mode in plain refine; do
  if [ $mode = plain ]; then cfg="--iters 600"; else cfg="--subject spheres --pose-noise 0.05 --iters 600 --pose-step 4"; fi
  rocprofv3 --kernel-trace --stats -d /tmp/prof/e2e_$mode -- python $ROOT/tools/train_synthetic.py $cfg > /tmp/prof/e2e_$mode.log 2>/tmp/prof/e2e_$mode.err
  python $ROOT/tools/rocprof_summary.py /tmp/prof/e2e_${mode}_kt.txt /tmp/prof/e2e_$mode > /dev/null 2>&1
  { echo "# rocprofv3 --kernel-trace --stats -- python tools/train_synthetic.py $cfg"; grep "^{" /tmp/prof/e2e_$mode.log | python -c "
import sys,json
d=json.loads(sys.stdin.read()); print('# it/s under the profiler: %.1f  (wall of the training loop %.3f s for %d iterations; the trace also holds the dataset build, the checkpoint renders and the test render)' % (d['it_per_s'], d['iters']/d['it_per_s'], d['iters']))"
    cat /tmp/prof/e2e_${mode}_kt.txt
  } > $O/e2e_${mode}_kernel_stats.txt 2>&1
  head -30 $O/e2e_${mode}_kernel_stats.txt | cut -c1-150
done
