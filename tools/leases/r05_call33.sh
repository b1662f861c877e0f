#!/bin/bash
# round 5, call 33: rocprofv3 --kernel-trace --stats of the DRIVER'S bench command itself (headline kernel's average launch duration
# next to the line's avg_launch_ms)
ROOT=$GRAFT_REPO_ROOT; O=$ROOT/gpurun_out/r05_call33; mkdir -p $O /tmp/prof
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d /tmp/prof/drv -- python $ROOT/bench.py --gpus 1 --steps 20 --warmup 5 > /tmp/prof/drv.log 2>/tmp/prof/drv.err
python $ROOT/tools/rocprof_summary.py /tmp/prof/drv_kt.txt /tmp/prof/drv > /dev/null 2>&1
{ echo "# rocprofv3 --kernel-trace --stats -- python bench.py --gpus 1 --steps 20 --warmup 5   (the driver's command; all default extras ran under the trace)"
  tail -1 /tmp/prof/drv.log | python -c "
import sys,json
d=json.loads(sys.stdin.read()); r=d['roofline']; print('# the line of this run: value %.0f rays/s, ms_per_step %.3f, roofline.avg_launch_ms %.3f, frac %.4f, traffic_source: %s' % (d['value'], d['ms_per_step'], r['avg_launch_ms'], r['frac'], r.get('traffic_source')))"
  head -24 /tmp/prof/drv_kt.txt; } > $O/driver_command_kernel_stats.txt 2>&1
head -12 $O/driver_command_kernel_stats.txt | cut -c1-200
