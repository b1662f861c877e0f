#!/bin/bash
# one PMC pass on the GPU box; prints the per-kernel averages for kernels matching $KERNEL_RE.  usage: tools/pmc_one.sh TAG "CTR1 CTR2" -- cmd...
TAG=$1; CTRS=$2; shift; shift; shift
ROOT=$(cd "$(dirname "$0")/.." && pwd)
mkdir -p /tmp/prof; cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d /tmp/prof/${TAG}_kt -- "$@" > /tmp/prof/${TAG}_kt.log 2>&1
rocprofv3 --kernel-trace --pmc $CTRS -d /tmp/prof/${TAG}_pmc -- "$@" > /tmp/prof/${TAG}_pmc.log 2>&1
python $ROOT/tools/rocprof_summary.py /tmp/prof/${TAG}.txt /tmp/prof/${TAG}_kt /tmp/prof/${TAG}_pmc > /dev/null
grep -E "${KERNEL_RE:-anerf::k_}" -A8 /tmp/prof/${TAG}.txt | grep -v "at::" | cut -c1-110
