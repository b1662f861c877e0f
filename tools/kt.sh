#!/bin/bash
# kernel-trace one command on the GPU box and print the top kernels (raw DBs stay in /tmp).  usage: tools/kt.sh TAG -- cmd...
TAG=$1; shift; shift
ROOT=$(cd "$(dirname "$0")/.." && pwd)
mkdir -p /tmp/prof; cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d /tmp/prof/${TAG}_kt -- "$@" > /tmp/prof/${TAG}_kt.log 2>&1
python $ROOT/tools/rocprof_summary.py /tmp/prof/${TAG}.txt /tmp/prof/${TAG}_kt > /dev/null; head -${KT_LINES:-6} /tmp/prof/${TAG}.txt | cut -c1-110
