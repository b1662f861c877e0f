#!/bin/bash
# extra PMC groups (instruction fetch / per-class issue cycles) on one command.  usage: tools/pmc_extra.sh TAG -- <command...>
TAG=$1; shift; shift
ROOT=$(cd "$(dirname "$0")/.." && pwd)
mkdir -p $ROOT/gpurun_out /tmp/prof
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d /tmp/prof/${TAG}_kt -- "$@" > /tmp/prof/${TAG}_kt.log 2>/tmp/prof/${TAG}_kt.err
dirs=""
for grp in "SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE" "SQ_IFETCH SQ_IFETCH_LEVEL SQC_TC_INST_REQ SQC_TC_STALL" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA" "SQ_ACTIVE_INST_MISC SQ_VALU_MFMA_COEXEC_CYCLES SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_SALU" \
           "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_INSTS_SALU"; do
  name=$(echo $grp | cut -d' ' -f1)
  rocprofv3 --kernel-trace --pmc $grp -d /tmp/prof/${TAG}_pmc_$name -- "$@" > /tmp/prof/${TAG}_pmc_$name.log 2>&1
  tail -2 /tmp/prof/${TAG}_pmc_$name.log
  dirs="$dirs /tmp/prof/${TAG}_pmc_$name"
done
python $ROOT/tools/rocprof_summary.py $ROOT/gpurun_out/${TAG}_summary.txt /tmp/prof/${TAG}_kt $dirs
