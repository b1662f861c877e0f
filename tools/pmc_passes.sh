#!/bin/bash
# Kernel-trace + PMC passes used for profiles/ (one counter group per pass, --kernel-trace only; gpurun refuses
# --pmc together with the sys/hip/hsa trace domains).  Raw SQLite outputs stay in /tmp on the GPU box (they exceed
# the 64 MiB gpurun_out quota); only the text summary is written to gpurun_out/TAG_summary.txt.
#   usage: tools/pmc_passes.sh TAG -- <command...>
TAG=$1; shift; shift
ROOT=$(cd "$(dirname "$0")/.." && pwd)
mkdir -p $ROOT/gpurun_out /tmp/prof
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d /tmp/prof/${TAG}_kt -- "$@" > /tmp/prof/${TAG}_kt.log 2>/tmp/prof/${TAG}_kt.err
dirs=""
for grp in "FETCH_SIZE" "WRITE_SIZE" "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_INSTS_MFMA SQ_INSTS_VALU GRBM_GUI_ACTIVE" \
           "SQ_WAIT_ANY SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_WAIT_INST_LDS"; do
  name=$(echo $grp | cut -d' ' -f1)
  rocprofv3 --kernel-trace --pmc $grp -d /tmp/prof/${TAG}_pmc_$name -- "$@" > /tmp/prof/${TAG}_pmc_$name.log 2>&1
  dirs="$dirs /tmp/prof/${TAG}_pmc_$name"
done
python $ROOT/tools/rocprof_summary.py $ROOT/gpurun_out/${TAG}_summary.txt /tmp/prof/${TAG}_kt $dirs
grep -a '"metric"' /tmp/prof/${TAG}_kt.log > $ROOT/gpurun_out/${TAG}_bench.json
