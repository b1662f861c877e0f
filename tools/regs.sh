#!/bin/bash
# per-kernel register / scratch usage of one HIP source (compile only).  usage: tools/regs.sh a-nerf_amd/csrc/anerf_mlp.hip [extra -D flags]
SRC=$1; shift
D=$(dirname $SRC)
/opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -fPIC -I$D/../../include -I$D "$@" -Rpass-analysis=kernel-resource-usage -c $SRC -o /tmp/regs_$$.o 2>&1 | python3 -c "
import sys,re,subprocess
cur=None;rows=[]
for l in sys.stdin:
    m=re.search(r'Function Name: (\S+)',l)
    if m:
        cur={'name':subprocess.run(['c++filt',m.group(1)],capture_output=True,text=True).stdout.strip()[:72]}; rows.append(cur); continue
    m=re.search(r'remark:\s+([A-Za-z ]+?)(?: \[[^\]]*\])?: (\d+)',l)
    if m and cur is not None: cur[m.group(1).strip()]=m.group(2)
for r in rows: print(f\"{r['name']:74s} V{r.get('VGPRs','?'):>4} A{r.get('AGPRs','?'):>4} S{r.get('TotalSGPRs','?'):>4} scratch{r.get('ScratchSize','?'):>5} vspill{r.get('VGPRs Spill','?'):>4} sspill{r.get('SGPRs Spill','?'):>4}\")
"
rm -f /tmp/regs_$$.o
