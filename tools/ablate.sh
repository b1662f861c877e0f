#!/bin/bash
# Build ablation variants of the forward kernel (timing experiments only; outputs are WRONG by construction).
# usage: tools/ablate.sh NAME "-DFLAG ..."   ->  tools/exp/libanerf_NAME.so
set -e
cd "$(dirname "$0")/../a-nerf_amd/csrc"
mkdir -p ../../tools/exp
OUT=../../tools/exp
/opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -fPIC -I../../include -I. $2 -c anerf_mlp.hip -o $OUT/mlp_$1.o 2>/dev/null
/opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -fPIC -I../../include -I. $2 -c anerf_mlp_b3.hip -o $OUT/mlpb3_$1.o 2>/dev/null
/opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -fPIC -I../../include -I. $2 -c anerf_capi.hip -o $OUT/capi_$1.o 2>/dev/null
/opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -fPIC -I../../include -I. $2 -c anerf_mlp_bwd.hip -o $OUT/bwd_$1.o 2>/dev/null
/opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -fPIC -I../../include -I. $2 -c anerf_gemm.hip -o $OUT/gemm_$1.o 2>/dev/null
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $OUT/capi_$1.o anerf_aux.o $OUT/mlp_$1.o $OUT/mlpb3_$1.o $OUT/bwd_$1.o $OUT/gemm_$1.o anerf_pose.o anerf_optim.o anerf_fk.o anerf_step.o -o $OUT/libanerf_$1.so
echo built $OUT/libanerf_$1.so
