#!/bin/bash
# Development aid: build variants/lib_NAME.so with extra flags for kernels_align3.hip
#   bash tools/build_variant.sh NAME [-DFLAG ...]      (run with tools/bench_variant.py)
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
C=$ROOT/nadavca_amd/csrc
NAME=$1; shift
mkdir -p $ROOT/variants /tmp/vb_$NAME
cd /tmp/vb_$NAME
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -Wno-unused-function \
  -mllvm -amdgpu-sched-strategy=max-ilp "$@" --save-temps -c $C/kernels_align3.hip -o a3.o
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o $ROOT/variants/lib_$NAME.so $C/api.o $C/kernels_plan.o \
  $C/kernels_align.o a3.o $C/kernels_align4.o $C/kernels_ell.o $C/kernels_consensus.o $C/kernels_renorm.o
echo "built $NAME: $(grep 'align3_kernelILi2ELi4E.*num_vgpr' kernels_align3-hip-amdgcn-amd-amdhsa-gfx950.s | sed 's/.*num_vgpr, /vgpr /')"
