#!/bin/bash
# Development aid: build variants/lib_NAME.so from the product objects with kernels_ell.hip recompiled with extra flags
#   bash tools/build_ell_variant.sh NAME [-DFLAG ...]    (SRC=other.hip: another source in its place)
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
C=$ROOT/nadavca_amd/csrc
NAME=$1; shift
mkdir -p $ROOT/variants /tmp/vbe_$NAME
cd /tmp/vbe_$NAME
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -Wno-unused-function -I$C \
  ${SCHED--mllvm -amdgpu-sched-strategy=iterative-ilp} -DNVK_VARIANT_BUILD "$@" --save-temps -c ${SRC:-$C/kernels_ell.hip} -o ell.o
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o $ROOT/variants/lib_$NAME.so $C/api.o $C/pipeline.o $C/kernels_plan.o \
  $C/kernels_align.o $C/kernels_align3.o ell.o $C/kernels_consensus.o $C/kernels_renorm.o $C/kernels_splfit.o
S=$(ls *gfx950.s | head -1)
echo "built $NAME: $(grep 'ell_kernelILi2ELb1ELi8EEE.*num_vgpr' $S | sed 's/.*num_vgpr, /vgpr /') scratch $(grep -c 'scratch_' $S)"
