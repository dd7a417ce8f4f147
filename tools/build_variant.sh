#!/bin/bash
# Development aid: build variants/lib_NAME.so with extra flags for kernels_align3.hip
#   bash tools/build_variant.sh NAME [-DFLAG ...]      (run with tools/bench_variant.py)
#   FTZ= bash tools/build_variant.sh NAME ...          without the FP64 denormal flush of the product build
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
C=$ROOT/nadavca_amd/csrc
NAME=$1; shift
mkdir -p $ROOT/variants /tmp/vb_$NAME
cd /tmp/vb_$NAME
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -Wno-unused-function \
  ${SCHED--mllvm -amdgpu-sched-strategy=max-ilp} ${FTZ--fdenormal-fp-math=preserve-sign} -DNVK_VARIANT_BUILD "$@" --save-temps -c ${SRC:-$C/kernels_align3.hip} -o a3.o
API=$C/api.o
if [ -n "$APIDEBUG" ]; then  # api.hip with the debug switches (NADAVCA_ALIGN3_NORETRY leaves flagged reads alone)
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -Wno-unused-function -DNVK_VARIANT_BUILD -DNVK_DEBUG_SWITCHES -c $C/api.hip -o api_dbg.o
  API=api_dbg.o
fi
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o $ROOT/variants/lib_$NAME.so $API $C/pipeline.o $C/kernels_plan.o \
  $C/kernels_align.o a3.o $C/kernels_ell.o $C/kernels_consensus.o $C/kernels_renorm.o $C/kernels_splfit.o
echo "built $NAME: $(grep 'align3_kernelILi2ELi4ELb[01]E.*num_vgpr' kernels_align3-hip-amdgcn-amd-amdhsa-gfx950.s | sed 's/.*num_vgpr, /vgpr /')"
