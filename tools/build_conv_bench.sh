#!/bin/bash
# ABL="1 2 4" tools/build_conv_bench.sh also builds timing-only ablation variants (see LS_CONV_ABL in ls_conv.hip)
# builds variants/conv_bench (timing) and variants/conv_bench_prof (stage stamps) from the current ls_conv.hip
set -e
cd "$(dirname "$0")/.."
mkdir -p variants
F="--offload-arch=gfx950 -O3 -std=c++17 -fno-slp-vectorize -w -I livelyspeaker_amd/csrc -I include"
/opt/rocm/bin/hipcc $F tools/conv_bench.cpp livelyspeaker_amd/csrc/ls_conv.hip livelyspeaker_amd/csrc/ls_gemm.hip livelyspeaker_amd/csrc/ls_train_kernels.hip -o variants/conv_bench &
/opt/rocm/bin/hipcc $F -DLS_CONV_PROF -fgpu-rdc tools/conv_bench.cpp livelyspeaker_amd/csrc/ls_conv.hip livelyspeaker_amd/csrc/ls_gemm.hip livelyspeaker_amd/csrc/ls_train_kernels.hip -o variants/conv_bench_prof &
/opt/rocm/bin/hipcc $F $BWD_FLAGS tools/conv_bwd_bench.cpp livelyspeaker_amd/csrc/ls_conv.hip livelyspeaker_amd/csrc/ls_train_kernels.hip -o variants/conv_bwd_bench${BWD_NAME:+_$BWD_NAME} &
for abl in $BWD_ABL; do     # BWD_ABL="1 2 4": timing-only ablations of the weight-gradient kernel (LS_WG_ABL in ls_conv.hip)
  /opt/rocm/bin/hipcc $F -DLS_WG_ABL=$abl tools/conv_bwd_bench.cpp livelyspeaker_amd/csrc/ls_conv.hip livelyspeaker_amd/csrc/ls_train_kernels.hip -o variants/conv_bwd_bench_abl$abl &
done
for v in $VARIANTS; do      # VARIANTS="name:-DMACRO=1 ..."  ->  variants/conv_bench_<name>
  /opt/rocm/bin/hipcc $F ${v#*:} tools/conv_bench.cpp livelyspeaker_amd/csrc/ls_conv.hip livelyspeaker_amd/csrc/ls_gemm.hip livelyspeaker_amd/csrc/ls_train_kernels.hip -o variants/conv_bench_${v%%:*} &
done
for abl in $ABL; do
  /opt/rocm/bin/hipcc $F -DLS_CONV_ABL=$abl tools/conv_bench.cpp livelyspeaker_amd/csrc/ls_conv.hip livelyspeaker_amd/csrc/ls_gemm.hip livelyspeaker_amd/csrc/ls_train_kernels.hip -o variants/conv_bench_abl$abl &
done
wait
