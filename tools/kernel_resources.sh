#!/bin/bash
# Register / scratch usage of every kernel in one source file (hipcc -Rpass-analysis=kernel-resource-usage):
#   tools/kernel_resources.sh livelyspeaker_amd/csrc/ls_step.hip [-DNAME ...]
cd "$(dirname "$0")/.." || exit 1
src="$1"; shift
/opt/rocm/bin/hipcc -c --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fno-slp-vectorize -Iinclude -Ilivelyspeaker_amd/csrc "$@" "$src" -o /dev/null \
    -Rpass-analysis=kernel-resource-usage 2>&1 | grep -E "Function Name|    VGPRs:|ScratchSize|VGPRs Spill|Occupancy" |
  sed -E 's/.*remark: [^ ]+ +//; s/ \[-Rpass.*//' | paste - - - - - | c++filt | sed -E 's/Function Name: //; s/\(ls::StepArgs\)//'
