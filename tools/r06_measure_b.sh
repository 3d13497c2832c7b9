#!/bin/bash
out=gpurun_out/r06f; mkdir -p $out; export TMPDIR=/tmp
ls /sys/class/drm/ > $out/drm_cards.txt 2>&1
for spec in "ted 512 fused fp32" "ted 256 pass4 fp32" "ted 128 pass fp32" "ted 512 fused bf16x3" "ted 256 pass4 bf16x3" "ted 64 coop2 fp32" "beat 32 coop8 fp32"; do
  set -- $spec
  python tools/power_trace.py "$1 B=$2 $3 $4" -- python tools/busy_loop.py $1 $2 $3 $4 5 >> $out/power_trace.txt 2>&1
done
for spec in "ted 64 coop2" "beat 64 coop2" "ted 4 coop2" "beat 32 coop8" "beat 32 coop4" "ted 32 coop8"; do
  set -- $spec
  python tools/coop_profile.py $1 $2 $3 2>&1 | grep -v amdgpu.ids >> $out/coop_profile.txt
done
