#!/bin/bash
# Round 6, first measurement set (GPU box): hand-off probe, clock / power telemetry, counters of the sample-split slicings.
out=gpurun_out/r06e; mkdir -p $out; export TMPDIR=/tmp
variants/handoff_probe > $out/handoff_probe.txt 2>&1
for spec in "ted 512 fused fp32" "ted 256 pass4 fp32" "ted 128 pass fp32" "ted 512 fused bf16x3" "ted 256 pass4 bf16x3" "ted 64 coop2 fp32" "ted 32 coop8 fp32" "beat 32 coop8 fp32"; do
  set -- $spec
  python tools/power_trace.py "$1 B=$2 $3 $4" -- python tools/busy_loop.py $1 $2 $3 $4 5 >> $out/power_trace.txt 2>&1
  python tools/busy_loop.py $1 $2 $3 $4 1 2>&1 | grep -v amdgpu.ids >> $out/power_trace_ms.txt
done
C="SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INSTS_VALU FETCH_SIZE WRITE_SIZE"
tools/prof_call.sh $out/coop8_beat32 "$C" -- python tools/coop_time.py beat 20 32 coop8
tools/prof_call.sh $out/coop4_beat32 "$C" -- python tools/coop_time.py beat 20 32 coop4
tools/prof_call.sh $out/coop2_beat64 "$C" -- python tools/coop_time.py beat 20 64 coop2
tools/prof_call.sh $out/coop2_ted64 "$C" -- python tools/coop_time.py ted 20 64 coop2
rm -rf $out/*/kt $out/*/pmc_*/
