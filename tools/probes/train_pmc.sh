#!/bin/bash
# PMC passes of the training step's two fused mixer kernels (GPU box): bash tools/probes/train_pmc.sh
cd ${GRAFT_REPO_ROOT:-.}; export TMPDIR=/tmp
CMD="python tools/train_perf.py ted 512 4"
mkdir -p gpurun_out/trp
timeout 600 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE --kernel-trace -d gpurun_out/trp/p1 -o pmc -- $CMD > gpurun_out/trp/p1.log 2>&1
timeout 600 rocprofv3 --pmc SQ_INSTS_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD --kernel-trace -d gpurun_out/trp/p2 -o pmc -- $CMD > gpurun_out/trp/p2.log 2>&1
timeout 600 rocprofv3 --pmc FETCH_SIZE --kernel-trace -d gpurun_out/trp/p3 -o pmc -- $CMD > gpurun_out/trp/p3.log 2>&1
timeout 600 rocprofv3 --pmc WRITE_SIZE --kernel-trace -d gpurun_out/trp/p4 -o pmc -- $CMD > gpurun_out/trp/p4.log 2>&1
for p in p1 p2 p3 p4; do for k in "ls::k_mixer_bwd" "ls::k_step" "ls::k_gemm_tr<false, false, true" "ls::k_conv"; do python profiles/dispatch_summary.py "$(ls gpurun_out/trp/$p/*.db | head -1)" "$k" | grep "ls::" ; done > gpurun_out/trp/$p.md 2>>gpurun_out/trp/$p.log; done
rm -rf gpurun_out/trp/p?/
cat gpurun_out/trp/p*.md | cut -c1-330
