cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
CMD="python bench.py --dataset beat150 --batch 256 --no-extra-legs --no-cpu-baseline --no-parity --no-traffic-pass --steps 1 --warmup 1 --diffusion-steps 10"
mkdir -p gpurun_out/tm
timeout 600 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --kernel-trace -d gpurun_out/tm/p1 -o pmc -- $CMD > gpurun_out/tm/p1.log 2>&1
timeout 600 rocprofv3 --pmc SQ_WAIT_INST_LDS SQ_INSTS_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT --kernel-trace -d gpurun_out/tm/p2 -o pmc -- $CMD > gpurun_out/tm/p2.log 2>&1
timeout 600 rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INSTS_LDS GRBM_GUI_ACTIVE --kernel-trace -d gpurun_out/tm/p3 -o pmc -- $CMD > gpurun_out/tm/p3.log 2>&1
for p in p1 p2 p3; do python profiles/dispatch_summary.py "$(ls gpurun_out/tm/$p/*.db | head -1)" "ls::k_long_tokmix" > gpurun_out/tm/$p.md 2>>gpurun_out/tm/$p.log; python profiles/dispatch_summary.py "$(ls gpurun_out/tm/$p/*.db | head -1)" "ls::k_gemm_dma" >> gpurun_out/tm/$p.md 2>>gpurun_out/tm/$p.log; done
cat gpurun_out/tm/p*.md | cut -c1-400
