#!/bin/bash
# Run on the GPU box (through gpurun): the round's measurement set.  Everything lands under gpurun_out/$1/.
out="gpurun_out/$1"; mkdir -p "$out"
export TMPDIR=/tmp
python -m pytest tests -m gpu -x -q > "$out/pytest_gpu.log" 2>&1; tail -3 "$out/pytest_gpu.log"
python bench.py > "$out/bench_default.json" 2> "$out/bench_default.err"; echo "default rc=$?"
python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29621 bench.py --gpus 1 --global-batch 512 \
    --no-extra-legs --no-cpu-baseline > "$out/bench_strong_n1.json" 2> "$out/bench_strong_n1.err"; echo "strong rc=$?"
python bench.py --dataset beat --batch 256 --no-extra-legs --no-cpu-baseline > "$out/bench_beat256.json" 2> "$out/bench_beat256.err"; echo "beat rc=$?"
python bench.py --dataset beat150 --batch 32 --no-extra-legs --no-cpu-baseline --steps 2 > "$out/bench_beat150_b32.json" 2> "$out/bench_beat150_b32.err"; echo "beat150/32 rc=$?"
python bench.py --dataset beat150 --batch 256 --no-extra-legs --no-cpu-baseline --steps 1 --diffusion-steps 200 > "$out/bench_beat150_b256.json" 2> "$out/bench_beat150_b256.err"; echo "beat150/256 rc=$?"
python bench.py --respacing ddim100 --no-extra-legs --no-cpu-baseline --steps 5 > "$out/bench_ddim100_full.json" 2> "$out/bench_ddim100_full.err"; echo "ddim100 rc=$?"
python bench.py --batch 4 --diffusion-steps 50 --no-extra-legs --no-cpu-baseline --steps 20 > "$out/bench_config1_shape.json" 2> "$out/bench_config1.err"; echo "config1 rc=$?"
python bench.py --gpus 2 --ranks-share-device --batch 256 --legs lively --no-cpu-baseline --steps 2 > "$out/bench_two_ranks_one_gpu.json" 2> "$out/bench_two_ranks_one_gpu.err"; echo "2 ranks / 1 GPU rc=$?"
python tools/coop_time.py ted 30 4,8,16,24,32,40,48,64,72,80,88,96,112,128,144,160,176,192,224,256,288,320,352,384,416,448,480,512 coop8,coop4,coop2,batch,pass,fused,auto 2>&1 | grep -v amdgpu.ids > "$out/tvb_ted.txt"
python tools/coop_time.py beat 30 4,16,32,48,64,80,96,128,160,192,256 coop8,coop4,coop2,batch,pass,fused,auto 2>&1 | grep -v amdgpu.ids > "$out/tvb_beat.txt"
python tools/bf16x3_time.py auto,fused,pass 64,128,256,384,512 32,128,256 2>&1 | grep -v amdgpu.ids > "$out/bf16x3_time.txt"
python bench.py --gpus 8 --ranks-share-device --batch 64 --legs none --no-cpu-baseline --steps 2 --warmup 1 > "$out/bench_eight_ranks_one_gpu.json" 2> "$out/bench_eight_ranks.err"; echo "8 ranks / 1 GPU rc=$?"
python bench.py --gpus 8 --launcher threads --ranks-share-device --batch 64 --path pass --steps 2 --warmup 1 > "$out/bench_threads_eight_handles.json" 2> "$out/bench_threads8.err"; echo "threads launcher x8 rc=$?"
python bench.py --gpus 2 --launcher threads --ranks-share-device --batch 256 --steps 2 > "$out/bench_threads_two_handles.json" 2> "$out/bench_threads.err"; echo "threads launcher rc=$?"
tools/prof_call.sh "$out/kstep" "SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_COEXEC_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT GRBM_GUI_ACTIVE" -- python tools/coop_time.py ted 20 512 fused
tools/prof_call.sh "$out/coop" "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_INSTS_VALU FETCH_SIZE WRITE_SIZE" -- python tools/coop_time.py beat 20 32 coop
tools/prof_call.sh "$out/coop2_ted64" "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_INSTS_VALU FETCH_SIZE WRITE_SIZE" -- python tools/coop_time.py ted 20 64 coop
# the one-pass-per-workgroup kernel: one workgroup per CU (B = 128) and two (B = 256); GRBM_GUI_ACTIVE / 8 / kernel time = the clock the chip held
tools/prof_call.sh "$out/pass128" "SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA GRBM_GUI_ACTIVE SQ_WAIT_INST_ANY FETCH_SIZE WRITE_SIZE" -- python tools/coop_time.py ted 20 128 pass
tools/prof_call.sh "$out/pass256" "SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA GRBM_GUI_ACTIVE SQ_WAIT_INST_ANY FETCH_SIZE WRITE_SIZE" -- python tools/coop_time.py ted 20 256 pass
tools/prof_call.sh "$out/fused256" "GRBM_GUI_ACTIVE" -- python tools/coop_time.py ted 20 256 fused
tools/prof_call.sh "$out/pass256_bf16x3" "SQ_VALU_MFMA_BUSY_CYCLES SQ_VALU_MFMA_COEXEC_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA GRBM_GUI_ACTIVE" -- python tools/bf16x3_time.py pass 256 256
tools/prof_call.sh "$out/fused256_bf16x3" "SQ_VALU_MFMA_BUSY_CYCLES SQ_VALU_MFMA_COEXEC_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA GRBM_GUI_ACTIVE" -- python tools/bf16x3_time.py fused 256 256
LS_PROF=0 python tools/wg_place.py ted 256 2>&1 | grep -v amdgpu.ids > "$out/wg_place_pass_ted256.txt"
LS_PROF=0 LS_PROF_PRECISION=bf16x3 python tools/wg_place.py ted 256 2>&1 | grep -v amdgpu.ids > "$out/wg_place_pass_ted256_bf16x3.txt"
[ -x variants/conv_bench ] && variants/conv_bench 512 > "$out/conv_bench.txt" 2>&1
[ -x variants/conv_bench_prof ] && variants/conv_bench_prof 512 > "$out/conv_bench_prof.txt" 2>&1
[ -x variants/conv_bwd_bench ] && variants/conv_bwd_bench 512 > "$out/conv_bwd_bench.txt" 2>&1
python tools/sag_time.py > "$out/sag_time.txt" 2>&1
python tools/train_perf.py ted 512 8 2>&1 | tail -4 > "$out/train_perf.txt"
python bench.py --scale 1.0 --no-extra-legs --no-cpu-baseline > "$out/bench_scale1.json" 2> "$out/bench_scale1.err"; echo "scale1 rc=$?"
rocprofv3 --kernel-trace --stats -d "$out/kt" -o kt -- python bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-parity --no-extra-legs > "$out/kt_bench.json" 2> "$out/kt.log"
python profiles/summarize_rocprof.py "$(ls $out/kt/*.db | head -1)" "kernel trace of bench.py --steps 1 --warmup 1 (headline workload)" "$out/kt_bench.json" > "$out/kt_bench.md"
python profiles/dispatch_summary.py "$(ls $out/kt/*.db | head -1)" "ls::" > "$out/kt_dispatch.md"
tools/prof_call.sh "$out/lively" "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE FETCH_SIZE WRITE_SIZE" -- python examples/livelyspeaker_ted.py 512
for b in 4 8 16 32 40 64 96 128; do python tools/mix_time.py $b 30 coop; python tools/mix_time.py $b 30 batch; done 2>&1 | grep -v amdgpu.ids > "$out/mix_time.txt"
tools/prof_call.sh "$out/mixer" "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_VALU SQ_INSTS_MFMA SQ_WAIT_INST_ANY SQ_ACTIVE_INST_LDS FETCH_SIZE WRITE_SIZE" -- python tools/mix_time.py 32 20 coop
tools/prof_call.sh "$out/long" "" -- python bench.py --dataset beat150 --batch 32 --no-extra-legs --no-cpu-baseline --no-parity --no-traffic-pass --steps 1 --warmup 1 --diffusion-steps 20
tools/prof_call.sh "$out/long256" "" -- python bench.py --dataset beat150 --batch 256 --no-extra-legs --no-cpu-baseline --no-parity --no-traffic-pass --steps 1 --warmup 1 --diffusion-steps 20
tools/prof_call.sh "$out/train" "" -- python tools/train_perf.py ted 512 4
tools/prof_call.sh "$out/prepare" "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE" -- python tools/prepare_only.py
rm -rf "$out"/kt "$out"/*/kt "$out"/*/pmc_*/
for f in "$out"/bench_*.json; do python - "$f" <<'PY'
import json, sys
try:
    r = json.load(open(sys.argv[1]))
    print(sys.argv[1].split("/")[-1], r["value"], r["ms_per_step"], r["roofline"].get("kernel_ms", r["roofline"].get("kernel_ms_device0")), r["roofline"]["frac"], (r.get("parity_in_run") or {}).get("max_abs_diff"), (r.get("shard_check") or {}).get("bitwise_equal"))
except Exception as e:
    print(sys.argv[1], "ERR", e)
PY
done
