#!/bin/bash
# Run on the GPU box: FETCH_SIZE / WRITE_SIZE passes of the three k_step workloads bench.py quotes traffic for, then refresh
# profiles/k_step_traffic.json (copy it back from gpurun_out/traffic/).
out=gpurun_out/traffic; mkdir -p $out
common="--steps 1 --warmup 0 --diffusion-steps 40 --no-cpu-baseline --no-parity --no-extra-legs"
tools/prof_call.sh $out/ted "FETCH_SIZE WRITE_SIZE" -- python bench.py $common
tools/prof_call.sh $out/ted1 "FETCH_SIZE WRITE_SIZE" -- python bench.py $common --scale 1.0
tools/prof_call.sh $out/beat "FETCH_SIZE WRITE_SIZE" -- python bench.py $common --dataset beat --batch 256
python tools/update_traffic.py ted:512 "k_step<35, 1, 27, 0, 0, 0>" 512 $out/ted "tools/traffic_measure.sh (40 launches, bench.py --diffusion-steps 40)"
python tools/update_traffic.py ted:512:single_pass "k_step<35, 1, 27, 0, 0, 1>" 256 $out/ted1 "tools/traffic_measure.sh (--scale 1.0)"
python tools/update_traffic.py beat:256 "k_step<36, 2, 282, 0, 0, 0>" 256 $out/beat "tools/traffic_measure.sh (--dataset beat --batch 256)"
cp profiles/k_step_traffic.json $out/
rm -rf $out/*/kt $out/*/pmc_*/
