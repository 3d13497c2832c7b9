#!/bin/bash
# Run on the GPU box (through gpurun): kernel trace + one PMC pass per counter of a short command, summarised per (kernel, grid).
#   tools/prof_call.sh OUTDIR "COUNTER1 COUNTER2 ..." -- python examples/livelyspeaker_ted.py 512
# Counters are collected in their own runs with --kernel-trace only (never combined with sys / hip / hsa traces).
out="$1"; counters="$2"; shift 3
mkdir -p "$out"
export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d "$out/kt" -o kt -- "$@" > "$out/kt.log" 2>&1
python profiles/dispatch_summary.py "$out"/kt/kt_results.db > "$out/kt.md" 2>> "$out/kt.log" || python profiles/dispatch_summary.py "$(ls "$out"/kt/*.db | head -1)" > "$out/kt.md"
for c in $counters; do
  timeout 600 rocprofv3 --pmc "$c" --kernel-trace -d "$out/pmc_$c" -o pmc -- "$@" > "$out/pmc_$c.log" 2>&1
  python profiles/dispatch_summary.py "$(ls "$out"/pmc_"$c"/*.db | head -1)" "ls::" > "$out/pmc_$c.md" 2>> "$out/pmc_$c.log"
done
