#!/bin/bash
# Sample the shader clock and the package power while a command keeps the GPU busy:  tools/clock_watch.sh <out file> -- <command>
out="$1"; shift; shift
( "$@" > /dev/null 2>&1 ) &
pid=$!
sleep 1.5
for i in 1 2 3 4 5 6; do
  /opt/rocm/bin/rocm-smi --showclocks --showpower 2>/dev/null | grep -E "sclk|Power|mclk" | tr '\n' ' '; echo
  sleep 0.5
done > "$out" 2>&1
wait $pid
cat "$out"
