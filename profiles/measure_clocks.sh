#!/bin/bash
# Shader clock and package power while bench.py runs (read-only rocm-smi polling every 0.5 s; the steady-state rows are the ones taken
# inside the timed region).  Usage: bash profiles/measure_clocks.sh <out-prefix> [bench flags]
out=$1; shift
python bench.py --cpu-sample 0 --steps 1200 "$@" > ${out}_bench.json 2> ${out}_bench.err &
pid=$!
: > ${out}_smi.txt
while kill -0 $pid 2>/dev/null; do
  echo "t $(date +%s.%N)" >> ${out}_smi.txt
  rocm-smi -d 0 --showclocks --showpower 2>/dev/null | grep -i "sclk\|mclk\|fclk\|Power" >> ${out}_smi.txt
  sleep 0.5
done
wait $pid
