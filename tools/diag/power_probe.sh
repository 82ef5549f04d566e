#!/bin/bash
# samples board power / clocks with rocm-smi while bench.py runs (is the step power-limited?)
mkdir -p gpurun_out
python bench.py --no-cpu-baseline --steps 500 --warmup 3 > gpurun_out/power_bench.json 2>/dev/null &
BP=$!
: > gpurun_out/power_samples.txt
while kill -0 $BP 2>/dev/null; do
  rocm-smi --showpower --showclocks 2>/dev/null | grep -E "Power \(W\)|sclk" | sed 's/.*: //' | tr '\n' ' ' >> gpurun_out/power_samples.txt
  echo >> gpurun_out/power_samples.txt
  sleep 0.3
done
tail -1 gpurun_out/power_bench.json | cut -c1-200
sort -k3 -n -t' ' gpurun_out/power_samples.txt | uniq -c | sort -k1 -n | tail -15
