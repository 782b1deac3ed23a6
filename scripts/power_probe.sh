#!/bin/bash
# samples rocm-smi power / clocks while the bench runs (evidence for the power-limited GEMM clock)
python bench.py --steps 60 --warmup 3 --no-cpu-baseline --no-kernel-timers > gpurun_out/power_bench.log 2>&1 &
BP=$!
while kill -0 $BP 2>/dev/null; do
  rocm-smi --showpower --showclocks --showtemp 2>/dev/null | grep -E "Package Power|sclk|junction" | sed -e 's/GPU\[0\]\s*: //' | tr '\n' ';'; echo
  sleep 0.4
done > gpurun_out/power_samples.txt
tail -1 gpurun_out/power_bench.log | cut -c1-140
sort -t: -k4 -n gpurun_out/power_samples.txt | awk -F'Power \\(W\\): ' '{print $2+0, $0}' | sort -n | tail -8 | cut -c1-220
wc -l gpurun_out/power_samples.txt
