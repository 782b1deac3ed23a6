#!/bin/bash
for shape in 16 32 16 32; do
  scripts/_bin/mfma_power $shape 5 > gpurun_out/mfma_$shape.log 2>&1 &
  P=$!
  sleep 2.5
  rocm-smi --showpower --showclocks 2>/dev/null | grep -E "Package Power|sclk" | sed -e 's/GPU\[0\]\s*: //' | tr '\n' ';'; echo
  wait $P; cat gpurun_out/mfma_$shape.log
  sleep 2
done
