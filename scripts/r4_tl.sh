#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4o
for a in "1" "1 streams=2" "1 gemm_ln_xpf=0" "1 streams=2 gemm_ln_xpf=0" "0"; do echo "== $a"; python scripts/tower_trace.py $a 2>&1 | grep "tower pass"; done > gpurun_out/r4o/streams.txt 2>&1
cat gpurun_out/r4o/streams.txt
