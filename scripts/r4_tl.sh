#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4q
python scripts/lnfold_skew_timeline.py > gpurun_out/r4q/skew_timeline.txt 2>&1
cat gpurun_out/r4q/skew_timeline.txt
timeout 600 python -m pytest tests/test_kernels_gpu.py -x -q -m gpu -k "224" 2>&1 | tail -3
python scripts/launch_census2.py > gpurun_out/r4q/census.txt 2>&1; head -1 gpurun_out/r4q/census.txt
