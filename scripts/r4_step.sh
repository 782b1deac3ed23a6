#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4h
python -m pytest tests/test_step_gpu.py tests/test_dp_gpu.py tests/test_kernels_gpu.py -x -q -m gpu -k "not gemm_persistent" 2>&1 | tail -15 > gpurun_out/r4h/tests_step.log
tail -6 gpurun_out/r4h/tests_step.log
python scripts/launch_census2.py 2>&1 | grep -v "amdgpu.ids\|Warning\|_warn_once" > gpurun_out/r4h/census2b.txt
head -2 gpurun_out/r4h/census2b.txt
