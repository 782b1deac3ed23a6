#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4h
python scripts/launch_census2.py 2>&1 | grep -v "amdgpu.ids\|Warning\|_warn_once" > gpurun_out/r4h/census2.txt
cat gpurun_out/r4h/census2.txt
