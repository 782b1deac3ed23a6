#!/bin/bash
R=$GRAFT_REPO_ROOT
cd $R
bash scripts/make_profiles.sh r4 > gpurun_out/r4_make_profiles.log 2>&1
bash scripts/make_profiles.sh r4_c4 --config c4 --batch 4 > gpurun_out/r4_c4_make_profiles.log 2>&1
ls $R/gpurun_out/profiles_new | grep r4
