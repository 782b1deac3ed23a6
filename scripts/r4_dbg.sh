#!/bin/bash
cd $GRAFT_REPO_ROOT
export AMD_SERIALIZE_KERNEL=3
for args in "10240 1024 1024 1 0" "10240 1024 1024 1 1" "10240 1024 1024 0 1" "10272 1024 1024 1 1" "9216 1024 512 1 1" "8192 1024 1024 1 1"; do
  echo "== $args"; timeout 120 python scripts/lnfold_dbg.py $args 2>&1 | grep -v amdgpu.ids | tail -3
done
