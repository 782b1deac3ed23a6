#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4j
rm -f gpurun_out/parity_measured.jsonl
python -m pytest tests/ -q -m gpu 2>&1 | tail -15 > gpurun_out/r4j/tests_all.log
tail -6 gpurun_out/r4j/tests_all.log
python __graft_entry__.py smoke 2>&1 | tail -2
