cd $GRAFT_REPO_ROOT
rm -f gpurun_out/parity_measured.jsonl
python -m pytest tests -m gpu -q > gpurun_out/r6_gputest_full.txt 2>&1; tail -4 gpurun_out/r6_gputest_full.txt > gpurun_out/r6_gputest_tail.txt
bash scripts/make_profiles.sh r6 > gpurun_out/make_profiles_r6.log 2>&1
bash scripts/make_profiles.sh r6_c4 --config c4 --batch 4 > gpurun_out/make_profiles_r6c4.log 2>&1
cd /tmp; python3 $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 5 > $GRAFT_REPO_ROOT/gpurun_out/r6_bench_line.json 2> $GRAFT_REPO_ROOT/gpurun_out/r6_bench_line.err
cat $GRAFT_REPO_ROOT/gpurun_out/r6_gputest_tail.txt; tail -c 300 $GRAFT_REPO_ROOT/gpurun_out/r6_bench_line.json
