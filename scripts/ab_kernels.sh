R=$GRAFT_REPO_ROOT
for round in 1 2; do
for v in "w8:--tune gemm256=3" "w4:"; do
  label=${v%%:*}; flags=${v#*:}
  line=$(python3 $R/bench.py --steps 20 --warmup 4 --no-cpu-baseline --no-mcq --no-c4 --no-power $flags 2>/dev/null | tail -1)
  echo "$label: $(echo $line | python3 -c 'import json,sys; d=json.loads(sys.stdin.read()); a=d["attention_roofline"]; print(d["value"], "clips/s | gemm", d["roofline"]["achieved"], d["roofline"]["isolated"]["achieved"], "| add_ln", a["add_ln"]["avg_launch_us"], a["add_ln"]["isolated"]["avg_launch_us"], "| space", a["space_attn"]["avg_launch_us"], a["space_attn"]["isolated"]["avg_launch_us"], "| time", a["time_attn"]["avg_launch_us"], a["time_attn"]["isolated"]["avg_launch_us"])')"
done; done
