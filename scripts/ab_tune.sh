#!/bin/bash
# same-session A/B of hh_set_tuning knobs at step level: alternates `bench.py --tune ...` variants, 3 rounds, 20 timed steps each
# usage: ab_tune.sh "label1:--tune a=1" "label2:" ...      (the part after the colon is appended to the bench command line)
R=$GRAFT_REPO_ROOT
for round in 1 2 3; do
  for v in "$@"; do
    label=${v%%:*}; flags=${v#*:}
    line=$(python3 $R/bench.py --steps 20 --warmup 4 --no-cpu-baseline --no-mcq --no-c4 --no-power --no-kernel-timers $flags 2>/dev/null | tail -1)
    echo "round $round $label: $(echo $line | python3 -c 'import json,sys; d=json.loads(sys.stdin.read()); print(d["value"], "clips/s", d["ms_per_step"], "ms/step  p50", d["step_stats"]["ms_per_step_p50"])')"
  done
done
