#!/bin/bash
# batch-size / configuration sweep of the train step (pipelined, LayerNorm fold): does every B run, what does it deliver -> gpurun_out/sweep/sweep.txt
cd $GRAFT_REPO_ROOT
O=gpurun_out/sweep; mkdir -p $O; : > $O/sweep.txt
F="--no-mcq --no-c4 --no-cpu-baseline --no-power --no-variants --no-kernel-timers"
for b in 1 2 3 5 8 12 16 24 40 48 64; do
  r=$(timeout 300 python bench.py --steps 6 --warmup 2 --batch $b $F 2>$O/err_$b.log | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['loss'], d.get('selfcheck',{}).get('encoder_bit_identical_clips_before_last'))" 2>&1 | tail -1)
  echo "c2 B=$b: $r" | tee -a $O/sweep.txt
done
for b in 1 2 3 6 8; do
  r=$(timeout 300 python bench.py --config c4 --steps 5 --warmup 2 --batch $b $F 2>$O/err_c4_$b.log | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['loss'])" 2>&1 | tail -1)
  echo "c4 B=$b: $r" | tee -a $O/sweep.txt
done
