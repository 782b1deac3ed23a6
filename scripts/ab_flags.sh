# A/B of bench.py flags in ONE session (box-to-box variance is 8-9 %): usage  bash scripts/ab_flags.sh "<flags A>" "<flags B>" [rounds] [steps]
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
N=${3:-2}
S=${4:-20}
for i in $(seq $N); do
for F in "$1" "$2"; do
python3 $R/bench.py --steps $S --warmup 5 --no-cpu-baseline --no-mcq --no-c4 --no-variants --no-selfcheck --no-kernel-timers $F 2>/dev/null | python3 -c "import json,sys; d=json.loads([l for l in sys.stdin.read().splitlines() if l.startswith(chr(123))][-1]); print('%-28s' % sys.argv[1], d['value'], d['ms_per_step'], (d.get('power') or {}).get('mean_w'))" "[$F]"
done
done
