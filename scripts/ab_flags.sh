# A/B of bench.py flags in ONE session (box-to-box variance is +-3 %): usage  bash scripts/ab_flags.sh "<flags A>" "<flags B>" [rounds]
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
N=${3:-2}
for i in $(seq $N); do
for F in "$1" "$2"; do
python3 $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-mcq --no-kernel-timers $F 2>/dev/null | python3 -c "import json,sys; d=json.loads([l for l in sys.stdin.read().splitlines() if l.startswith(chr(123))][-1]); print('%-28s' % sys.argv[1], d['value'], d['ms_per_step'])" "[$F]"
done
done
