# in-step and isolated time of the space attention kernel per hh_set_tuning("space_mfma32") value (bench.py's own library-side timers), and what
# the choice does to the DECODER stream's span inside the pipelined step (a persistent attention kernel holds every CU for its whole duration)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for M in 0 1 2 0 1 2; do
python3 $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-mcq --no-c4 --no-variants --no-selfcheck --no-power --tune space_mfma32=$M 2>/dev/null | python3 -c "
import json,sys
d=json.loads([l for l in sys.stdin.read().splitlines() if l.startswith(chr(123))][-1])
a=d['attention_roofline']['space_attn']; c=d['config']
print('space_mfma32=%s  %.2f clips/s  %s  in step %.1f us  isolated %.1f us | time attention in step %.1f us | decoder stream in step %.1f ms (%.2f of the step), alone %.1f ms | GEMM stream %.1f ms' % (sys.argv[1], d['value'], a['kernel'], a['avg_launch_us'], a['isolated']['avg_launch_us'], d['attention_roofline']['time_attn']['avg_launch_us'], c['decoder_in_step_ms'], c['decoder_in_step_frac'], c['decoder_alone_ms'], d['roofline']['stream_ms_per_step']))" $M
done
