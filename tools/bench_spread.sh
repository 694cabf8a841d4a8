# run-to-run spread of the two headline numbers on one box: the fusion stream at the driver's region length (--steps 20 --warmup 5) and
# at the default 200 steps, and the captured training step -- five processes each.  Usage (gpurun): bash tools/bench_spread.sh
for i in 1 2 3 4 5; do
  python3 bench.py --only-fusion --no-profile --steps 20 --warmup 5 2>/dev/null | python3 -c "
import sys,json
d=json.loads([l for l in sys.stdin if l.startswith(chr(123))][-1]); print('fusion steps=20 ', round(d['value']), 'frames/s', round(d['ms_per_step']*1e3,2), 'us  regions', [round(x,3) for x in d['region_ms']])"
done
for i in 1 2 3 4 5; do
  python3 bench.py --only-fusion --no-profile 2>/dev/null | python3 -c "
import sys,json
d=json.loads([l for l in sys.stdin if l.startswith(chr(123))][-1]); print('fusion steps=200', round(d['value']), 'frames/s', round(d['ms_per_step']*1e3,2), 'us')"
done
for i in 1 2 3 4 5; do
  python3 bench.py --train-only 2>/dev/null | python3 -c "
import sys,json
t=json.loads([l for l in sys.stdin if l.startswith(chr(123))][-1])['train']; print('train step', round(t['ms_per_step'],2), 'ms  host enqueue', round(t['host_enqueue_ms_per_step'],2), 'ms')"
done
