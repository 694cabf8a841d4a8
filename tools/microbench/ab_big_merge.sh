# hash path (unbounded workspace): merged launches of round 5 on / off (MMF_NO_BIG_MERGE), three runs each.  Usage (gpurun): bash tools/microbench/ab_big_merge.sh
for v in 1 0; do export MMF_NO_BIG_MERGE=$v; for i in 1 2 3; do python3 bench.py --unbounded-only --steps 100 --warmup 60 2>/dev/null | python3 -c "
import sys,json
d=json.loads([l for l in sys.stdin if l.startswith(chr(123))][-1])['unbounded_workspace']; print('NO_BIG_MERGE=$v', round(d['frames_per_s']), 'frames/s', round(d['ms_per_step']*1e3,1), 'us; pipelined', d.get('pipelined') and round(d['pipelined']['frames_per_s']), {k['kernel'][:22]: round(k['avg_us_per_frame'],1) for k in d['per_kernel'] if k['avg_us_per_frame']})"; done; done
