# hash-path leg for several library builds: bash tools/microbench/ab_unbounded_libs.sh libmmfusion.so libmmfusion_x.so
for lib in "$@"; do export MMF_LIB=$lib; for i in 1 2 3; do python3 bench.py --unbounded-only --steps 100 --warmup 60 2>/dev/null | python3 -c "
import sys,json
d=json.loads([l for l in sys.stdin if l.startswith(chr(123))][-1])['unbounded_workspace']; print('$lib', round(d['frames_per_s']), 'frames/s; pipelined', d.get('pipelined') and round(d['pipelined']['frames_per_s']), {k['kernel'][:22]: round(k['avg_us_per_frame'],1) for k in d['per_kernel'] if k['avg_us_per_frame'] and 'tsdf' in k['kernel']})"; done; done
