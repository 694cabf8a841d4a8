for i in 1 2 3; do python bench.py --no-train --no-infer --no-backproj --no-ref-shape --cpu-sample 0 2>/dev/null | python -c "
import sys,json
d=json.loads([l for l in sys.stdin if l.startswith(chr(123))][-1]); print(round(d['value']), round(d['ms_per_step']*1e3,1), {k:round(v,1) for k,v in d['kernel_us_per_launch'].items() if v})"; done
