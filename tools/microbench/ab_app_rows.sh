for v in 1 0; do export MMF_NO_APP_ROWS=$v; for i in 1 2 3; do python bench.py --only-fusion --no-profile 2>/dev/null | python -c "
import sys,json
d=json.loads([l for l in sys.stdin if l.startswith(chr(123))][-1]); print('NO_APP_ROWS=$v', round(d['value']), 'frames/s; undeferred', round(d['roofline']['legs'].get('undeferred_fps',0)), d['roofline']['legs'].get('undeferred_launches'))"; done; done
