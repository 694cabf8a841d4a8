# experiment r05: a frame's deferred tail (gating + rows) on a second stream beside the next frame's launches (MMF_SIDE_STREAM=1) vs hosted
# as roles of the next frame's launches 1 and 3 (default).  Usage (gpurun): bash tools/microbench/ab_side_stream.sh
for v in 0 1; do export MMF_SIDE_STREAM=$v; for i in 1 2 3; do python3 bench.py --only-fusion --no-profile 2>/dev/null | python3 -c "
import sys,json
d=json.loads([l for l in sys.stdin if l.startswith(chr(123))][-1]); print('SIDE_STREAM=$v BL', round(d['value']), 'frames/s', round(d['ms_per_step']*1e3,2), 'us')"; done
python3 bench.py --ref-shape-only 2>/dev/null | python3 -c "
import sys,json
d=json.loads([l for l in sys.stdin if l.startswith(chr(123))][-1])['reference_shape']; print('SIDE_STREAM=$v REF', round(d['frames_per_s']), 'frames/s; pipelined', round(d['pipelined']['frames_per_s']), 'lowres pipelined ms', round(d['from_backbone_output_pipelined']['fused_lowres_ms'],4))"; done
