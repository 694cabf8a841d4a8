for v in 0 1 2 3; do echo "variant $v"; MMF_DEBUG_FPS_VARIANT=$v python tools/time_fps.py; done
