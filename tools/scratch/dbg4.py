import sys, os, time
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tools"))
import numpy as np, torch
import profile_model_inputs as P
from nvblox_mindmap_amd.mapping.helpers.nvblox_input_helpers import frame_inputs_from_sample
device = torch.device("cuda:0")
cfg, C, frames, samples, ex, facade = P.build("bl", device, 12)
rows = []
for i in range(80):
    fr, smp = frames[i % 12], samples[i % 12]
    ex.next, ex.low = fr["features"], fr["lowres"]
    t0 = time.perf_counter()
    facade.decay()
    t1 = time.perf_counter()
    inp = frame_inputs_from_sample(smp, 0)
    t2 = time.perf_counter()
    facade._update_reconstruction(*inp, "pov")
    t3 = time.perf_counter()
    rows.append((i, (t1-t0)*1e3, (t2-t1)*1e3, (t3-t2)*1e3))
for r in rows:
    if max(r[1:]) > 1.0 or r[0] < 3: print("iter %d decay %.3f inputs %.3f update %.3f" % r)
