import sys, torch, time
sys.path.insert(0, '/root/repo')
from nvblox_mindmap_amd.diffuser_actor import DiffuserActorConfig
from nvblox_mindmap_amd.diffuser_actor import layers as LY, train_attention as TA, train_ops as TO
from nvblox_mindmap_amd.training import GraphedTrainStep, build_model, synthetic_batch
cfg = DiffuserActorConfig(data_type="rgbd_and_mesh", image_size=(128, 128), feature_dim=768)
batch = synthetic_batch(cfg, 8, "cuda", num_vertices=512, seed=3)
for on in (True, False):
    TA.ENABLED = TO.ENABLED = LY.FUSED_ROTARY_TRAINING = on
    torch.manual_seed(0)
    model = build_model(cfg, device="cuda")
    g = GraphedTrainStep(cfg, model, batch, lr=1e-3)
    torch.manual_seed(1)
    t0 = time.time(); ls = []
    for i in range(150):
        ls.append(g.step(batch, batch).clone())
    torch.cuda.synchronize()
    L = torch.stack(ls).cpu()
    print("kernels" if on else "torch  ", "time %.1fs" % (time.time() - t0), "loss first10 %.4f last10 %.4f" % (float(L[:10, 0].mean()), float(L[-10:, 0].mean())), [round(float(x), 4) for x in L[-10:].mean(0)])
