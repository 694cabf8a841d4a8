import math, torch, sys
sys.path.insert(0, '/root/repo')
from nvblox_mindmap_amd.diffuser_actor.train_attention import train_attention
torch.manual_seed(0)
def ref(q, k, v, mask, H):
    B, Lq, D = q.shape; Lk = k.shape[1]; hd = D // H
    qh = q.view(B, Lq, H, hd).transpose(1, 2).double(); kh = k.view(B, Lk, H, hd).transpose(1, 2).double(); vh = v.view(B, Lk, H, hd).transpose(1, 2).double()
    s = qh @ kh.transpose(-1, -2) / math.sqrt(hd)
    if mask is not None: s = s.masked_fill(mask[:, None, None, :], float("-inf"))
    return (s.softmax(-1) @ vh).transpose(1, 2).reshape(B, Lq, D)
for (B, H, hd, Lq, Lk, masked, chunk) in [(2, 8, 15, 616, 616, True, False), (3, 4, 16, 70, 200, False, True), (1, 2, 8, 64, 129, True, True), (32, 8, 15, 616, 616, True, True)]:
    D = H * hd
    q = torch.randn(B, Lq, D, device="cuda", requires_grad=True)
    if chunk:
        kv = torch.randn(B, Lk, 2 * D, device="cuda", requires_grad=True); k, v = kv.chunk(2, dim=-1)
    else:
        k = torch.randn(B, Lk, D, device="cuda", requires_grad=True); v = torch.randn(B, Lk, D, device="cuda", requires_grad=True)
    mask = None
    if masked:
        mask = torch.zeros(B, Lk, dtype=torch.bool, device="cuda"); mask[:, Lk - 37:] = True; mask[0, 5] = True
    g = torch.randn(B, Lq, D, device="cuda")
    out = train_attention(q, k, v, mask, H)
    (out * g).sum().backward()
    got = [out.detach().double(), q.grad.double()] + ([kv.grad.double()] if chunk else [k.grad.double(), v.grad.double()])
    q2 = q.detach().clone().requires_grad_(True)
    if chunk:
        kv2 = kv.detach().clone().requires_grad_(True); k2, v2 = kv2.chunk(2, dim=-1)
    else:
        k2 = k.detach().clone().requires_grad_(True); v2 = v.detach().clone().requires_grad_(True)
    o2 = ref(q2, k2, v2, mask, H)
    (o2 * g.double()).sum().backward()
    want = [o2.detach(), q2.grad.double()] + ([kv2.grad.double()] if chunk else [k2.grad.double(), v2.grad.double()])
    errs = [float((a - b).abs().max() / b.abs().max()) for a, b in zip(got, want)]
    print((B, H, hd, Lq, Lk, masked, chunk), ["%.2e" % e for e in errs])
# timing at the policy's shape
import time
B, H, hd, L = 32, 8, 15, 616; D = H * hd
q = torch.randn(B, L, D, device="cuda", requires_grad=True); kv = torch.randn(B, L, 2 * D, device="cuda", requires_grad=True); k, v = kv.chunk(2, -1)
mask = torch.zeros(B, L, dtype=torch.bool, device="cuda"); mask[:, -10:] = True
g = torch.randn(B, L, D, device="cuda")
def run_mine():
    o = train_attention(q, k, v, mask, H); o.backward(g)
import torch.nn.functional as F
def run_sdpa():
    qh = F.pad(q.view(B, L, H, hd).transpose(1, 2), (0, 1)); kh = F.pad(k.reshape(B, L, H, hd).transpose(1, 2), (0, 1)); vh = F.pad(v.reshape(B, L, H, hd).transpose(1, 2), (0, 1))
    o = F.scaled_dot_product_attention(qh, kh, vh, attn_mask=(~mask)[:, None, None, :], scale=1 / math.sqrt(hd))[..., :hd].transpose(1, 2).reshape(B, L, D); o.backward(g)
for name, fn in (("mine", run_mine), ("sdpa", run_sdpa)):
    for _ in range(3): fn()
    torch.cuda.synchronize(); a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(10): fn()
    b.record(); torch.cuda.synchronize()
    print(name, "fwd+bwd ms", a.elapsed_time(b) / 10)
