import torch, time, math
import torch.nn.functional as F
torch.manual_seed(0)
def timed(fn, n=10):
    fn(); torch.cuda.synchronize(); t0=time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter()-t0)/n*1e3
for (B,H,Lq,Lk,d) in [(32,8,616,616,15),(32,8,6,3072,15)]:
    q=torch.randn(B,H,Lq,d,device="cuda",requires_grad=True); k=torch.randn(B,H,Lk,d,device="cuda",requires_grad=True); v=torch.randn(B,H,Lk,d,device="cuda",requires_grad=True)
    pad=torch.rand(B,Lk,device="cuda")<0.2; pad[:,0]=False
    mask=(~pad)[:,None,None,:]
    def run(q,k,v,mask,scale=None):
        o=F.scaled_dot_product_attention(q,k,v,attn_mask=mask,scale=scale); o.sum().backward(); return o
    t_math=timed(lambda: run(q,k,v,mask))
    ref=F.scaled_dot_product_attention(q,k,v,attn_mask=mask).detach()
    # padded to 16
    def padded(mask_kind):
        qp=F.pad(q,(0,1)); kp=F.pad(k,(0,1)); vp=F.pad(v,(0,1))
        if mask_kind=="bool": m=mask
        elif mask_kind=="float": m=torch.zeros(B,1,1,Lk,device="cuda").masked_fill(~mask,float("-inf"))
        else: m=torch.zeros(B,1,1,Lk,device="cuda").masked_fill(~mask,float("-inf")).expand(B,H,Lq,Lk)
        o=F.scaled_dot_product_attention(qp,kp,vp,attn_mask=m,scale=1/math.sqrt(d))[...,:d]; o.sum().backward(); return o
    for kind in ("bool","float","float_expanded"):
        try:
            t=timed(lambda: padded(kind)); err=(padded(kind).detach()-ref).abs().max().item()
            print((B,H,Lq,Lk,d), kind, "padded16 %.2f ms"%t, "vs math %.2f ms"%t_math, "maxerr %.2e"%err)
        except Exception as e:
            print(kind, "failed:", str(e)[:120])
