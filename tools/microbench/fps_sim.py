import numpy as np, sys
def sim(x, npts, G, K):
    N = x.shape[0]
    per = (N + G - 1)//G
    d = np.full(N, np.inf, np.float32)
    cur = 0; picks=[0]; exch=0
    # initial: dist vs start
    def upd(c):
        nonlocal d
        dd = ((x - x[c])**2).sum(1).astype(np.float32)
        d = np.minimum(d, dd)
    upd(0)
    while len(picks) < npts:
        exch += 1
        # per-group top-K by (d, -idx)
        cands=[]; T=-1.0
        for g in range(G):
            lo, hi = g*per, min(N,(g+1)*per)
            idx = np.arange(lo,hi)
            order = np.lexsort((idx, -d[lo:hi]))[:K]
            cands += list(idx[order])
            if len(order)==K: T = max(T, d[idx[order[-1]]])
        cands = np.array(cands)
        n=0
        while len(picks) < npts and n < K:
            dc = d[cands]
            j = np.lexsort((cands, -dc))[0]
            if n>0 and not (dc[j] >= T): break
            a = cands[j]; picks.append(a); upd(a); n+=1
    return picks, exch
rng = np.random.default_rng(0)
N,C=3072,120
for name,x in (("gauss", rng.standard_normal((N,C)).astype(np.float32)), ("lowrank", (rng.standard_normal((N,6))@rng.standard_normal((6,C))).astype(np.float32)), ("clustered", (rng.standard_normal((16,C))[rng.integers(0,16,N)]+0.1*rng.standard_normal((N,C))).astype(np.float32))):
    ref,_ = sim(x,614,1,1)
    for G,K in ((16,2),(16,3),(16,4),(12,2)):
        p,e = sim(x,614,G,K)
        print(name,G,K,"exchanges",e,"picks/exch %.2f"%(613/e), "same", p==ref)
