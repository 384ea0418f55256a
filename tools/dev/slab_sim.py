import numpy as np
def rays(W, steps, yaw, pitch, dist, tx, ty):
    eye=dist*np.array([np.cos(pitch)*np.sin(yaw), np.sin(pitch), np.cos(pitch)*np.cos(yaw)])
    front=-eye/np.linalg.norm(eye); up0=np.array([0,1.,0]); right=np.cross(front,up0); right/=np.linalg.norm(right); up=np.cross(right,front)
    tanf=np.tan(np.deg2rad(45)/2); step=1.0/steps
    xs=(tx*8+np.arange(64)%8); ys=(ty*8+np.arange(64)//8)
    d=front[None]+(2*(xs+0.5)/W-1)[:,None]*tanf*right[None]+(2*(ys+0.5)/W-1)[:,None]*tanf*up[None]
    d/=np.linalg.norm(d,axis=1,keepdims=True)
    t1=(-0.5-eye)/d; t2=(0.5-eye)/d
    tmin=np.maximum(np.max(np.minimum(t1,t2),axis=1),0); tmax=np.min(np.maximum(t1,t2),axis=1)
    if not np.any(tmax>=tmin): return None
    n=int(np.max((tmax-tmin)/step))+1
    t=tmin[:,None]+np.arange(n)[None]*step; valid=t<=tmax[:,None]
    p=eye[None,None]+d[:,None,:]*t[:,:,None]+0.5
    return p,valid,d

def choose(cells, cand, N, d0, policy):
    # cells: [64,3] int ghost-extended coords in [0,N]; cand: bool mask of lanes to choose from
    idx=np.flatnonzero(cand); r0=idx[0]; c0=cells[r0]
    inC0=(cells==c0).all(axis=1)
    others=cand&~inC0
    if others.any():
        r1=np.flatnonzero(others)[0]; c1=cells[r1]
        a=int(np.flatnonzero(c1!=c0)[0]); up=c1[a]>c0[a]
    else:
        a=int(np.argmax(np.abs(d0))) if policy.get('dom',True) else 0
        up=(d0[a]>0) if policy.get('dom',True) else True
        if up and c0[a]>=N: up=False
        if (not up) and c0[a]<=0: up=True
    lower=c0.copy()
    if not up: lower[a]-=1
    return a,lower
def inslab(cells,a,lower):
    rel=cells-lower[None]
    ok=np.ones(len(cells),bool)
    for j in range(3):
        ok&= ((rel[:,j]==0)|((rel[:,j]==1)&(j==a)))
    return ok
def sim(N,policy,ntiles=300,seed=0,W=1024,steps=512):
    rng=np.random.default_rng(seed)
    tot=fast=resel=extra=0
    for _ in range(ntiles):
        r=rays(W,steps,0.7,0.4,1.6,rng.integers(0,W//8),rng.integers(0,W//8))
        if r is None: continue
        p,valid,d=r
        cells=np.clip(np.floor(p*N+0.5),0,N).astype(int)  # [64,n,3]
        res=None
        for s in range(valid.shape[1]):
            v=valid[:,s]
            if not v.any(): break
            tot+=1; c=cells[:,s]
            if res is not None and inslab(c,*res)[v].all(): fast+=1; continue
            resel+=1
            if policy.get('first_uncovered',False) and res is not None:
                cand=v&~inslab(c,*res)
                # choose r0 among uncovered, partner among all valid
                idx=np.flatnonzero(cand); r0=idx[0]; c0=c[r0]; inC0=(c==c0).all(axis=1); others=v&~inC0
                if others.any():
                    c1=c[np.flatnonzero(others)[0]]; a=int(np.flatnonzero(c1!=c0)[0]); up=c1[a]>c0[a]; lower=c0.copy(); 
                    if not up: lower[a]-=1
                    res=(a,lower)
                else: res=choose(c,v,N,d[np.flatnonzero(v)[0]],policy)
            else:
                res=choose(c,v,N,d[np.flatnonzero(v)[0]],policy)
            rem=v&~inslab(c,*res)
            while rem.any():
                extra+=1
                tmp=choose(c,rem,N,d[np.flatnonzero(rem)[0]],policy)
                rem&=~inslab(c,*tmp)
    print(f"N={N} {policy}: fast {fast/tot:.3f} reselect {resel/tot:.3f} extra passes/step {extra/tot:.3f}")
for N in (16,32):
    sim(N,dict(dom=False)); sim(N,dict(dom=True)); sim(N,dict(dom=True,first_uncovered=True))
