import numpy as np
def sim(N=16, W=1024, steps=512, yaw=0.7, pitch=0.4, dist=1.6, ntiles=400, seed=0):
    rng=np.random.default_rng(seed)
    eye=dist*np.array([np.cos(pitch)*np.sin(yaw), np.sin(pitch), np.cos(pitch)*np.cos(yaw)])
    front=-eye/np.linalg.norm(eye); up0=np.array([0,1.,0]); right=np.cross(front,up0); right/=np.linalg.norm(right); up=np.cross(right,front)
    tanf=np.tan(np.deg2rad(45)/2)
    step=1.0/steps
    tot=0; slow=0; clampzone=0; multi=0; third=0
    for _ in range(ntiles):
        tx=rng.integers(0,W//8); ty=rng.integers(0,W//8)
        xs=(tx*8+np.arange(64)%8); ys=(ty*8+np.arange(64)//8)
        ndcx=2*(xs+0.5)/W-1; ndcy=2*(ys+0.5)/W-1
        d=front[None]+ndcx[:,None]*tanf*right[None]+ndcy[:,None]*tanf*up[None]
        d/=np.linalg.norm(d,axis=1,keepdims=True)
        inv=1/d
        t1=(-0.5-eye)*inv; t2=(0.5-eye)*inv
        tmin=np.max(np.minimum(t1,t2),axis=1); tmax=np.min(np.maximum(t1,t2),axis=1)
        tmin=np.maximum(tmin,0)
        if not np.any(tmax>=tmin): continue
        n=int(np.max((tmax-tmin)/step))+1
        i=np.arange(n)
        t=tmin[:,None]+i[None]*step      # lane, step
        valid=t<=tmax[:,None]
        p=eye[None,None]+d[:,None,:]*t[:,:,None]+0.5   # unit box coords
        f=np.clip(p*N-0.5,0,N-1)
        c0=np.minimum(np.floor(f),N-2)
        unclamped=p*N-0.5
        inclamp=((unclamped<0)|(unclamped>N-1)).any(axis=2)
        cell=(c0[...,2]*(N-1)+c0[...,1])*(N-1)+c0[...,0]
        changed=np.zeros_like(valid); changed[:,1:]=cell[:,1:]!=cell[:,:-1]
        anyvalid=valid.any(axis=0)
        tot+=anyvalid.sum()
        slow+=((changed&valid).any(axis=0)&anyvalid).sum()
        clampzone+=((inclamp&valid).any(axis=0)).sum()
        for s in range(n):
            if anyvalid[s]:
                u=np.unique(cell[valid[:,s],s]); 
                if len(u)>1: multi+=1
                if len(u)>2: third+=1
    print(f"N={N} steps={steps}: wave steps {tot}, with a crossing {slow/tot:.3f}, any lane in clamp zone {clampzone/tot:.3f}, >1 cell {multi/tot:.3f}, >2 cells {third/tot:.3f}")
sim(16); sim(32)

def sim2(N=16, W=1024, steps=512, yaw=0.7, pitch=0.4, dist=1.6, ntiles=300, seed=0):
    rng=np.random.default_rng(seed)
    eye=dist*np.array([np.cos(pitch)*np.sin(yaw), np.sin(pitch), np.cos(pitch)*np.cos(yaw)])
    front=-eye/np.linalg.norm(eye); up0=np.array([0,1.,0]); right=np.cross(front,up0); right/=np.linalg.norm(right); up=np.cross(right,front)
    tanf=np.tan(np.deg2rad(45)/2); step=1.0/steps
    cat=np.zeros(6); tot=0
    for _ in range(ntiles):
        tx=rng.integers(0,W//8); ty=rng.integers(0,W//8)
        xs=(tx*8+np.arange(64)%8); ys=(ty*8+np.arange(64)//8)
        d=front[None]+(2*(xs+0.5)/W-1)[:,None]*tanf*right[None]+(2*(ys+0.5)/W-1)[:,None]*tanf*up[None]
        d/=np.linalg.norm(d,axis=1,keepdims=True)
        t1=(-0.5-eye)/d; t2=(0.5-eye)/d
        tmin=np.maximum(np.max(np.minimum(t1,t2),axis=1),0); tmax=np.min(np.maximum(t1,t2),axis=1)
        if not np.any(tmax>=tmin): continue
        n=int(np.max((tmax-tmin)/step))+1
        t=tmin[:,None]+np.arange(n)[None]*step; valid=t<=tmax[:,None]
        p=eye[None,None]+d[:,None,:]*t[:,:,None]+0.5
        f=np.clip(p*N-0.5,0,N-1); c0=np.minimum(np.floor(f),N-2)
        cell=(c0[...,2]*(N-1)+c0[...,1])*(N-1)+c0[...,0]
        changed=np.zeros_like(valid); changed[:,1:]=cell[:,1:]!=cell[:,:-1]
        # A/B tracking: keep (cA,cB) as long as all valid lanes in {cA,cB}; else reselect (slow)
        cA=cB=-1
        for s in range(n):
            v=valid[:,s]
            if not v.any(): break
            tot+=1
            cs=cell[v,s]; u=np.unique(cs)
            covered=np.isin(cs,[cA,cB]).all()
            if covered:
                if len(u)==1: cat[0]+=1      # single cell, table resident
                else: cat[1]+=1              # two cells, both resident
            else:
                cat[2]+=1                    # reselect
                # choose: keep the cell with most lanes among current, plus next
                vals,cnt=np.unique(cs,return_counts=True); o=np.argsort(-cnt)
                cA=vals[o[0]]; cB=vals[o[1]] if len(o)>1 else cA
                if len(u)>2: cat[3]+=1
    print(f"N={N}: steps {tot}: resident single {cat[0]/tot:.3f}, resident two {cat[1]/tot:.3f}, reselect {cat[2]/tot:.3f} (of which >2 cells {cat[3]/tot:.3f})")
sim2(16); sim2(32)
